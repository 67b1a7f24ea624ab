// Device half of the cross-clip tracker (mdqe/tracking/OverTracker.py).
// The bank holds a running SUM of mask logits per (instance, frame); a saved instance's mask on an
// overlapping frame is sigmoid(sum/n_present) > 0.5 <=> sum > 0, so the hard-mask IoU of
// OverTracker._get_siou (:92-113) only needs sign tests -- no 0/1 matrices are materialised.
#include "common.h"

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  __syncthreads();
  return r;
}

// out[i, j] += (inter, |saved_i|, |input_j|) over n contiguous floats per row pair.  The pixel range is cut into
// gridDim.z chunks so that even a 1x1 pair fills the chip; partial counts are integers, so the float atomicAdd
// accumulation is exact and order-independent (deterministic).  `out` is zeroed by the launcher.
__global__ void __launch_bounds__(256)
trk_siou_kernel(const float* __restrict__ saved, long saved_stride, const float* __restrict__ inp, long inp_stride,
                long n, float* __restrict__ out, int n_in) {
  __shared__ float sh[4];
  const int i = blockIdx.y, j = blockIdx.x;
  const long per = ((n / 4 + gridDim.z - 1) / gridDim.z) * 4;
  const long k0 = (long)blockIdx.z * per, k1 = min(n, k0 + per);
  const float* a = saved + (long)i * saved_stride;
  const float* b = inp + (long)j * inp_stride;
  float ci = 0.f, ca = 0.f, cb = 0.f;
  for (long k = k0 + (long)threadIdx.x * 4; k < k1; k += 1024) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(a + k);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(b + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool pa = va[e] > 0.f, pb = vb[e] > 0.f;
      ci += (pa && pb) ? 1.f : 0.f; ca += pa ? 1.f : 0.f; cb += pb ? 1.f : 0.f;
    }
  }
  ci = block_sum(ci, sh); ca = block_sum(ca, sh); cb = block_sum(cb, sh);
  if (threadIdx.x == 0) {
    float* o = out + ((long)i * n_in + j) * 3;
    atomicAdd(o, ci); atomicAdd(o + 1, ca); atomicAdd(o + 2, cb);
  }
}

extern "C" int mdqe_trk_siou_f32(const float* saved, long saved_stride, int n_saved, const float* inp, long inp_stride,
                                 int n_in, long n, float* out3, void* stream) {
  MDQE_REQUIRE(n_saved >= 0 && n_in >= 0 && n >= 0 && n % 4 == 0 && saved_stride % 4 == 0 && inp_stride % 4 == 0);
  if (n_saved == 0 || n_in == 0) return MDQE_OK;
  MDQE_CHECK_PTR(saved); MDQE_CHECK_PTR(inp); MDQE_CHECK_PTR(out3);
  MDQE_REQUIRE((((uintptr_t)saved | (uintptr_t)inp) & 15) == 0);
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out3, 0, (size_t)n_saved * n_in * 3 * sizeof(float), st) != hipSuccess) return MDQE_ELAUNCH;
  int chunks = 1;
  const long pairs = (long)n_saved * n_in;
  if (pairs < 512) { chunks = (int)(512 / pairs); const long maxc = (n / 4 + 1023) / 1024; if (chunks > maxc) chunks = (int)maxc; if (chunks < 1) chunks = 1; }
  hipLaunchKernelGGL(trk_siou_kernel, dim3(n_in, n_saved, chunks), dim3(256), 0, st, saved, saved_stride, inp,
                     inp_stride, n, out3, n_in);
  return mdqe_launch_status();
}

struct TrkIdx { int r[128]; int c[128]; };

// sum[r[k], :n] += src[c[k], :n] ; cnt[r[k], f] += 1 for f < nf   (OverTracker._update_memory :65-76)
__global__ void __launch_bounds__(256)
trk_accumulate_kernel(float* __restrict__ sum, long sum_stride, float* __restrict__ cnt, long cnt_stride,
                      const float* __restrict__ src, long src_stride, long n, int nf, TrkIdx idx) {
  const int k = blockIdx.y;
  float* d = sum + (long)idx.r[k] * sum_stride;
  const float* s = src + (long)idx.c[k] * src_stride;
  for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += (long)gridDim.x * blockDim.x * 4) {
    f32x4 v = *reinterpret_cast<f32x4*>(d + e);
    v += *reinterpret_cast<const f32x4*>(s + e);
    *reinterpret_cast<f32x4*>(d + e) = v;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < nf) cnt[(long)idx.r[k] * cnt_stride + threadIdx.x] += 1.f;
}

extern "C" int mdqe_trk_accumulate_f32(float* sum, long sum_stride, float* cnt, long cnt_stride, const float* src,
                                       long src_stride, long n, int nf, const int* r_host, const int* c_host, int count,
                                       void* stream) {
  MDQE_REQUIRE(count >= 0 && count <= 128 && n >= 0 && n % 4 == 0 && nf >= 0 && nf <= 256);
  MDQE_REQUIRE(sum_stride % 4 == 0 && src_stride % 4 == 0);
  if (count == 0 || n == 0) return MDQE_OK;
  MDQE_CHECK_PTR(sum); MDQE_CHECK_PTR(cnt); MDQE_CHECK_PTR(src); MDQE_CHECK_PTR(r_host); MDQE_CHECK_PTR(c_host);
  MDQE_REQUIRE((((uintptr_t)sum | (uintptr_t)src) & 15) == 0);
  TrkIdx idx;
  for (int i = 0; i < 128; ++i) { idx.r[i] = 0; idx.c[i] = 0; }
  for (int i = 0; i < count; ++i) { idx.r[i] = r_host[i]; idx.c[i] = c_host[i]; }
  mdqe_clear_error();
  long bx = (n / 4 + 255) / 256; if (bx > 64) bx = 64;
  hipLaunchKernelGGL(trk_accumulate_kernel, dim3((unsigned)bx, count), dim3(256), 0, (hipStream_t)stream, sum, sum_stride, cnt,
                     cnt_stride, src, src_stride, n, nf, idx);
  return mdqe_launch_status();
}
