// Multi-scale deformable attention sampling for gfx950 (MI355X), forward only.
//
// Arithmetic follows the reference kernel (mdqe/models/ops/src/cuda/ms_deform_im2col_cuda.cuh
// :33-84 bilinear, :237-299 main loop): pixel = loc*size - 0.5; a sample contributes only if
// -1 < h < H and -1 < w < W; each of the four corners is zero outside the map.
//
// Mapping (CDNA4): one lane owns VEC consecutive channels of one (b,q,head) -- with D=32 and
// VEC=4 a 64-lane wave is exactly one query (8 heads x 8 lanes), every corner fetch is one
// global_load_dwordx4 per lane (1 KiB per wave-instruction in eight 128-B segments) and the
// output row [b,q,:] is written as one contiguous 1 KiB store.  value is read through L1/L2 (the
// per-frame value map, 5.2 MB at 360p, is L2/MALL resident); loc/attn are streamed once.
#include "common.h"

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T ldv(const float* p) {
  return *reinterpret_cast<const typename VecT<VEC>::T*>(p);
}

// LP > 0: compile-time L*P (fully unrolled); LP == 0: runtime loops.
template <int VEC, int LC, int PC>
__global__ void __launch_bounds__(256)
msda_fwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                const float* __restrict__ attn, int B, int S, int M, int D, int G, int Lr, int Q, int Pr,
                float scale, float* __restrict__ out, long total) {
  typedef typename VecT<VEC>::T V;
  const int L = LC > 0 ? LC : Lr;
  const int P = PC > 0 ? PC : Pr;
  const int DV = D / VEC;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % DV);
    long t = idx / DV;
    const int m = (int)(t % M);
    t /= M;                                   // t = b*Q + q
    const int b = (int)(t / Q);
    const long samp = (t * M + m) * (long)(L * P);   // index of (b,q,m,0,0)
    const float* lp = loc + samp * 2;
    const float* wp = attn + samp;
    const int rs = M * D;                     // stride between pixels
    const float* vb = value + (long)b * S * rs + m * D + cv * VEC;
    V acc = {0};
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[(g * L + l) * 2];
        const int W = (int)shapes[(g * L + l) * 2 + 1];
        const float* vl = vb + level_start[g * L + l] * (long)rs;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float lx = lp[(l * P + p) * 2];
          const float ly = lp[(l * P + p) * 2 + 1];
          const float aw = wp[l * P + p];
          const float h_im = ly * H - 0.5f;
          const float w_im = lx * W - 0.5f;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const int h_low = (int)floorf(h_im);
            const int w_low = (int)floorf(w_im);
            const float lh = h_im - h_low, lw = w_im - w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
            const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
            const float* p00 = vl + ((long)h_low * W + w_low) * rs;
            V v1 = {0}, v2 = {0}, v3 = {0}, v4 = {0};
            if (h0 && w0) v1 = ldv<VEC>(p00);
            if (h0 && w1) v2 = ldv<VEC>(p00 + rs);
            if (h1 && w0) v3 = ldv<VEC>(p00 + (long)W * rs);
            if (h1 && w1) v4 = ldv<VEC>(p00 + (long)W * rs + rs);
            const V val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
            acc += val * aw;
          }
        }
      }
    }
    *reinterpret_cast<V*>(out + t * rs + m * D + cv * VEC) = acc * scale;
  }
}

template <int VEC>
static int launch_msda(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                       const float* attn, int B, int S, int M, int D, int G, int L, int Q, int P, float scale,
                       float* out, hipStream_t st) {
  const long total = (long)B * Q * M * (D / VEC);
  if (total == 0) return MDQE_OK;
  const int block = 256;
  long nb = (total + block - 1) / block;
  if (nb > 256L * 64) nb = 256L * 64;          // grid-stride beyond 64 blocks per CU
  if (L == 4 && P == 4)
    hipLaunchKernelGGL((msda_fwd_kernel<VEC, 4, 4>), dim3((unsigned)nb), dim3(block), 0, st, value, shapes, level_start,
                       loc, attn, B, S, M, D, G, L, Q, P, scale, out, total);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<VEC, 0, 0>), dim3((unsigned)nb), dim3(block), 0, st, value, shapes, level_start,
                       loc, attn, B, S, M, D, G, L, Q, P, scale, out, total);
  return mdqe_launch_status();
}

extern "C" int mdqe_msda_forward_grouped_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                             const float* loc, const float* attn, int B, int S, int M, int D, int G,
                                             int L, int Q, int P, float scale, float* out, void* stream) {
  MDQE_REQUIRE(B >= 0 && S >= 0 && M > 0 && D > 0 && G > 0 && L > 0 && Q >= 0 && P > 0);
  if ((long)B * Q == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(shapes); MDQE_CHECK_PTR(level_start);
  MDQE_CHECK_PTR(loc); MDQE_CHECK_PTR(attn); MDQE_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  const bool al16 = (((uintptr_t)value | (uintptr_t)out) & 15) == 0;
  const bool al8 = (((uintptr_t)value | (uintptr_t)out) & 7) == 0;
  if (D % 4 == 0 && al16) return launch_msda<4>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
  if (D % 2 == 0 && al8) return launch_msda<2>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
  return launch_msda<1>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
}

extern "C" int mdqe_msda_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                     const float* loc, const float* attn, int B, int S, int M, int D, int L, int Q,
                                     int P, float* out, void* stream) {
  return mdqe_msda_forward_grouped_f32(value, shapes, level_start, loc, attn, B, S, M, D, 1, L, Q, P, 1.0f, out, stream);
}
