// Per-instance statistics of the dynamic mask logits in ONE pass over HBM (inference_clip,
// mdqe/mdqe.py:387-413): blank test, mask-quality numerator/denominator at full stride-4 resolution,
// and the half-resolution soft / hard maps + their sums that feed the soft-IoU NMS matrix.
#include "common.h"

__global__ void __launch_bounds__(256)
mask_row_stats_kernel(const float* __restrict__ x, int T, int H, int W, int t_step, float* __restrict__ stats,
                      float* __restrict__ soft_h, float* __restrict__ hard_h) {
  __shared__ float sh[5][4];
  const int r = blockIdx.x;
  const long P = (long)T * H * W;
  const int Hh = H / 2, Wh = W / 2;
  const int Th = (T + t_step - 1) / t_step;
  const long Ph = (long)Th * Hh * Wh;
  const float* xr = x + (long)r * P;
  float any = 0.f, qn = 0.f, qd = 0.f, ss = 0.f, hs = 0.f;
  for (long i = threadIdx.x; i < P; i += blockDim.x) {
    const float v = xr[i];
    const bool pos = v > 0.f;
    const float s = 1.0f / (1.0f + expf(-v));
    const bool hard = s > 0.5f;                       // the reference thresholds the sigmoid (mdqe.py:412)
    if (pos) any = 1.f;
    if (hard) { qn += s; qd += 1.f; }
    const int xx = (int)(i % W); const long t2 = i / W; const int yy = (int)(t2 % H); const int tt = (int)(t2 / H);
    if (((xx | yy) & 1) == 0 && (xx >> 1) < Wh && (yy >> 1) < Hh && (tt % t_step) == 0) {
      const long o = (long)r * Ph + ((long)(tt / t_step) * Hh + (yy >> 1)) * Wh + (xx >> 1);
      soft_h[o] = s; hard_h[o] = hard ? 1.f : 0.f;
      ss += s; hs += hard ? 1.f : 0.f;
    }
  }
  float v[5] = {any, qn, qd, ss, hs};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = (k == 0) ? fmaxf(a, __shfl_xor(a, o, 64)) : a + __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    float a = sh[k][0];
    for (int w = 1; w < 4; ++w) a = (k == 0) ? fmaxf(a, sh[k][w]) : a + sh[k][w];
    stats[(long)r * 5 + k] = a;
  }
}

extern "C" int mdqe_mask_row_stats_f32(const float* logits, int n, int T, int H, int W, int t_step, float* stats5,
                                       float* soft_h, float* hard_h, void* stream) {
  MDQE_REQUIRE(n >= 0 && T > 0 && H > 1 && W > 1 && (t_step == 1 || t_step == 2));
  if (n == 0) return MDQE_OK;
  MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(stats5); MDQE_CHECK_PTR(soft_h); MDQE_CHECK_PTR(hard_h);
  mdqe_clear_error();
  hipLaunchKernelGGL(mask_row_stats_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, T, H, W, t_step, stats5, soft_h,
                     hard_h);
  return mdqe_launch_status();
}
