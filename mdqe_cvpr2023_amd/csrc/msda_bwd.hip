// Multi-scale deformable attention, backward (the other half of the reference's native op:
// ms_deform_attn_backward, mdqe/models/ops/src/vision.cpp:15, src/cuda/ms_deform_attn_cuda.cu:83-153, kernels
// ms_deformable_col2im_gpu_kernel_* and ms_deform_attn_col2im_bilinear, src/cuda/ms_deform_im2col_cuda.cuh:87-234, 301-920).
//
//   grad_value[b, pix, m, :] += w_corner * attn * g          (four corners of every in-range sample; atomics)
//   grad_attn[b,q,m,l,p]      = sum_d g[d] * bilinear(value)[d]
//   grad_loc[b,q,m,l,p,(x,y)] = (W, H) * attn * sum_d g[d] * d bilinear / d (w, h)
// with g = grad_output[b,q,m,:], pixel = loc*size - 0.5, a sample contributing only if -1 < h < H and -1 < w < W.
//
// Mapping: one lane owns one sample (b,q,m,l,p) and walks the head's D channels in float4 steps: the three
// reductions over d stay in registers (the reference reduces across a block through shared memory, one thread per
// channel), every corner read is a 16-B load of a line the L2 holds, and grad_value takes hardware float atomics
// (global_atomic_add_f32, no CAS loop).  Not on the eval hot path; kept simple.
#include "common.h"

__global__ void __launch_bounds__(256)
msda_bwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ level_start,
                const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ gout,
                int B, int S, int M, int D, int L, int Q, int P, float* __restrict__ gvalue, float* __restrict__ gloc,
                float* __restrict__ gattn, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx / P;                         // idx = sample (b,q,m,l,p)
    const int l = (int)(t % L);
    t /= L;                                   // t = (b*Q + q)*M + m
    const int m = (int)(t % M);
    const long bq = t / M;
    const int b = (int)(bq / Q);
    const int H = (int)shapes[l * 2], W = (int)shapes[l * 2 + 1];
    const float lx = loc[idx * 2], ly = loc[idx * 2 + 1];
    const float aw = attn[idx];
    const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
    float ga = 0.f, gx = 0.f, gy = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - h_low, lw = w_im - w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
      const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
      const int rs = M * D;
      const long base = ((long)b * S + level_start[l]) * rs + (long)m * D;
      const long o00 = base + ((long)h_low * W + w_low) * rs;
      const long o01 = o00 + rs, o10 = o00 + (long)W * rs, o11 = o10 + rs;
      const float* g = gout + bq * rs + (long)m * D;
      const float w1c = hh * hw * aw, w2c = hh * lw * aw, w3c = lh * hw * aw, w4c = lh * lw * aw;
      for (int d = 0; d < D; ++d) {
        const float gd = g[d];
        const float v1 = (h0 && w0) ? value[o00 + d] : 0.f;
        const float v2 = (h0 && w1) ? value[o01 + d] : 0.f;
        const float v3 = (h1 && w0) ? value[o10 + d] : 0.f;
        const float v4 = (h1 && w1) ? value[o11 + d] : 0.f;
        ga += gd * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        gy += gd * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);          // d/dh
        gx += gd * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);          // d/dw
        if (h0 && w0) unsafeAtomicAdd(gvalue + o00 + d, w1c * gd);
        if (h0 && w1) unsafeAtomicAdd(gvalue + o01 + d, w2c * gd);
        if (h1 && w0) unsafeAtomicAdd(gvalue + o10 + d, w3c * gd);
        if (h1 && w1) unsafeAtomicAdd(gvalue + o11 + d, w4c * gd);
      }
    }
    gattn[idx] = ga;
    gloc[idx * 2] = (float)W * aw * gx;
    gloc[idx * 2 + 1] = (float)H * aw * gy;
  }
}

extern "C" int mdqe_msda_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                                      const float* attn, const float* grad_out, int B, int S, int M, int D, int L, int Q,
                                      int P, float* grad_value, float* grad_loc, float* grad_attn, void* stream) {
  MDQE_REQUIRE(B >= 0 && S >= 0 && M > 0 && D > 0 && L > 0 && Q >= 0 && P > 0);
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  if ((long)B * S > 0) MDQE_CHECK_PTR(grad_value);      // empty tensors (B == 0) come with NULL pointers: accepted, as the reference does
  if ((long)B * S > 0 && hipMemsetAsync(grad_value, 0, (size_t)B * S * M * D * sizeof(float), st) != hipSuccess) return MDQE_ELAUNCH;
  const long total = (long)B * Q * M * L * P;
  if (total == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(shapes); MDQE_CHECK_PTR(level_start); MDQE_CHECK_PTR(loc); MDQE_CHECK_PTR(attn);
  MDQE_CHECK_PTR(grad_out); MDQE_CHECK_PTR(grad_loc); MDQE_CHECK_PTR(grad_attn);
  long nb = (total + 255) / 256; if (nb > 256L * 64) nb = 256L * 64;
  hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, value, shapes, level_start, loc, attn, grad_out,
                     B, S, M, D, L, Q, P, grad_value, grad_loc, grad_attn, total);
  return mdqe_launch_status();
}
