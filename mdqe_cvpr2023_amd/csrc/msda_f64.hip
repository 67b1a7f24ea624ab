// Multi-scale deformable attention in float64 -- the other half of the reference's dtype dispatch
// (AT_DISPATCH_FLOATING_TYPES: mdqe/models/ops/src/cuda/ms_deform_attn_cuda.cu:64 forward, :134 backward).  The eval path runs
// fp32 (callers cast, functions/ms_deform_attn_func.py:24); float64 is what the reference's own test script exercises
// (mdqe/models/ops/test.py:32-44 forward, :63-86 gradcheck), so the drop-in has to take it.
//
// Arithmetic as ms_deform_im2col_cuda.cuh:33-84 (bilinear), :237-299 (forward), :87-234 + :301-920 (backward) with scalar_t = double:
// pixel = loc*size - 0.5; a sample contributes only if -1 < h < H and -1 < w < W; corners outside the map are zero.
// Mapping: forward -- one lane per output channel (b, q, m, d), samples walked in (l, p) order, so the sum order is the reference's;
// backward -- one lane per sample (b, q, m, l, p) walking the head's channels, the three d-reductions in registers, grad_value by
// hardware double atomics (global_atomic_add_f64).  Simple by intent: not on the eval hot path.
#include "common.h"

__global__ void __launch_bounds__(256)
msda_fwd_f64_kernel(const double* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ level_start,
                    const double* __restrict__ loc, const double* __restrict__ attn, int S, int M, int D, int L, int Q, int P,
                    double* __restrict__ out, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D);
    long t = idx / D;
    const int m = (int)(t % M);
    t /= M;                                   // t = b*Q + q
    const long b = t / Q;
    const long samp = (t * M + m) * (long)(L * P);
    const long rs = (long)M * D;
    const double* vb = value + b * S * rs + (long)m * D + d;
    double acc = 0.0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[l * 2], W = (int)shapes[l * 2 + 1];
      const double* vl = vb + level_start[l] * rs;
      for (int p = 0; p < P; ++p) {
        const double lx = loc[(samp + l * P + p) * 2], ly = loc[(samp + l * P + p) * 2 + 1];
        const double aw = attn[samp + l * P + p];
        const double h_im = ly * H - 0.5, w_im = lx * W - 0.5;
        if (h_im > -1.0 && w_im > -1.0 && h_im < (double)H && w_im < (double)W) {
          const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
          const double lh = h_im - h_low, lw = w_im - w_low;
          const double hh = 1.0 - lh, hw = 1.0 - lw;
          const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
          const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
          const double* p00 = vl + ((long)h_low * W + w_low) * rs;
          const double v1 = (h0 && w0) ? p00[0] : 0.0;
          const double v2 = (h0 && w1) ? p00[rs] : 0.0;
          const double v3 = (h1 && w0) ? p00[(long)W * rs] : 0.0;
          const double v4 = (h1 && w1) ? p00[(long)W * rs + rs] : 0.0;
          acc += ((hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4) * aw;
        }
      }
    }
    out[idx] = acc;
  }
}

__global__ void __launch_bounds__(256)
msda_bwd_f64_kernel(const double* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ level_start,
                    const double* __restrict__ loc, const double* __restrict__ attn, const double* __restrict__ gout,
                    int S, int M, int D, int L, int Q, int P, double* __restrict__ gvalue, double* __restrict__ gloc,
                    double* __restrict__ gattn, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx / P;                         // idx = sample (b,q,m,l,p)
    const int l = (int)(t % L);
    t /= L;                                   // t = (b*Q + q)*M + m
    const int m = (int)(t % M);
    const long bq = t / M;
    const long b = bq / Q;
    const int H = (int)shapes[l * 2], W = (int)shapes[l * 2 + 1];
    const double lx = loc[idx * 2], ly = loc[idx * 2 + 1];
    const double aw = attn[idx];
    const double h_im = ly * H - 0.5, w_im = lx * W - 0.5;
    double ga = 0.0, gx = 0.0, gy = 0.0;
    if (h_im > -1.0 && w_im > -1.0 && h_im < (double)H && w_im < (double)W) {
      const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
      const double lh = h_im - h_low, lw = w_im - w_low;
      const double hh = 1.0 - lh, hw = 1.0 - lw;
      const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
      const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
      const long rs = (long)M * D;
      const long base = (b * S + level_start[l]) * rs + (long)m * D;
      const long o00 = base + ((long)h_low * W + w_low) * rs;
      const long o01 = o00 + rs, o10 = o00 + (long)W * rs, o11 = o10 + rs;
      const double* g = gout + bq * rs + (long)m * D;
      const double w1c = hh * hw * aw, w2c = hh * lw * aw, w3c = lh * hw * aw, w4c = lh * lw * aw;
      for (int d = 0; d < D; ++d) {
        const double gd = g[d];
        const double v1 = (h0 && w0) ? value[o00 + d] : 0.0;
        const double v2 = (h0 && w1) ? value[o01 + d] : 0.0;
        const double v3 = (h1 && w0) ? value[o10 + d] : 0.0;
        const double v4 = (h1 && w1) ? value[o11 + d] : 0.0;
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
    gloc[idx * 2] = (double)W * aw * gx;
    gloc[idx * 2 + 1] = (double)H * aw * gy;
  }
}

extern "C" int mdqe_msda_forward_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                                     const double* attn, int B, int S, int M, int D, int L, int Q, int P, double* out, void* stream) {
  MDQE_REQUIRE(B >= 0 && S >= 0 && M > 0 && D > 0 && L > 0 && Q >= 0 && P > 0);
  const long total = (long)B * Q * M * D;
  if (total == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(shapes); MDQE_CHECK_PTR(level_start); MDQE_CHECK_PTR(loc); MDQE_CHECK_PTR(attn); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  long nb = (total + 255) / 256; if (nb > 256L * 64) nb = 256L * 64;
  hipLaunchKernelGGL(msda_fwd_f64_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, value, shapes, level_start, loc, attn,
                     S, M, D, L, Q, P, out, total);
  return mdqe_launch_status();
}

extern "C" int mdqe_msda_backward_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                                      const double* attn, const double* grad_out, int B, int S, int M, int D, int L, int Q, int P,
                                      double* grad_value, double* grad_loc, double* grad_attn, void* stream) {
  MDQE_REQUIRE(B >= 0 && S >= 0 && M > 0 && D > 0 && L > 0 && Q >= 0 && P > 0);
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  if ((long)B * S > 0) MDQE_CHECK_PTR(grad_value);      // empty tensors (B == 0) come with NULL pointers: accepted, as the reference does
  if ((long)B * S > 0 && hipMemsetAsync(grad_value, 0, (size_t)B * S * M * D * sizeof(double), st) != hipSuccess) return MDQE_ELAUNCH;
  const long total = (long)B * Q * M * L * P;
  if (total == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(shapes); MDQE_CHECK_PTR(level_start); MDQE_CHECK_PTR(loc); MDQE_CHECK_PTR(attn);
  MDQE_CHECK_PTR(grad_out); MDQE_CHECK_PTR(grad_loc); MDQE_CHECK_PTR(grad_attn);
  long nb = (total + 255) / 256; if (nb > 256L * 64) nb = 256L * 64;
  hipLaunchKernelGGL(msda_bwd_f64_kernel, dim3((unsigned)nb), dim3(256), 0, st, value, shapes, level_start, loc, attn, grad_out,
                     S, M, D, L, Q, P, grad_value, grad_loc, grad_attn, total);
  return mdqe_launch_status();
}
