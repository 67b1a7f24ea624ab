// Depthwise 5x5 kernels of the mask-feature head for C == 256 (one wave = the 64 float4 channel groups of a pixel, so a
// lane's 25 filter taps live in registers for the whole kernel and every bounds decision is wave-uniform):
//
//  * dw5_c256_kernel: plain depthwise 5x5 (DepthwiseSeparableConv2d.depthwise, segmentation.py:92-98,112), two
//    horizontally adjacent outputs per step (30 loads for 2 outputs instead of 50).
//  * dw5_up2_c256_kernel: ConvTranspose2d(k=1, s=2, output_padding=1, groups=C) -> depthwise 5x5 (segmentation.py:28-29,59)
//    in one pass.  The upsampled tensor is virtual: up[2i,2j] = x[i,j]*tw + tb, every other position = tb.  A 2x2 output
//    quad (2a+dy, 2b+dx) touches only x[a-1..a+1][b-1..b+1] and uses each of the 25 taps exactly once:
//    out = bias + tb * (sum of the in-bounds taps) + tw * (sum over the even-even taps of w * x)
//    -- 9 loads and 25 float4 FMAs per quad instead of 100 tap visits.
#include "common.h"

namespace {
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
}

__global__ void __launch_bounds__(256, 2)
dw5_c256_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ y,
                int NI, int H, int W) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  f32x4 w[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) w[t] = ld4(wt + t * C + lane * 4);
  const f32x4 bv = ld4(bias + lane * 4);
  const int Wp = (W + 1) / 2;                         // output pairs per row
  const long total = (long)NI * H * Wp;
  for (long i = wid; i < total; i += nw) {
    const int pw_ = (int)(i % Wp); const long t = i / Wp;
    const int hh = (int)(t % H), img = (int)(t / H);
    const int w0 = pw_ * 2;
    f32x4 a0 = bv, a1 = bv;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int ih = hh - 2 + kh;
      if (ih < 0 || ih >= H) continue;               // wave-uniform
      const float* row = x + (((long)img * H + ih) * W) * C + lane * 4;
      f32x4 X[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int iw = w0 - 2 + c;
        X[c] = (iw >= 0 && iw < W) ? ld4(row + (long)iw * C) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) { a0 += X[kw] * w[kh * 5 + kw]; a1 += X[kw + 1] * w[kh * 5 + kw]; }
    }
    float* o = y + (((long)img * H + hh) * W + w0) * C + lane * 4;
    *reinterpret_cast<f32x4*>(o) = a0;
    if (w0 + 1 < W) *reinterpret_cast<f32x4*>(o + C) = a1;
  }
}

__global__ void __launch_bounds__(256, 2)
dw5_up2_c256_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                    const float* __restrict__ tw, const float* __restrict__ tb, float* __restrict__ y, int NI, int Hs, int Ws) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  const int H = 2 * Hs, W = 2 * Ws;
  f32x4 w[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) w[t] = ld4(wt + t * C + lane * 4);
  const f32x4 bv = ld4(bias + lane * 4), twv = ld4(tw + lane * 4), tbv = ld4(tb + lane * 4);
  f32x4 wall = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 25; ++t) wall += w[t];
  const f32x4 cin = bv + tbv * wall;                  // constant part of an interior output

  const long total = (long)NI * Hs * Ws;
  for (long i = wid; i < total; i += nw) {
    const int b = (int)(i % Ws); const long t = i / Ws;
    const int a = (int)(t % Hs), img = (int)(t / Hs);
    f32x4 o00 = {0.f, 0.f, 0.f, 0.f}, o01 = o00, o10 = o00, o11 = o00;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ia = a - 1 + r;
      if (ia < 0 || ia >= Hs) continue;              // wave-uniform
      const float* row = x + (((long)img * Hs + ia) * Ws) * C + lane * 4;
      f32x4 X[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int ib = b - 1 + c;
        X[c] = (ib >= 0 && ib < Ws) ? ld4(row + (long)ib * C) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // output (dy,dx) of the quad uses taps kh = 2r - dy, kw = 2c - dx
#pragma unroll
      for (int c = 0; c < 3; ++c) o00 += X[c] * w[(2 * r) * 5 + 2 * c];
#pragma unroll
      for (int c = 1; c < 3; ++c) o01 += X[c] * w[(2 * r) * 5 + 2 * c - 1];
      if (r >= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o10 += X[c] * w[(2 * r - 1) * 5 + 2 * c];
#pragma unroll
        for (int c = 1; c < 3; ++c) o11 += X[c] * w[(2 * r - 1) * 5 + 2 * c - 1];
      }
    }
    f32x4 c00 = cin, c01 = cin, c10 = cin, c11 = cin;
    if (a == 0 || a == Hs - 1 || b == 0 || b == Ws - 1) {          // border quad (wave-uniform): only in-bounds taps count
      auto cst = [&](int hh, int ww) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
          const int ih = hh - 2 + kh;
          if (ih < 0 || ih >= H) continue;
#pragma unroll
          for (int kw = 0; kw < 5; ++kw) {
            const int iw = ww - 2 + kw;
            if (iw >= 0 && iw < W) s += w[kh * 5 + kw];
          }
        }
        return bv + tbv * s;
      };
      c00 = cst(2 * a, 2 * b); c01 = cst(2 * a, 2 * b + 1); c10 = cst(2 * a + 1, 2 * b); c11 = cst(2 * a + 1, 2 * b + 1);
    }
    float* o = y + (((long)img * H + 2 * a) * W + 2 * b) * C + lane * 4;
    *reinterpret_cast<f32x4*>(o) = c00 + twv * o00;
    *reinterpret_cast<f32x4*>(o + C) = c01 + twv * o01;
    *reinterpret_cast<f32x4*>(o + (long)W * C) = c10 + twv * o10;
    *reinterpret_cast<f32x4*>(o + (long)W * C + C) = c11 + twv * o11;
  }
}

extern "C" int mdqe_dwconv5x5_c256_f32(const float* x, const float* wt, const float* bias, float* y, int NI, int H, int W,
                                       void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && W > 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(wt); MDQE_CHECK_PTR(bias); MDQE_CHECK_PTR(y);
  mdqe_clear_error();
  const long total = (long)NI * H * ((W + 1) / 2);
  long nb = (total + 3) / 4; if (nb > 256 * 8) nb = 256 * 8;
  hipLaunchKernelGGL(dw5_c256_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, NI, H, W);
  return mdqe_launch_status();
}

// x [NI,Hs,Ws,256] -> y [NI,2Hs,2Ws,256] = depthwise5x5(transposed-conv-x2(x)); wt [25,256], bias/tw/tb [256]
extern "C" int mdqe_dwconv5x5_up2_c256_f32(const float* x, const float* wt, const float* bias, const float* tw, const float* tb,
                                           float* y, int NI, int Hs, int Ws, void* stream) {
  MDQE_REQUIRE(NI >= 0 && Hs > 0 && Ws > 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(wt); MDQE_CHECK_PTR(bias); MDQE_CHECK_PTR(tw); MDQE_CHECK_PTR(tb); MDQE_CHECK_PTR(y);
  mdqe_clear_error();
  const long total = (long)NI * Hs * Ws;
  long nb = (total + 3) / 4; if (nb > 256 * 8) nb = 256 * 8;
  hipLaunchKernelGGL(dw5_up2_c256_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, wt, bias, tw, tb, y, NI, Hs, Ws);
  return mdqe_launch_status();
}
