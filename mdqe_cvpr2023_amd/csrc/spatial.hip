// Small NHWC spatial kernels of the backbone stem and the mask-feature head (HBM/L2-streaming,
// 16-B lane accesses, one lane = 4 channels of one output pixel).
#include "common.h"

// ---- a1+a2+stem: normalise + zero-pad + 7x7/s2 im2col --------------------------------------------
// frames: NI images, CHW (uint8 or fp32), each h x w (un-padded); padded canvas Hp x Wp (multiple of
// 32, mdqe/mdqe.py:65,318).  x = (frame - mean)/std (mdqe/mdqe.py:176-178,480); padding is zero in
// normalised space.  Output rows m=(img,oh,ow), K = 7*7*3 = 147 padded to 160 floats, k = (kh,kw,c).
template <typename T>
__global__ void __launch_bounds__(256)
stem_im2col_kernel(const T* __restrict__ frames, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                   float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out, int OH, int OW) {
  const long total = (long)NI * OH * OW * 40;          // 40 float4 per row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q4 = (int)(i % 40);
    const long m = i / 40;
    const int ow = (int)(m % OW); const long t = m / OW; const int oh = (int)(t % OH); const int img = (int)(t / OH);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = q4 * 4 + e;
      float val = 0.f;
      if (k < 147) {
        const int c = k % 3, tap = k / 3, kw = tap % 7, kh = tap / 7;
        const int ih = oh * 2 - 3 + kh, iw = ow * 2 - 3 + kw;
        if (ih >= 0 && ih < h && iw >= 0 && iw < w) {           // inside the real image (padding area is 0)
          const float raw = (float)frames[(long)img * frame_stride + ((long)c * h + ih) * w + iw];
          const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
          const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
          val = (raw - mean) / sd;
        }
      }
      v[e] = val;
    }
    *reinterpret_cast<f32x4*>(out + m * 160 + q4 * 4) = v;
  }
}

extern "C" int mdqe_stem_im2col_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                                    const float* mean3_host, const float* std3_host, float* out, void* stream) {
  MDQE_REQUIRE(NI >= 0 && h > 0 && w > 0 && Hp >= h && Wp >= w && Hp % 2 == 0 && Wp % 2 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(frames); MDQE_CHECK_PTR(out); MDQE_CHECK_PTR(mean3_host); MDQE_CHECK_PTR(std3_host);
  mdqe_clear_error();
  const int OH = Hp / 2, OW = Wp / 2;
  const long total = (long)NI * OH * OW * 40;
  long nb = (total + 255) / 256; if (nb > 256 * 16) nb = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
  if (is_u8)
    hipLaunchKernelGGL((stem_im2col_kernel<unsigned char>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned char*)frames,
                       frame_stride, NI, h, w, Hp, Wp, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                       std3_host[2], out, OH, OW);
  else
    hipLaunchKernelGGL((stem_im2col_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)frames, frame_stride, NI,
                       h, w, Hp, Wp, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], out,
                       OH, OW);
  return mdqe_launch_status();
}

// ---- 3x3 / stride 2 / pad 1 max pool, NHWC (ResNet stem) ------------------------------------------
__global__ void __launch_bounds__(256)
maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int NI, int H, int W, int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)NI * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n); long t = i / c4n;
    const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int img = (int)(t / OH);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)img * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(y + (((long)img * OH + oh) * OW + ow) * C + c4 * 4) = m;
  }
}

extern "C" int mdqe_maxpool3x3s2_nhwc_f32(const float* x, float* y, int NI, int H, int W, int C, void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(y);
  mdqe_clear_error();
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)NI * OH * OW * (C / 4);
  long nb = (total + 255) / 256; if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, NI, H, W, C, OH, OW);
  return mdqe_launch_status();
}

// ---- y = a + nearest_upsample(b) (MaskHead FPN merge, segmentation.py:47-55) -----------------------
// a,y: [NI,H,W,C]; b: [NI,Hb,Wb,C]; src index = min(floor(dst * (float)Hb/H), Hb-1) (F.interpolate 'nearest').
__global__ void __launch_bounds__(256)
upsample_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int NI, int H, int W,
                    int Hb, int Wb, int C) {
  const int c4n = C / 4;
  const float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
  const long total = (long)NI * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n); long t = i / c4n;
    const int ww = (int)(t % W); t /= W; const int hh = (int)(t % H); const int img = (int)(t / H);
    const int hb = min((int)floorf(hh * sh), Hb - 1), wb = min((int)floorf(ww * sw), Wb - 1);
    const f32x4 va = *reinterpret_cast<const f32x4*>(a + (((long)img * H + hh) * W + ww) * C + c4 * 4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(b + (((long)img * Hb + hb) * Wb + wb) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(y + (((long)img * H + hh) * W + ww) * C + c4 * 4) = va + vb;
  }
}

extern "C" int mdqe_upsample_nearest_add_nhwc_f32(const float* a, const float* b, float* y, int NI, int H, int W, int Hb,
                                                  int Wb, int C, void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && W > 0 && Hb > 0 && Wb > 0 && C > 0 && C % 4 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(a); MDQE_CHECK_PTR(b); MDQE_CHECK_PTR(y);
  mdqe_clear_error();
  const long total = (long)NI * H * W * (C / 4);
  long nb = (total + 255) / 256; if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, b, y, NI, H, W, Hb, Wb, C);
  return mdqe_launch_status();
}

// ---- depthwise KxK conv, stride 1, pad K/2, NHWC; weights [K*K][C] (tap-major), bias [C] -----------
// (DepthwiseSeparableConv2d.depthwise, segmentation.py:92-98,112).
// `up2`: the input is the *virtual* output of the depthwise ConvTranspose2d(k=1,s=2,output_padding=1)
// (segmentation.py:28-29,59): up[n,2i,2j,c] = x[n,i,j,c]*tw[c] + tb[c], every other position = tb[c];
// it is generated on the fly so the 2x-upsampled tensor is never written to HBM.
template <int K>
__global__ void __launch_bounds__(256)
dwconv_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ y,
              int NI, int H, int W, int C, int up2, const float* __restrict__ tw, const float* __restrict__ tb) {
  // H, W are the OUTPUT (= virtual input) sizes; when up2 the stored input is (H/2) x (W/2)
  const int c4n = C / 4;
  const long total = (long)NI * H * W * c4n;
  const int Hs = up2 ? H / 2 : H, Ws = up2 ? W / 2 : W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n); long t = i / c4n;
    const int ww = (int)(t % W); t /= W; const int hh = (int)(t % H); const int img = (int)(t / H);
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    f32x4 twv = {0, 0, 0, 0}, tbv = {0, 0, 0, 0};
    if (up2) { twv = *reinterpret_cast<const f32x4*>(tw + c4 * 4); tbv = *reinterpret_cast<const f32x4*>(tb + c4 * 4); }
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int ih = hh - K / 2 + kh;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int iw = ww - K / 2 + kw;
        if (iw < 0 || iw >= W) continue;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + (long)(kh * K + kw) * C + c4 * 4);
        f32x4 v;
        if (up2) {
          if (((ih | iw) & 1) == 0)
            v = *reinterpret_cast<const f32x4*>(x + (((long)img * Hs + (ih >> 1)) * Ws + (iw >> 1)) * C + c4 * 4) * twv + tbv;
          else
            v = tbv;
        } else {
          v = *reinterpret_cast<const f32x4*>(x + (((long)img * Hs + ih) * Ws + iw) * C + c4 * 4);
        }
        acc += v * wv;
      }
    }
    *reinterpret_cast<f32x4*>(y + (((long)img * H + hh) * W + ww) * C + c4 * 4) = acc;
  }
}

extern "C" int mdqe_dwconv5x5_nhwc_f32(const float* x, const float* wt, const float* bias, float* y, int NI, int H, int W,
                                       int C, int up2, const float* tw, const float* tb, void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  MDQE_REQUIRE(!up2 || (H % 2 == 0 && W % 2 == 0));
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(wt); MDQE_CHECK_PTR(bias); MDQE_CHECK_PTR(y);
  if (up2) { MDQE_CHECK_PTR(tw); MDQE_CHECK_PTR(tb); }
  mdqe_clear_error();
  const long total = (long)NI * H * W * (C / 4);
  long nb = (total + 255) / 256; if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL((dwconv_kernel<5>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, NI, H, W, C, up2,
                     tw, tb);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Eval-time frame resize on the device: ResizeShortestEdgeClip -> ResizeTransform -> for uint8 frames
// PIL Image.resize(BILINEAR) (mdqe/data/augmentation.py:364-389, mdqe/data/dataset_mapper.py:252-258).  Pillow's separable
// resampling, bit for bit: horizontal pass first, each pass = sum of uint8 taps x 22-bit fixed-point coefficients starting at
// 1 << 21, >> 22, clipped to uint8; the coefficient tables (triangle filter stretched by max(scale, 1), normalised in double
// precision, quantised) are built on the host.  One thread per output value; the horizontal pass is evaluated on the fly for
// the few source rows a vertical tap window needs (no intermediate image).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
resize_pil_bilinear_kernel(const unsigned char* __restrict__ in, long in_img_stride, int C, int H, int W, int oh, int ow,
                           const int* __restrict__ xmin, const int* __restrict__ xcnt, const int* __restrict__ xk, int kxs,
                           const int* __restrict__ ymin, const int* __restrict__ ycnt, const int* __restrict__ yk, int kys,
                           unsigned char* __restrict__ out, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % ow); long t = i / ow;
    const int Y = (int)(t % oh); t /= oh;
    const int c = (int)(t % C); const long img = t / C;
    const unsigned char* src = in + img * in_img_stride + (long)c * H * W;
    const int x0 = xmin[X], nx = xcnt[X], y0 = ymin[Y], ny = ycnt[Y];
    const int* kx = xk + (long)X * kxs;
    const int* ky = yk + (long)Y * kys;
    int acc = 1 << 21;
    for (int ty = 0; ty < ny; ++ty) {
      const unsigned char* row = src + (long)(y0 + ty) * W + x0;
      int hv;
      if (ow == W) {
        hv = row[X - x0];                            // no horizontal pass when the width is unchanged (Pillow skips it)
      } else {
        int hs = 1 << 21;
        for (int tx = 0; tx < nx; ++tx) hs += (int)row[tx] * kx[tx];
        hv = min(max(hs >> 22, 0), 255);
      }
      if (oh == H) { acc = hv; break; }              // no vertical pass when the height is unchanged
      acc += hv * ky[ty];
    }
    out[i] = (unsigned char)(oh == H ? acc : min(max(acc >> 22, 0), 255));
  }
}

extern "C" int mdqe_resize_pil_bilinear_u8(const unsigned char* in, long in_img_stride, int NI, int C, int H, int W, int oh, int ow,
                                           const int* xmin, const int* xcnt, const int* xk, int kxs, const int* ymin,
                                           const int* ycnt, const int* yk, int kys, unsigned char* out, void* stream) {
  MDQE_REQUIRE(NI >= 0 && C > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && kxs > 0 && kys > 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(in); MDQE_CHECK_PTR(out); MDQE_CHECK_PTR(xmin); MDQE_CHECK_PTR(xcnt); MDQE_CHECK_PTR(xk);
  MDQE_CHECK_PTR(ymin); MDQE_CHECK_PTR(ycnt); MDQE_CHECK_PTR(yk);
  mdqe_clear_error();
  const long total = (long)NI * C * oh * ow;
  long nb = (total + 255) / 256; if (nb > 256 * 64) nb = 256 * 64;
  hipLaunchKernelGGL(resize_pil_bilinear_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, in_img_stride, C, H, W, oh,
                     ow, xmin, xcnt, xk, kxs, ymin, ycnt, yk, kys, out, total);
  return mdqe_launch_status();
}
