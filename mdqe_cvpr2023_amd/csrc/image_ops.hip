// COCO single-image branch, post-decoder stage (MDQE.inference_image, mdqe/mdqe.py:486-556) on the low-resolution
// mask logits [n, Hm, Wm] of the centre frame.  aligned_bilinear (util/misc.py:485-507) is evaluated in closed form --
// full-resolution pixel p reads the source at max(p - f/2, 0)/f, clamped to the map -- so the x`factor` maps are never
// materialised:
//   image_mask_stats : per mask, over the crop [:h,:w] of the up-sampled logits: sum(sigmoid * [sigmoid > 0.5]),
//                      count(sigmoid > 0.5) (mask-quality score, :512-516) and the tight box of (logit > 0) (:526)
//   image_final_masks: bilinear resize (align_corners=False, torchvision/F.interpolate semantics) of the cropped
//                      up-sampled logits to (Ho, Wo), > 0  (:545-547)
#include "common.h"

__device__ __forceinline__ float ab_sample(const float* __restrict__ m, int Hm, int Wm, int factor, int Y, int X) {
  const float fy = (float)max(Y - factor / 2, 0) / (float)factor, fx = (float)max(X - factor / 2, 0) / (float)factor;
  const int y0 = min((int)fy, Hm - 1), x0 = min((int)fx, Wm - 1);
  const int y1 = min(y0 + 1, Hm - 1), x1 = min(x0 + 1, Wm - 1);
  const float ly = fy - y0, lx = fx - x0;
  const float top = m[y0 * Wm + x0] * (1.f - lx) + m[y0 * Wm + x1] * lx;
  const float bot = m[y1 * Wm + x0] * (1.f - lx) + m[y1 * Wm + x1] * lx;
  return top * (1.f - ly) + bot * ly;
}

// one block per mask; out[k] = {num, den, xmin, ymin, xmax, ymax} (box of logit > 0; xmin > xmax when empty)
__global__ void __launch_bounds__(256)
image_mask_stats_kernel(const float* __restrict__ lg, int Hm, int Wm, int factor, int h, int w, float* __restrict__ out) {
  const float* m = lg + (long)blockIdx.x * Hm * Wm;
  float num = 0.f, den = 0.f;
  int x0 = w, y0 = h, x1 = -1, y1 = -1;
  for (int i = threadIdx.x; i < h * w; i += 256) {
    const int Y = i / w, X = i - Y * w;
    const float v = ab_sample(m, Hm, Wm, factor, Y, X);
    const float p = 1.0f / (1.0f + expf(-v));
    if (p > 0.5f) { num += p; den += 1.f; }
    if (v > 0.f) { x0 = min(x0, X); y0 = min(y0, Y); x1 = max(x1, X); y1 = max(y1, Y); }
  }
  __shared__ float sf[2][256];
  __shared__ int si[4][256];
  sf[0][threadIdx.x] = num; sf[1][threadIdx.x] = den;
  si[0][threadIdx.x] = x0; si[1][threadIdx.x] = y0; si[2][threadIdx.x] = x1; si[3][threadIdx.x] = y1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sf[0][threadIdx.x] += sf[0][threadIdx.x + s]; sf[1][threadIdx.x] += sf[1][threadIdx.x + s];
      si[0][threadIdx.x] = min(si[0][threadIdx.x], si[0][threadIdx.x + s]); si[1][threadIdx.x] = min(si[1][threadIdx.x], si[1][threadIdx.x + s]);
      si[2][threadIdx.x] = max(si[2][threadIdx.x], si[2][threadIdx.x + s]); si[3][threadIdx.x] = max(si[3][threadIdx.x], si[3][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* o = out + (long)blockIdx.x * 6;
    o[0] = sf[0][0]; o[1] = sf[1][0];
    o[2] = (float)si[0][0]; o[3] = (float)si[1][0]; o[4] = (float)si[2][0]; o[5] = (float)si[3][0];
  }
}

__global__ void __launch_bounds__(256)
image_final_masks_kernel(const float* __restrict__ lg, const int* __restrict__ idx, int Hm, int Wm, int factor, int h, int w,
                         int Ho, int Wo, unsigned char* __restrict__ out, long total) {
  const float sy = (float)h / (float)Ho, sx = (float)w / (float)Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % Wo); long t = i / Wo;
    const int Y = (int)(t % Ho); const int k = (int)(t / Ho);
    const float* m = lg + (long)idx[k] * Hm * Wm;
    // F.interpolate(mode="bilinear", align_corners=False): src = max((dst + 0.5) * scale - 0.5, 0)
    const float fy = fmaxf(((float)Y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)X + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float v00 = ab_sample(m, Hm, Wm, factor, y0, x0), v01 = ab_sample(m, Hm, Wm, factor, y0, x1);
    const float v10 = ab_sample(m, Hm, Wm, factor, y1, x0), v11 = ab_sample(m, Hm, Wm, factor, y1, x1);
    const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    out[i] = v > 0.f ? 1 : 0;
  }
}

extern "C" int mdqe_image_mask_stats_f32(const float* logits, int n, int Hm, int Wm, int factor, int h, int w, float* stats,
                                         void* stream) {
  MDQE_REQUIRE(n >= 0 && Hm > 0 && Wm > 0 && factor >= 1 && h > 0 && w > 0 && h <= Hm * factor && w <= Wm * factor);
  if (n == 0) return MDQE_OK;
  MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(stats);
  mdqe_clear_error();
  hipLaunchKernelGGL(image_mask_stats_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, Hm, Wm, factor, h, w, stats);
  return mdqe_launch_status();
}

extern "C" int mdqe_image_final_masks_u8(const float* logits, int n_sel, const int* idx_dev, int Hm, int Wm, int factor, int h,
                                         int w, int Ho, int Wo, unsigned char* out, void* stream) {
  MDQE_REQUIRE(n_sel >= 0 && Hm > 0 && Wm > 0 && factor >= 1 && h > 0 && w > 0 && Ho > 0 && Wo > 0);
  MDQE_REQUIRE(h <= Hm * factor && w <= Wm * factor);
  if (n_sel == 0) return MDQE_OK;
  MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(idx_dev); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  const long total = (long)n_sel * Ho * Wo;
  long nb = (total + 255) / 256; if (nb > 256 * 64) nb = 256 * 64;
  hipLaunchKernelGGL(image_final_masks_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, logits, idx_dev, Hm, Wm,
                     factor, h, w, Ho, Wo, out, total);
  return mdqe_launch_status();
}
