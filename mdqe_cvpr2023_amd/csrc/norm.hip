// LayerNorm and NHWC GroupNorm for gfx950.  Both are HBM-streaming kernels: 16-B lane accesses,
// one wave per LayerNorm row, two deterministic passes (partials -> apply) for GroupNorm.
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The reduction tree of the LayerNorm EPILOGUE of the 64x256 GEMM tile (gemm_k16.hip: a lane's two 4-column chunks 32 columns apart,
// then the 8 lanes of a 32-column group, then the four 64-column wave slices as (w0 + w1) + (w2 + w3)), replayed on one wave whose
// lane i holds columns 4i .. 4i+3 of a 256-wide row: chunk i pairs with chunk i^8 first, then xor 1, 2, 4, then 16, 32.  With it
// `linear -> layernorm_kernel` and the fused kernel give the SAME BITS, so which of the two a launch takes (a question of how many
// rows there are to fill the chip, ops.linear_ln) never changes a result.
__device__ __forceinline__ float wave_sum_c256(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}

// y[r,:] = LN(x[r,:] + res[r,:]) * gamma + beta   (nn.LayerNorm eps=1e-5; transformer_enc.py:103-108,
// transformer_dec.py:345-358).  Optional second output y2 = y + add2[(r % add2_mod), :].
// SwinMap (ws > 0): x rows are in WINDOW order; `post` and `y` are the NHWC map [B, H, W, C] -- row r goes to its pixel (window reverse +
// un-shift + crop, swin_transformer_v2.py:273-287), rows of the zero padding are dropped: y[pix] = post[pix] + LN(x[r]).
struct SwinMap { int ws, shift, H, W; };

template <int NCH>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                 const float* __restrict__ beta, const float* __restrict__ post, float* __restrict__ y, long rows, int C, float eps,
                 SwinMap sm) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long r = wid; r < rows; r += nw) {
    long ro = r;                                 // row of `post` / `y`
    if (sm.ws > 0) {
      const int ws = sm.ws, nWx = (sm.W + ws - 1) / ws, nWy = (sm.H + ws - 1) / ws;
      const int ix = (int)(r % ws); long t = r / ws;
      const int iy = (int)(t % ws); t /= ws;
      const int wx = (int)(t % nWx); t /= nWx;
      const int wy = (int)(t % nWy); const long b = t / nWy;
      const int yy = (wy * ws + iy + sm.shift) % (nWy * ws), xx = (wx * ws + ix + sm.shift) % (nWx * ws);
      if (yy >= sm.H || xx >= sm.W) continue;    // (wave-uniform: one wave per row)
      ro = (b * sm.H + yy) * sm.W + xx;
    }
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < C) {
        v[i] = *reinterpret_cast<const f32x4*>(x + r * C + c);
        if (res != nullptr) v[i] += *reinterpret_cast<const f32x4*>(res + r * C + c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const bool tree256 = NCH == 1 && C == 256;
    const float mean = (tree256 ? wave_sum_c256(s) : wave_sum(s)) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < C) {
        const f32x4 d = v[i] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
    const float rstd = rsqrtf((tree256 ? wave_sum_c256(q) : wave_sum(q)) / C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < C) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o = (v[i] - mean) * rstd * g + b;
        if (post != nullptr) o += *reinterpret_cast<const f32x4*>(post + ro * C + c);
        *reinterpret_cast<f32x4*>(y + ro * C + c) = o;
      }
    }
  }
}

static int layernorm_impl(const float* x, const float* res, const float* gamma, const float* beta, const float* post, float* y,
                          long rows, int C, float eps, void* stream, SwinMap sm = SwinMap{0, 0, 0, 0}) {
  MDQE_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= 2048);
  if (rows == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(gamma); MDQE_CHECK_PTR(beta); MDQE_CHECK_PTR(y);
  mdqe_clear_error();
  long nb = (rows + 3) / 4;
  if (nb > 256 * 32) nb = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (C <= 256) hipLaunchKernelGGL((layernorm_kernel<1>), dim3((unsigned)nb), dim3(256), 0, st, x, res, gamma, beta, post, y, rows, C, eps, sm);
  else if (C <= 512) hipLaunchKernelGGL((layernorm_kernel<2>), dim3((unsigned)nb), dim3(256), 0, st, x, res, gamma, beta, post, y, rows, C, eps, sm);
  else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<4>), dim3((unsigned)nb), dim3(256), 0, st, x, res, gamma, beta, post, y, rows, C, eps, sm);
  else hipLaunchKernelGGL((layernorm_kernel<8>), dim3((unsigned)nb), dim3(256), 0, st, x, res, gamma, beta, post, y, rows, C, eps, sm);
  return mdqe_launch_status();
}

extern "C" int mdqe_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                  long rows, int C, float eps, void* stream) {
  return layernorm_impl(x, res, gamma, beta, nullptr, y, rows, C, eps, stream);
}

// y = LN(x)*gamma + beta + post   (SwinV2 res-post-norm, swin_transformer_v2.py:287-288)
extern "C" int mdqe_layernorm_post_f32(const float* x, const float* gamma, const float* beta, const float* post, float* y,
                                       long rows, int C, float eps, void* stream) {
  MDQE_CHECK_PTR(post);
  return layernorm_impl(x, nullptr, gamma, beta, post, y, rows, C, eps, stream);
}

// out[pix] = shortcut[pix] + LN(rows[r]) * gamma + beta with r the window-order row of pixel pix: norm1 of a Swin block, window reverse,
// un-shift, crop and the residual add in ONE pass (swin_transformer_v2.py:273-287); rows [B * Hp * Wp, C]; shortcut / out [B, H, W, C]
// (out may alias shortcut: every pixel is written by exactly one row).
extern "C" int mdqe_layernorm_swin_scatter_f32(const float* rows_, const float* gamma, const float* beta, const float* shortcut, float* out,
                                               int B, int H, int W, int C, int ws, int shift, float eps, void* stream) {
  MDQE_REQUIRE(B >= 0 && H > 0 && W > 0 && ws > 0 && shift >= 0 && shift < ws);
  MDQE_CHECK_PTR(shortcut);
  const long Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  return layernorm_impl(rows_, nullptr, gamma, beta, shortcut, out, (long)B * Hp * Wp, C, eps, stream, SwinMap{ws, shift, H, W});
}

// ---------------------------------------------------------------------------------------------
// GroupNorm over NHWC [NI, HW, C] (ldx floats between pixels), G groups of C/G channels
// (nn.GroupNorm eps 1e-5: models/mdqe.py:36,42; segmentation.py:21-26,104-105).
// Pass 1: block (chunk, img) accumulates per-channel sum / sum-of-squares over its pixel chunk,
//         folds channels into groups and writes partial[img][chunk][g][2] (double).
// Pass 2: every block re-derives mean/rstd from the partials (<= 64 chunks) and applies
//         y = act((x-mean)*rstd*gamma+beta) with 16-B accesses.  Deterministic (no atomics).
// ---------------------------------------------------------------------------------------------
#define GN_MAX_CHUNKS 64

__global__ void __launch_bounds__(256)
gn_partial_kernel(const float* __restrict__ x, long ldx, long xis, int HW, int C, int G, int nchunks, double* __restrict__ part) {
  __shared__ float sh_s[1024], sh_q[1024];      // [rsub][channel] partial sums of the block's threads
  __shared__ double ch_s[1024], ch_q[1024];     // per channel
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int c4n = C / 4;                        // float4 columns (<= 256)
  const int tpr = 256 / c4n;                    // pixel rows handled in parallel
  const int col = threadIdx.x % c4n, rsub = threadIdx.x / c4n;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rsub < tpr) {
    const float* base = x + (long)img * xis + col * 4;
    for (int p = p0 + rsub; p < p1; p += tpr) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)p * ldx);
#pragma unroll
      for (int k = 0; k < 4; ++k) { s[k] += v[k]; q[k] += v[k] * v[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh_s[rsub * C + col * 4 + k] = s[k]; sh_q[rsub * C + col * 4 + k] = q[k]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < tpr; ++r) { a += (double)sh_s[r * C + c]; b += (double)sh_q[r * C + c]; }
    ch_s[c] = a; ch_q[c] = b;
  }
  __syncthreads();
  const int cpg = C / G;
  if (threadIdx.x < G) {
    double a = 0.0, b = 0.0;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { a += ch_s[c]; b += ch_q[c]; }
    double* o = part + (((long)img * nchunks + chunk) * G + threadIdx.x) * 2;
    o[0] = a; o[1] = b;
  }
}

__global__ void __launch_bounds__(256)
gn_apply_kernel(const float* __restrict__ x, long ldx, long xis, float* __restrict__ y, long ldy, long yis, int HW, int C, int G, int nchunks,
                const double* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                float eps, int act) {
  __shared__ float s_mean[64], s_rstd[64];
  const int img = blockIdx.y;
  if (threadIdx.x < G) {
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) {
      const double* o = part + (((long)img * nchunks + c) * G + threadIdx.x) * 2;
      s += o[0]; q += o[1];
    }
    const double n = (double)HW * (C / G);
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    s_mean[threadIdx.x] = (float)mean;
    s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int c4n = C / 4, cpg = C / G;
  const long total = (long)HW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % c4n);
    const long p = i / c4n;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long)img * xis + p * ldx + col * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + col * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gi = (col * 4 + k) / cpg;
      o[k] = mdqe_act((v[k] - s_mean[gi]) * s_rstd[gi] * g[k] + b[k], act);
    }
    *reinterpret_cast<f32x4*>(y + (long)img * yis + p * ldy + col * 4) = o;
  }
}

extern "C" long mdqe_groupnorm_workspace_bytes(int NI, int G) { return (long)NI * GN_MAX_CHUNKS * G * 2 * sizeof(double); }

extern "C" int mdqe_groupnorm_nhwc_f32(const float* x, long ldx, long x_img_stride, float* y, long ldy, long y_img_stride,
                                       int NI, int HW, int C, int G,
                                       const float* gamma, const float* beta, float eps, int act, void* workspace,
                                       void* stream) {
  MDQE_REQUIRE(NI >= 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && C % 4 == 0 && C <= 1024);
  MDQE_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0);
  if (x_img_stride <= 0) x_img_stride = (long)HW * ldx;
  if (y_img_stride <= 0) y_img_stride = (long)HW * ldy;
  MDQE_REQUIRE(x_img_stride % 4 == 0 && y_img_stride % 4 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(y); MDQE_CHECK_PTR(gamma); MDQE_CHECK_PTR(beta); MDQE_CHECK_PTR(workspace);
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  int nchunks = (HW + 255) / 256;
  if (nchunks > GN_MAX_CHUNKS) nchunks = GN_MAX_CHUNKS;
  if (nchunks < 1) nchunks = 1;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, NI), dim3(256), 0, st, x, ldx, x_img_stride, HW, C, G, nchunks, (double*)workspace);
  int rc = mdqe_launch_status();
  if (rc) return rc;
  long blocks = ((long)HW * (C / 4) + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, NI), dim3(256), 0, st, x, ldx, x_img_stride, y, ldy, y_img_stride, HW, C, G, nchunks,
                     (const double*)workspace, gamma, beta, eps, act);
  return mdqe_launch_status();
}
