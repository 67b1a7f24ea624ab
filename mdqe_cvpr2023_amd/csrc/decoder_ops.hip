// Decoder-side kernels of the MDQE path for gfx950: small-sequence multi-head attention, grid-guided
// query selection, multi-level content sampling, and the fused final-mask kernel.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// nn.MultiheadAttention core for short sequences (Q <= 256 tokens, head dim D <= 64, D % 4 == 0):
//   o[b,q,h,:] = softmax_k( (q[b,q,h,:] * D^-0.5) . k[b,k,h,:] ) @ v[b,k,h,:]
// (transformer_dec.py:348-353,397-402; torch scales q before the product).  One block per (batch, head):
// K and V of the head live in LDS (2 x Q x D floats, 50 KB at Q=196, D=32), one thread per query row,
// single-pass online softmax; every lane reads the same K/V row -> LDS broadcast reads.
// qk: [B*Q, ldqk] with q at column h*D and k at column C + h*D; v: [B*Q, ldv]; o: [B*Q, ldo].
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
mha_small_kernel(const float* __restrict__ qk, long ldqk, const float* __restrict__ v, long ldv, float* __restrict__ o,
                 long ldo, int Q, int C, int nh) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sK = sm;
  float* sV = sm + Q * D;
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const long row0 = (long)b * Q;
  for (int i = threadIdx.x; i < Q * (D / 4); i += blockDim.x) {
    const int r = i / (D / 4), c4 = i % (D / 4);
    *reinterpret_cast<f32x4*>(sK + r * D + c4 * 4) = *reinterpret_cast<const f32x4*>(qk + (row0 + r) * ldqk + C + h * D + c4 * 4);
    *reinterpret_cast<f32x4*>(sV + r * D + c4 * 4) = *reinterpret_cast<const f32x4*>(v + (row0 + r) * ldv + h * D + c4 * 4);
  }
  __syncthreads();
  const int r = threadIdx.x;
  if (r >= Q) return;
  const float scale = rsqrtf((float)D);
  float q[D], acc[D];
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qk + (row0 + r) * ldqk + h * D + c);
    q[c] = t[0] * scale; q[c + 1] = t[1] * scale; q[c + 2] = t[2] * scale; q[c + 3] = t[3] * scale;
    acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Q; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 kk = *reinterpret_cast<const f32x4*>(sK + j * D + c);
      s += q[c] * kk[0] + q[c + 1] * kk[1] + q[c + 2] * kk[2] + q[c + 3] * kk[3];
    }
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn);
    const float p = expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + j * D + c);
      acc[c] = acc[c] * corr + p * vv[0]; acc[c + 1] = acc[c + 1] * corr + p * vv[1];
      acc[c + 2] = acc[c + 2] * corr + p * vv[2]; acc[c + 3] = acc[c + 3] * corr + p * vv[3];
    }
    m = mn;
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < D; c += 4)
    *reinterpret_cast<f32x4*>(o + (row0 + r) * ldo + h * D + c) = f32x4{acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv};
}

// MFMA form (D = 32 or 24, Q <= 208): the scheme of window_attn_mfma_kernel (swin.hip) without normalisation / bias:
// NWV waves per (batch, head) (7 for D = 32, 4 for D = 24, by measurement), K and V in LDS (rows padded to 36 floats), S^T tiles = K_tile . (q * D^-0.5)^T on
// v_mfma_f32_16x16x4_f32 with the head dimension walked as d = (D/4)*(lane>>4) + t, probabilities kept in the
// accumulator registers and fed straight into the P . V product (its k index follows the same key order).
// VG: V is read straight from global memory (L1 / L2) instead of LDS: the block then holds 30 KB instead of 60 KB of LDS and five
// blocks instead of two share a CU -- the softmax between the two products of one wave hides behind the MFMAs of more others.
template <int D, int NWV, bool VG = false>
__global__ void __launch_bounds__(64 * NWV)
mha_small_mfma_kernel(const float* __restrict__ qk, long ldqk, const float* __restrict__ v, long ldv, float* __restrict__ o,
                      long ldo, int Q, int C, int nh) {
  constexpr int LDK = 36, MAXT = 13, DG = D / 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NT = (Q + 15) / 16, NP = NT * 16;
  float* sK = sm;                                    // [NP][36] (rows >= Q zero)
  float* sV = sm + (VG ? 0 : NP * LDK);
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const long row0 = (long)b * Q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, g = lane >> 4;
  for (int i = tid; i < NP * 8; i += 64 * NWV) {
    const int r = i >> 3, c4 = i & 7;
    f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
    if (r < Q && c4 * 4 < D) {
      kk = *reinterpret_cast<const f32x4*>(qk + (row0 + r) * ldqk + C + h * D + c4 * 4);
      vv = *reinterpret_cast<const f32x4*>(v + (row0 + r) * ldv + h * D + c4 * 4);
    }
    *reinterpret_cast<f32x4*>(sK + r * LDK + c4 * 4) = kk;
    if (!VG) *reinterpret_cast<f32x4*>(sV + r * LDK + c4 * 4) = vv;
  }
  __syncthreads();
  const float* vg = v + row0 * ldv + h * D + lc;     // VG: V[key][d = lc] = vg[key * ldv] (and d = 16 + lc at +16)
  const bool d1 = 16 + lc < D;
  const float sc = rsqrtf((float)D) * 1.4426950408889634f;
  for (int rt = wave; rt < NT; rt += NWV) {
    const int i0 = rt * 16;
    const int qi = min(i0 + lc, Q - 1);
    float q[DG];
    {
      const float* qp = qk + (row0 + qi) * ldqk + h * D + DG * g;
#pragma unroll
      for (int t = 0; t < DG; t += 2) { const f32x2 a = *reinterpret_cast<const f32x2*>(qp + t); q[t] = a[0] * sc; q[t + 1] = a[1] * sc; }
    }
    f32x4 sT[MAXT];
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < MAXT; ++jt) {
      if (jt < NT) {
        const float* kr = sK + (jt * 16 + lc) * LDK + DG * g;
        float ka[DG];
#pragma unroll
        for (int t = 0; t < DG; t += 2) { const f32x2 a = *reinterpret_cast<const f32x2*>(kr + t); ka[t] = a[0]; ka[t + 1] = a[1]; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < DG; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], q[t], acc, 0, 0, 0);
        const int j = jt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (j + r >= Q) acc[r] = -INFINITY;
          mx = fmaxf(mx, acc[r]);
        }
        sT[jt] = acc;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    float lsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MAXT; ++jt)
      if (jt < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float pv = __builtin_amdgcn_exp2f(sT[jt][r] - mx); sT[jt][r] = pv; lsum += pv; }
      }
    lsum += __shfl_xor(lsum, 16); lsum += __shfl_xor(lsum, 32);
    const float linv = 1.f / lsum;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0, p0 = o0, p1 = o0;      // even / odd key tiles: four independent MFMA chains
#pragma unroll
    for (int jt = 0; jt < MAXT; ++jt)
      if (jt < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v0, v1;
          if (VG) {                                  // keys >= Q carry probability 0: any finite row will do
            const float* vr = vg + (long)min(jt * 16 + 4 * g + r, Q - 1) * ldv;
            v0 = vr[0]; v1 = d1 ? vr[16] : 0.f;
          } else {
            const float* vr = sV + (jt * 16 + 4 * g + r) * LDK + lc;
            v0 = vr[0]; v1 = vr[16];                 // columns >= D are zero in LDS
          }
          if (jt & 1) {
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], v0, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], v1, p1, 0, 0, 0);
          } else {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], v0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], v1, o1, 0, 0, 0);
          }
        }
      }
    o0 += p0; o1 += p1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float li = __shfl(linv, 4 * g + r);
      const int qrow = i0 + 4 * g + r;
      if (qrow < Q) {
        float* op = o + (row0 + qrow) * ldo + h * D + lc;
        op[0] = o0[r] * li;
        if (16 + lc < D) op[16] = o1[r] * li;
      }
    }
  }
}

static int g_mha_variant = 1;   // 1: MFMA form where it applies, 0: scalar form everywhere (tools/ A/B)
extern "C" int mdqe_debug_mha_variant(int v) { g_mha_variant = v; return MDQE_OK; }

extern "C" int mdqe_mha_small_f32(const float* qk, long ldqk, const float* v, long ldv, float* o, long ldo, int B, int Q,
                                  int C, int nh, void* stream) {
  MDQE_REQUIRE(B >= 0 && Q > 0 && Q <= 256 && nh > 0 && C % nh == 0 && ldqk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0);
  const int D = C / nh;
  MDQE_REQUIRE(D == 32 || D == 24 || D == 16 || D == 8);
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(qk); MDQE_CHECK_PTR(v); MDQE_CHECK_PTR(o);
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  if ((D == 32 || D == 24) && Q <= 208 && g_mha_variant != 0) {
    const size_t smem2 = (size_t)2 * ((Q + 15) / 16 * 16) * 36 * sizeof(float);
    // 13 row tiles of 16 queries over NWV waves.  Measured (tools/attn_bench.py, B = 148): D = 32: 3 waves 117 us, 4: 100, 7: 86,
    // 13: 100; D = 24: 4 waves win.  g_mha_variant 1 = that choice; 2 -> 3 waves, 3 -> 7, 4 -> 13, 5 -> 4 (A/B)
#define LM(DD, NW_) hipLaunchKernelGGL((mha_small_mfma_kernel<DD, NW_>), dim3(B * nh), dim3(64 * NW_), smem2, st, qk, ldqk, v, ldv, o, ldo, Q, C, nh)
    const int nwv = g_mha_variant == 2 ? 3 : g_mha_variant == 3 ? 7 : g_mha_variant == 4 ? 13 : g_mha_variant == 5 ? 4 : (D == 32 ? 7 : 4);
    if (g_mha_variant >= 6 && g_mha_variant <= 8) {     // A/B: V from global memory (half the LDS), 7 / 4 / 3 waves
      const size_t smem1 = (size_t)((Q + 15) / 16 * 16) * 36 * sizeof(float);
#define LG(DD, NW_) hipLaunchKernelGGL((mha_small_mfma_kernel<DD, NW_, true>), dim3(B * nh), dim3(64 * NW_), smem1, st, qk, ldqk, v, ldv, o, ldo, Q, C, nh)
      if (D == 32) { if (g_mha_variant == 6) LG(32, 7); else if (g_mha_variant == 7) LG(32, 4); else LG(32, 3); }
      else         { if (g_mha_variant == 6) LG(24, 7); else if (g_mha_variant == 7) LG(24, 4); else LG(24, 3); }
#undef LG
      return mdqe_launch_status();
    }
    if (D == 32) { if (nwv == 3) LM(32, 3); else if (nwv == 7) LM(32, 7); else if (nwv == 13) LM(32, 13); else LM(32, 4); }
    else         { if (nwv == 3) LM(24, 3); else if (nwv == 7) LM(24, 7); else if (nwv == 13) LM(24, 13); else LM(24, 4); }
#undef LM
    return mdqe_launch_status();
  }
  const size_t smem = (size_t)2 * Q * D * sizeof(float);
#define L(DD) do { (void)hipFuncSetAttribute((const void*)mha_small_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((mha_small_kernel<DD>), dim3(B * nh), dim3(256), smem, st, qk, ldqk, v, ldv, o, ldo, Q, C, nh); } while (0)
  if (D == 32) L(32); else if (D == 24) L(24); else if (D == 16) L(16); else L(8);
#undef L
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Grid-guided query selection (transformer_dec.py:81-109): s = max_k sigmoid(conf[img,y,x,k]);
// bilinear resize (align_corners=False) to (H_up,W_up) = multiples of the nb x nb grid; per cell the FIRST
// maximum wins; coords = (fmod(idx,W_up)/W_up, (idx/W_up)/H_up) with TRUE division (:105-106).
// Pass 1 writes the score map; pass 2: one wave per (img, cell).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
score_max_kernel(const float* __restrict__ conf, long n, int K, float* __restrict__ score) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, conf[i * K + k]);
    score[i] = 1.0f / (1.0f + expf(-m));                 // sigmoid is monotone: max(sigmoid) = sigmoid(max)
  }
}

__global__ void __launch_bounds__(64)
cell_argmax_kernel(const float* __restrict__ score, int H, int W, int H_up, int W_up, int nb, float* __restrict__ coords) {
  const int cell = blockIdx.x % (nb * nb), img = blockIdx.x / (nb * nb);
  const int cy = cell / nb, cx = cell % nb;
  const int r = H_up / nb, t = W_up / nb;
  const float sh = (float)H / (float)H_up, sw = (float)W / (float)W_up;
  const float* s = score + (long)img * H * W;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int p = threadIdx.x; p < r * t; p += 64) {
    const int oy = cy * r + p / t, ox = cx * t + p % t;
    float fy = ((float)oy + 0.5f) * sh - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)ox + 0.5f) * sw - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float v = (1.f - ly) * ((1.f - lx) * s[y0 * W + x0] + lx * s[y0 * W + x1]) +
                    ly * ((1.f - lx) * s[y1 * W + x0] + lx * s[y1 * W + x1]);
    if (v > best || (v == best && p < bidx)) { best = v; bidx = p; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  if (threadIdx.x == 0) {
    const int row = cy * r + bidx / t, col = cx * t + bidx % t;
    const float idx = (float)(row * W_up + col);
    float* c = coords + ((long)img * nb * nb + cell) * 2;
    c[0] = fmodf(idx, (float)W_up) / (float)W_up;
    c[1] = (idx / (float)W_up) / (float)H_up;
  }
}

extern "C" int mdqe_query_select_f32(const float* conf, int NI, int H, int W, int K, int nb, float* score_ws, float* coords,
                                     void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && W > 0 && K > 0 && nb > 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(conf); MDQE_CHECK_PTR(score_ws); MDQE_CHECK_PTR(coords);
  mdqe_clear_error();
  const int H_up = (2 * H / nb + 1) * nb, W_up = (2 * W / nb + 1) * nb;
  const long n = (long)NI * H * W;
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(score_max_kernel, dim3((unsigned)blocks), dim3(256), 0, st, conf, n, K, score_ws);
  int rc = mdqe_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(cell_argmax_kernel, dim3(NI * nb * nb), dim3(64), 0, st, score_ws, H, W, H_up, W_up, nb, coords);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Content of the selected queries (transformer_dec.py:171-179): mean over levels of
// grid_sample(level, 2*coord-1, bilinear, padding_mode=border, align_corners=False) on channels-last tokens
// tokens [NI, N, C]; level l occupies rows start[l] .. start[l]+H_l*W_l.
// ------------------------------------------------------------------------------------------------
struct SampLevels { int H[8]; int W[8]; int start[8]; int n; };

__global__ void __launch_bounds__(256)
sample_levels_mean_kernel(const float* __restrict__ tok, long N, int C, const float* __restrict__ coords, int Qn, SampLevels lv,
                          float* __restrict__ out, long total) {
  const int c4n = C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const long t = i / c4n;                              // img*Qn + q
    const long img = t / Qn;
    const float cx = coords[t * 2], cy = coords[t * 2 + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < lv.n; ++l) {
      const int H = lv.H[l], W = lv.W[l];
      float ix = ((2.f * cx - 1.f + 1.f) * W - 1.f) * 0.5f;      // grid_sampler unnormalize, align_corners=False
      float iy = ((2.f * cy - 1.f + 1.f) * H - 1.f) * 0.5f;
      ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));                // border: clip coordinates
      iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
      const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
      const float lx = ix - x0, ly = iy - y0;
      const float* base = tok + (img * N + lv.start[l]) * C + c4 * 4;
      f32x4 v = (1.f - ly) * (1.f - lx) * *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0) * C);
      if (x0 + 1 < W) v += (1.f - ly) * lx * *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0 + 1) * C);
      if (y0 + 1 < H) v += ly * (1.f - lx) * *reinterpret_cast<const f32x4*>(base + ((long)(y0 + 1) * W + x0) * C);
      if (x0 + 1 < W && y0 + 1 < H) v += ly * lx * *reinterpret_cast<const f32x4*>(base + ((long)(y0 + 1) * W + x0 + 1) * C);
      acc += v;
    }
    *reinterpret_cast<f32x4*>(out + t * C + c4 * 4) = acc / (float)lv.n;
  }
}

extern "C" int mdqe_sample_levels_mean_f32(const float* tokens, int NI, long N, int C, const float* coords, int Qn,
                                           const int* lvH_host, const int* lvW_host, const int* lvStart_host, int n_levels,
                                           float* out, void* stream) {
  MDQE_REQUIRE(NI >= 0 && N > 0 && C > 0 && C % 4 == 0 && Qn > 0 && n_levels > 0 && n_levels <= 8);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(tokens); MDQE_CHECK_PTR(coords); MDQE_CHECK_PTR(out);
  MDQE_CHECK_PTR(lvH_host); MDQE_CHECK_PTR(lvW_host); MDQE_CHECK_PTR(lvStart_host);
  SampLevels lv;
  lv.n = n_levels;
  for (int i = 0; i < 8; ++i) { lv.H[i] = 1; lv.W[i] = 1; lv.start[i] = 0; }
  for (int i = 0; i < n_levels; ++i) { lv.H[i] = lvH_host[i]; lv.W[i] = lvW_host[i]; lv.start[i] = lvStart_host[i]; }
  mdqe_clear_error();
  const long total = (long)NI * Qn * (C / 4);
  long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(sample_levels_mean_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, tokens, N, C, coords, Qn, lv,
                     out, total);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Final masks (mdqe/mdqe.py:357-358 + 458-462): out[i,f,Y,X] = sigmoid(aligned_bilinear_x4(logits)[sy,sx]) > 0.5 with
// (sy,sx) = nearest source pixel of the crop [:h,:w] for an output of (Ho,Wo):  sy = min(floor(Y*h/Ho), h-1).
// aligned_bilinear (util/misc.py:485-507) in closed form: pixel p reads source (max(p - f/2, 0))/f, clamped to the map.
// logits [n, F, Hm, Wm] (mean logits of one tracker window); out uint8 [n, F_total, Ho, Wo] written at frame f_off.
// ------------------------------------------------------------------------------------------------
// one output pixel of the final mask (shared by the dense and the RLE form: identical arithmetic, identical bits)
__device__ __forceinline__ int final_mask_pixel(const float* __restrict__ m, int Hm, int Wm, int factor, int h, int w, float sy_scale,
                                                float sx_scale, int Y, int X) {
  const int sy = min((int)floorf(Y * sy_scale), h - 1), sx = min((int)floorf(X * sx_scale), w - 1);
  const float fy = (float)max(sy - factor / 2, 0) / (float)factor, fx = (float)max(sx - factor / 2, 0) / (float)factor;
  const int y0 = min((int)fy, Hm - 1), x0 = min((int)fx, Wm - 1);
  const int y1 = min(y0 + 1, Hm - 1), x1 = min(x0 + 1, Wm - 1);
  const float ly = fy - y0, lx = fx - x0;
  const float top = m[y0 * Wm + x0] * (1.f - lx) + m[y0 * Wm + x1] * lx;
  const float bot = m[y1 * Wm + x0] * (1.f - lx) + m[y1 * Wm + x1] * lx;
  const float v = top * (1.f - ly) + bot * ly;
  const float p = 1.0f / (1.0f + expf(-v));
  return p > 0.5f ? 1 : 0;
}

__global__ void __launch_bounds__(256)
final_mask_kernel(const float* __restrict__ lg, int Fw, int Hm, int Wm, int factor, int h, int w, int Ho, int Wo,
                  unsigned char* __restrict__ out, long out_inst_stride, int f_off, const int* __restrict__ inst_idx, long total) {
  const float sy_scale = (float)h / (float)Ho, sx_scale = (float)w / (float)Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % Wo); long t = i / Wo;
    const int Y = (int)(t % Ho); t /= Ho;
    const int f = (int)(t % Fw); const int k = (int)(t / Fw);
    const float* m = lg + ((long)inst_idx[k] * Fw + f) * Hm * Wm;
    out[(long)k * out_inst_stride + ((long)(f_off + f) * Ho + Y) * Wo + X] = (unsigned char)final_mask_pixel(m, Hm, Wm, factor, h, w, sy_scale, sx_scale, Y, X);
  }
}

// ------------------------------------------------------------------------------------------------
// Final masks straight to COCO run-length form (the result writer's
// mask_util.encode(np.array(mask[:, :, None], order="F")), mdqe/data/ytvis_eval.py:307-312, i.e. cocoapi rleEncode): the
// mask of (instance k, frame f) is never materialised -- one block walks its pixels in COLUMN-major order, every thread a
// contiguous segment, evaluating final_mask_pixel on the fly, and emits the positions p where the value differs from
// p-1 (value before the first pixel = 0).  Runs are the differences of consecutive positions (host).  Two sweeps: count
// per thread -> block scan -> write.  pos [n_sel*Fw, cap], n_pos [n_sel*Fw] (may exceed cap: the host then falls back).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
final_mask_rle_kernel(const float* __restrict__ lg, int Fw, int Hm, int Wm, int factor, int h, int w, int Ho, int Wo,
                      const int* __restrict__ inst_idx, int cap, int* __restrict__ pos, int* __restrict__ n_pos) {
  const int k = blockIdx.x / Fw, f = blockIdx.x - k * Fw;
  const float* m = lg + ((long)inst_idx[k] * Fw + f) * Hm * Wm;
  const float sy_scale = (float)h / (float)Ho, sx_scale = (float)w / (float)Wo;
  const int total = Ho * Wo;
  const int seg = (total + 255) / 256;
  const int p0 = min((int)threadIdx.x * seg, total), p1 = min(p0 + seg, total);
  int prev0 = 0;
  if (p0 > 0 && p0 < total) prev0 = final_mask_pixel(m, Hm, Wm, factor, h, w, sy_scale, sx_scale, (p0 - 1) % Ho, (p0 - 1) / Ho);
  int cnt = 0, prev = prev0;
  for (int p = p0; p < p1; ++p) {
    const int v = final_mask_pixel(m, Hm, Wm, factor, h, w, sy_scale, sx_scale, p % Ho, p / Ho);
    cnt += (v != prev);
    prev = v;
  }
  __shared__ int sc[256];
  sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                  // inclusive scan
    const int add = (int)threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
    __syncthreads();
    sc[threadIdx.x] += add;
    __syncthreads();
  }
  int off = sc[threadIdx.x] - cnt;
  if (threadIdx.x == 255) n_pos[blockIdx.x] = sc[255];
  int* out = pos + (long)blockIdx.x * cap;
  prev = prev0;
  for (int p = p0; p < p1; ++p) {
    const int v = final_mask_pixel(m, Hm, Wm, factor, h, w, sy_scale, sx_scale, p % Ho, p / Ho);
    if (v != prev) { if (off < cap) out[off] = p; ++off; }
    prev = v;
  }
}

extern "C" int mdqe_final_masks_rle(const float* logits, int n_sel, const int* inst_idx_dev, int Fw, int Hm, int Wm, int factor,
                                    int h, int w, int Ho, int Wo, int cap, int* pos, int* n_pos, void* stream) {
  MDQE_REQUIRE(n_sel >= 0 && Fw >= 0 && Hm > 0 && Wm > 0 && factor >= 1 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && cap > 0);
  MDQE_REQUIRE(h <= Hm * factor && w <= Wm * factor && (long)Ho * Wo < 0x7FFFFFFFL);
  if (n_sel == 0 || Fw == 0) return MDQE_OK;
  MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(inst_idx_dev); MDQE_CHECK_PTR(pos); MDQE_CHECK_PTR(n_pos);
  mdqe_clear_error();
  hipLaunchKernelGGL(final_mask_rle_kernel, dim3((unsigned)(n_sel * Fw)), dim3(256), 0, (hipStream_t)stream, logits, Fw, Hm, Wm, factor,
                     h, w, Ho, Wo, inst_idx_dev, cap, pos, n_pos);
  return mdqe_launch_status();
}

extern "C" int mdqe_final_masks_u8(const float* logits, int n_sel, const int* inst_idx_dev, int Fw, int Hm, int Wm, int factor,
                                   int h, int w, int Ho, int Wo, unsigned char* out, long out_inst_stride, int f_off,
                                   void* stream) {
  MDQE_REQUIRE(n_sel >= 0 && Fw >= 0 && Hm > 0 && Wm > 0 && factor >= 1 && h > 0 && w > 0 && Ho > 0 && Wo > 0);
  MDQE_REQUIRE(h <= Hm * factor && w <= Wm * factor);
  if (n_sel == 0 || Fw == 0) return MDQE_OK;
  MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(inst_idx_dev); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  const long total = (long)n_sel * Fw * Ho * Wo;
  long nb = (total + 255) / 256; if (nb > 256 * 64) nb = 256 * 64;
  hipLaunchKernelGGL(final_mask_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, logits, Fw, Hm, Wm, factor, h, w, Ho,
                     Wo, out, out_inst_stride, f_off, inst_idx_dev, total);
  return mdqe_launch_status();
}
