// fp32 NT GEMM / NHWC implicit-GEMM convolution on the gfx950 matrix cores.
//
//   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )          A, W, C fp32; exact fp32 arithmetic
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate; bit-for-bit an fmaf chain, 157 TFLOP/s peak)
// is used because the path's parity bar is fp32 (north star: 1e-3 against the fp32 CPU path).
//
// Serves every Linear / 1x1 conv / KxK conv on the hot path:
//   * plain mode:  A(m,k) = A[m*lda + k]
//   * conv mode:   m = (img, oh, ow), k = (kh, kw, cin) over an NHWC input; zero padding comes
//                  from the buffer-descriptor bounds check (out-of-range lanes return 0).
// W is [N][K] row-major (nn.Linear layout; convs are pre-packed [Cout][KH][KW][Cin]).
//
// Structure (CDNA4): block tile BM x BN, K-step 32 floats (one 128-B line per row).  Tiles go
// HBM/L2 -> LDS with buffer_load ... lds (16 B per lane, 1 KiB per wave-instruction, no VGPR round
// trip), double buffered; the load of step k+1 is issued before the MFMAs of step k.  LDS rows are
// 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the *source* side (the LDS write
// of an LDS-DMA is lane-linear) and on the ds_read_b128 side, which makes every 16-lane read group
// hit 16 distinct 16-B slots (conflict-free).  One ds_read_b128 per operand feeds four MFMAs:
// lanes 0-31 carry k = 4c..4c+3 of their row, lanes 32-63 carry the next four.
#include "common.h"

#include "gemm_params.h"

template <int BM, int BN, int WM, int WN, bool CONV>
__global__ void __launch_bounds__(64 * WM * WN)
gemm_nt_f32_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int BK = 32;
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int ROWS = BM + BN;              // A rows then W rows in one LDS image
  constexpr int NINST = ROWS / 8;            // 1-KiB LDS-DMA instructions per stage
  constexpr int IPW = NINST / NW;            // per wave
  static_assert(NINST % NW == 0, "tile rows must split evenly over waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * ROWS * 32 floats

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: blocks that share an A row-panel run on the same XCD (same L2).
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, i = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int bm = bid / nbn, bn = bid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);

  // Per-lane source bookkeeping for this lane's IPW rows (fixed over the K loop).
  // instruction j of this wave covers image rows (wave*IPW + j)*8 .. +7; lane -> row += lane>>3, chunk' = lane&7,
  // source chunk = (lane&7) ^ ((row>>1)&7).  voff[j] = byte offset of (row, k = 4*chunk); the K-step rides on the
  // instruction's scalar offset (plain rows, W rows) or on the filter-tap offset (conv A rows; kept in the VGPR offset
  // because a padded pixel's base may be "negative" and the scalar offset is not part of the bounds check).
  unsigned voff[IPW];
  int ih0[IPW], iw0[IPW], kch[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int irow = (wave * IPW + j) * 8 + (lane >> 3);
    kch[j] = ((lane & 7) ^ ((irow >> 1) & 7)) * 4;
    ih0[j] = 0; iw0[j] = 0;
    if (irow < BM) {
      int m = m0 + irow; if (m > p.M - 1) m = p.M - 1;
      if (CONV) {
        const int ow = m % p.OW; const int t = m / p.OW; const int oh = t % p.OH; const int img = t / p.OH;
        ih0[j] = oh * p.stride - p.pad; iw0[j] = ow * p.stride - p.pad;
        voff[j] = (unsigned)(((long)img * p.img_stride + ((long)ih0[j] * p.Wd + iw0[j]) * p.Cin) * 4) + (unsigned)(kch[j] * 4);   // may wrap
      } else {
        voff[j] = (unsigned)((long)m * p.lda * 4) + (unsigned)(kch[j] * 4);
      }
    } else {
      int n = n0 + irow - BM; if (n > p.N - 1) n = p.N - 1;
      voff[j] = (unsigned)((long)n * p.K * 4) + (unsigned)(kch[j] * 4);
    }
  }

  const int kbeg = p.ksplit > 1 ? blockIdx.y * p.kchunk : 0;
  const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const int kt0 = kbeg / BK;
  const int nk = (kend - kbeg + BK - 1) / BK;
  // conv: filter tap of the K-step about to be issued (the 32-float K-step lies inside one tap: Cin % 32 == 0, host-checked)
  int t_kh = 0, t_kw = 0, t_c = 0;
  if (CONV) { const int tap = kbeg / p.Cin; t_c = kbeg - tap * p.Cin; t_kh = tap / p.KW; t_kw = tap - t_kh * p.KW; }

  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    const bool ktail = k0 + BK > p.K;              // only the last K-step of a ragged K (plain mode) checks lanes against K
    float* base = lds + buf * (ROWS * BK);
    int tap_off = 0;
    if (CONV) tap_off = ((t_kh * p.Wd + t_kw) * p.Cin + t_c) * 4;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int irow0 = (wave * IPW + j) * 8;        // wave-uniform
      unsigned off = voff[j];
      if (irow0 < BM) {
        if (CONV) {
          const int ih = ih0[j] + t_kh, iw = iw0[j] + t_kw;
          const bool ok = (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.Wd);
          off = ok ? off + (unsigned)tap_off : OOB_OFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + irow0 * BK), 16, off, 0, 0, 0);
        } else {
          if (ktail && k0 + kch[j] >= p.K) off = OOB_OFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + irow0 * BK), 16, off, k0 * 4, 0, 0);
        }
      } else {
        if (ktail && k0 + kch[j] >= p.K) off = OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + irow0 * BK), 16, off, k0 * 4, 0, 0);
      }
    }
    if (CONV) { t_c += BK; if (t_c >= p.Cin) { t_c = 0; if (++t_kw == p.KW) { t_kw = 0; ++t_kh; } } }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  issue(kt0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA for step kt has landed
    __syncthreads();                                   // ... and everybody else's; all reads of the other buffer are done
    if (kt + 1 < nk) issue(kt0 + kt + 1, (kt + 1) & 1);
    const float* sA = lds + (kt & 1) * (ROWS * BK);
    const float* sW = sA + BM * BK;
    // fragments of sub-step kk+1 are fetched before the MFMAs of kk are issued (LDS latency behind 16 MFMAs)
    f32x4 fa_[2][MT], fb_[2][NT];
    auto frag = [&](int kk, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * (BM / WM) + i * 32 + lr;
        const int ch = (kk * 2 + lh) ^ ((row >> 1) & 7);
        fa_[slot][i] = *reinterpret_cast<const f32x4*>(sA + row * BK + ch * 4);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * (BN / WN) + j * 32 + lr;
        const int ch = (kk * 2 + lh) ^ ((row >> 1) & 7);
        fb_[slot][j] = *reinterpret_cast<const f32x4*>(sW + row * BK + ch * 4);
      }
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) frag(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[kk & 1][i][s], fb_[kk & 1][j][s], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // acc[i][j][r] is C(row = (r&3) + 8*(r>>2) + 4*lh, col = lr) of a 32x32 sub-tile.  The tile is restaged
  // through LDS (free after the K loop) so that every lane then owns 4 consecutive columns of a row:
  // bias / residual / C move as 16-B lane accesses, 512 B contiguous per row across 32 lanes.
  __syncthreads();                                   // every wave is done reading the last K-step
  float* sC = lds;                                   // [BM][BN] floats (== 2 stage buffers when BM == BN)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = wn * (BN / WN) + j * 32 + lr;
        sC[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();
  constexpr int C4 = BN / 4;                         // float4 per tile row
  constexpr int NV = BM * C4 / (64 * NW);            // float4 per thread
  if (p.ksplit > 1) {                                // raw partial tile -> workspace; epilogue runs in the reduce pass
    float* w = p.ws + (long)blockIdx.y * p.M * p.N;
    for (int it = 0; it < NV; ++it) {
      const int idx = it * (64 * NW) + tid;
      const int row = idx / C4, c4 = idx - row * C4;
      const int m = m0 + row, n = n0 + c4 * 4;
      if (m >= p.M) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) w[(long)m * p.N + n + e] = sC[row * BN + c4 * 4 + e];
    }
    return;
  }
  const bool vec = p.vec_ok;
  int rr0 = 0;
  if (p.residual != nullptr && p.res_mod > 0) rr0 = m0 % p.res_mod;
#pragma unroll 1
  for (int it = 0; it < NV; ++it) {                   // (rolled: see gemm_k16.hip's epilogue)
    const int idx = it * (64 * NW) + tid;
    const int row = idx / C4, c4 = idx - row * C4;
    const int m = m0 + row, n = n0 + c4 * 4;
    if (m >= p.M || n >= p.N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * BN + c4 * 4);
    const bool full = vec && (n + 3 < p.N);
    long rrow = m;
    if (p.res_mod > 0) { int t = rr0 + row; while (t >= p.res_mod) t -= p.res_mod; rrow = t; }
    const bool masked = p.rowmask != nullptr && p.rowmask[m];
    if (full) {
      if (p.bias != nullptr) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (p.residual != nullptr) rv = *reinterpret_cast<const f32x4*>(p.residual + rrow * p.ldr + n);
      if (p.res_first) v += rv;
      mdqe_act4(v, p.act, [&](int e) { return p.act_cols <= 0 || n + e < p.act_cols; });
      if (!p.res_first) v += rv;
      if (masked) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.mask_cols) v[e] = 0.f;
      }
      *reinterpret_cast<f32x4*>(p.C + (long)m * p.ldc + n) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) break;
        float x = v[e] + (p.bias != nullptr ? p.bias[n + e] : 0.f);
        const float rv = p.residual != nullptr ? p.residual[rrow * p.ldr + n + e] : 0.f;
        if (p.res_first) x += rv;
        if (p.act != MDQE_ACT_NONE && (p.act_cols <= 0 || n + e < p.act_cols)) x = mdqe_act(x, p.act);
        if (!p.res_first) x += rv;
        if (masked && n + e < p.mask_cols) x = 0.f;
        p.C[(long)m * p.ldc + n + e] = x;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(const GemmParams p) {
  const long total = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.N), n = (int)(i - (long)m * p.N);
    float x = 0.f;
    for (int s = 0; s < p.ksplit; ++s) x += p.ws[(long)s * total + i];       // fixed order: deterministic
    if (p.bias != nullptr) x += p.bias[n];
    float rv = 0.f;
    if (p.residual != nullptr) rv = p.residual[(long)(p.res_mod > 0 ? m % p.res_mod : m) * p.ldr + n];
    if (p.res_first) x += rv;
    if (p.act != MDQE_ACT_NONE && (p.act_cols <= 0 || n < p.act_cols)) x = mdqe_act(x, p.act);
    if (!p.res_first) x += rv;
    if (p.rowmask != nullptr && n < p.mask_cols && p.rowmask[m]) x = 0.f;
    p.C[(long)m * p.ldc + n] = x;
  }
}

template <int BM, int BN, int WM, int WN, bool CONV>
static int launch_gemm_(const GemmParams& p, hipStream_t st) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  size_t smem = 2 * (BM + BN) * 32 * sizeof(float);
  if (smem < (size_t)BM * BN * sizeof(float)) smem = (size_t)BM * BN * sizeof(float);   // epilogue restage
  auto kern = gemm_nt_f32_kernel<BM, BN, WM, WN, CONV>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nbm * nbn, p.ksplit > 1 ? p.ksplit : 1), dim3(64 * WM * WN), smem, st, p);
  int rc = mdqe_launch_status();
  if (rc || p.ksplit <= 1) return rc;
  long nb = ((long)p.M * p.N + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
  return mdqe_launch_status();
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmParams& p, hipStream_t st) {
  return p.conv ? launch_gemm_<BM, BN, WM, WN, true>(p, st) : launch_gemm_<BM, BN, WM, WN, false>(p, st);
}

static unsigned long long* g_gemm_stamps = nullptr;   // tools/ only: in-kernel phase stamps of the f16x3w kernel
extern "C" int mdqe_debug_gemm_stamps(void* buf) { g_gemm_stamps = (unsigned long long*)buf; return MDQE_OK; }

static int g_gemm_tile_rule = 0;      // tools/ A/B of the auto tile rule
extern "C" int mdqe_debug_gemm_tile_rule(int v) { g_gemm_tile_rule = v; return MDQE_OK; }
static int g_gemm_rows_dot = 1;       // tools/ A/B: 0 = N <= 8 products stay on the MFMA tiles
extern "C" int mdqe_debug_gemm_rows_dot(int v) { g_gemm_rows_dot = v; return MDQE_OK; }
static int g_gemm_variant = 2;         // fp32 kernel form: 0 = K-step 32 (gemm.hip), 1 = K-step 16 (gemm_k16.hip), 2 = by shape
extern "C" int mdqe_debug_gemm_variant(int v) { g_gemm_variant = v; return MDQE_OK; }

static int g_gemm_precision_all = 0;  // 0: exact fp32 MFMA; 1: f16x3 split on the 128-row tiles (gemm_f16x3.hip); 2: ONE f16 pass where the weight has planes -- process-wide
// The calling THREAD's override (-1 = none): a region of the model (the reference's autocast regions, engine.Engine.amp) switches
// the mode for its own launches without changing what another host thread -- the sharded schedule's tracker replay -- launches
// meanwhile.  Every reader below goes through the macro.
static thread_local int tl_gemm_precision = -1;
#define g_gemm_precision (tl_gemm_precision >= 0 ? tl_gemm_precision : g_gemm_precision_all)
extern "C" int mdqe_set_gemm_precision(int mode) {
  if (mode != 0 && mode != 1 && mode != 2) return MDQE_EINVAL;
  g_gemm_precision_all = mode;
  return MDQE_OK;
}
extern "C" int mdqe_set_gemm_precision_thread(int mode) {
  if (mode != -1 && mode != 0 && mode != 1 && mode != 2) return MDQE_EINVAL;
  tl_gemm_precision = mode;
  return MDQE_OK;
}
extern "C" int mdqe_get_gemm_precision(void) { return g_gemm_precision; }

static int dispatch_gemm(GemmParams& p, int tile, hipStream_t st) {
  p.stamps = g_gemm_stamps;
  p.vec_ok = ((((uintptr_t)p.C | (uintptr_t)p.bias | (uintptr_t)p.residual) & 15) == 0) && (p.ldc % 4 == 0) &&
             (p.residual == nullptr || p.ldr % 4 == 0);
  // tile: 0 = auto.  Measured on the K-step-16 kernel (tools/tile_sweep*.py): the 128x128 tile only pays when the grid is
  // many times the 1024 resident blocks AND the tile has depth or width to amortise (N >= 1024 or K >= 1024); mid-size
  // problems -- the decoder's 21168-row GEMMs, res4/res5 3x3 convs -- run 15-30 % faster on 64x64 tiles (8 waves/SIMD).
  if (tile == 0) {
    const long b128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    // f16x3w / f16w have their own tiling -- but only products the pre-split-weight kernels can take (the same predicate as below): an
    // ineligible one (ragged K, 64 < N < 128, unaligned C, split-K) keeps the fp32 heuristics' tile instead of a forced 128x128
    const bool planes_ok = p.Wh != nullptr && p.K % 32 == 0 && p.N >= 128 && p.N % 4 == 0 && p.vec_ok && p.ksplit <= 1;
    if (g_gemm_precision != 0 && planes_ok && b128 >= 192) tile = 1;
    else if (p.N <= 64) tile = (p.M >= 4096) ? 2 : 3;
    else if ((g_gemm_tile_rule & 1) && b128 >= 2000 && p.N >= 1024 && p.N < 2048 && p.K <= 256 && !p.conv) tile = 9;   // short K, wide N -> 64x128
    else if ((g_gemm_tile_rule & 2) && p.conv && p.KH == 3 && b128 < 2000 && b128 >= 400 && p.N >= 256 && p.K < 4096) tile = 9;   // mid-grid 3x3 convs
    else if (b128 >= 2000 && (p.N >= 768 || p.K >= 1024)) tile = 1;        // (round 4: 768-wide products too -- Swin-L's K = 192 FFN1: 782 vs 794 us)
    else if (b128 >= 2000 && (p.N > 256 || p.K > 256)) tile = 2;
    else if (b128 >= 2000 && p.N >= 128 && !p.conv) tile = 2;               // (round 4, with the few-instruction epilogue: [153000,256]x[256,256] 167 vs 174 us,
    else tile = 3;                                                           //  Swin-L's [907200,192]x[192,192] 590 vs 616 us; before it 64x64 won: 288 vs 329 us)
  }
  if (tile == 1 && g_gemm_precision == 2 && p.Wh != nullptr && p.K % 32 == 0 && p.N >= 128 && p.N % 4 == 0 && p.vec_ok && p.ksplit <= 1) {
    // mode 2 ("f16"): ONE f16 MFMA pass on the round-to-nearest plane of a constant weight (the third plane of mdqe_f16x3_split_f32);
    // every product without planes, or too small / ragged for the 128-row tile, stays exact fp32 below (more accurate, never less)
    GemmParams q = p;
    q.Wh = (const char*)p.Wh + (long)p.N * p.K * 4;                      // planes: hi | lo | rn, N*K halves each
    const int n256 = (p.N + 255) / 256 * 256, n128 = (p.N + 127) / 128 * 128;
    const int bn = (n256 > n128 || (long)((p.M + 127) / 128) * (n256 / 256) < 400) ? 128 : 256;
    return mdqe_launch_gemm_f16w(q, bn, st);
  }
  if (tile == 1 && g_gemm_precision == 1 && p.Wh != nullptr && p.K % 32 == 0 && p.N >= 128 && p.N % 4 == 0 &&
      p.vec_ok && p.ksplit <= 1) {
    // constant weights with pre-split planes: 128 x BN tile, BN by column waste, then by grid size
    const int n256 = (p.N + 255) / 256 * 256, n128 = (p.N + 127) / 128 * 128;
    int bn = 256;
    if (n256 > n128 || (long)((p.M + 127) / 128) * (n256 / 256) < 200) bn = 128;
    else {
      // the kernel is persistent, one block per CU: tiles are dealt in rounds of g_num_cus, and a 128-column tile costs ~0.55 of a
      // 256-column one (profiles/r04_f16x3w_one_site_ab.txt, columns "rule 8" / "rule 12") -- few-round grids take the narrow tile
      // when that saves a partial round (300 tiles: 2 rounds -> 3 half rounds, 93.6 -> 82.7 us)
      const long t256 = (long)((p.M + 127) / 128) * (n256 / 256), cus = 256;
      const long r256 = (t256 + cus - 1) / cus, r128 = (2 * t256 + cus - 1) / cus;
      if (r256 <= 4 && 0.55 * (double)r128 < 0.97 * (double)r256) bn = 128;
    }
    int rc = mdqe_launch_gemm_f16x3w(p, bn, st);
    if (rc || p.ksplit <= 1) return rc;
    long nb = ((long)p.M * p.N + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
    return mdqe_launch_status();
  }
  if ((tile == 1 || tile == 2) && g_gemm_precision == 1) {
    int rc = mdqe_launch_gemm_f16x3(p, tile, st);
    if (rc || p.ksplit <= 1) return rc;
    long nb = ((long)p.M * p.N + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
    return mdqe_launch_status();
  }
  // K-step 16 (4+ blocks per CU) wins almost everywhere.  (Rounds 1-3: very deep-K convs on few rows kept the 32-wide step; with the
  // round-4 K loop and epilogue the K-step-16 kernel is ahead there too -- res5's 3x3 convs 419 against 460 us -- except on the one
  // narrow, few-row shape below: 139 against 146 us.)
  // (the window-order A map of mdqe_gemm_nt_swin_f32 exists in the K-step-16 kernel only: never the K-step-32 form, whatever the debug variant)
  if (p.swin_ws > 0 || g_gemm_variant == 1 || (g_gemm_variant == 2 && !(p.conv && p.K >= 2048 && p.K < 4096 && p.M <= 16384 && p.N <= 256))) {
    int rc = mdqe_launch_gemm_k16(p, tile, st);
    if (rc || p.ksplit <= 1) return rc;
    long nb = ((long)p.M * p.N + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
    return mdqe_launch_status();
  }
  switch (tile) {
    case 1: return launch_gemm<128, 128, 2, 2>(p, st);
    case 2: return launch_gemm<128, 64, 2, 2>(p, st);
    case 3: case 7: case 8: case 9: return launch_gemm<64, 64, 2, 2>(p, st);
    default: return MDQE_EINVAL;
  }
}

// qkv = window_partition(roll(pad(x))) W^T + bias without the partitioned copy: the Swin block's first product reads its A rows straight
// from the NHWC map through the window order (SwinTransformerBlock.forward, swin_transformer_v2.py:236-262 -> WindowAttention.forward
// :153-155).  X [B, H, Wd, lda >= K]; C [B * Hp * Wp, ldc] in window order (Hp, Wp: H, Wd rounded up to multiples of ws).  Exact fp32
// MFMA only (the split-precision kernels do not know the row map: MDQE_EINVAL in that mode; the host then partitions first).
extern "C" int mdqe_gemm_nt_swin_f32(const float* X, long lda, const float* W, const float* bias, float* C, long ldc, int B, int H, int Wd,
                                     int ws, int shift, int N, int K, void* stream) {
  MDQE_REQUIRE(B >= 0 && H > 0 && Wd > 0 && ws > 0 && shift >= 0 && shift < ws && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && lda >= K &&
               ldc >= N);
  if (g_gemm_precision != 0) return MDQE_EINVAL;
  const int Hp = (H + ws - 1) / ws * ws, Wp = (Wd + ws - 1) / ws * ws;
  const long Ml = (long)B * Hp * Wp;
  if (Ml == 0) return MDQE_OK;
  MDQE_REQUIRE(Ml < 0x7FFFFFFFL);
  MDQE_CHECK_PTR(X); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C);
  MDQE_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0);
  const long ab = (((long)B * H * Wd - 1) * lda + K) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  GemmParams p = {};
  p.A = X; p.W = W; p.C = C; p.M = (int)Ml; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.conv = 0;
  p.bias = bias; p.act = MDQE_ACT_NONE;
  p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.ksplit = 1; p.kchunk = K;
  p.swin_ws = ws; p.swin_shift = shift; p.swin_H = H; p.swin_W = Wd;
  mdqe_clear_error();
  return dispatch_gemm(p, 0, (hipStream_t)stream);
}

// C = LayerNorm(A W^T + bias + residual) * gamma + beta over N == 256 columns, ONE kernel (64x256 tile: the block owns whole
// rows, statistics in the epilogue).  Always exact fp32 MFMA.  C may alias the residual.
extern "C" int mdqe_gemm_ln_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N,
                                int K, const float* residual, long ldr, const float* gamma, const float* beta, float eps,
                                void* stream) {
  MDQE_REQUIRE(M >= 0 && N == 256 && K > 0 && K % 4 == 0 && lda % 4 == 0 && lda >= K && ldc >= N && ldc % 4 == 0);
  if (M == 0) return MDQE_OK;
  MDQE_CHECK_PTR(A); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C); MDQE_CHECK_PTR(gamma); MDQE_CHECK_PTR(beta);
  MDQE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)gamma |
                 (uintptr_t)beta) & 15) == 0);
  MDQE_REQUIRE(residual == nullptr || (ldr >= N && ldr % 4 == 0));
  const long ab = ((long)(M - 1) * lda + K) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  // (the epilogue addresses C and the residual through 32-bit buffer offsets)
  MDQE_REQUIRE((long)M * ldc * 4 < 0xFFFF0000L && (residual == nullptr || (long)M * ldr * 4 < 0xFFFF0000L));
  GemmParams p = {};
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.conv = 0;
  p.bias = bias; p.residual = residual; p.ldr = ldr; p.act = MDQE_ACT_NONE;
  p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.ksplit = 1; p.kchunk = K; p.vec_ok = 1;
  p.ln_g = gamma; p.ln_b = beta; p.ln_eps = eps;
  mdqe_clear_error();
  return mdqe_launch_gemm_k16(p, 6, (hipStream_t)stream);
}

// ... with a SECOND LayerNorm of the result in the same epilogue: C = LN(A W^T + bias + residual) * gamma + beta and
// C2 = LN(C) * gamma2 + beta2 -- the decoder's `x = norm3(x + ffn(x))` followed by the shared `decoder_norm(x)` that feeds the box head
// (transformer_dec.py:352-358,492-495).  C2 equals mdqe_layernorm_f32 applied to C bit for bit (same reduction tree).
extern "C" int mdqe_gemm_ln2_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N,
                                 int K, const float* residual, long ldr, const float* gamma, const float* beta, const float* gamma2,
                                 const float* beta2, float* C2, long ldc2, float eps, void* stream) {
  MDQE_REQUIRE(M >= 0 && N == 256 && K > 0 && K % 4 == 0 && lda % 4 == 0 && lda >= K && ldc >= N && ldc % 4 == 0 && ldc2 >= N && ldc2 % 4 == 0);
  if (M == 0) return MDQE_OK;
  MDQE_CHECK_PTR(A); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C); MDQE_CHECK_PTR(gamma); MDQE_CHECK_PTR(beta);
  MDQE_CHECK_PTR(gamma2); MDQE_CHECK_PTR(beta2); MDQE_CHECK_PTR(C2);
  MDQE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)gamma |
                 (uintptr_t)beta | (uintptr_t)gamma2 | (uintptr_t)beta2 | (uintptr_t)C2) & 15) == 0);
  MDQE_REQUIRE(residual == nullptr || (ldr >= N && ldr % 4 == 0));
  const long ab = ((long)(M - 1) * lda + K) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  MDQE_REQUIRE((long)M * ldc * 4 < 0xFFFF0000L && (long)M * ldc2 * 4 < 0xFFFF0000L && (residual == nullptr || (long)M * ldr * 4 < 0xFFFF0000L));
  GemmParams p = {};
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.conv = 0;
  p.bias = bias; p.residual = residual; p.ldr = ldr; p.act = MDQE_ACT_NONE;
  p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.ksplit = 1; p.kchunk = K; p.vec_ok = 1;
  p.ln_g = gamma; p.ln_b = beta; p.ln_eps = eps;
  p.ln2_g = gamma2; p.ln2_b = beta2; p.C2 = C2; p.ldc2 = ldc2;
  mdqe_clear_error();
  return mdqe_launch_gemm_k16(p, 6, (hipStream_t)stream);
}


// ---- N <= 8 output columns (the decoder's box head 256 -> 4 and time weights 256 -> 1, on 31 360 rows a pass): a 128x64 MFMA tile
// computes 64 columns to keep 4, and the launch is bound by reading A anyway.  Here a wave takes 4 rows at a time, a lane 4
// consecutive k of each (one coalesced 1-KB read per row and 256 k), the N weight rows come from L1, and the partial dot products
// are summed across the wave with xor-shuffles.  fp32 FMA arithmetic; the summation order differs from the MFMA kernel's
// (the same rows always take the same path: N and K decide, not M).
template <int NN>
__global__ void __launch_bounds__(256)
rows_dot_kernel(const float* __restrict__ A, long lda, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ C,
                long ldc, int M, int K, int act, int act_cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long r0 = wave * 4; r0 < M; r0 += nwaves * 4) {
    float acc[4][NN];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[r][n] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 256) {
      f32x4 a[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = r0 + r < M ? r0 + r : M - 1;
        a[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + row * lda + k0 + lane * 4));
      }
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)n * K + k0 + lane * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[r][n] = fmaf(a[r][3], w[3], fmaf(a[r][2], w[2], fmaf(a[r][1], w[1], fmaf(a[r][0], w[0], acc[r][n]))));
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        float v = acc[r][n];
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
        acc[r][n] = v;
      }
    if (lane < 4 && r0 + lane < M) {               // lane r writes row r0 + r
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        float v = lane == 0 ? acc[0][n] : lane == 1 ? acc[1][n] : lane == 2 ? acc[2][n] : acc[3][n];
        if (bias != nullptr) v += bias[n];
        if (act != MDQE_ACT_NONE && (act_cols <= 0 || n < act_cols)) v = mdqe_act(v, act);
        C[(r0 + lane) * ldc + n] = v;
      }
    }
  }
}

extern "C" int mdqe_gemm_nt_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc,
                                int M, int N, int K, int act, int act_cols, const float* residual, long ldr, int res_mod,
                                int res_first, const unsigned char* rowmask, int mask_cols, int tile, int ksplit,
                                float* splitk_ws, const void* w_split, void* stream) {
  MDQE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && lda >= K && ldc >= N);
  if (M == 0) return MDQE_OK;
  MDQE_CHECK_PTR(A); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C);
  MDQE_REQUIRE((((uintptr_t)A | (uintptr_t)W) & 15) == 0);
  const long ab = ((long)(M - 1) * lda + K) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  GemmParams p = {};
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.conv = 0;
  p.bias = bias; p.residual = residual; p.ldr = ldr; p.res_mod = res_mod; p.res_first = res_first; p.rowmask = rowmask; p.mask_cols = mask_cols;
  p.act = act; p.act_cols = act_cols; p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
  p.ksplit = 1; p.kchunk = K; p.ws = nullptr;
  if (w_split != nullptr) { p.Wh = w_split; p.Wl = (const char*)w_split + (long)N * K * 2; }
  if (ksplit > 1) {
    MDQE_CHECK_PTR(splitk_ws);
    int kc = (K + ksplit - 1) / ksplit; kc = (kc + 31) / 32 * 32;
    const int ks = (K + kc - 1) / kc;
    if (ks > 1) { p.ksplit = ks; p.kchunk = kc; p.ws = splitk_ws; }
  }
  mdqe_clear_error();
  if (N <= 8 && K % 256 == 0 && tile == 0 && p.ksplit <= 1 && residual == nullptr && rowmask == nullptr && g_gemm_rows_dot) {
    const long waves = ((long)M + 3) / 4;
    long nb = (waves + 3) / 4;
    if (nb > 256L * 8) nb = 256L * 8;
    hipStream_t st = (hipStream_t)stream;
#define ROWS_DOT(NN) hipLaunchKernelGGL((rows_dot_kernel<NN>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, W, bias, C, ldc, M, K, act, act_cols)
    switch (N) {
      case 1: ROWS_DOT(1); break; case 2: ROWS_DOT(2); break; case 3: ROWS_DOT(3); break; case 4: ROWS_DOT(4); break;
      case 5: ROWS_DOT(5); break; case 6: ROWS_DOT(6); break; case 7: ROWS_DOT(7); break; default: ROWS_DOT(8); break;
    }
#undef ROWS_DOT
    return mdqe_launch_status();
  }
  return dispatch_gemm(p, tile, (hipStream_t)stream);
}


// C[m, n < side_cols] += side[m, 0:4] . side_w[n, 0:4] as a pass of its own: what the GEMM kernels without the in-epilogue
// side term (split-precision tiles, the K-step-32 form) are followed by.
__global__ void __launch_bounds__(256)
side_add_kernel(const float* __restrict__ side, const float* __restrict__ side_w, float* __restrict__ C, long ldc, long M, int cols) {
  const int c4n = cols / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M * c4n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / c4n; const int n = (int)(i % c4n) * 4;
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(side + m * 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(C + m * ldc + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(side_w + (long)(n + e) * 4);
      v[e] += (s4[0] * w4[0] + s4[1] * w4[1]) + (s4[2] * w4[2] + s4[3] * w4[3]);
    }
    *reinterpret_cast<f32x4*>(C + m * ldc + n) = v;
  }
}

// C = A W^T + bias, and C[:, n < side_cols] += side [M, 4] x side_w[n, 0:4]^T -- a projection of `x + pos` where pos is itself a
// linear function of four numbers per row (the decoder's query position embedding, point2pos_proj(box centre),
// mdqe/models/transformer_dec.py:469,480,495,503 feeding :348-353, :397-402 and the sampling-offset / attention-weight projections
// of ms_deform_attn.py): `(x + pos) W^T = x W^T + box (W P)^T + W b_P`, with `W P` [N, 4] and the bias folded on the host once.
// One launch instead of add + GEMM, the [M, C] position tensor is never written; several projections of the same x with and
// without the position (q, k | v) become ONE product with side_cols marking the columns that take it.
extern "C" int mdqe_gemm_nt_side_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K,
                                     const float* side, const float* side_w, int side_cols, const void* w_split, void* stream) {
  MDQE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && lda >= K && ldc >= N && ldc % 4 == 0);
  MDQE_REQUIRE(side_cols >= 0 && side_cols <= N && side_cols % 4 == 0 && N % 4 == 0);
  if (M == 0) return MDQE_OK;
  MDQE_CHECK_PTR(A); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C);
  if (side_cols > 0) { MDQE_CHECK_PTR(side); MDQE_CHECK_PTR(side_w); }
  MDQE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)side | (uintptr_t)side_w) & 15) == 0);
  const long ab = ((long)(M - 1) * lda + K) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  GemmParams p = {};
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.conv = 0;
  p.bias = bias; p.act = MDQE_ACT_NONE; p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
  p.ksplit = 1; p.kchunk = K; p.ws = nullptr;
  if (w_split != nullptr) { p.Wh = w_split; p.Wl = (const char*)w_split + (long)N * K * 2; }
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  const bool in_epilogue = side_cols > 0 && g_gemm_precision == 0 && g_gemm_variant != 0;      // the K-step-16 fp32 kernel takes it itself
  if (in_epilogue) { p.side = side; p.side_w = side_w; p.side_cols = side_cols; }
  const int rc = dispatch_gemm(p, 0, st);
  if (rc != MDQE_OK || in_epilogue || side_cols == 0) return rc;
  long nb = ((long)M * (side_cols / 4) + 255) / 256; if (nb > 256L * 16) nb = 256L * 16;
  hipLaunchKernelGGL(side_add_kernel, dim3((unsigned)nb), dim3(256), 0, st, side, side_w, C, ldc, (long)M, side_cols);
  return mdqe_launch_status();
}

// C = act([A1 | A2'] W^T + bias): the last 1x1 conv of a ResNet bottleneck and its projection shortcut as ONE product -- W = [W3 | Ws]
// along K, bias = b3 + bs -- so the shortcut's output (as wide as the block's output) is never written and read back.  A1: [M, K1]
// rows (pitch lda1); A2: the block's input, NHWC [NI, H2, W2, lda2 >= K2], read at pixel (oh*stride, ow*stride) for output row
// m = (img, oh, ow) (the shortcut's stride).  Exact fp32 MFMA only (the split-precision mode keeps the two-launch form).
extern "C" int mdqe_gemm_nt_cat2_f32(const float* A1, long lda1, int K1, const float* A2, long lda2, int K2, int NI, int OH, int OW,
                                     int H2, int W2, int stride, const float* W, const float* bias, float* C, long ldc, int N, int act,
                                     void* stream) {
  MDQE_REQUIRE(NI >= 0 && OH > 0 && OW > 0 && H2 > 0 && W2 > 0 && stride > 0 && N > 0 && K1 > 0 && K2 > 0);
  MDQE_REQUIRE(K1 % 16 == 0 && K2 % 16 == 0 && lda1 % 4 == 0 && lda2 % 4 == 0 && lda1 >= K1 && lda2 >= K2 && ldc >= N);
  MDQE_REQUIRE((long)(OH - 1) * stride < H2 && (long)(OW - 1) * stride < W2);
  const long Ml = (long)NI * OH * OW;
  MDQE_REQUIRE(Ml < 0x7FFFFFFFL);
  if (Ml == 0) return MDQE_OK;
  MDQE_CHECK_PTR(A1); MDQE_CHECK_PTR(A2); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(C);
  MDQE_REQUIRE((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)W) & 15) == 0);
  const int M = (int)Ml, K = K1 + K2;
  const long ab = ((long)(M - 1) * lda1 + K1) * 4, a2b = (((long)NI * H2 * W2 - 1) * lda2 + K2) * 4, wb = (long)N * K * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && a2b < 0xFFFFFFF0L && wb < 0xFFFFFFF0L);
  GemmParams p = {};
  p.A = A1; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda1; p.ldc = ldc; p.conv = 0;
  p.A2 = A2; p.lda2 = lda2; p.K1 = K1; p.a2_bytes = (unsigned)a2b; p.OH = OH; p.OW = OW; p.H = H2; p.Wd = W2; p.stride = stride;
  p.bias = bias; p.act = act; p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.ksplit = 1; p.kchunk = K;
  p.stamps = g_gemm_stamps;
  p.vec_ok = ((((uintptr_t)C | (uintptr_t)bias) & 15) == 0) && (ldc % 4 == 0);
  const long b128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  int tile = 3;                                          // the plain GEMM's rule (dispatch_gemm)
  if (N <= 64) tile = (M >= 4096) ? 2 : 3;
  else if (b128 >= 2000 && (N >= 1024 || K >= 1024)) tile = 1;
  else if (b128 >= 2000 && (N > 256 || K > 256)) tile = 2;
  mdqe_clear_error();
  return mdqe_launch_gemm_k16(p, tile, (hipStream_t)stream);
}

extern "C" int mdqe_conv2d_nhwc_f32(const float* X, long x_img_stride, const float* Wt, const float* bias, float* Y, long ldy,
                                    int NI, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                    int act, const float* residual, long ldr, int res_first, int tile, const void* w_split,
                                    int ksplit, float* splitk_ws, void* stream) {
  MDQE_REQUIRE(NI >= 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  MDQE_REQUIRE(Cin % 32 == 0);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
  MDQE_REQUIRE(OH > 0 && OW > 0 && ldy >= Cout);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(X); MDQE_CHECK_PTR(Wt); MDQE_CHECK_PTR(Y);
  MDQE_REQUIRE((((uintptr_t)X | (uintptr_t)Wt) & 15) == 0);
  if (x_img_stride <= 0) x_img_stride = (long)H * Wd * Cin;
  MDQE_REQUIRE(x_img_stride % 4 == 0 && x_img_stride >= (long)H * Wd * Cin);
  const long ab = ((long)(NI - 1) * x_img_stride + (long)H * Wd * Cin) * 4, wb = (long)Cout * KH * KW * Cin * 4;
  MDQE_REQUIRE(ab < 0xFFFFFFF0L && wb < 0xFFFFFFF0L && (long)NI * OH * OW < 0x7FFFFFFFL);
  GemmParams p = {};
  p.A = X; p.W = Wt; p.C = Y; p.M = NI * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.lda = 0; p.ldc = ldy;
  p.conv = 1; p.H = H; p.Wd = Wd; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.bias = bias; p.residual = residual; p.ldr = ldr; p.res_mod = 0; p.res_first = res_first; p.img_stride = x_img_stride; p.rowmask = nullptr; p.mask_cols = 0;
  p.act = act; p.act_cols = 0; p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
  p.ksplit = 1; p.kchunk = p.K; p.ws = nullptr;
  if (ksplit > 1) {                                  // deep-K conv on few output pixels: spread K over the CUs (chunks of whole taps x 32)
    MDQE_CHECK_PTR(splitk_ws);
    int kc = (p.K + ksplit - 1) / ksplit; kc = (kc + 31) / 32 * 32;
    const int ks = (p.K + kc - 1) / kc;
    if (ks > 1) { p.ksplit = ks; p.kchunk = kc; p.ws = splitk_ws; }
  }
  if (w_split != nullptr && p.ksplit <= 1) { p.Wh = w_split; p.Wl = (const char*)w_split + (long)Cout * p.K * 2; }
  mdqe_clear_error();
  return dispatch_gemm(p, tile, (hipStream_t)stream);
}
