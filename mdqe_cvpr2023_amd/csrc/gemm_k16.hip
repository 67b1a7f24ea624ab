// fp32 NT GEMM / implicit-GEMM conv, K-step 16 variant of gemm.hip (same arithmetic: v_mfma_f32_32x32x2_f32, exact fp32).
//
// Why a second form: with 32-float K-steps a 128x128 block needs 64 KB of LDS (two stages, and the same again for the
// epilogue restage), so only two blocks -- two waves per SIMD -- fit a CU, and whenever one of them is in its prologue or
// epilogue the other wave alone cannot keep the matrix pipe busy.  Here a stage is 16 floats deep (64-B rows, 16 rows per
// 1-KiB LDS-DMA instruction, 16-B chunks XOR-swizzled by (row>>2)&3 -> conflict-free ds_read_b128) and the epilogue is
// restaged per WAVE through a private 4-KB slice (32x32 sub-tile at a time, no block barrier): 32 KB per 128x128 block,
// four blocks = four waves per SIMD resident, so prologues, epilogues and barrier waits of one block hide behind the
// MFMAs of the others.
//
// Round 4 -- the rule everything below is written to: v_mfma_f32_32x32x2_f32 runs at the VECTOR rate and holds the SIMD's vector
// issue, so a vector instruction of ANY wave on the SIMD is matrix time lost (in-kernel stamps: beside three waves in their K loops a
// wave in its epilogue issues about one instruction per MFMA; counters: MFMA pipe busy 0.775 -> 0.880 with 61 % fewer non-MFMA vector
// instructions, profiles/r04_pmc_gemm_epilogue.txt).  Hence: the K loop unrolled by its two LDS stages (fragment reads = base +
// immediate) with the ragged-K check behind a uniform branch; interior tiles of plain products on a few-instruction epilogue (buffer
// addressing with scalar offsets, one specialised copy per form behind uniform branches); the LayerNorm epilogue on buffer resources
// whose range check replaces the row guards.  Uniform conditions are branches, never selects.
#include "common.h"
#include "gemm_params.h"
#include <type_traits>

// NS = LDS stages.  2 (one K-step of look-ahead) is the default: several blocks share a SIMD and one block's load latency hides
// behind the others' MFMAs.  NS = 4 (three K-steps in flight, `s_waitcnt vmcnt` counting the stages issued behind the one about
// to be consumed) was built for launches of about one block per CU (rocprof: 14 us for the 16 K-steps of a
// [5292,256]x[256,256] product) and measured at no gain, so it stays behind mdqe_debug_gemm_stages(4).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// tools/ A/B: mdqe_debug_gemm_fast_epilogue(0) sends every tile through the general epilogue (GemmParams::fast_epi)
__device__ __forceinline__ bool g_k16_fast_epilogue_on(const GemmParams& p) { return p.fast_epi != 0; }

template <int BM, int BN, int WM, int WN, bool CONV, bool LN = false, int NS = 2, bool CAT = false>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN > 4) ? 2 : 4)
gemm_nt_f32_k16_kernel(const GemmParams p) {
  static_assert(!CAT || (!CONV && !LN), "cat mode: plain tiles only");
  constexpr int NW = WM * WN;
  constexpr int BK = 16;
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int ROWS = BM + BN;              // A rows then W rows in one LDS image
  constexpr int NINST = ROWS / 16;           // 1-KiB LDS-DMA instructions per stage (16 rows x 64 B)
  constexpr int IPW = NINST / NW;            // per wave
  static_assert(NINST % NW == 0 && BM % 16 == 0, "tile rows must split evenly over waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // max(2 * ROWS * 16, NW * 1024) floats

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  unsigned long long t_start = 0, t_pro = 0, t_loop = 0;   // tools/gemm_stamps.py only (100 MHz wall clock)
  if (p.stamps) t_start = wall_clock64();
  if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 1024 && blockIdx.y == 0) {
    // The blocks that share a CU start together and, tiles being equal, stay in lockstep for the whole launch: all of them are in
    // their epilogue at the same time and the matrix pipe idles.  The k-th block of a CU's first round (dispatch fills the 256 CUs
    // once per 256 blocks) therefore waits k x stagger; blocks start when a predecessor retires, so the offset carries on.
    // Measured (tools/gemm_stagger_ab.py, r02): +9 % on FFN1 + GELU when the same GEMM is launched back to back (100 -> 109 TF),
    // nothing inside the pipeline (107 TF either way, tools/frame_gemm_table.py; bench 700 vs 699 frames/s): there the blocks
    // of a launch start as the previous, different kernel's blocks retire and are staggered already.  Off by default.
    const unsigned long long until = wall_clock64() + (unsigned long long)((blockIdx.x >> 8) * p.stagger);
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
  }

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
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CAT ? p.A2 : p.A), 0, CAT ? p.a2_bytes : p.a_bytes, 0x00020000);

  // instruction j of this wave covers image rows (wave*IPW + j)*16 .. +15; lane -> row += lane>>2, LDS chunk lane&3,
  // source chunk = (lane&3) ^ ((row>>2)&3).  voff[j] = byte offset of (row, k = 4*chunk); the K-step rides on the scalar
  // offset (plain rows, W rows) or on the filter-tap offset in the VGPR (conv A rows: a padded pixel's base may wrap).
  unsigned voff[IPW], voff2[IPW];
  int ih0[IPW], iw0[IPW], kch[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int irow = (wave * IPW + j) * 16 + (lane >> 2);
    kch[j] = ((lane & 3) ^ ((irow >> 2) & 3)) * 4;
    ih0[j] = 0; iw0[j] = 0;
    if (irow < BM) {
      int m = m0 + irow; if (m > p.M - 1) m = p.M - 1;
      if (CONV) {
        const int ow = m % p.OW; const int t = m / p.OW; const int oh = t % p.OH; const int img = t / p.OH;
        ih0[j] = oh * p.stride - p.pad; iw0[j] = ow * p.stride - p.pad;
        voff[j] = (unsigned)(((long)img * p.img_stride + ((long)ih0[j] * p.Wd + iw0[j]) * p.Cin) * 4) + (unsigned)(kch[j] * 4);   // may wrap
      } else if (!CAT && !LN && p.swin_ws > 0) {   // window-ordered row m -> its pixel of the [B, H, W, lda] map (or zeros)
        const int ws = p.swin_ws, nWx = (p.swin_W + ws - 1) / ws, nWy = (p.swin_H + ws - 1) / ws;
        const int ix = m % ws; int t = m / ws;
        const int iy = t % ws; t /= ws;
        const int wx = t % nWx; t /= nWx;
        const int wy = t % nWy; const int b = t / nWy;
        const int y = (wy * ws + iy + p.swin_shift) % (nWy * ws), x = (wx * ws + ix + p.swin_shift) % (nWx * ws);
        voff[j] = (y < p.swin_H && x < p.swin_W) ? (unsigned)((((long)b * p.swin_H + y) * p.swin_W + x) * p.lda * 4) + (unsigned)(kch[j] * 4) : OOB_OFF;
      } else {
        voff[j] = (unsigned)((long)m * p.lda * 4) + (unsigned)(kch[j] * 4);
        if constexpr (CAT) {                       // the second operand's row: pixel (oh*stride, ow*stride) of image img
          const int ow = m % p.OW; const int t = m / p.OW; const int oh = t % p.OH; const int img = t / p.OH;
          const long r2 = ((long)img * p.H + (long)oh * p.stride) * p.Wd + (long)ow * p.stride;
          voff2[j] = (unsigned)(r2 * p.lda2 * 4) + (unsigned)(kch[j] * 4);
        }
      }
      if constexpr (!CAT) voff2[j] = 0;
    } else {
      voff2[j] = 0;
      int n = n0 + irow - BM; if (n > p.N - 1) n = p.N - 1;
      voff[j] = (unsigned)((long)n * p.K * 4) + (unsigned)(kch[j] * 4);
    }
  }

  const int kbeg = p.ksplit > 1 ? blockIdx.y * p.kchunk : 0;
  const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const int kt0 = kbeg / BK;
  const int nk = (kend - kbeg + BK - 1) / BK;
  int t_kh = 0, t_kw = 0, t_c = 0;                 // conv: filter tap of the K-step about to be issued (Cin % 16 == 0)
  if (CONV) { const int tap = kbeg / p.Cin; t_c = kbeg - tap * p.Cin; t_kh = tap / p.KW; t_kw = tap - t_kh * p.KW; }

  auto issue_impl = [&](int kt, int buf, auto tail_) __attribute__((always_inline)) {
    constexpr bool ktail = decltype(tail_)::value;   // the last K-step of a ragged K checks lanes against K
    const int k0 = kt * BK;
    float* base = lds + buf * (ROWS * BK);
    int tap_off = 0;
    if (CONV) tap_off = ((t_kh * p.Wd + t_kw) * p.Cin + t_c) * 4;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int irow0 = (wave * IPW + j) * 16;       // wave-uniform
      unsigned off = voff[j];
      if (irow0 < BM) {
        if (CONV) {
          const int ih = ih0[j] + t_kh, iw = iw0[j] + t_kw;
          const bool ok = (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.Wd);
          off = ok ? off + (unsigned)tap_off : OOB_OFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + irow0 * BK), 16, off, 0, 0, 0);
        } else if (CAT && k0 >= p.K1) {              // (wave-uniform: K1 is a multiple of the K-step)
          off = voff2[j];
          if (ktail && k0 + kch[j] >= p.K) off = OOB_OFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (__attribute__((address_space(3))) void*)(base + irow0 * BK), 16, off, (k0 - p.K1) * 4, 0, 0);
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
  // Two copies behind a UNIFORM branch: folded into one, the ragged-K check is a compare + select per LDS-DMA address in EVERY step,
  // and every vector instruction of this loop is matrix-pipe time (the f32 MFMA runs at the vector rate).  The empty asm keeps the
  // optimiser from merging the copies back.
  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
    if constexpr (CONV) { issue_impl(kt, buf, std::false_type{}); }          // (Cin % 16 == 0: a conv's K is never ragged)
    else if (kt * BK + BK > p.K) { asm volatile("; ragged last K-step" ::: "memory"); issue_impl(kt, buf, std::true_type{}); }
    else issue_impl(kt, buf, std::false_type{});
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (floats): row*16 + ((chunk ^ (row>>2)&3) << 2), chunk = 2*kk + lh
  int fa[MT], fb[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int row = wm * (BM / WM) + i * 32 + lr; fa[i] = row * BK + ((lh ^ ((row >> 2) & 3)) << 2); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int row = BM + wn * (BN / WN) + j * 32 + lr; fb[j] = row * BK + ((lh ^ ((row >> 2) & 3)) << 2); }
  int fa1[MT], fb1[NT];                            // the second 8-wide half of a K-step (chunk bit 1 flipped)
#pragma unroll
  for (int i = 0; i < MT; ++i) fa1[i] = fa[i] ^ 8;
#pragma unroll
  for (int j = 0; j < NT; ++j) fb1[j] = fb[j] ^ 8;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(kt0 + s, s);
  if (p.stamps) t_pro = wall_clock64();
  // One K-step on LDS stage S (compile-time: the fragment reads of a stage are then `base register + immediate`, where a run-time
  // stage costs one vector add per fragment address per step).
  auto kstep2 = [&](int kt, auto S_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    wait_vmcnt<0>();
    __syncthreads();
    if (kt + 1 < nk) issue(kt0 + kt + 1, S ^ 1);
    const float* sI = lds + S * (ROWS * BK);
    f32x4 a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const f32x4*>(sI + fa[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) b0[j] = *reinterpret_cast<const f32x4*>(sI + fb[j]);
#pragma unroll
    for (int i = 0; i < MT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(sI + fa1[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) b1[j] = *reinterpret_cast<const f32x4*>(sI + fb1[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][s], b0[j][s], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][s], b1[j][s], acc[i][j], 0, 0, 0);
  };
  if constexpr (NS == 2) {
    for (int kt = 0; kt < nk; kt += 2) {
      kstep2(kt, std::integral_constant<int, 0>{});
      if (kt + 1 < nk) kstep2(kt + 1, std::integral_constant<int, 1>{});
    }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's LDS-DMA for step kt has landed: at most the loads of the stages issued after it may still be in flight
    if constexpr (NS == 2) {
      wait_vmcnt<0>();
    } else if constexpr (NS == 3) {
      if (nk - 1 - kt >= 1) wait_vmcnt<IPW>(); else wait_vmcnt<0>();     // step kt + 1 may still be in flight
    } else {
      const int behind = nk - 1 - kt;                  // stages already issued behind step kt: min(NS - 2, behind)
      if (behind >= 2) wait_vmcnt<2 * IPW>(); else if (behind == 1) wait_vmcnt<IPW>(); else wait_vmcnt<0>();
    }
    __syncthreads();                                   // ... and everybody else's; all reads of the buffer refilled next are done
    if (kt + NS - 1 < nk) issue(kt0 + kt + NS - 1, (kt + NS - 1) % NS);
    const float* sI = lds + (kt % NS) * (ROWS * BK);
    f32x4 a0[MT], b0[NT], a1[MT], b1[NT];              // both 8-wide halves up front: the second hides behind 16 MFMAs
#pragma unroll
    for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const f32x4*>(sI + fa[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) b0[j] = *reinterpret_cast<const f32x4*>(sI + fb[j]);
#pragma unroll
    for (int i = 0; i < MT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(sI + (fa[i] ^ 8));
#pragma unroll
    for (int j = 0; j < NT; ++j) b1[j] = *reinterpret_cast<const f32x4*>(sI + (fb[j] ^ 8));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][s], b0[j][s], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][s], b1[j][s], acc[i][j], 0, 0, 0);
  }

  if (p.stamps) t_loop = wall_clock64();
  if constexpr (LN) {
    // ---- LayerNorm epilogue (BN == N: the block owns whole rows): y = LN(acc + bias + residual) * gamma + beta.
    // Sub-tiles are restaged as below (a lane gets 4 consecutive columns of 4 rows per sub-tile) but stay in registers;
    // row sums go across the 8 lanes of a row, then across the WN waves through LDS; two passes (mean, then centred
    // squares) as in layernorm_kernel.  C may alias the residual: a wave only rewrites the elements it read.
    static_assert(!LN || (WM == 1 && BN == 256), "LN epilogue: one wave row, 256 columns");
    __syncthreads();
    float* sC = lds + wave * 1024;
    float* red1 = lds + NW * 1024;
    float* red2 = red1 + NW * BM;
    f32x4 vv[MT][NT][4];
    float rs[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it) rs[i][it] = 0.f;
    const int c4 = lane & 7, rsub = lane >> 3;
    // (round 4) residual rows and the stores of C / C2 go through buffer resources sized to M rows: ONE per-lane byte offset + a scalar
    // offset per (sub-tile, trip, column block), and the hardware's range check stands in for every `m < M` test -- a row past the end
    // reads zeros and its store is dropped.  No 64-bit address arithmetic, no exec-mask branches: each vector instruction here is matrix
    // time of the three other waves on the SIMD (see the fast epilogue below).  The launcher guarantees M * ld * 4 < 4 GB.
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual != nullptr ? p.residual : p.C), 0,
                                                       (int)(unsigned)((long)p.M * (p.residual != nullptr ? p.ldr : p.ldc) * 4), 0x00020000);
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)(unsigned)((long)p.M * p.ldc * 4), 0x00020000);
    const unsigned vR = (unsigned)((rsub * p.ldr + c4 * 4) * 4), vC = (unsigned)((rsub * p.ldc + c4 * 4) * 4);
    auto stage = [&](auto i_, auto j_) __attribute__((always_inline)) {
      constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int n = wn * 64 + j * 32 + c4 * 4;
      const f32x4 bv = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * 32 + c4 * 4) + bv;
        if (p.residual != nullptr && m0 + i * 32 + it * 8 < p.M) {  // (wave-uniform; a group wholly past M is never addressed, see the stores)
          const int so = (int)((((long)m0 + i * 32 + it * 8) * p.ldr + wn * 64 + j * 32) * 4);
          v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vR, so, 0));
        }
        vv[i][j][it] = v;
        rs[i][it] += (v[0] + v[1]) + (v[2] + v[3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    stage(I0{}, I0{}); stage(I0{}, I1{}); stage(I1{}, I0{}); stage(I1{}, I1{});
    float mean[MT][4], rstd[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float x = rs[i][it];
        x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4);
        if (c4 == 0) red1[wave * BM + i * 32 + it * 8 + rsub] = x;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = i * 32 + it * 8 + rsub;
        mean[i][it] = ((red1[row] + red1[BM + row]) + (red1[2 * BM + row] + red1[3 * BM + row])) * (1.f / 256.f);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 d = vv[i][j][it] - mean[i][it];
          q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
        if (c4 == 0) red2[wave * BM + row] = q;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = i * 32 + it * 8 + rsub;
        rstd[i][it] = rsqrtf(((red2[row] + red2[BM + row]) + (red2[2 * BM + row] + red2[3 * BM + row])) * (1.f / 256.f) + p.ln_eps);
      }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wn * 64 + j * 32 + c4 * 4;
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_g + n);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + n);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          vv[i][j][it] = (vv[i][j][it] - mean[i][it]) * rstd[i][it] * g + b;
          // Rows past M: the scalar offset `so` carries the 8-row group's base, the lane offset vC the row within the group, and the buffer
          // range check is `vC >= num_records - so` in unsigned arithmetic -- sound only while so < num_records.  A group that starts at or
          // past row M (last tile of a ragged M) is therefore skipped by this wave-uniform branch instead of being left to the check; a
          // group that straddles M has so < num_records and its lanes past M are dropped by the hardware.
          if (m0 + i * 32 + it * 8 >= p.M) continue;
          const int so = (int)((((long)m0 + i * 32 + it * 8) * p.ldc + n - c4 * 4) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, vv[i][j][it]), rsC, vC, so, 0);
          __builtin_amdgcn_sched_barrier(0);                       // (store-data hazard, see the fast epilogue)
          asm volatile("s_nop 1" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.ln2_g != nullptr) {
      // second LayerNorm, of the values just written (they are still in vv): the same two passes over the same reduction tree --
      // a lane's two chunks in stage order, the 8 lanes of a 32-column group, the four wave slices as (w0 + w1) + (w2 + w3)
      __syncthreads();                               // every wave has read red1 / red2 of the first LayerNorm
      const auto rsC2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.C2, 0, (int)(unsigned)((long)p.M * p.ldc2 * 4), 0x00020000);
      const unsigned vC2 = (unsigned)((rsub * p.ldc2 + c4 * 4) * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float x = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) { const f32x4 v = vv[i][j][it]; x += (v[0] + v[1]) + (v[2] + v[3]); }
          x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4);
          if (c4 == 0) red1[wave * BM + i * 32 + it * 8 + rsub] = x;
        }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = i * 32 + it * 8 + rsub;
          mean[i][it] = ((red1[row] + red1[BM + row]) + (red1[2 * BM + row] + red1[3 * BM + row])) * (1.f / 256.f);
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f32x4 d = vv[i][j][it] - mean[i][it];
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
          }
          q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
          if (c4 == 0) red2[wave * BM + row] = q;
        }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = i * 32 + it * 8 + rsub;
          rstd[i][it] = rsqrtf(((red2[row] + red2[BM + row]) + (red2[2 * BM + row] + red2[3 * BM + row])) * (1.f / 256.f) + p.ln_eps);
        }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = wn * 64 + j * 32 + c4 * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln2_g + n);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln2_b + n);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            if (m0 + i * 32 + it * 8 >= p.M) continue;              // (as above: never rely on the range check with so >= num_records)
            const f32x4 y2 = (vv[i][j][it] - mean[i][it]) * rstd[i][it] * g + b;
            const int so = (int)((((long)m0 + i * 32 + it * 8) * p.ldc2 + n - c4 * 4) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y2), rsC2, vC2, so, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
    return;
  }

  // ---- epilogue: per-wave restage of one 32x32 sub-tile at a time through the wave's own 4-KB LDS slice ----------------
  // acc[i][j][r] is C(row = (r&3) + 8*(r>>2) + 4*lh, col = lr) of the sub-tile; after the restage a lane owns 4 consecutive
  // columns of a row and 8 lanes cover one 128-B line of C.
  __syncthreads();                                   // every wave is done reading the last K-step
  float* sC = lds + wave * 1024;
  const bool vec = p.vec_ok;

  // ---- fast path (round 4): an interior tile of a plain product -- bias, none / ReLU / GELU on every column, optional full residual.
  // The f32 MFMA runs at the vector rate on the SIMD's own lanes: while the three other waves of a SIMD are in their K loops, a wave
  // in its epilogue gets a vector issue slot about once per MFMA (in-kernel stamps: 30 us for the ~2000 vector instructions of the
  // general epilogue below, 16 K-steps take 46), and every vector instruction it does issue is matrix time lost.  So this path
  // spends as few as it can: the C / residual addresses are ONE per-lane byte offset into a buffer resource + a scalar offset per
  // (sub-tile, trip) -- no per-store address arithmetic, no bounds compares (the tile is interior), no masks, no side term; the
  // residual rows of the NEXT sub-tile are requested before this sub-tile's stores (vmcnt retires loads and stores in issue order:
  // a load issued behind a store cannot be waited for without waiting for that store's acknowledgement).  Same arithmetic per
  // element as the general path (which edge tiles of the same launch take): equal bits.
  if constexpr (!LN) {
    // a residual that repeats every res_mod rows (the encoder's position table): its row is m % res_mod -- a uniform shift of the
    // tile's rows unless the tile straddles a period (1 tile in 40 at 360p: general path)
    const int rshift = p.res_mod > 0 ? m0 - m0 % p.res_mod : 0;
    const long res_rows = p.res_mod > 0 ? p.res_mod : p.M;
    const long c_bytes = (long)p.M * p.ldc * 4, r_bytes = p.residual != nullptr ? res_rows * p.ldr * 4 : 0;
    bool fast = vec && p.ksplit <= 1 && (p.side == nullptr || (MT * NT == 1 && p.act == MDQE_ACT_NONE && p.residual == nullptr && p.side_cols % 4 == 0)) && p.act_cols <= 0 &&
                (p.res_mod <= 0 || (p.residual != nullptr && m0 - rshift + BM <= p.res_mod)) &&
                (p.act == MDQE_ACT_NONE || p.act == MDQE_ACT_RELU || p.act == MDQE_ACT_GELU) && m0 + BM <= p.M && n0 + BN <= p.N &&
                c_bytes < 0xFFFF0000L && r_bytes < 0xFFFF0000L && g_k16_fast_epilogue_on(p);      // (buffer offsets are unsigned 32-bit)
    if (fast && p.rowmask != nullptr) {
      // masked rows (padding tokens, ~6 % of the rows at 360p): the decision is per WAVE -- nothing below is block-wide -- so a wave
      // whose 32 / 64 rows hold none takes the fast path and the others the general one
      constexpr int WR = BM / WM;
      const bool mrow = lane < WR && p.rowmask[m0 + wm * WR + lane] != 0;
      fast = __builtin_amdgcn_ballot_w64(mrow) == 0;
    }
    if (fast) {
      const int c4 = lane & 7, r8 = lane >> 3;
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)(unsigned)c_bytes, 0x00020000);
      const auto rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual != nullptr ? p.residual : p.C), 0, (int)(unsigned)(p.residual != nullptr ? r_bytes : c_bytes), 0x00020000);
      const int row0 = m0 + wm * (BM / WM), col0 = n0 + wn * (BN / WN);
      const unsigned vC = (unsigned)((r8 * p.ldc + c4 * 4) * 4), vR = (unsigned)((r8 * p.ldr + c4 * 4) * 4);
      const bool has_res = p.residual != nullptr, has_bias = p.bias != nullptr;
      f32x4 bj[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bj[j] = has_bias ? *reinterpret_cast<const f32x4*>(p.bias + col0 + j * 32 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      auto load_res = [&](int i, int j, f32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int so = (int)((((long)row0 - rshift + i * 32 + it * 8) * p.ldr + col0 + j * 32) * 4);
          r[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vR, so, 0));
        }
      };
      // one specialised copy per (activation, residual, bias) combination that the per-frame stages use, chosen by a UNIFORM branch:
      // a select between "with" and "without" costs vector instructions on every value, a branch around the other copies none
      auto run = [&](auto act_, auto res_, auto bias_, auto side_) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_)::value;
        constexpr bool RES = decltype(res_)::value, BIAS = decltype(bias_)::value, SIDE = decltype(side_)::value;
        f32x4 rbuf[2][4];
        if constexpr (RES) load_res(0, 0, rbuf[0]);
        // rank-4 side term (the decoder's position embeddings folded into the q / k / offset products): the wave's 64 / 32 side rows
        // [row, 4] are requested up front, one per (sub-tile row block, trip); a lane's 4 columns' side weights once per column block
        f32x4 srow[SIDE ? MT : 1][4];
        if constexpr (SIDE) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int it = 0; it < 4; ++it)
              srow[i][it] = *reinterpret_cast<const f32x4*>(p.side + (long)(row0 + i * 32 + it * 8 + r8) * 4);
        }
        auto fsub = [&](auto i_, auto j_, auto s_) __attribute__((always_inline)) {
          constexpr int i = decltype(i_)::value, j = decltype(j_)::value, sidx = decltype(s_)::value;
          constexpr int nsub = MT * NT;
#pragma unroll
          for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[i][j][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          if constexpr (RES && sidx + 1 < nsub) {        // the next sub-tile's residual rows, BEFORE this one's stores
            constexpr int s1 = sidx + 1;
            constexpr int i1 = (NT > 1) ? s1 / NT : s1, j1 = (NT > 1) ? s1 % NT : 0;
            load_res(i1, j1, rbuf[s1 & 1]);
          }
          f32x4 sw[SIDE ? 4 : 1];
          bool sidecol = false;
          if constexpr (SIDE) {
            sidecol = col0 + j * 32 + c4 * 4 < p.side_cols;          // (side_cols % 4 == 0: a lane's 4 columns together)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              sw[e] = sidecol ? *reinterpret_cast<const f32x4*>(p.side_w + (long)(col0 + j * 32 + c4 * 4 + e) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            f32x4 x = *reinterpret_cast<const f32x4*>(sC + (it * 8 + r8) * 32 + c4 * 4);
            if constexpr (BIAS) x += bj[j];
            if constexpr (SIDE) {
              if (sidecol) {
                const f32x4 s4 = srow[i][it];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] += (s4[0] * sw[e][0] + s4[1] * sw[e][1]) + (s4[2] * sw[e][2] + s4[3] * sw[e][3]);
              }
            }
            if constexpr (RES) { if (p.res_first) x += rbuf[sidx & 1][it]; }
            if constexpr (ACT == MDQE_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = x[e] > 0.f ? x[e] : 0.f;
            } else if constexpr (ACT == MDQE_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = mdqe_gelu(x[e]);
            }
            if constexpr (RES) { if (!p.res_first) x += rbuf[sidx & 1][it]; }
            const int so = (int)((((long)row0 + i * 32 + it * 8) * p.ldc + col0 + j * 32) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, x), rsC, vC, so, 0);
            // The next trip's first vector instruction may write the registers this 16-byte store is still reading (hipcc 7.2 places a
            // v_pk_add_f32 into v[26:27] directly behind `buffer_store_dwordx4 v[26:29] ... s8 offen` and inserts no wait state --
            // its hazard table exempts stores with an SGPR offset; on gfx950 the second register of the store then carries the NEW
            // value in lanes 12-15 of every 16).  Two wait states, fenced so that nothing is scheduled into them.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        };
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
        using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
        if constexpr (MT == 1 && NT == 1) { fsub(J0{}, J0{}, J0{}); }
        else if constexpr (MT == 1 && NT == 2) { fsub(J0{}, J0{}, J0{}); fsub(J0{}, J1{}, J1{}); }
        else if constexpr (MT == 2 && NT == 1) { fsub(J0{}, J0{}, J0{}); fsub(J1{}, J0{}, J1{}); }
        else { fsub(J0{}, J0{}, J0{}); fsub(J0{}, J1{}, J1{}); fsub(J1{}, J0{}, J2{}); fsub(J1{}, J1{}, J3{}); }
      };
      using AN = std::integral_constant<int, MDQE_ACT_NONE>; using AR = std::integral_constant<int, MDQE_ACT_RELU>;
      using AG = std::integral_constant<int, MDQE_ACT_GELU>;
      using T = std::true_type; using F = std::false_type;
      // (the combinations of the per-frame stages: bias always; ReLU with / without residual -- ResNet; GELU without -- FFN1;
      // none with / without -- projections.  Anything else takes the general path below.)
      bool done = true;
      if (!has_bias) done = false;
      else if (MT * NT == 1 && p.side != nullptr) {
        // (tiles of ONE sub-tile per wave only -- what the decoder's 29 008-row products run on; the 128 x 128 tile has no
        // registers to spare for the side rows)
        if constexpr (MT * NT == 1) run(AN{}, F{}, T{}, T{});
      }
      else if (p.act == MDQE_ACT_RELU) { if (has_res) run(AR{}, T{}, T{}, F{}); else run(AR{}, F{}, T{}, F{}); }
      else if (p.act == MDQE_ACT_GELU) { if (has_res) done = false; else run(AG{}, F{}, T{}, F{}); }
      else { if (has_res) run(AN{}, T{}, T{}, F{}); else run(AN{}, F{}, T{}, F{}); }
      if (done) {
        if (p.stamps && tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          unsigned long long* o = p.stamps + (long)blockIdx.x * 4;
          o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = wall_clock64();
        }
        return;
      }
    }
  }
  auto sub = [&](auto i_, auto j_) __attribute__((always_inline)) {
      constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // (rolled, round 4: the body -- bias, side term, residual, activation, mask, the ragged-edge path -- is long, and sixteen unrolled
      // copies of it per kernel made the epilogue an instruction-cache problem)
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx >> 3, c4 = idx & 7;
        const int m = m0 + wm * (BM / WM) + i * 32 + row, n = n0 + wn * (BN / WN) + j * 32 + c4 * 4;
        if (m >= p.M || n >= p.N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * 32 + c4 * 4);
        if (p.ksplit > 1) {                            // raw partial -> workspace; the epilogue runs in the reduce pass
          float* w = p.ws + (long)blockIdx.y * p.M * p.N + (long)m * p.N + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) w[e] = v[e];
          continue;
        }
        const long rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
        const bool masked = p.rowmask != nullptr && p.rowmask[m];
        if (vec && (n + 3 < p.N)) {
          if (p.bias != nullptr) v += *reinterpret_cast<const f32x4*>(p.bias + n);
          if (p.side != nullptr && n < p.side_cols) {        // rank-4 side term (side_cols % 4 == 0)
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.side + (long)m * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.side_w + (long)(n + e) * 4);
              v[e] += (s4[0] * w4[0] + s4[1] * w4[1]) + (s4[2] * w4[2] + s4[3] * w4[3]);
            }
          }
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
          // ragged edge / unaligned operands: element by element, ROLLED (the vector rotates through the loop so that no register is
          // indexed dynamically) -- the generic activation switch, tanhf included, appears once here instead of four times
#pragma unroll 1
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) break;
            const float ve = v[0];
            v = f32x4{v[1], v[2], v[3], ve};
            float x = ve + (p.bias != nullptr ? p.bias[n + e] : 0.f);
            if (p.side != nullptr && n + e < p.side_cols) {
              const float* s4 = p.side + (long)m * 4; const float* w4 = p.side_w + (long)(n + e) * 4;
              x += (s4[0] * w4[0] + s4[1] * w4[1]) + (s4[2] * w4[2] + s4[3] * w4[3]);
            }
            const float rv = p.residual != nullptr ? p.residual[rrow * p.ldr + n + e] : 0.f;
            if (p.res_first) x += rv;
            if (p.act != MDQE_ACT_NONE && (p.act_cols <= 0 || n + e < p.act_cols)) x = mdqe_act(x, p.act);
            if (!p.res_first) x += rv;
            if (masked && n + e < p.mask_cols) x = 0.f;
            p.C[(long)m * p.ldc + n + e] = x;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  sub(I0{}, I0{});
  if constexpr (NT > 1) sub(I0{}, I1{});
  if constexpr (MT > 1) sub(I1{}, I0{});
  if constexpr (MT > 1 && NT > 1) sub(I1{}, I1{});
  if (p.stamps && tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = p.stamps + (long)blockIdx.x * 4;
    o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = wall_clock64();
  }
}

static int g_k16_fast_epi = 1;     // tools/ A/B: 0 = every tile through the general epilogue
extern "C" int mdqe_debug_gemm_fast_epilogue(int v) { g_k16_fast_epi = v ? 1 : 0; return MDQE_OK; }
static int g_k16_stagger = 0;      // tools/ A/B: first-round start offset between the blocks of a CU, in 10-ns ticks
extern "C" int mdqe_debug_gemm_stagger(int v) { g_k16_stagger = v; return MDQE_OK; }
static int g_k16_lds_pad = 0;      // tools/ A/B: extra dynamic LDS bytes per block (caps the blocks per CU: occupancy experiments)
extern "C" int mdqe_debug_gemm_lds_pad(int v) { g_k16_lds_pad = v > 0 ? v : 0; return MDQE_OK; }
static int g_k16_stages = 0;       // tools/ A/B: 0 = by grid size, 2 / 4 = forced
extern "C" int mdqe_debug_gemm_stages(int v) { g_k16_stages = v; return MDQE_OK; }

template <int BM, int BN, int WM, int WN, bool CONV, bool LN, int NS, bool CAT = false>
static int launch_k16_ns_(const GemmParams& p_in, hipStream_t st) {
  GemmParams p = p_in;
  p.stagger = g_k16_stagger;
  p.fast_epi = g_k16_fast_epi;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  size_t smem = (size_t)NS * (BM + BN) * 16 * sizeof(float);
  if (smem < (size_t)WM * WN * 4096) smem = (size_t)WM * WN * 4096;        // per-wave epilogue slices
  auto kern = gemm_nt_f32_k16_kernel<BM, BN, WM, WN, CONV, LN, NS, CAT>;
  if (g_k16_lds_pad > 0) {                                                 // tools/ only
    smem += (size_t)g_k16_lds_pad;
    if (smem > 64 * 1024 && mdqe_allow_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 256) != hipSuccess) return MDQE_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(nbm * nbn, p.ksplit > 1 ? p.ksplit : 1), dim3(64 * WM * WN), smem, st, p);
  return mdqe_launch_status();
}

template <int BM, int BN, int WM, int WN, bool CONV, bool LN = false>
static int launch_k16_(const GemmParams& p, hipStream_t st) {
  if constexpr (!LN && !CONV && BM * BN >= 128 * 64 && BM * BN <= 128 * 128) {
    // tools/ only (mdqe_debug_gemm_stages(3)): three LDS stages for the large plain tiles -- two K-steps of look-ahead (48 KB per 128x128
    // block: three blocks per CU instead of four); measured in tools/gemm_stages3_ab.py
    if (g_k16_stages == 3) return launch_k16_ns_<BM, BN, WM, WN, CONV, LN, 3>(p, st);
  }
  if constexpr (!LN && BM * BN <= 64 * 64) {
    // small tiles on a small grid (about one block per CU or less per SIMD wave slot): deep look-ahead
    const long blocks = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.ksplit > 1 ? p.ksplit : 1);
    // measured (tools/gemm_stages_ab.py, r02): no gain -- 14.8 vs 14.9 us on [5292,256]x[256,256], 10 us even for 16 blocks: the
    // floor of these launches is not the look-ahead depth.  Kept as a measured alternative behind the debug switch.
    (void)blocks;
    if (g_k16_stages == 4) return launch_k16_ns_<BM, BN, WM, WN, CONV, LN, 4>(p, st);
  }
  return launch_k16_ns_<BM, BN, WM, WN, CONV, LN, 2>(p, st);
}

// tile: 1 128x128, 2 128x64, 3 64x64 (as gemm.hip), 7 32x64, 8 32x128, 9 64x128; the split-K reduce pass is launched by the caller
int mdqe_launch_gemm_k16(const GemmParams& p, int tile, hipStream_t st) {
  if (p.side != nullptr && (p.conv || p.ksplit > 1 || tile == 6 || p.A2 != nullptr)) return MDQE_EINVAL;   // side term: plain tiles only
  if (p.swin_ws > 0 && (p.conv || p.ksplit > 1 || tile == 6 || p.A2 != nullptr)) return MDQE_EINVAL;        // window gather: plain tiles only
  if (p.A2 != nullptr) {                                // cat mode: two A operands side by side along K
    if (p.conv || p.ksplit > 1) return MDQE_EINVAL;
    switch (tile) {
      case 1: return launch_k16_ns_<128, 128, 2, 2, false, false, 2, true>(p, st);
      case 2: return launch_k16_ns_<128, 64, 2, 2, false, false, 2, true>(p, st);
      case 3: return launch_k16_ns_<64, 64, 2, 2, false, false, 2, true>(p, st);
      default: return MDQE_EINVAL;
    }
  }
  switch (tile) {
    case 1: return p.conv ? launch_k16_<128, 128, 2, 2, true>(p, st) : launch_k16_<128, 128, 2, 2, false>(p, st);
    case 2: return p.conv ? launch_k16_<128, 64, 2, 2, true>(p, st) : launch_k16_<128, 64, 2, 2, false>(p, st);
    case 3: return p.conv ? launch_k16_<64, 64, 2, 2, true>(p, st) : launch_k16_<64, 64, 2, 2, false>(p, st);
    case 4: return p.conv ? MDQE_EINVAL : launch_k16_<64, 256, 1, 4, false>(p, st);      // full 256-wide rows per block
    case 5: return p.conv ? MDQE_EINVAL : launch_k16_<128, 256, 2, 4, false>(p, st);
    case 7: return p.conv ? launch_k16_<32, 64, 1, 2, true>(p, st) : launch_k16_<32, 64, 1, 2, false>(p, st);   // small problems:
    case 8: return p.conv ? launch_k16_<32, 128, 1, 2, true>(p, st) : launch_k16_<32, 128, 1, 2, false>(p, st); // more, shorter blocks
    case 9: return p.conv ? launch_k16_<64, 128, 2, 2, true>(p, st) : launch_k16_<64, 128, 2, 2, false>(p, st);
    case 6:                                             // 64x256 with the LayerNorm epilogue (mdqe_gemm_ln_f32)
      if (p.conv || p.N != 256 || p.ksplit > 1 || !p.vec_ok || p.ln_g == nullptr || p.ln_b == nullptr) return MDQE_EINVAL;
      return launch_k16_<64, 256, 1, 4, false, true>(p, st);
    default: return MDQE_EINVAL;
  }
}
