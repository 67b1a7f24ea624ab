// f16x3 GEMM / implicit-GEMM conv with PRE-SPLIT constant weights  (fp32 activations in, fp32 out).
//
//   x = hi + lo'/2048,  hi = f16_rtz(x),  lo' = f16_rtz((x - hi) * 2048)          (see gemm_f16x3.hip)
//   A.B ~= Ah.Bh + (Ah.Bl' + Al'.Bh)/2048        three v_mfma_f32_32x32x16_f16, fp32 accumulation
//
// What differs from gemm_f16x3.hip (which splits both operands in-kernel, keeps one K-step in flight and is
// bound by VALU issue + exposed load latency):
//   * W is constant on this path, so its hi / lo' planes ([N][K] f16 each) are produced once at load
//     time (mdqe_f16x3_split_f32) and stream HBM/L2 -> LDS by buffer_load ... lds (3-stage ring, two
//     K-steps in flight), no VGPRs, no VALU;
//   * the block tile is 128 x BN (BN = 256 or 128) on 8 waves (2 x 4, 64 x BN/4 per wave): the fp32 A
//     tile is split in registers exactly once per element (v_fma_mix form) and amortised over BN columns;
//     A travels through a 3-deep register ring (three K-steps of HBM latency cover);
//   * K-step addressing is scalar (buffer soffset) in plain mode; conv mode walks the filter taps with
//     scalar state instead of dividing per step;
//   * one raw s_barrier per K-step with counted vmcnt (no fence: prefetches stay in flight across it);
//   * the epilogue restages each wave's 32 x (BN/4) half tile through its OWN LDS slice (no block
//     barrier), then bias / residual / activation / mask and 16-B stores as in gemm.hip.
// Tile order, conv addressing, split-K and the epilogue semantics are those of gemm.hip.  K % 32 == 0.
#include "common.h"
#include "gemm_params.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 pkrtz2(float a, float b) { return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b)); }

// ---- one-time weight split ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
f16x3_split_kernel(const float* __restrict__ w, long n, _Float16* __restrict__ hi, _Float16* __restrict__ lo, _Float16* __restrict__ rn) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (long)gridDim.x * blockDim.x * 2) {
    const float a = w[i], b = (i + 1 < n) ? w[i + 1] : 0.f;
    const h2 h = pkrtz2(a, b);
    const h2 l = pkrtz2((a - (float)h[0]) * 2048.f, (b - (float)h[1]) * 2048.f);
    hi[i] = h[0]; lo[i] = l[0]; rn[i] = (_Float16)a;                   // rn: round-to-nearest-even, what `.half()` gives (mode 2)
    if (i + 1 < n) { hi[i + 1] = h[1]; lo[i + 1] = l[1]; rn[i + 1] = (_Float16)b; }
  }
}

extern "C" int mdqe_f16x3_split_f32(const float* w, long n, void* planes, void* stream) {
  MDQE_REQUIRE(n > 0);
  MDQE_CHECK_PTR(w); MDQE_CHECK_PTR(planes);
  long nb = (n / 2 + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  mdqe_clear_error();
  hipLaunchKernelGGL(f16x3_split_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, n,
                     (_Float16*)planes, (_Float16*)planes + n, (_Float16*)planes + 2 * n);
  return mdqe_launch_status();
}

// ---- GEMM -----------------------------------------------------------------------------------------
// Persistent: gridDim.x = min(#tiles, #CUs) blocks; block b walks tiles b, b + G, b + 2G, ... (XCD-aware order) as ONE flat
// sequence of K-steps.  The LDS / register rings simply run on across tile boundaries, so the first K-steps of the next
// tile are already in flight while the finished tile's epilogue issues its stores, and stores drain under the next tile's
// MFMAs (they are never waited for explicitly; the counted vmcnt of a later K-step covers them in issue order).
// The MFMAs are issued with the operands swapped (weights as A, activations as B): D[n][m] puts four consecutive n of one
// row m in each lane's accumulator quad, i.e. a 16-B piece of a C row -- the epilogue needs no LDS restage.
// Requires K % 32 == 0 (conv: Cin % 32 == 0), N % 4 == 0 and 16-B aligned C / bias / residual (vec_ok), no split-K.
// SINGLE (mode 2, "f16"): ONE f16 MFMA pass -- operands rounded to nearest f16 (the activations in registers, the weights from their
// pre-rounded plane), fp32 accumulation, fp32 out: the arithmetic of the reference's fp16-autocast regions on a GPU (train_net.py:207) with
// an fp32 instead of an fp16 result.  Same pipeline; a stage holds one A and one B plane (72 instead of 144 KB at BN = 256: two blocks per CU).
template <int BN, bool CONV, bool SINGLE = false>
__global__ void __launch_bounds__(512, 2)
gemm_nt_f16x3w_kernel(const GemmParams p) {
  constexpr int BM = 128, BK = 32, NW = 8;
  constexpr int WTN = BN / 4;                    // wave tile: 64 x WTN
  constexpr int MT = 2, NT = WTN / 32;
  constexpr int A_PLANE = BM * BK * 2;           // bytes
  constexpr int B_PLANE = BN * BK * 2;
  constexpr int NP = SINGLE ? 1 : 2;             // planes per operand and stage
  constexpr int STAGE = NP * A_PLANE + NP * B_PLANE;
  constexpr int IB = BN / 16 / NW;               // 1-KiB LDS-DMA instructions per wave per B plane
  static_assert(BN % 128 == 0, "BN");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
  if (p.stamps) t_start = __builtin_amdgcn_s_memtime();

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int nk = p.K / BK;
  const int G = gridDim.x;
  const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;        // >= 1 (G <= ntiles)
  const int total = my_tiles * nk;
  const int xq = ntiles / 8, xr = ntiles % 8;
  auto tile_mn = [&](int it, int& m0, int& n0) __attribute__((always_inline)) {                       // XCD-aware order: an XCD owns a contiguous tile range
    int v = (int)blockIdx.x + it * G;
    const int xcd = v % 8, i = v / 8;
    v = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + i;
    const int bm = v / nbn;
    m0 = bm * BM; n0 = (v - bm * nbn) * BN;
  };

  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wh, 0, p.w_bytes / 2, 0x00020000);
  const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wl, 0, p.w_bytes / 2, 0x00020000);

  // ---- issue cursors: A (register ring, 3 K-steps ahead of the MFMAs) and B (LDS-DMA, 2 ahead) ------------------------
  // A: thread -> rows (tid>>3) and 64 + (tid>>3), float4 q8 = tid&7 of the 32-float K-step.
  // plain: voffset fixed per tile, the K-step rides on the scalar offset.  conv: voffset = pixel base + filter-tap offset.
  const int q8 = tid & 7;
  int itA = 0, ktA = 0, itB = 0, ktB = 0;
  unsigned arow[2]; int ih0[2], iw0[2];
  int t_kh = 0, t_kw = 0, t_c = 0;                                     // conv: filter tap of the A cursor
  auto setup_a = [&](int it) __attribute__((always_inline)) {
    int m0, n0; tile_mn(it, m0, n0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = m0 + j * 64 + (tid >> 3); if (m > p.M - 1) m = p.M - 1;
      ih0[j] = 0; iw0[j] = 0;
      if (CONV) {
        const int ow = m % p.OW; const int t = m / p.OW; const int oh = t % p.OH; const int img = t / p.OH;
        ih0[j] = oh * p.stride - p.pad; iw0[j] = ow * p.stride - p.pad;
        arow[j] = (unsigned)(((long)img * p.img_stride + ((long)ih0[j] * p.Wd + iw0[j]) * p.Cin) * 4) + (unsigned)(q8 * 16);
      } else {
        arow[j] = (unsigned)((long)m * p.lda * 4) + (unsigned)(q8 * 16);
      }
    }
    t_kh = 0; t_kw = 0; t_c = 0;
  };
  unsigned wrow[IB];
  auto setup_b = [&](int it) __attribute__((always_inline)) {
    int m0, n0; tile_mn(it, m0, n0);
#pragma unroll
    for (int j = 0; j < IB; ++j) {       // instruction j of this wave covers tile rows (wave*IB + j)*16 .. +15; lane -> row += lane>>2, LDS chunk lane&3
      const int irow = (wave * IB + j) * 16 + (lane >> 2);
      int n = n0 + irow; if (n > p.N - 1) n = p.N - 1;
      wrow[j] = (unsigned)((long)n * p.K * 2) + (unsigned)((((lane & 3) ^ ((irow >> 2) & 3))) * 16);
    }
  };
  setup_a(0);
  setup_b(0);

  f32x4 stg[3][2];                       // A register ring: slot r holds the two float4 of flat K-step s with s % 3 == r
  auto load_a = [&](auto slot_) __attribute__((always_inline)) {        // loads the K-step under the A cursor, then advances it (parks on the last one at the end)
    constexpr int slot = decltype(slot_)::value;
    if (CONV) {
      const int tap_off = ((t_kh * p.Wd + t_kw) * p.Cin + t_c) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ih = ih0[j] + t_kh, iw = iw0[j] + t_kw;
        const bool ok = (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.Wd);
        const unsigned off = ok ? arow[j] + (unsigned)tap_off : OOB_OFF;
        stg[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
      }
    } else {
      const int so = ktA * (BK * 4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned off = arow[j];
        stg[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, so, 0));
      }
    }
    if (ktA + 1 < nk) {
      ++ktA;
      if (CONV) { t_c += BK; if (t_c >= p.Cin) { t_c = 0; if (++t_kw == p.KW) { t_kw = 0; ++t_kh; } } }
    } else if (itA + 1 < my_tiles) {
      ktA = 0; ++itA; setup_a(itA);
    }
  };
  auto issue_b = [&](int buf) __attribute__((always_inline)) {          // 2*IB LDS-DMA instructions for the K-step under the B cursor, then advances it
    const int so = ktB * (BK * 2);
    char* base = lds + buf * STAGE + NP * A_PLANE;
#pragma unroll
    for (int j = 0; j < IB; ++j) {
      const unsigned off = wrow[j];      // (a local: passing the captured array element straight to the builtin loses the host stub, hipcc 7.2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH, (__attribute__((address_space(3))) void*)(base + (wave * IB + j) * 1024), 16, off, so, 0, 0);
      if constexpr (!SINGLE)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (__attribute__((address_space(3))) void*)(base + B_PLANE + (wave * IB + j) * 1024), 16, off, so, 0, 0);
    }
    if (ktB + 1 < nk) ++ktB;
    else if (itB + 1 < my_tiles) { ktB = 0; ++itB; setup_b(itB); }
  };
  // LDS position (in halves) of this thread's 4-half group inside an A plane, per row j
  int apos[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = j * 64 + (tid >> 3);
    apos[j] = row * BK + ((((q8 >> 1) ^ ((row >> 2) & 3)) << 3) | ((q8 & 1) << 2));
  }
  auto split_store = [&](auto slot_, int buf) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_)::value;
    _Float16* hi = reinterpret_cast<_Float16*>(lds + buf * STAGE);
    _Float16* lo = hi + BM * BK;
    if constexpr (SINGLE) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 v = stg[slot][j];
        *reinterpret_cast<h4*>(hi + apos[j]) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};     // round to nearest even
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 v = stg[slot][j];
      const h2 a = pkrtz2(v[0], v[1]);
      const h2 b = pkrtz2(v[2], v[3]);
      // (v - hi) * 2048 == fma(hi, -2048, v * 2048): every step exact; the f16 source feeds v_fma_mix_f32 directly
      const h2 c = pkrtz2(__builtin_fmaf((float)a[0], -2048.f, v[0] * 2048.f), __builtin_fmaf((float)a[1], -2048.f, v[1] * 2048.f));
      const h2 d = pkrtz2(__builtin_fmaf((float)b[0], -2048.f, v[2] * 2048.f), __builtin_fmaf((float)b[1], -2048.f, v[3] * 2048.f));
      *reinterpret_cast<h4*>(hi + apos[j]) = h4{a[0], a[1], b[0], b[1]};
      *reinterpret_cast<h4*>(lo + apos[j]) = h4{c[0], c[1], d[0], d[1]};
    }
  };

  f32x16 acc[MT][NT], acx[MT][NT];                       // hi.hi  and  (hi.lo' + lo'.hi), joined as acc + acx/2048
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }
  };
  zero_acc();

  // ---- epilogue of tile `it` (registers -> global, no LDS): lane owns row m = .. + lr and, per accumulator quad g,
  //      columns n .. n+3 with n = .. + 8g + 4*lh ------------------------------------------------------------------------
  auto epilogue = [&](int it) __attribute__((always_inline)) {
    int m0, n0; tile_mn(it, m0, n0);
    const float inv = 1.0f / 2048.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * 64 + i * 32 + lr;
      const bool mok = m < p.M;
      const bool masked = mok && p.rowmask != nullptr && p.rowmask[m];
      const long rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * WTN + j * 32 + 8 * g + 4 * lh;
          if (!mok || n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = SINGLE ? acc[i][j][4 * g + e] : acc[i][j][4 * g + e] + acx[i][j][4 * g + e] * inv;
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
        }
    }
  };

  // fragment positions (halves) for g = 0; g = 1 flips chunk bit 1 (XOR 16 halves)
  int fa[MT], fb[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int row = wm * 64 + i * 32 + lr; fa[i] = row * BK + ((lh ^ ((row >> 2) & 3)) << 3); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int row = wn * WTN + j * 32 + lr; fb[j] = row * BK + ((lh ^ ((row >> 2) & 3)) << 3); }
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

  int itC = 0, ktC = 0;                                  // compute cursor
  // One flat K-step s on LDS stage S (s % 3 == S).  On entry: stage S complete (A planes stored + B planes landed, barrier
  // passed), B planes of step s+1 in flight, A registers of steps s+1, s+2 in flight.  VMEM issue order per step is
  // [epilogue stores, first step of a tile only] [B(s+2): 2*IB] ... [A(s+3): 2], so "B(s+1) has landed" is implied by
  // vmcnt(2*IB + 4) at the end of the step (older epilogue traffic only makes that wait conservative).
  auto kstep = [&](auto S_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value, S1 = (S + 1) % 3, S2 = (S + 2) % 3;
    if (ktC == 0 && itC > 0) { epilogue(itC - 1); zero_acc(); }
    issue_b(S2);                                         // stage S2 was last read in step s-1 (barrier since)
    const _Float16* sAh = reinterpret_cast<const _Float16*>(lds + S * STAGE);
    const _Float16* sAl = sAh + BM * BK;
    const _Float16* sBh = sAh + NP * BM * BK;
    const _Float16* sBl = sBh + BN * BK;
    if constexpr (SINGLE) {
      h8 a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) { a0[i] = *reinterpret_cast<const h8*>(sAh + fa[i]); a1[i] = *reinterpret_cast<const h8*>(sAh + (fa[i] ^ 16)); }
#pragma unroll
      for (int j = 0; j < NT; ++j) { b0[j] = *reinterpret_cast<const h8*>(sBh + fb[j]); b1[j] = *reinterpret_cast<const h8*>(sBh + (fb[j] ^ 16)); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0[j], a0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
      split_store(std::integral_constant<int, S1>{}, S1);
      load_a(S_);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IB + 4) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (++ktC == nk) { ktC = 0; ++itC; }
      return;
    }
    // LDS fragment reads run one half K-step ahead of their MFMAs: all of half 0 and the hi planes of half 1 up front,
    // the lo planes of half 1 once half 0 has retired some registers (48 fragment VGPRs instead of 64).
    h8 ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { ah0[i] = *reinterpret_cast<const h8*>(sAh + fa[i]); al0[i] = *reinterpret_cast<const h8*>(sAl + fa[i]); }
#pragma unroll
    for (int j = 0; j < NT; ++j) { bh0[j] = *reinterpret_cast<const h8*>(sBh + fb[j]); bl0[j] = *reinterpret_cast<const h8*>(sBl + fb[j]); }
#pragma unroll
    for (int i = 0; i < MT; ++i) ah1[i] = *reinterpret_cast<const h8*>(sAh + (fa[i] ^ 16));
#pragma unroll
    for (int j = 0; j < NT; ++j) bh1[j] = *reinterpret_cast<const h8*>(sBh + (fb[j] ^ 16));
    __builtin_amdgcn_sched_barrier(0);
    // weights as the MFMA "A" operand: D[n][m]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0[j], ah0[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0[j], ah0[i], acx[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) al1[i] = *reinterpret_cast<const h8*>(sAl + (fa[i] ^ 16));
#pragma unroll
    for (int j = 0; j < NT; ++j) bl1[j] = *reinterpret_cast<const h8*>(sBl + (fb[j] ^ 16));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0[j], al0[i], acx[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1[j], ah1[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1[j], ah1[i], acx[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1[j], al1[i], acx[i][j], 0, 0, 0);
    split_store(std::integral_constant<int, S1>{}, S1);  // A(s+1): registers -> planes of stage S1 (last read in step s-2)
    load_a(S_);                                          // A(s+3) into slot S (consumed by the split of step s-1)
    // raw barrier: __syncthreads() carries a workgroup fence that would drain the in-flight LDS-DMAs of stage S2
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * IB + 4) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (++ktC == nk) { ktC = 0; ++itC; }
  };

  load_a(I0{});
  load_a(I1{});
  issue_b(0);
  split_store(I0{}, 0);
  issue_b(1);
  load_a(I2{});
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP * IB + 2) : "memory");    // B(0) landed; B(1), A(2) may be in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.stamps) t_pro = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < total; s += 3) {
    kstep(I0{});
    if (s + 1 < total) kstep(I1{});
    if (s + 2 < total) kstep(I2{});
  }
  if (p.stamps) t_loop = __builtin_amdgcn_s_memtime();
  epilogue(my_tiles - 1);
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + (long)blockIdx.x * 4;
    o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
  }
}

static int g_num_cus = 0;

template <int BN, bool CONV, bool SINGLE = false>
static int launch_f16x3w(const GemmParams& p, hipStream_t st) {
  const int nbm = (p.M + 127) / 128, nbn = (p.N + BN - 1) / BN;
  const size_t smem = (size_t)3 * (SINGLE ? 1 : 2) * (128 * 32 * 2 + BN * 32 * 2);       // 3 stages x (A hi[, A lo], B hi[, B lo])
  auto kern = gemm_nt_f16x3w_kernel<BN, CONV, SINGLE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cus = n;
  }
  const int per_cu = smem * 2 <= 160 * 1024 ? 2 : 1;                            // blocks that fit a CU's LDS
  int grid = nbm * nbn; if (grid > g_num_cus * per_cu) grid = g_num_cus * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, p);
  return mdqe_launch_status();
}

// bn: 256 or 128.  Caller guarantees K % 32 == 0, N % 4 == 0, vec_ok, ksplit <= 1 and Wh/Wl set.
// mode 2: one f16 pass; p.Wh must point at the round-to-nearest plane (the third plane of mdqe_f16x3_split_f32).
int mdqe_launch_gemm_f16w(const GemmParams& p, int bn, hipStream_t st) {
  if (p.conv) return bn == 256 ? launch_f16x3w<256, true, true>(p, st) : launch_f16x3w<128, true, true>(p, st);
  return bn == 256 ? launch_f16x3w<256, false, true>(p, st) : launch_f16x3w<128, false, true>(p, st);
}

int mdqe_launch_gemm_f16x3w(const GemmParams& p, int bn, hipStream_t st) {
  if (p.conv) return bn == 256 ? launch_f16x3w<256, true>(p, st) : launch_f16x3w<128, true>(p, st);
  return bn == 256 ? launch_f16x3w<256, false>(p, st) : launch_f16x3w<128, false>(p, st);
}
