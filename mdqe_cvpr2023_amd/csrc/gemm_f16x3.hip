// fp32-in / fp32-out NT GEMM + implicit-GEMM conv on the f16 matrix cores ("f16x3" split precision).
//
// Every fp32 operand x is split on the fly into  hi = f16(x)  and  lo = f16((x - hi) * 2048)  (both
// round-toward-zero, x - hi is exact in fp32), so x = hi + lo/2048 up to 2^-21 |x|.  The product
//     A.B ~= Ah.Bh + (Ah.Bl + Al.Bh) / 2048
// takes three v_mfma_f32_32x32x16_f16 (each 16x the rate of the fp32 MFMA) with fp32 accumulation; the
// dropped Al.Bl term is 2^-22 relative.  Every f16 x f16 product is exact in fp32, so the result differs
// from an fp32 GEMM only by the 2^-21 operand representation -- ~1e-6 relative, the same order as the
// summation-order noise of fp32 itself (measured in tests/test_kernels_gpu.py).  Requires |x| < 32752.
//
// Operands stay fp32 in HBM (nothing else in the pipeline changes): global -> registers (buffer loads,
// zero fill for padding/K tail) -> split in registers (VALU, overlapped with the MFMAs of the previous
// K-step) -> LDS as two f16 planes (64-B rows, 16-B chunks XOR-swizzled by (row>>2)&3: conflict-free
// ds_read_b128) -> MFMA.  Same tile order, epilogue and conv addressing as gemm.hip.
#include "common.h"
#include "gemm_params.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 pkrtz(float a, float b) { return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b)); }

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, 2)
gemm_nt_f16x3_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int NT_ = 64 * NW;
  constexpr int BK = 32;
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int ROWS = BM + BN;
  constexpr int NLD = ROWS * 8 / NT_;          // float4 loads per thread per K-step (8 float4 per row)
  constexpr int RPP = NT_ / 8;                 // rows covered per pass
  constexpr int PLANE = ROWS * BK;             // halves per plane
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  _Float16* lds = reinterpret_cast<_Float16*>(lds_f);     // stage s: [hi plane][lo plane], 2 stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

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

  const int q8 = tid & 7;                      // float4 index inside the 32-float K-step
  unsigned rowoff[NLD];
  int ih0[NLD], iw0[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int irow = j * RPP + (tid >> 3);
    ih0[j] = 0; iw0[j] = 0;
    if (irow < BM) {
      int m = m0 + irow; if (m > p.M - 1) m = p.M - 1;
      if (p.conv) {
        const int ow = m % p.OW; const int t = m / p.OW; const int oh = t % p.OH; const int img = t / p.OH;
        ih0[j] = oh * p.stride - p.pad; iw0[j] = ow * p.stride - p.pad;
        rowoff[j] = (unsigned)(((long)img * p.img_stride + ((long)ih0[j] * p.Wd + iw0[j]) * p.Cin) * 4);
      } else {
        rowoff[j] = (unsigned)((long)m * p.lda * 4);
      }
    } else {
      int n = n0 + irow - BM; if (n > p.N - 1) n = p.N - 1;
      rowoff[j] = (unsigned)((long)n * p.K * 4);
    }
  }

  f32x4 stg[NLD];
  auto load_regs = [&](int kt) {
    const int k0 = kt * BK;
    int tap_off = 0, kh = 0, kw = 0;
    if (p.conv) {
      const int tap = k0 / p.Cin; const int cin0 = k0 - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW;
      tap_off = ((kh * p.Wd + kw) * p.Cin + cin0) * 4;
    }
    const int kk = k0 + q8 * 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const bool isA = (j * RPP) < BM;          // compile-time per j (RPP divides BM)
      unsigned off;
      if (isA && p.conv) {
        const int ih = ih0[j] + kh, iw = iw0[j] + kw;
        const bool ok = (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.Wd);
        off = ok ? rowoff[j] + (unsigned)tap_off + (unsigned)(q8 * 16) : OOB_OFF;
      } else {
        off = kk < p.K ? rowoff[j] + (unsigned)(kk * 4) : OOB_OFF;
      }
      stg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isA ? rsA : rsW, off, 0, 0));
    }
  };
  auto split_store = [&](int buf) {
    _Float16* hi = lds + buf * (2 * PLANE);
    _Float16* lo = hi + PLANE;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row = j * RPP + (tid >> 3);
      const f32x4 v = stg[j];
      const h2 a = pkrtz(v[0], v[1]);
      const h2 b = pkrtz(v[2], v[3]);
      const h2 c = pkrtz((v[0] - (float)a[0]) * 2048.f, (v[1] - (float)a[1]) * 2048.f);
      const h2 d = pkrtz((v[2] - (float)b[0]) * 2048.f, (v[3] - (float)b[1]) * 2048.f);
      const int pos = row * BK + ((((q8 >> 1) ^ ((row >> 2) & 3)) << 3) | ((q8 & 1) << 2));
      *reinterpret_cast<h4*>(hi + pos) = h4{a[0], a[1], b[0], b[1]};
      *reinterpret_cast<h4*>(lo + pos) = h4{c[0], c[1], d[0], d[1]};
    }
  };

  f32x16 acc[MT][NT], acx[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }

  const int kbeg = p.ksplit > 1 ? blockIdx.y * p.kchunk : 0;
  const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const int kt0 = kbeg / BK;
  const int nk = (kend - kbeg + BK - 1) / BK;
  const int lr = lane & 31, lh = lane >> 5;

  load_regs(kt0);
  split_store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_regs(kt0 + kt + 1);          // in flight during the MFMAs below
    const _Float16* sH = lds + (kt & 1) * (2 * PLANE);
    const _Float16* sL = sH + PLANE;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      h8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * (BM / WM) + i * 32 + lr;
        const int pos = row * BK + (((2 * g + lh) ^ ((row >> 2) & 3)) << 3);
        ah[i] = *reinterpret_cast<const h8*>(sH + pos);
        al[i] = *reinterpret_cast<const h8*>(sL + pos);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = BM + wn * (BN / WN) + j * 32 + lr;
        const int pos = row * BK + (((2 * g + lh) ^ ((row >> 2) & 3)) << 3);
        bh[j] = *reinterpret_cast<const h8*>(sH + pos);
        bl[j] = *reinterpret_cast<const h8*>(sL + pos);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) split_store((kt + 1) & 1);        // the other buffer: last read one barrier ago
    __syncthreads();
  }

  // ---- epilogue (as gemm.hip): restage the fp32 tile through LDS, float4 per lane -----------------
  float* sC = lds_f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = wn * (BN / WN) + j * 32 + lr;
        sC[row * BN + col] = acc[i][j][r] + acx[i][j][r] * (1.0f / 2048.f);
      }
  __syncthreads();
  constexpr int C4 = BN / 4;
  constexpr int NV = BM * C4 / NT_;
  if (p.ksplit > 1) {
    float* w = p.ws + (long)blockIdx.y * p.M * p.N;
    for (int it = 0; it < NV; ++it) {
      const int idx = it * NT_ + tid;
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
#pragma unroll 4
  for (int it = 0; it < NV; ++it) {
    const int idx = it * NT_ + tid;
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

template <int BM, int BN>
static int launch_f16x3(const GemmParams& p, hipStream_t st) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  size_t smem = (size_t)2 * 2 * (BM + BN) * 32 * sizeof(_Float16);     // 2 stages x (hi, lo) planes
  if (smem < (size_t)BM * BN * sizeof(float)) smem = (size_t)BM * BN * sizeof(float);
  auto kern = gemm_nt_f16x3_kernel<BM, BN, 2, 2>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nbm * nbn, p.ksplit > 1 ? p.ksplit : 1), dim3(256), smem, st, p);
  return mdqe_launch_status();
}

int mdqe_launch_gemm_f16x3(const GemmParams& p, int tile, hipStream_t st) {
  return tile == 2 ? launch_f16x3<128, 64>(p, st) : launch_f16x3<128, 128>(p, st);
}
