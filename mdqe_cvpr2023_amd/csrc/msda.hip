// Multi-scale deformable attention sampling for gfx950 (MI355X), forward (backward: msda_bwd.hip).
//
// Arithmetic follows the reference kernel (mdqe/models/ops/src/cuda/ms_deform_im2col_cuda.cuh
// :33-84 bilinear, :237-299 main loop): pixel = loc*size - 0.5; a sample contributes only if
// -1 < h < H and -1 < w < W; each of the four corners is zero outside the map.
//
// Mapping (CDNA4): one lane owns VEC consecutive channels of one (b,q,head) -- with D=32 and
// VEC=4 a 64-lane wave is exactly one query (8 heads x 8 lanes), every corner fetch is one
// global_load_dwordx4 per lane (1 KiB per wave-instruction in eight 128-B segments) and the
// output row [b,q,:] is written as one contiguous 1 KiB store.  value is read through L1/L2 (the
// per-frame value map, 5.2 MB at 360p, is L2/MALL resident); loc/attn are streamed once.
#include "common.h"

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T ldv(const float* p) {
  return *reinterpret_cast<const typename VecT<VEC>::T*>(p);
}

// LP > 0: compile-time L*P (fully unrolled); LP == 0: runtime loops.
template <int VEC, int LC, int PC>
__global__ void __launch_bounds__(256)
msda_fwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                const float* __restrict__ attn, int B, int S, int M, int D, int G, int Lr, int Q, int Pr,
                float scale, float* __restrict__ out, long total) {
  typedef typename VecT<VEC>::T V;
  const int L = LC > 0 ? LC : Lr;
  const int P = PC > 0 ? PC : Pr;
  const int DV = D / VEC;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % DV);
    long t = idx / DV;
    const int m = (int)(t % M);
    t /= M;                                   // t = b*Q + q
    const int b = (int)(t / Q);
    const long samp = (t * M + m) * (long)(L * P);   // index of (b,q,m,0,0)
    const float* lp = loc + samp * 2;
    const float* wp = attn + samp;
    const int rs = M * D;                     // stride between pixels
    const float* vb = value + (long)b * S * rs + m * D + cv * VEC;
    V acc = {0};
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[(g * L + l) * 2];
        const int W = (int)shapes[(g * L + l) * 2 + 1];
        const float* vl = vb + level_start[g * L + l] * (long)rs;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float lx = lp[(l * P + p) * 2];
          const float ly = lp[(l * P + p) * 2 + 1];
          const float aw = wp[l * P + p];
          const float h_im = ly * H - 0.5f;
          const float w_im = lx * W - 0.5f;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const int h_low = (int)floorf(h_im);
            const int w_low = (int)floorf(w_im);
            const float lh = h_im - h_low, lw = w_im - w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
            const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
            const float* p00 = vl + ((long)h_low * W + w_low) * rs;
            V v1 = {0}, v2 = {0}, v3 = {0}, v4 = {0};
            if (h0 && w0) v1 = ldv<VEC>(p00);
            if (h0 && w1) v2 = ldv<VEC>(p00 + rs);
            if (h1 && w0) v3 = ldv<VEC>(p00 + (long)W * rs);
            if (h1 && w1) v4 = ldv<VEC>(p00 + (long)W * rs + rs);
            const V val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
            acc += val * aw;
          }
        }
      }
    }
    *reinterpret_cast<V*>(out + t * rs + m * D + cv * VEC) = acc * scale;
  }
}

// ---- v2 (D == 32, L*P == 16): cooperative sample set-up (the scheme of msda_fused_v2_kernel) -------------------------
// The 8 lanes that own the 32 channels of one (query, head) all need the same 16 sample descriptors; the kernel above
// recomputes them in every lane (~60 VALU per sample) and is VALU-bound at 0.9 TB/s of compulsory traffic.  Here lane j of
// the group prepares samples 2j and 2j+1 only -- location -> four corner byte offsets and four (bilinear x attention)
// weights -- and publishes them in LDS; then every lane walks the 16 samples with two broadcast ds_read_b128 and four
// bounds-checked buffer loads each (a corner outside the map carries an out-of-range offset and reads as 0: no branches).
#define MSDA_OOB 0xF0000000u
template <int L, int P>
__global__ void __launch_bounds__(256)
msda_fwd_v2_kernel(const float* __restrict__ value, unsigned value_bytes, const int64_t* __restrict__ shapes,
                   const int64_t* __restrict__ level_start, const float* __restrict__ loc, const float* __restrict__ attn,
                   int S, int M, int G, int Q, float scale, float* __restrict__ out, long total) {
  constexpr int LP = L * P;                    // 16
  constexpr int D = 32, DV = 8;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) unsigned soff[32][LP][4];     // [group in block][sample][corner byte offset]
  __shared__ __attribute__((aligned(16))) float swgt[32][LP][4];        // [group in block][sample][corner weight]
  __shared__ int sH[16], sW[16], sS[16];
  if ((int)threadIdx.x < G * L) {
    sH[threadIdx.x] = (int)shapes[threadIdx.x * 2]; sW[threadIdx.x] = (int)shapes[threadIdx.x * 2 + 1];
    sS[threadIdx.x] = (int)level_start[threadIdx.x];
  }
  __syncthreads();
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)value, 0, value_bytes, 0x00020000);
  const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
  const long ldv = (long)M * D;
  // XCD-aware block order: blocks p and p+8 share an XCD (round-robin dispatch).  The first 8*floor(B/8) batch elements are
  // dealt one per XCD (element b entirely on XCD b mod 8: its value map, 5.2 MB at 360p, is fetched into ONE 4-MB L2 instead
  // of all eight); the remainder -- and everything when B < 16 -- keeps the plain order so that no XCD idles.
  const int per_b = Q * M * DV;                // lanes per batch element
  const int nbq = (per_b + 255) / 256;         // blocks per batch element
  const int Bn = (int)(total / per_b);
  const int full = Bn >= 16 ? (Bn / 8) * 8 : 0;
  int b, blk;
  if ((int)blockIdx.x < full * nbq) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int fi = slot / nbq;
    b = xcd + 8 * fi; blk = slot - fi * nbq;
  } else {
    const int r = (int)blockIdx.x - full * nbq;
    b = full + r / nbq; blk = r - (r / nbq) * nbq;
  }
  {
    const int within = blk * 256 + (int)threadIdx.x;
    if (b >= Bn || within >= per_b) return;
    long t = (long)b * Q + within / (M * DV);  // t = b*Q + q
    const int m = (within / DV) % M;
    const long samp = (t * M + m) * (long)LP;  // index of (b,q,m,0,0)
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(loc + (samp + 2 * j) * 2);       // (x,y) of samples 2j, 2j+1
    const f32x2 a2 = *reinterpret_cast<const f32x2*>(attn + samp + 2 * j);
    const unsigned lane_off = (unsigned)((m * D + j * 4) * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = 2 * j + k;
        const int l = i / P;
        const float lx = l4[2 * k], ly = l4[2 * k + 1];
        const float aw = a2[k];
        const int H = sH[g * L + l], W = sW[g * L + l];
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
        const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool h0 = in && h_low >= 0, h1 = in && h_low + 1 <= H - 1;
        const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
        const long prow = (long)b * S + sS[g * L + l] + (long)h_low * W + w_low;       // pixel row of the (low, low) corner
        const unsigned base = (unsigned)(prow * ldv * 4);
        const unsigned dW = (unsigned)((long)W * ldv * 4), d1 = (unsigned)(ldv * 4);
        u32x4 offv;
        f32x4 wv;
        offv[0] = (h0 && w0) ? base : MSDA_OOB;
        offv[1] = (h0 && w1) ? base + d1 : MSDA_OOB;
        offv[2] = (h1 && w0) ? base + dW : MSDA_OOB;
        offv[3] = (h1 && w1) ? base + dW + d1 : MSDA_OOB;
        wv[0] = hh * hw * aw; wv[1] = hh * lw * aw; wv[2] = lh * hw * aw; wv[3] = lh * lw * aw;
        *reinterpret_cast<u32x4*>(&soff[grp][i][0]) = offv;
        *reinterpret_cast<f32x4*>(&swgt[grp][i][0]) = wv;
      }
      __builtin_amdgcn_wave_barrier();         // the 8 lanes of a group sit in one wave: in-order LDS suffices
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const u32x4 offv = *reinterpret_cast<const u32x4*>(&soff[grp][i][0]);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&swgt[grp][i][0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offv[c] + lane_off, 0, 0));
          acc += v * wv[c];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    *reinterpret_cast<f32x4*>(out + t * ldv + m * D + j * 4) = acc * scale;
  }
}


// ---- v3 (one group, D == 32, 16 samples): the coarse levels of a (batch element, head) staged in LDS -- the op-ABI twin of
// msda_fused_v3_kernel (msda_fused.hip).  The level table lives in DEVICE memory here (the reference's ABI), so the host cannot
// size the staging area from it: it grants `budget_px` pixel rows of LDS from S alone (a /2 pyramid's two coarsest levels are
// S * 5/85) and every block stages the longest suffix of levels that fits (none, if the table is not a pyramid: then it is v2
// with another block map).  Same lane mapping and order of operations as v2: equal bits.
template <int L, int P, int NT>
__global__ void __launch_bounds__(NT)
msda_fwd_v3_kernel(const float* __restrict__ value, unsigned value_bytes, const int64_t* __restrict__ shapes,
                   const int64_t* __restrict__ level_start, const float* __restrict__ loc, const float* __restrict__ attn,
                   int B, int S, int M, int Q, int budget_px, int chunk, int nchunk, float scale, float* __restrict__ out) {
  constexpr int LP = L * P, D = 32, HS = LP / 2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  float* stage = smem3;
  const int wave = threadIdx.x >> 6;
  unsigned* soff = reinterpret_cast<unsigned*>(smem3 + (long)(budget_px + 1) * D) + wave * (8 * (HS + 1) * 4 * 2);
  float* swgt = reinterpret_cast<float*>(soff + 8 * (HS + 1) * 4);
  __shared__ int sH[L], sW[L], sS[L], sB[L + 1], sLS;       // sB: first staged row of level l
  if (threadIdx.x == 0) {
    int px = 0, LS = L;
    for (int l = 0; l < L; ++l) { sH[l] = (int)shapes[l * 2]; sW[l] = (int)shapes[l * 2 + 1]; sS[l] = (int)level_start[l]; sB[l] = 0; }
    while (LS > 0) {
      const long n = (long)sH[LS - 1] * sW[LS - 1];
      if (n <= 0 || px + n > budget_px) break;
      --LS; px += (int)n;
    }
    int o = 0;
    for (int l = LS; l < L; ++l) { sB[l] = o; o += sH[l] * sW[l]; }
    sB[L] = o;
    sLS = LS;
  }
  const int nbq = M * nchunk;
  const int full = B >= 16 ? (B / 8) * 8 : 0;
  int b, blk;
  if ((int)blockIdx.x < full * nbq) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int fi = slot / nbq;
    b = xcd + 8 * fi; blk = slot - fi * nbq;
  } else {
    const int r = (int)blockIdx.x - full * nbq;
    b = full + r / nbq; blk = r - (r / nbq) * nbq;
  }
  if (b >= B) return;
  const int m = blk % M, ck = blk / M;
  const long ldv = (long)M * D;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)value, 0, value_bytes, 0x00020000);
  __syncthreads();
  const int LS = sLS, npx = sB[L];
  for (int idx = threadIdx.x; idx < (npx + 1) * 8; idx += NT) {
    const int px = idx >> 3, c = idx & 7;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (px < npx) {
      int l = LS;
      while (l + 1 < L && px >= sB[l + 1]) ++l;
      v = *reinterpret_cast<const f32x4*>(value + ((long)b * S + sS[l] + (px - sB[l])) * ldv + m * D + c * 4);
    }
    *reinterpret_cast<f32x4*>(stage + (long)px * D + c * 4) = v;         // row npx: zeros
  }
  __syncthreads();
  const int grp = (threadIdx.x >> 3) & 7, j = threadIdx.x & 7;
  const unsigned lane_off = (unsigned)((m * D + j * 4) * 4), lane_lds = (unsigned)(j * 16);
  const unsigned zero_row = (unsigned)(npx * D * 4);
  const int q_end = min(Q, (ck + 1) * chunk);
  for (int q0 = ck * chunk; q0 < q_end; q0 += NT / 8) {
    const int q = q0 + (threadIdx.x >> 3);
    const bool live = q < q_end;
    const long t = (long)b * Q + (live ? q : q_end - 1);
    const long samp = (t * M + m) * (long)LP;
    const f32x4 l4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(loc + (samp + 2 * j) * 2));   // read once: past the caches
    const f32x2 a2 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(attn + samp + 2 * j));
    u32x4 offv[2];
    f32x4 wv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = 2 * j + k;
      const int l = i / P;
      const float lx = l4[2 * k], ly = l4[2 * k + 1];
      const float aw = a2[k];
      const int H = sH[l], W = sW[l];
      const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
      const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - h_low, lw = w_im - w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool h0 = in && h_low >= 0, h1 = in && h_low + 1 <= H - 1;
      const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
      unsigned base, dW, d1, oob;
      if (l >= LS) {
        base = (unsigned)((sB[l] + h_low * W + w_low) * (D * 4));
        dW = (unsigned)(W * D * 4); d1 = (unsigned)(D * 4); oob = zero_row;
      } else {
        const long prow = (long)b * S + sS[l] + (long)h_low * W + w_low;
        base = (unsigned)(prow * ldv * 4);
        dW = (unsigned)((long)W * ldv * 4); d1 = (unsigned)(ldv * 4); oob = MSDA_OOB;
      }
      offv[k][0] = (h0 && w0) ? base : oob;
      offv[k][1] = (h0 && w1) ? base + d1 : oob;
      offv[k][2] = (h1 && w0) ? base + dW : oob;
      offv[k][3] = (h1 && w1) ? base + dW + d1 : oob;
      wv[k][0] = hh * hw * aw; wv[k][1] = hh * lw * aw; wv[k][2] = lh * hw * aw; wv[k][3] = lh * lw * aw;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((j >> 2) == half) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int slot = 2 * (j & 3) + k;
          *reinterpret_cast<u32x4*>(soff + (grp * (HS + 1) + slot) * 4) = offv[k];
          *reinterpret_cast<f32x4*>(swgt + (grp * (HS + 1) + slot) * 4) = wv[k];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        const int l = (half * HS + s) / P;        // compile-time
        const u32x4 o = *reinterpret_cast<const u32x4*>(soff + (grp * (HS + 1) + s) * 4);
        const f32x4 w = *reinterpret_cast<const f32x4*>(swgt + (grp * (HS + 1) + s) * 4);
        if (l >= LS) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stage) + o[c] + lane_lds);
            acc += v * w[c];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o[c] + lane_off, 0, 0));
            acc += v * w[c];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (live) __builtin_nontemporal_store(acc * scale, reinterpret_cast<f32x4*>(out + t * ldv + m * D + j * 4));
  }
}

static int g_msda_op_staged = 1;   // tools/ A/B: 0 = every corner through the texture path (v2)
extern "C" int mdqe_debug_msda_op_staged(int v) { g_msda_op_staged = v; return MDQE_OK; }

template <int VEC>
static int launch_msda(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                       const float* attn, int B, int S, int M, int D, int G, int L, int Q, int P, float scale,
                       float* out, hipStream_t st) {
  const long total = (long)B * Q * M * (D / VEC);
  if (total == 0) return MDQE_OK;
  const int block = 256;
  long nb = (total + block - 1) / block;
  if (nb > 256L * 64) nb = 256L * 64;          // grid-stride beyond 64 blocks per CU
  if (L == 4 && P == 4)
    hipLaunchKernelGGL((msda_fwd_kernel<VEC, 4, 4>), dim3((unsigned)nb), dim3(block), 0, st, value, shapes, level_start,
                       loc, attn, B, S, M, D, G, L, Q, P, scale, out, total);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<VEC, 0, 0>), dim3((unsigned)nb), dim3(block), 0, st, value, shapes, level_start,
                       loc, attn, B, S, M, D, G, L, Q, P, scale, out, total);
  return mdqe_launch_status();
}

extern "C" int mdqe_msda_forward_grouped_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                             const float* loc, const float* attn, int B, int S, int M, int D, int G,
                                             int L, int Q, int P, float scale, float* out, void* stream) {
  MDQE_REQUIRE(B >= 0 && S >= 0 && M > 0 && D > 0 && G > 0 && L > 0 && Q >= 0 && P > 0);
  if ((long)B * Q == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(shapes); MDQE_CHECK_PTR(level_start);
  MDQE_CHECK_PTR(loc); MDQE_CHECK_PTR(attn); MDQE_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  const bool al16 = (((uintptr_t)value | (uintptr_t)out) & 15) == 0;
  const long vbytes = (long)B * S * M * D * 4;
  if (D == 32 && L * P == 16 && (L == 4 || L == 2) && G * L <= 16 && al16 && vbytes > 0 && vbytes < 0xF0000000L &&
      (((uintptr_t)loc | (uintptr_t)attn) & 15) == 0) {
    if (g_msda_op_staged && G == 1 && L == 4 && S >= 64) {
      // v3: coarse levels in LDS.  Budget from S alone (the level table is device memory): the two coarsest levels of a /2 pyramid
      // are 5/85 of the tokens; at most 960 rows = 120 KB beside the descriptors (36 KB) in the 160-KB LDS
      long budget = S / 16 + 8;
      if (budget > 960) budget = 960;
      const long desc = 16L * 8 * (8 + 1) * 4 * 2 * 4;
      const long staged = (budget + 1) * 32 * 4;
      const int chunk = staged < 45 * 1024 ? 128 : staged < 80 * 1024 ? 256 : 512;      // as msda_fused_v3 (measured there)
      const int nchunk = (Q + chunk - 1) / chunk;
      auto kern = msda_fwd_v3_kernel<4, 4, 1024>;
      if (mdqe_allow_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 256) != hipSuccess) return MDQE_ELAUNCH;
      hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * M * nchunk)), dim3(1024), (size_t)(staged + desc), st, value, (unsigned)vbytes, shapes,
                         level_start, loc, attn, B, S, M, Q, (int)budget, chunk, nchunk, scale, out);
      return mdqe_launch_status();
    }
    const long total = (long)B * Q * M * 8;
    const long nb = (long)B * (((long)Q * M * 8 + 255) / 256);                 // exact grid: blocks per batch element x B
    if (L == 4) hipLaunchKernelGGL((msda_fwd_v2_kernel<4, 4>), dim3((unsigned)nb), dim3(256), 0, st, value, (unsigned)vbytes, shapes,
                                   level_start, loc, attn, S, M, G, Q, scale, out, total);
    else hipLaunchKernelGGL((msda_fwd_v2_kernel<2, 8>), dim3((unsigned)nb), dim3(256), 0, st, value, (unsigned)vbytes, shapes,
                            level_start, loc, attn, S, M, G, Q, scale, out, total);
    return mdqe_launch_status();
  }
  const bool al8 = (((uintptr_t)value | (uintptr_t)out) & 7) == 0;
  if (D % 4 == 0 && al16) return launch_msda<4>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
  if (D % 2 == 0 && al8) return launch_msda<2>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
  return launch_msda<1>(value, shapes, level_start, loc, attn, B, S, M, D, G, L, Q, P, scale, out, st);
}

extern "C" int mdqe_msda_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                     const float* loc, const float* attn, int B, int S, int M, int D, int L, int Q,
                                     int P, float* out, void* stream) {
  return mdqe_msda_forward_grouped_f32(value, shapes, level_start, loc, attn, B, S, M, D, 1, L, Q, P, 1.0f, out, stream);
}
