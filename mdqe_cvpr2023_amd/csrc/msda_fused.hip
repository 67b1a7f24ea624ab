// Fused MSDeformAttn core for the in-repo pipeline (gfx950).
//
// The reference materialises sampling_locations [B,Q,M,L,P,2] and softmaxed attention_weights
// [B,Q,M,L,P] in HBM between PyTorch ops and its native kernel
// (mdqe/models/ops/modules/ms_deform_attn.py:141-170 / 198-233).  Here one kernel consumes the raw
// outputs of the projection GEMM -- value, sampling offsets (or grid deltas) and attention logits
// live side by side in one [rows, ld] buffer -- and does
//     loc   = ref_xy + off/8                                        (encoder, pred_offsets=True, :155)
//     loc   = ref_xy + (grid*0.5*wh + clamp(delta, +-8*wh))/8       (decoder, pred_offsets=False, :146-155)
//     w     = softmax over the L*P logits of (query, head)           (:157-161)
//     out   = scale * sum_g sum_l sum_p w * bilinear_zero_pad(value_{g,l}, loc)    (native op + mean, :235)
// Thread mapping as in msda.hip: a lane owns 4 channels of one (query, head); a wave64 = one query
// when M*D/4 = 64.  Level geometry comes in by value (no device table reads).
#include "common.h"
#include <type_traits>

struct MsdaLevels { int H[16]; int W[16]; int start[16]; };

template <int L, int P>
__global__ void __launch_bounds__(256)
msda_fused_kernel(const float* __restrict__ value, long ldv, long v_brows, const int* __restrict__ vidx,
                  const float* __restrict__ offs, long ldo, const float* __restrict__ logits, long ldl,
                  const float* __restrict__ ref, long ref_bstride, int ref_dim, int mode,
                  const float* __restrict__ grid, MsdaLevels lv,
                  int B, int M, int D, int G, int Q, float scale, float* __restrict__ out, long ldout, long total) {
  constexpr int LP = L * P;
  const int DV = D / 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % DV);
    long t = idx / DV;
    const int m = (int)(t % M);
    t /= M;                                    // t = b*Q + q
    const int b = (int)(t / Q);
    const int q = (int)(t - (long)b * Q);
    // ---- per-(query, head) scalars: 2*LP offsets + LP logits --------------------------------------
    float off[2 * LP], lg[LP];
    {
      const float* op = offs + t * ldo + m * (2 * LP);
      const float* lp = logits + t * ldl + m * LP;
#pragma unroll
      for (int i = 0; i < 2 * LP / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(op + i * 4);
        off[i * 4] = v[0]; off[i * 4 + 1] = v[1]; off[i * 4 + 2] = v[2]; off[i * 4 + 3] = v[3];
      }
#pragma unroll
      for (int i = 0; i < LP / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lp + i * 4);
        lg[i * 4] = v[0]; lg[i * 4 + 1] = v[1]; lg[i * 4 + 2] = v[2]; lg[i * 4 + 3] = v[3];
      }
    }
    float mx = lg[0];
#pragma unroll
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
    const float inv = 1.0f / sum;
    const float* rp = ref + (long)b * ref_bstride + (long)q * ref_dim;
    const float rx = rp[0], ry = rp[1];
    float bw = 0.f, bh = 0.f;
    if (mode == 1) { bw = rp[2]; bh = rp[3]; }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long brow = vidx != nullptr ? (long)vidx[b] * v_brows : (long)b * v_brows;
    const float* vb = value + brow * ldv + m * D + cv * 4;
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = l * P + p;
        float ox = off[2 * i], oy = off[2 * i + 1];
        if (mode == 1) {
          const float gx = grid[(m * LP + i) * 2], gy = grid[(m * LP + i) * 2 + 1];
          ox = fminf(fmaxf(ox, -bw * 8.f), bw * 8.f);
          oy = fminf(fmaxf(oy, -bh * 8.f), bh * 8.f);
          ox = gx * 0.5f * bw + ox;
          oy = gy * 0.5f * bh + oy;
        }
        const float lx = rx + ox / 8.f, ly = ry + oy / 8.f;
        const float aw = lg[i] * inv;
        for (int g = 0; g < G; ++g) {
          const int H = lv.H[g * L + l], W = lv.W[g * L + l];
          const float* vl = vb + (long)lv.start[g * L + l] * ldv;
          const float h_im = ly * H - 0.5f;
          const float w_im = lx * W - 0.5f;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const int h_low = (int)floorf(h_im);
            const int w_low = (int)floorf(w_im);
            const float lh = h_im - h_low, lw = w_im - w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool h0 = h_low >= 0, h1 = h_low + 1 <= H - 1;
            const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
            const float* p00 = vl + ((long)h_low * W + w_low) * ldv;
            f32x4 v1 = {0, 0, 0, 0}, v2 = {0, 0, 0, 0}, v3 = {0, 0, 0, 0}, v4 = {0, 0, 0, 0};
            if (h0 && w0) v1 = *reinterpret_cast<const f32x4*>(p00);
            if (h0 && w1) v2 = *reinterpret_cast<const f32x4*>(p00 + ldv);
            if (h1 && w0) v3 = *reinterpret_cast<const f32x4*>(p00 + (long)W * ldv);
            if (h1 && w1) v4 = *reinterpret_cast<const f32x4*>(p00 + (long)W * ldv + ldv);
            const f32x4 val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
            acc += val * aw;
          }
        }
      }
    }
    *reinterpret_cast<f32x4*>(out + t * ldout + m * D + cv * 4) = acc * scale;
  }
}

// ---- v2 (D == 32, L*P == 16): cooperative sample set-up --------------------------------------------------------------
// The 8 lanes that own the 32 channels of one (query, head) all need the same 16 sample descriptors.  v1 recomputes them in
// every lane (~60 VALU x 16 samples).  Here lane j of the group prepares samples 2j and 2j+1 only -- softmax across the
// group with 3 xor-shuffles, locations, the four corner byte offsets and the four (bilinear x attention) weights -- and
// publishes 8 floats per sample in LDS; then every lane walks the 16 samples with two broadcast ds_read_b128 and four
// bounds-checked buffer loads each (a corner outside the map carries an out-of-range offset and reads as 0: no branches).
#define MSDA_OOB 0xF0000000u
// cache-policy bits of the gather's buffer loads (aux: 1 = sc0, 2 = nt, 16 = sc1); 0 = default policy.  tools/msda_aux_ab.sh builds the
// library once per policy and times the three launches (profiles/r06_msda_aux_ab.txt)
#ifndef MSDA_GATHER_AUX
#define MSDA_GATHER_AUX 0
#endif
#define MSDA_DEFAULT_VARIANT 0
// MAP: how a block's 32 (query, head) groups are chosen inside a batch element.  0: 4 consecutive queries x 8 heads.
// 1: 32 consecutive queries of ONE head -- a head samples along its own direction (ms_deform_attn.py:81-87), so neighbouring
// queries of the same head read neighbouring pixels of the same 128-B head slice: the block's lines overlap in L1.
// 2: as 1 with the 32 queries an 8 x 4 patch of their level (overlap in y as well).
// DD = head width: 32 (8 lanes x 4 channels) or 24 (Swin-L's hidden 192: the group keeps its 8 lanes for the sample set-up, lanes
// 6 and 7 carry out-of-range offsets in the gather -- they read zeros -- and do not store).
template <int L, int P, int MAP, int WPE, int DD = 32>
__global__ void __launch_bounds__(256, WPE)
msda_fused_v2_kernel(const float* __restrict__ value, long value_bytes, long ldv, long v_brows, const int* __restrict__ vidx,
                     const float* __restrict__ offs, long ldo, const float* __restrict__ logits, long ldl,
                     const float* __restrict__ ref, long ref_bstride, int ref_dim, int mode,
                     const float* __restrict__ grid, MsdaLevels lv,
                     int B, int M, int G, int Q, float scale, float* __restrict__ out, long ldout, long total, int xcd_order) {
  constexpr int LP = L * P;                    // 16
  constexpr int D = DD, DV = 8;               // DV: lanes of a (query, head) group (D / 4 of them carry channels)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // [group in block][sample (+1 pad)][corner]: the pad makes the group stride 272 B, so the two groups served by one 16-lane
  // pass of a ds_read/write_b128 fall on disjoint banks (a 256-B stride put every group on the same 4 banks: r01 PMC showed
  // ~20 % of the kernel's cycles in LDS bank conflicts)
  __shared__ __attribute__((aligned(16))) unsigned soff[32][LP + 1][4];  // corner byte offsets
  __shared__ __attribute__((aligned(16))) float swgt[32][LP + 1][4];     // corner weights (bilinear x attention)
  __shared__ int sH[16], sW[16], sS[16];                                 // level table (indexed with a runtime level)
  if (threadIdx.x < 16) { sH[threadIdx.x] = lv.H[threadIdx.x]; sW[threadIdx.x] = lv.W[threadIdx.x]; sS[threadIdx.x] = lv.start[threadIdx.x]; }
  __syncthreads();
  const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
  // XCD-aware block order: blocks p and p+8 share an XCD (round-robin dispatch).  The first 8*floor(B/8) batch elements are
  // dealt one per XCD (element b entirely on XCD b mod 8: its value map, 5.2 MB at 360p, is fetched into ONE 4-MB L2 instead
  // of all eight); the remainder -- and everything when B < 16 -- keeps the plain order so that no XCD idles.
  const int per_b = Q * M * DV;                // lanes per batch element
  int nbq;                                     // blocks per batch element
  if (MAP == 0) nbq = (per_b + 255) / 256;
  else if (MAP == 1) nbq = ((Q + 31) / 32) * M;
  else { nbq = 0; for (int l = 0; l < L; ++l) nbq += ((sW[l] + 7) / 8) * ((sH[l] + 3) / 4); nbq *= M; }
  const int Bn = B;
  const int full = (xcd_order && Bn >= 16) ? (Bn / 8) * 8 : 0;
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
    int q, m;
    if (MAP == 0) {
      const int within = blk * 256 + (int)threadIdx.x;
      if (b >= B || within >= per_b) return;
      q = within / (M * DV);
      m = (within / DV) % M;
    } else if (MAP == 1) {
      m = blk % M;
      q = (blk / M) * 32 + grp;
      if (b >= B || q >= Q) return;
    } else {
      m = blk % M;
      int tile = blk / M, l = 0, tw = (sW[0] + 7) / 8, nt = tw * ((sH[0] + 3) / 4);
      while (l + 1 < L && tile >= nt) { tile -= nt; ++l; tw = (sW[l] + 7) / 8; nt = tw * ((sH[l] + 3) / 4); }
      const int y = (tile / tw) * 4 + (grp >> 3), x = (tile % tw) * 8 + (grp & 7);
      if (b >= B || y >= sH[l] || x >= sW[l]) return;
      q = sS[l] + y * sW[l] + x;               // encoder: the queries ARE the tokens of the levels (one group, G == 1)
    }
    const long t = (long)b * Q + q;
    // ---- this lane's two samples: i0 = 2j, i1 = 2j+1
    // offsets / logits / output are touched once: streamed past the caches so that they do not evict the value map's lines
    const f32x4 o4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(offs + t * ldo + m * (2 * LP) + 4 * j));
    const f32x2 l2 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(logits + t * ldl + m * LP + 2 * j));
    float mx = fmaxf(l2[0], l2[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    const float e0 = expf(l2[0] - mx), e1 = expf(l2[1] - mx);
    float sm = e0 + e1;
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float inv = 1.0f / sm;
    const float* rp = ref + (long)b * ref_bstride + (long)q * ref_dim;
    const float rx = rp[0], ry = rp[1];
    float bw = 0.f, bh = 0.f;
    if (mode == 1) { bw = rp[2]; bh = rp[3]; }
    // The buffer resource starts at THIS element's value block (b is uniform over the block): the 32-bit offsets then span one
    // element's levels -- one frame, or the frames of a clip in the temporal form -- not the whole cache (4.3 GB at 640p)
    const long brow = __builtin_amdgcn_readfirstlane(vidx != nullptr ? vidx[b] : b) * v_brows;
    const long left = value_bytes - brow * ldv * 4;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(value + brow * ldv), 0, (unsigned)(left < 0xF0000000L ? left : 0xF0000000L), 0x00020000);
    const bool chan = j * 4 < D;
    const unsigned lane_off = chan ? (unsigned)((m * D + j * 4) * 4) : MSDA_OOB;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = 2 * j + k;
        const int l = i / P;
        float ox = o4[2 * k], oy = o4[2 * k + 1];
        if (mode == 1) {
          const float gx = grid[(m * LP + i) * 2], gy = grid[(m * LP + i) * 2 + 1];
          ox = fminf(fmaxf(ox, -bw * 8.f), bw * 8.f);
          oy = fminf(fmaxf(oy, -bh * 8.f), bh * 8.f);
          ox = gx * 0.5f * bw + ox;
          oy = gy * 0.5f * bh + oy;
        }
        const float lx = rx + ox / 8.f, ly = ry + oy / 8.f;
        const float aw = (k == 0 ? e0 : e1) * inv;
        const int H = sH[g * L + l], W = sW[g * L + l];
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
        const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool h0 = in && h_low >= 0, h1 = in && h_low + 1 <= H - 1;
        const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
        const long prow = sS[g * L + l] + (long)h_low * W + w_low;                    // pixel row of the (low, low) corner within the element
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
          const unsigned o = chan ? offv[c] + lane_off : MSDA_OOB;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, MSDA_GATHER_AUX));
          acc += v * wv[c];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (chan) __builtin_nontemporal_store(acc * scale, reinterpret_cast<f32x4*>(out + t * ldout + m * D + j * 4));
  }
}


// ---- v3 (encoder: mode 0, one group, the queries are the level tokens): the COARSE levels live in LDS -------------------
// A head's slice of a level is H*W rows of D floats: at 360p the two coarsest levels (12x20 + 6x10 tokens) are 38 KB per
// (frame, head), at 640p 115 KB -- they fit the 160-KB LDS, and half of every query's 16 samples fall on them.  A block of
// 1024 threads owns one (frame, head) and a run of queries: it stages the levels [LS, L) of its head once (coalesced 128-B
// rows), then walks its queries 128 at a time with exactly v2's lane mapping and arithmetic (8 lanes per (query, head), 4
// channels each, samples in order, corners in order -- bit-identical results); samples on a staged level read their four
// corners with ds_read_b128 instead of going through the texture path, which is what bounds v2 (TA busy 0.62-0.71, L1 hit 65 %).
// A corner outside the map points at a zero row behind the staged levels (v2: an out-of-range buffer offset that reads as 0).
// The sample descriptors take half of v2's LDS: the 16 samples go in two halves of 8 (same order).
// WPE: waves per SIMD the register allocation must allow (0: no constraint).  The natural allocation is 68 VGPRs = 7 waves per SIMD,
// ONE wave short of what two 1024-thread blocks per CU need (2 x 16 waves = 8 per SIMD): with WPE = 8 the same code takes 64 VGPRs
// (no spill) and the second block of a CU -- which the 38 + 37 KB of LDS per block always allowed -- becomes resident, so one block
// stages / sets up while the other gathers.
// LSC: the first staged level as a COMPILE-TIME constant (-1: the runtime argument).  With a runtime LS every sample's `level staged?`
// test is a (wave-uniform) branch, i.e. a basic-block boundary per sample: the compiler drains `vmcnt` at each one and a wave never has
// more than the 4 corner loads of ONE sample in flight -- in places one (round 4, from the ISA: `buffer_load; s_waitcnt vmcnt(0)` four
// times in a row).  With LSC the 8 samples of a half are straight-line code and the loads of several samples are in flight together.
// GS: samples whose corner loads are in flight together in the LSC form (4 = a whole level: 64 VGPRs of data, 126 in all -- one
// 1024-thread block per CU; 2 was measured in the 8-waves-per-SIMD build: 25 spilled registers, 737 us against 435, not instantiated).
template <int L, int P, int DD, int NT, int WPE = 0, int LSC = -1, int GS = 4>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE > 0 ? WPE : 1, WPE > 0 ? WPE : 8)))
msda_fused_v3_kernel(const float* __restrict__ value, long value_bytes, long ldv, long v_brows, const int* __restrict__ vidx,
                     const float* __restrict__ offs, long ldo, const float* __restrict__ logits, long ldl,
                     const float* __restrict__ ref, long ref_bstride, int ref_dim, int mode, const float* __restrict__ grid, MsdaLevels lv,
                     int B, int M, int Q, int LS_rt, int stage_px, int chunk, int nchunk, float scale,
                     float* __restrict__ out, long ldout, int xcd_order, int patch) {
  const int LS = LSC >= 0 ? LSC : LS_rt;
  constexpr int LP = L * P;                    // 16
  constexpr int D = DD, HS = LP / 2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [stage_px + 1 rows of D floats | per wave: 8 groups x (HS + 1) samples x 4 corner offsets | the same of weights]
  float* stage = smem;
  const int wave = threadIdx.x >> 6;
  unsigned* soff = reinterpret_cast<unsigned*>(smem + (long)(stage_px + 1) * D) + wave * (8 * (HS + 1) * 4 * 2);
  float* swgt = reinterpret_cast<float*>(soff + 8 * (HS + 1) * 4);
  __shared__ int sH[16], sW[16], sS[16];
  if (threadIdx.x < 16) { sH[threadIdx.x] = lv.H[threadIdx.x]; sW[threadIdx.x] = lv.W[threadIdx.x]; sS[threadIdx.x] = lv.start[threadIdx.x]; }
  // block -> (frame, head, run of queries); frames dealt one per XCD as in v2
  const int nbq = M * nchunk;
  const int full = (xcd_order && B >= 16) ? (B / 8) * 8 : 0;
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
  const long brow = __builtin_amdgcn_readfirstlane(vidx != nullptr ? vidx[b] : b) * v_brows;
  const long left = value_bytes - brow * ldv * 4;     // the resource starts at this element's value block (see v2)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(value + brow * ldv), 0, (unsigned)(left < 0xF0000000L ? left : 0xF0000000L), 0x00020000);
  __syncthreads();
  {                                             // stage the head's slice of the levels [LS, L): tokens sS[LS] .. sS[LS] + stage_px
    const int s0 = sS[LS];
    for (int idx = threadIdx.x; idx < (stage_px + 1) * 8; idx += NT) {
      const int px = idx >> 3, c = idx & 7;
      if (c * 4 >= D) continue;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (px < stage_px) v = *reinterpret_cast<const f32x4*>(value + (brow + s0 + px) * ldv + m * D + c * 4);
      *reinterpret_cast<f32x4*>(stage + (long)px * D + c * 4) = v;       // row stage_px: zeros
    }
  }
  __syncthreads();
  const int grp = (threadIdx.x >> 3) & 7, j = threadIdx.x & 7;             // group within the wave
  const bool chan = j * 4 < D;
  const unsigned lane_off = chan ? (unsigned)((m * D + j * 4) * 4) : MSDA_OOB;
  const unsigned lane_lds = chan ? (unsigned)(j * 16) : 0u;
  const unsigned zero_row = (unsigned)(stage_px * D * 4);
  // patch (round 6 experiment, OFF by default; encoder only, NT == 1024: the queries are the level tokens in raster order): the 128 queries
  // of an iteration as an 8 x 16 PATCH of their level instead of a run of 128 consecutive tokens (1.6 rows of level 0) -- a wave still takes
  // 8 consecutive tokens of a row, the block's 16 waves cover 8 rows x 16 columns whose samples land in one neighbourhood of the fine levels.
  // Same arithmetic per query: identical bits.  MEASURED SLOWER on the bench's own model and video (profiles/r06_msda_patch_ab.txt: 505 vs
  // 462 us per 40-frame 360p launch, 675 vs 656 at 640p, 221 vs 203 on Swin-L; 826 vs 833 frames/s): a head samples along its own direction
  // (ms_deform_attn.py:81-87) and a row run re-uses the texture path's lines along x better than a patch does, and partial patches of the
  // coarse levels idle lanes.  Kept behind MDQE_MSDA_PATCH=1 / mdqe_debug_msda_patch for the record.
  const int ppb = chunk / (NT / 8);
  const int q_end = patch ? Q : min(Q, (ck + 1) * chunk);
  for (int q0 = patch ? 0 : ck * chunk, it = 0; patch ? it < ppb : q0 < q_end; q0 += NT / 8, ++it) {
    int q = q0 + (threadIdx.x >> 3);
    bool live = q < q_end;
    if (patch) {
      int pid = ck * ppb + it, l = 0, npx = (sW[0] + 15) >> 4, np = npx * ((sH[0] + 7) >> 3);
      while (l + 1 < L && pid >= np) { pid -= np; ++l; npx = (sW[l] + 15) >> 4; np = npx * ((sH[l] + 7) >> 3); }
      if (pid >= np) break;                      // (uniform: behind the last patch of the last level)
      const int g = threadIdx.x >> 3;            // group in the block: row g / 16, column g % 16 of the patch
      const int y = (pid / npx) * 8 + (g >> 4), x = (pid % npx) * 16 + (g & 15);
      live = y < sH[l] && x < sW[l];
      q = sS[l] + min(y, sH[l] - 1) * sW[l] + min(x, sW[l] - 1);
    }
    const int qq = live ? q : (patch ? q : q_end - 1);         // idle groups shadow a live query and do not store
    const long t = (long)b * Q + qq;
    const f32x4 o4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(offs + t * ldo + m * (2 * LP) + 4 * j));
    const f32x2 l2 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(logits + t * ldl + m * LP + 2 * j));
    float mx = fmaxf(l2[0], l2[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    const float e0 = expf(l2[0] - mx), e1 = expf(l2[1] - mx);
    float sm = e0 + e1;
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float inv = 1.0f / sm;
    const float* rp = ref + (long)b * ref_bstride + (long)qq * ref_dim;
    const float rx = rp[0], ry = rp[1];
    float bw = 0.f, bh = 0.f;
    if (mode == 1) { bw = rp[2]; bh = rp[3]; }
    u32x4 offv[2];
    f32x4 wv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = 2 * j + k;
      const int l = i / P;
      float ox = o4[2 * k], oy = o4[2 * k + 1];
      if (mode == 1) {                           // decoder: grid pattern scaled by the box + clamped delta (as v2)
        const float gx = grid[(m * LP + i) * 2], gy = grid[(m * LP + i) * 2 + 1];
        ox = fminf(fmaxf(ox, -bw * 8.f), bw * 8.f);
        oy = fminf(fmaxf(oy, -bh * 8.f), bh * 8.f);
        ox = gx * 0.5f * bw + ox;
        oy = gy * 0.5f * bh + oy;
      }
      const float lx = rx + ox / 8.f, ly = ry + oy / 8.f;
      const float aw = (k == 0 ? e0 : e1) * inv;
      const int H = sH[l], W = sW[l];
      const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
      const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - h_low, lw = w_im - w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool h0 = in && h_low >= 0, h1 = in && h_low + 1 <= H - 1;
      const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
      unsigned base, dW, d1, oob;
      if (l >= LS) {                             // staged level: byte offsets into `stage`
        base = (unsigned)((sS[l] - sS[LS] + h_low * W + w_low) * (D * 4));
        dW = (unsigned)(W * D * 4); d1 = (unsigned)(D * 4); oob = zero_row;
      } else {
        const long prow = sS[l] + (long)h_low * W + w_low;
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
      if ((j >> 2) == half) {                    // lanes 0-3 hold samples 0-7, lanes 4-7 samples 8-15
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int slot = 2 * (j & 3) + k;
          *reinterpret_cast<u32x4*>(soff + (grp * (HS + 1) + slot) * 4) = offv[k];
          *reinterpret_cast<f32x4*>(swgt + (grp * (HS + 1) + slot) * 4) = wv[k];
        }
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (LSC >= 0) {
        // One LEVEL at a time (P == 4: samples 4g .. 4g+3 of the half): its 4 descriptors, then all 16 corner loads in flight together,
        // then the 16 multiply-adds in v2's order (sample by sample, corner by corner: the same bits).  The scheduling fences keep the
        // compiler from hoisting the next level's loads over this one's (128 live VGPRs of data and spills, as it does unfenced).
#pragma unroll
        for (int gq = 0; gq < HS / GS; ++gq) {
          constexpr int PP = GS;
          const int l = (half * HS + gq * GS) / P;   // compile-time (GS divides P: a group never straddles two levels)
          u32x4 o[PP];
          f32x4 w[PP], v[PP][4];
#pragma unroll
          for (int s = 0; s < PP; ++s) {
            o[s] = *reinterpret_cast<const u32x4*>(soff + (grp * (HS + 1) + gq * PP + s) * 4);
            w[s] = *reinterpret_cast<const f32x4*>(swgt + (grp * (HS + 1) + gq * PP + s) * 4);
          }
#pragma unroll
          for (int s = 0; s < PP; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (l >= LS) v[s][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stage) + o[s][c] + lane_lds);
              else v[s][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, chan ? o[s][c] + lane_off : MSDA_OOB, 0, MSDA_GATHER_AUX));
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < PP; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += v[s][c] * w[s][c];
          // (pin the chain here: `acc` is only stored under `chan && live`, and LLVM otherwise SINKS all 64 multiply-adds into that
          // block, keeping every loaded value alive -- and spilled -- until the end of the query)
          asm volatile("" : "+v"(acc));
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
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
              const unsigned a = chan ? o[c] + lane_off : MSDA_OOB;
              const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a, 0, MSDA_GATHER_AUX));
              acc += v * w[c];
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (chan && live) __builtin_nontemporal_store(acc * scale, reinterpret_cast<f32x4*>(out + t * ldout + m * D + j * 4));
  }
}

// ---- temporal form (decoder instance level, transformer_dec.py:378-390 -> ms_deform_attn.py:172-236): every (clip, query, head) takes
// F frames x P points, each looked up on ALL G levels of its frame and averaged (scale = 1/G).  The four frames' coarse levels do not
// fit the LDS together (4 x 38 KB at 360p), so the block walks the frames: stage frame f's coarse levels [LS, G) of its head, take
// the P points of frame f on the G levels (16 look-ups, the staged half of them through ds_read_b128), next frame -- the accumulator
// stays in registers, 2 barriers per frame.  Lane mapping of the set-up as v2 / v3 (lane j of a (query, head) group prepares look-ups
// 2j and 2j+1 of the phase: level (2j+k) / P, point (2j+k) % P).  Sum order: frame-major (v2: level-major) -- the same value up to fp32
// reassociation, held to v2 at 1e-5 (tests/test_kernels_gpu.py) and to the reference through the decoder goldens.
template <int F, int P, int G, int DD, int NT, int LSC = -1, int WPE = 0>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE > 0 ? WPE : 1, WPE > 0 ? WPE : 8)))
msda_fused_tp_kernel(const float* __restrict__ value, long value_bytes, long ldv, long v_brows, const int* __restrict__ vidx,
                     const float* __restrict__ offs, long ldo, const float* __restrict__ logits, long ldl,
                     const float* __restrict__ ref, long ref_bstride, const float* __restrict__ grid, MsdaLevels lv,
                     int B, int M, int Q, int LS_rt, int stage_px, int chunk, int nchunk, float scale,
                     float* __restrict__ out, long ldout) {
  const int LS = LSC >= 0 ? LSC : LS_rt;         // compile-time first staged level: straight-line gather, a level's loads in flight together (v3)
  constexpr int FP = F * P;                    // 16 (frame, point) samples share one softmax
  constexpr int GP = G * P;                    // 16 look-ups per frame phase
  constexpr int D = DD, HS = GP / 2;
  static_assert(FP == 16 && GP == 16 && P == 4, "set-up lanes are wired for 16 samples of 4 points");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* stage = smem;
  const int wave = threadIdx.x >> 6;
  unsigned* soff = reinterpret_cast<unsigned*>(smem + (long)(stage_px + 1) * D) + wave * (8 * (HS + 1) * 4 * 2);
  float* swgt = reinterpret_cast<float*>(soff + 8 * (HS + 1) * 4);
  __shared__ int sH[16], sW[16], sS[16];       // [level g][frame f] as the host table: index g * F + f
  if (threadIdx.x < 16) { sH[threadIdx.x] = lv.H[threadIdx.x]; sW[threadIdx.x] = lv.W[threadIdx.x]; sS[threadIdx.x] = lv.start[threadIdx.x]; }
  const int nbq = M * nchunk;
  const int b = blockIdx.x / nbq, blk = blockIdx.x % nbq;
  if (b >= B) return;
  const int m = blk % M, ck = blk / M;
  const long brow = __builtin_amdgcn_readfirstlane(vidx != nullptr ? vidx[b] : b) * v_brows;
  const long left = value_bytes - brow * ldv * 4;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(value + brow * ldv), 0, (unsigned)(left < 0xF0000000L ? left : 0xF0000000L), 0x00020000);
  __syncthreads();
  const int grp = (threadIdx.x >> 3) & 7, j = threadIdx.x & 7;
  const bool chan = j * 4 < D;
  const unsigned lane_off = chan ? (unsigned)((m * D + j * 4) * 4) : MSDA_OOB;
  const unsigned lane_lds = chan ? (unsigned)(j * 16) : 0u;
  const unsigned zero_row = (unsigned)(stage_px * D * 4);
  const int q_end = min(Q, (ck + 1) * chunk);
  for (int q0 = ck * chunk; q0 < q_end; q0 += NT / 8) {
    const int q = q0 + (threadIdx.x >> 3);
    const bool live = q < q_end;
    const int qq = live ? q : q_end - 1;
    const long t = (long)b * Q + qq;
    // softmax over the FP logits of (query, head): lane j holds samples 2j, 2j+1 (sample = frame * P + point)
    const f32x2 l2 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(logits + t * ldl + m * FP + 2 * j));
    float mx = fmaxf(l2[0], l2[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    float sm = expf(l2[0] - mx) + expf(l2[1] - mx);
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float inv = 1.0f / sm;
    const float* rp = ref + (long)b * ref_bstride + (long)qq * 4;
    const float rx = rp[0], ry = rp[1], bw = rp[2], bh = rp[3];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < F; ++f) {
      __syncthreads();                           // the previous frame's staged rows are no longer read
      {                                          // stage the head's slice of levels [LS, G) of frame f: stage_px consecutive tokens
        const int s0 = sS[LS * F + f];
        for (int idx = threadIdx.x; idx < (stage_px + 1) * 8; idx += NT) {
          const int px = idx >> 3, c = idx & 7;
          if (c * 4 >= D) continue;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (px < stage_px) v = *reinterpret_cast<const f32x4*>(value + (brow + s0 + px) * ldv + m * D + c * 4);
          *reinterpret_cast<f32x4*>(stage + (long)px * D + c * 4) = v;
        }
      }
      __syncthreads();
      // this lane's two look-ups of the phase: i = 2j + k -> level g = i / P, point p = i % P of frame f: sample s = f * P + p, whose
      // offsets and logit are the pair (2 * (s / 2), +1) -- one 16-B and one 8-B load, L1-resident after the first frame
      const int s_pair = f * P + ((2 * j) % P);   // even: samples s_pair (k = 0) and s_pair + 1 (k = 1)
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(offs + t * ldo + m * (2 * FP) + 2 * s_pair);
      const f32x2 lg = *reinterpret_cast<const f32x2*>(logits + t * ldl + m * FP + s_pair);
      u32x4 offv[2];
      f32x4 wv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = (2 * j + k) / P;
        const int s = s_pair + k;
        const float gx = grid[(m * FP + s) * 2], gy = grid[(m * FP + s) * 2 + 1];
        float ox = o4[2 * k], oy = o4[2 * k + 1];
        ox = fminf(fmaxf(ox, -bw * 8.f), bw * 8.f);
        oy = fminf(fmaxf(oy, -bh * 8.f), bh * 8.f);
        ox = gx * 0.5f * bw + ox;
        oy = gy * 0.5f * bh + oy;
        const float lx = rx + ox / 8.f, ly = ry + oy / 8.f;
        const float aw = expf(lg[k] - mx) * inv;
        const int H = sH[g * F + f], W = sW[g * F + f];
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
        const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool h0 = in && h_low >= 0, h1 = in && h_low + 1 <= H - 1;
        const bool w0 = w_low >= 0, w1 = w_low + 1 <= W - 1;
        unsigned base, dW, d1, oob;
        if (g >= LS) {
          base = (unsigned)((sS[g * F + f] - sS[LS * F + f] + h_low * W + w_low) * (D * 4));
          dW = (unsigned)(W * D * 4); d1 = (unsigned)(D * 4); oob = zero_row;
        } else {
          const long prow = sS[g * F + f] + (long)h_low * W + w_low;
          base = (unsigned)(prow * ldv * 4);
          dW = (unsigned)((long)W * ldv * 4); d1 = (unsigned)(ldv * 4); oob = MSDA_OOB;
        }
        offv[k][0] = (h0 && w0) ? base : oob;
        offv[k][1] = (h0 && w1) ? base + d1 : oob;
        offv[k][2] = (h1 && w0) ? base + dW : oob;
        offv[k][3] = (h1 && w1) ? base + dW + d1 : oob;
        wv[k][0] = hh * hw * aw; wv[k][1] = hh * lw * aw; wv[k][2] = lh * hw * aw; wv[k][3] = lh * lw * aw;
      }
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
        if constexpr (LSC >= 0) {
#pragma unroll
          for (int gq = 0; gq < HS / P; ++gq) {     // one level at a time: descriptors, 16 loads in flight, 16 multiply-adds in order
            const int g = (half * HS) / P + gq;     // compile-time
            u32x4 o[P];
            f32x4 w[P], v[P][4];
#pragma unroll
            for (int s = 0; s < P; ++s) {
              o[s] = *reinterpret_cast<const u32x4*>(soff + (grp * (HS + 1) + gq * P + s) * 4);
              w[s] = *reinterpret_cast<const f32x4*>(swgt + (grp * (HS + 1) + gq * P + s) * 4);
            }
#pragma unroll
            for (int s = 0; s < P; ++s)
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (g >= LS) v[s][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stage) + o[s][c] + lane_lds);
                else v[s][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, chan ? o[s][c] + lane_off : MSDA_OOB, 0, MSDA_GATHER_AUX));
              }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < P; ++s)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc += v[s][c] * w[s][c];
            asm volatile("" : "+v"(acc));           // (see msda_fused_v3_kernel: keeps LLVM from sinking the chain to the store)
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int s = 0; s < HS; ++s) {
            const int g = (half * HS + s) / P;      // compile-time
            const u32x4 o = *reinterpret_cast<const u32x4*>(soff + (grp * (HS + 1) + s) * 4);
            const f32x4 w = *reinterpret_cast<const f32x4*>(swgt + (grp * (HS + 1) + s) * 4);
            if (g >= LS) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stage) + o[c] + lane_lds);
                acc += v * w[c];
              }
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const unsigned a = chan ? o[c] + lane_off : MSDA_OOB;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a, 0, MSDA_GATHER_AUX));
                acc += v * w[c];
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (chan && live) __builtin_nontemporal_store(acc * scale, reinterpret_cast<f32x4*>(out + t * ldout + m * D + j * 4));
  }
}

static int g_msda_xcd_order = 1;   // tools/ A/B: 0 = plain block order in the fused kernel
extern "C" int mdqe_debug_msda_xcd_order(int v) { g_msda_xcd_order = v; return MDQE_OK; }
static int g_msda_variant = -1;    // tools/ A/B: block-to-query map (0, 1, 2) + 4 * (waves-per-SIMD hint 8 instead of none); -1 = by shape
extern "C" int mdqe_debug_msda_variant(int v) { g_msda_variant = v; return MDQE_OK; }
static int g_msda_stage_kb = 150;  // tools/ A/B: LDS budget (KB) of the levels staged by msda_fused_v3_kernel (how many coarse levels a block takes)
static int g_msda_dec_stage_kb = 72;   // the decoder's box-level launch: cap of that budget (two blocks per CU)
extern "C" int mdqe_debug_msda_stage_kb(int v) { g_msda_stage_kb = v > 0 ? v : 150; g_msda_dec_stage_kb = v > 0 ? v : 72; return MDQE_OK; }
extern "C" int mdqe_debug_msda_dec_stage_kb(int v) { g_msda_dec_stage_kb = v > 0 ? v : 72; return MDQE_OK; }   // the decoder's box-level launch only
static int g_msda_tp_staged = 1;   // tools/ A/B: 0 = the decoder's temporal launch stays on v2
extern "C" int mdqe_debug_msda_tp_staged(int v) { g_msda_tp_staged = v; return MDQE_OK; }
static int g_msda_patch = -1;      // tools/ A/B: 1 = the encoder's launch walks 8 x 16 query patches instead of token runs; -1 = MDQE_MSDA_PATCH (default 0)
extern "C" int mdqe_debug_msda_patch(int v) { g_msda_patch = v ? 1 : 0; return MDQE_OK; }
static int msda_patch_on() {
  if (g_msda_patch < 0) { const char* e = getenv("MDQE_MSDA_PATCH"); g_msda_patch = (e != nullptr && e[0] == '1') ? 1 : 0; }
  return g_msda_patch;
}
static int g_msda_dec_wpe8 = 0;    // tools/ A/B (round 6): 1 = the decoder's 832-thread launches in the 8-waves-per-SIMD build (<= 64 VGPRs: two blocks per CU by registers)
extern "C" int mdqe_debug_msda_dec_wpe8(int v) { g_msda_dec_wpe8 = v; return MDQE_OK; }     // (2: the temporal launch at 7 waves per SIMD, 72 VGPRs)
static int g_msda_dec_staged = 1;  // tools/ A/B: 0 = the decoder's box-level launch stays on v2 (the encoder keeps its default)
extern "C" int mdqe_debug_msda_dec_staged(int v) { g_msda_dec_staged = v; return MDQE_OK; }

extern "C" int mdqe_msda_fused_f32(const float* value, long ldv, long v_brows, const int* vidx, const float* offs, long ldo,
                                   const float* logits, long ldl, const float* ref, long ref_bstride, int ref_dim,
                                   int mode, const float* grid, const int* lvH_host, const int* lvW_host,
                                   const int* lvStart_host, int B, int M, int D, int G, int L, int Q, int P, float scale,
                                   float* out, long ldout, long value_rows, void* stream) {
  MDQE_REQUIRE(B >= 0 && M > 0 && D > 0 && D % 4 == 0 && G > 0 && L > 0 && P > 0 && Q >= 0 && G * L <= 16);
  MDQE_REQUIRE(ldv % 4 == 0 && ldo % 4 == 0 && ldl % 4 == 0 && ldout % 4 == 0 && (ref_dim == 2 || ref_dim == 4));
  MDQE_REQUIRE(mode == 0 || (mode == 1 && ref_dim == 4));
  if ((long)B * Q == 0) return MDQE_OK;
  MDQE_CHECK_PTR(value); MDQE_CHECK_PTR(offs); MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(ref); MDQE_CHECK_PTR(out);
  MDQE_CHECK_PTR(lvH_host); MDQE_CHECK_PTR(lvW_host); MDQE_CHECK_PTR(lvStart_host);
  if (mode == 1) MDQE_CHECK_PTR(grid);
  MDQE_REQUIRE((((uintptr_t)value | (uintptr_t)offs | (uintptr_t)logits | (uintptr_t)out) & 15) == 0);
  MsdaLevels lv;
  for (int i = 0; i < 16; ++i) { lv.H[i] = 1; lv.W[i] = 1; lv.start[i] = 0; }
  for (int i = 0; i < G * L; ++i) { lv.H[i] = lvH_host[i]; lv.W[i] = lvW_host[i]; lv.start[i] = lvStart_host[i]; }
  const long total = (long)B * Q * M * (D / 4);
  long nb = (total + 255) / 256;
  if (nb > 256L * 64) nb = 256L * 64;
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  // v2 (cooperative sample set-up, buffer loads): D == 32, 16 samples, value buffer addressable with 32-bit offsets
  const long vbytes = value_rows > 0 ? ((value_rows - 1) * ldv + (long)M * D) * 4 : 0;
  // what the 32-bit gather offsets must span: ONE element's levels (the kernels base their buffer resource at the element's block)
  long span_rows = 0;
  for (int i = 0; i < G * L; ++i) { const long e = (long)lv.start[i] + (long)lv.H[i] * lv.W[i]; if (e > span_rows) span_rows = e; }
  const long span = (span_rows * ldv + (long)M * D) * 4;
  if ((D == 32 || D == 24) && L * P == 16 && vbytes > 0 && span < 0xF0000000L) {
    // block-to-query map: the patch / row forms need the queries to be the level tokens in raster order (encoder: mode 0, G == 1,
    // Q == sum H*W); everything else keeps the plain order
    long ntok = 0;
    for (int l = 0; l < L; ++l) ntok += (long)lv.H[l] * lv.W[l];
    // default: the encoder (mode 0: the queries are the level tokens in raster order) takes map 1 -- 608 vs 635 us on the
    // 40-frame launch with unstructured offsets (tools/pmc_msda.py variants), no difference end to end; the decoder keeps map 0
    int var = g_msda_variant >= 0 ? g_msda_variant : (mode == 0 ? (1 | 8) : (MSDA_DEFAULT_VARIANT | 8));
    const bool enc_form = mode == 0 && ntok == Q && vidx == nullptr && ref_dim == 2;      // the queries are the level tokens
    // (24-wide heads -- Swin-L's 96-B rows -- gain nothing from the staged form in the decoder: 79-81 us staged against 73 us on the gather
    // form for 34 clips, tools/msda_dec_640p.py; a forced budget, tools/ only, still takes the staged kernel)
    const bool dec_form = mode == 1 && ref_dim == 4 && g_msda_dec_staged && (D == 32 || g_msda_stage_kb != 150);                                      // box-level decoder launch: Q queries per (clip, frame)
    if ((var & 8) && G == 1 && L == 4 && P == 4 && (enc_form || dec_form)) {
      // v3: the coarsest levels that fit beside the descriptors (2.3 KB per wave) in the 160-KB LDS are staged per (frame, head)
      int nt = (g_msda_variant >= 0 && (g_msda_variant & 128)) ? 512 : 1024;
      int chunk = 0;
      if (dec_form) {                            // few queries per (element, head): even runs of at most 128, threads to match (196 -> 2 x 98 on 13 waves)
        const int runs = (Q + 127) / 128;
        chunk = (Q + runs - 1) / runs;
        nt = chunk <= 64 ? 512 : chunk <= 104 ? 832 : 1024;
      }
      const long desc = (nt / 64) * 8L * (8 + 1) * 4 * 2 * 4;
      int LS = L;
      long px = 0;
      // the decoder's launch has few blocks per (frame, head): two blocks per CU (<= 72 KB each) beat one that stages more (640p: level 3
      // alone 72.5 us, levels 2 + 3 = 115 KB 79.8 us, tools/msda_dec_640p.py); the encoder's long query runs take what fits
      const long budget = (long)(dec_form && g_msda_stage_kb > g_msda_dec_stage_kb ? g_msda_dec_stage_kb : g_msda_stage_kb) * 1024;
      while (LS > 1 && ((px + (long)lv.H[LS - 1] * lv.W[LS - 1] + 1) * D * 4 + desc) <= budget) { --LS; px += (long)lv.H[LS] * lv.W[LS]; }
      if (LS < L) {
        const size_t smem = (size_t)((px + 1) * D * 4 + desc);
        // queries per block, by what a block stages (tools/pmc_msda.py variants, us per launch): 38 KB (360p, 40 frames) 128: 380,
        // 256: 420, 640: 436, 2048: 513; 50 KB (Swin-L 480p, D = 24) 128: 240, 256: 227, 512: 255; 115 KB (640p, one block per CU)
        // 256: 798, 512: 740, 1024: 755
        const long staged = (px + 1) * D * 4;
        if (chunk == 0) chunk = staged < 45 * 1024 ? 128 : staged < 80 * 1024 ? 256 : 512;
        if (g_msda_variant >= 0 && ((g_msda_variant >> 4) & 7)) chunk = 32 << (((g_msda_variant >> 4) & 7) - 1);       // tools/ sweep: 32 .. 2048
        // round 6 experiment (off by default, measured slower): the encoder's 1024-thread launch walks 8 x 16 query patches instead of token runs
        const bool patch = enc_form && nt == 1024 && chunk % 128 == 0 && !(g_msda_variant >= 0 && (g_msda_variant & 2048)) && msda_patch_on();
        long npatch = 0;
        for (int l = 0; l < L; ++l) npatch += (long)((lv.W[l] + 15) / 16) * ((lv.H[l] + 7) / 8);
        const int nchunk = patch ? (int)((npatch + chunk / 128 - 1) / (chunk / 128)) : (Q + chunk - 1) / chunk;
        const long nb3 = (long)B * M * nchunk;
        bool lds_ok = true;
        auto launch3 = [&](auto kern) {
          // per kernel FUNCTION (the six instantiations share this lambda's one operator()): keyed by pointer in mdqe_allow_lds
          if (mdqe_allow_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 256) != hipSuccess) { lds_ok = false; return; }   // nothing launched
          hipLaunchKernelGGL(kern, dim3((unsigned)nb3), dim3(nt), smem, st, value, vbytes, ldv, v_brows, vidx, offs, ldo, logits, ldl,
                             ref, ref_bstride, ref_dim, mode, grid, lv, B, M, Q, LS, (int)px, chunk, nchunk, scale, out, ldout, g_msda_xcd_order,
                             patch ? 1 : 0);
        };
        // two blocks of 1024 threads per CU need the 8-waves-per-SIMD build (variant bit 256 of the tools/ sweep turns it off)
        const bool wpe8 = !(g_msda_variant >= 0 && (g_msda_variant & 256)) && 2 * smem + 1024 <= 160 * 1024;
        // Which build (round 4; all forms give the same bits).  Alone (tools/msda_r04_ab.py, profiles/r04_msda_compile_time_level_ab.txt):
        // where two 1024-thread blocks fit a CU (the encoder at 360p, 2 x 75 KB) the runtime form in the 8-waves-per-SIMD build takes 435 us
        // against 454 (natural allocation, one block per CU) and 458 (compile-time level, a level's 16 loads in flight); elsewhere the
        // compile-time form is 0-5 % faster alone (decoder box level 103 against 108 us, Swin-L encoder 223 against 228, 640p equal).
        // INSIDE the pipeline (tools/msda_form_in_pipeline_ab.py, profiles/r04_msda_form_in_pipeline_ab.txt) its 126-VGPR blocks wait longer
        // for room beside the frame stream's GEMM blocks (decoder box level 614-632 us against 437-455) and frames/s are equal within
        // noise, single-GPU and sharded -- so the shipped choice is the runtime form everywhere, in the 8-waves build where that buys a
        // second block per CU.  variant bits of the tools/ sweep: 256 = no 8-waves build, 1024 = the compile-time form where LS is 2 or 3
        const bool force_ct = g_msda_variant >= 0 && (g_msda_variant & 1024);
        const bool two_blocks = nt == 1024 && wpe8;
        const int lsc = (force_ct && (LS == 2 || LS == 3)) ? LS : -1;
        auto pick = [&](auto lsc_) {
          constexpr int C = decltype(lsc_)::value;
          if constexpr (C < 0) {
            if (two_blocks) { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 1024, 8>); else launch3(msda_fused_v3_kernel<4, 4, 24, 1024, 8>); }
            else if (nt == 1024) { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 1024>); else launch3(msda_fused_v3_kernel<4, 4, 24, 1024>); }
            else if (nt == 832 && g_msda_dec_wpe8 && D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 832, 8>);
            else if (nt == 832) { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 832>); else launch3(msda_fused_v3_kernel<4, 4, 24, 832>); }
            else { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 512>); else launch3(msda_fused_v3_kernel<4, 4, 24, 512>); }
          } else {
            if (nt == 1024) { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 1024, 0, C>); else launch3(msda_fused_v3_kernel<4, 4, 24, 1024, 0, C>); }
            else if (nt == 832) { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 832, 0, C>); else launch3(msda_fused_v3_kernel<4, 4, 24, 832, 0, C>); }
            else { if (D == 32) launch3(msda_fused_v3_kernel<4, 4, 32, 512, 0, C>); else launch3(msda_fused_v3_kernel<4, 4, 24, 512, 0, C>); }
          }
        };
        if (lsc == 2) pick(std::integral_constant<int, 2>{});
        else if (lsc == 3) pick(std::integral_constant<int, 3>{});
        else pick(std::integral_constant<int, -1>{});
        if (!lds_ok) { (void)hipGetLastError(); return MDQE_ELAUNCH; }     // `out` was never written: not MDQE_OK (as msda.hip does)
        return mdqe_launch_status();
      }
    }
    if ((var & 8) && g_msda_tp_staged && mode == 1 && ref_dim == 4 && G == 4 && L == 4 && P == 4 && vidx != nullptr && Q <= 4096) {
      // temporal form: G levels of each of L frames; the table is [level g][frame f].  Needs every frame to carry the same pyramid,
      // contiguous within the frame (level g + 1 right behind level g), so that levels [LS, G) of a frame are one run of tokens
      bool ok = true;
      for (int g = 0; g < G && ok; ++g)
        for (int f = 0; f < L && ok; ++f) {
          ok = lv.H[g * L + f] == lv.H[g * L] && lv.W[g * L + f] == lv.W[g * L];
          if (g + 1 < G) ok = ok && lv.start[(g + 1) * L + f] == lv.start[g * L + f] + lv.H[g * L] * lv.W[g * L];
        }
      const int runs = (Q + 127) / 128;
      const int chunk = (Q + runs - 1) / runs;
      const int nt = chunk <= 64 ? 512 : chunk <= 104 ? 832 : 1024;
      const long desc = (nt / 64) * 8L * (8 + 1) * 4 * 2 * 4;
      int LS = G;
      long px = 0;
      while (ok && LS > 1 && ((px + (long)lv.H[(LS - 1) * L] * lv.W[(LS - 1) * L] + 1) * D * 4 + desc) <= 72L * 1024) { --LS; px += (long)lv.H[LS * L] * lv.W[LS * L]; }
      if (ok && LS < G) {                        // (<= 72 KB: two blocks per CU, one stages while the other gathers)
        const size_t smem = (size_t)((px + 1) * D * 4 + desc);
        const long nbt = (long)B * M * runs;
        bool lds_ok = true;
        auto launch_tp = [&](auto kern) {
          if (mdqe_allow_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 256) != hipSuccess) { lds_ok = false; return; }
          hipLaunchKernelGGL(kern, dim3((unsigned)nbt), dim3(nt), smem, st, value, vbytes, ldv, v_brows, vidx, offs, ldo, logits, ldl,
                             ref, ref_bstride, grid, lv, B, M, Q, LS, (int)px, chunk, runs, scale, out, ldout);
        };
        const int lsc = (g_msda_variant >= 0 && (g_msda_variant & 1024) && (LS == 2 || LS == 3)) ? LS : -1;     // (as the box-level launch: tools/ only)
        auto pick = [&](auto lsc_) {
          constexpr int C = decltype(lsc_)::value;
          if (nt == 1024) { if (D == 32) launch_tp(msda_fused_tp_kernel<4, 4, 4, 32, 1024, C>); else launch_tp(msda_fused_tp_kernel<4, 4, 4, 24, 1024, C>); }
          else if (nt == 832 && g_msda_dec_wpe8 == 1 && D == 32 && C < 0) launch_tp(msda_fused_tp_kernel<4, 4, 4, 32, 832, -1, 8>);
          else if (nt == 832 && g_msda_dec_wpe8 == 2 && D == 32 && C < 0) launch_tp(msda_fused_tp_kernel<4, 4, 4, 32, 832, -1, 7>);
          else if (nt == 832) { if (D == 32) launch_tp(msda_fused_tp_kernel<4, 4, 4, 32, 832, C>); else launch_tp(msda_fused_tp_kernel<4, 4, 4, 24, 832, C>); }
          else { if (D == 32) launch_tp(msda_fused_tp_kernel<4, 4, 4, 32, 512, C>); else launch_tp(msda_fused_tp_kernel<4, 4, 4, 24, 512, C>); }
        };
        if (lsc == 2) pick(std::integral_constant<int, 2>{});
        else if (lsc == 3) pick(std::integral_constant<int, 3>{});
        else pick(std::integral_constant<int, -1>{});
        if (!lds_ok) { (void)hipGetLastError(); return MDQE_ELAUNCH; }
        return mdqe_launch_status();
      }
    }
    int map = var & 3;
    if (map == 2 && !(mode == 0 && G == 1 && ntok == Q)) map = (mode == 0) ? 1 : 0;
    long nbq;
    if (map == 0) nbq = ((long)Q * M * 8 + 255) / 256;
    else if (map == 1) nbq = (long)((Q + 31) / 32) * M;
    else { nbq = 0; for (int l = 0; l < L; ++l) nbq += (long)((lv.W[l] + 7) / 8) * ((lv.H[l] + 3) / 4); nbq *= M; }
    const long nb2 = (long)B * nbq;                                            // exact grid: blocks per batch element x B
#define LAUNCH2D(LL, PP, MP, WP, DDD) hipLaunchKernelGGL((msda_fused_v2_kernel<LL, PP, MP, WP, DDD>), dim3((unsigned)nb2), dim3(256), 0, st, value, \
      vbytes, ldv, v_brows, vidx, offs, ldo, logits, ldl, ref, ref_bstride, ref_dim, mode, grid, lv, B, M, G, Q, scale, out, ldout, total, \
      g_msda_xcd_order)
#define LAUNCH2(LL, PP, MP, WP) do { if (D == 32) LAUNCH2D(LL, PP, MP, WP, 32); else LAUNCH2D(LL, PP, MP, WP, 24); } while (0)
#define LAUNCH2M(LL, PP) do { const bool w8 = (var & 4) != 0; \
      if (map == 0) { if (w8) LAUNCH2(LL, PP, 0, 8); else LAUNCH2(LL, PP, 0, 1); } \
      else if (map == 1) { if (w8) LAUNCH2(LL, PP, 1, 8); else LAUNCH2(LL, PP, 1, 1); } \
      else { if (w8) LAUNCH2(LL, PP, 2, 8); else LAUNCH2(LL, PP, 2, 1); } } while (0)
    if (L == 4 && P == 4) { LAUNCH2M(4, 4); return mdqe_launch_status(); }
    if (L == 2 && P == 8) { LAUNCH2M(2, 8); return mdqe_launch_status(); }
#undef LAUNCH2M
#undef LAUNCH2
  }
#define LAUNCH(LL, PP) hipLaunchKernelGGL((msda_fused_kernel<LL, PP>), dim3((unsigned)nb), dim3(256), 0, st, value, ldv, v_brows, vidx, \
    offs, ldo, logits, ldl, ref, ref_bstride, ref_dim, mode, grid, lv, B, M, D, G, Q, scale, out, ldout, total)
  if (L == 4 && P == 4) LAUNCH(4, 4);
  else if (L == 3 && P == 4) LAUNCH(3, 4);
  else if (L == 2 && P == 4) LAUNCH(2, 4);
  else if (L == 1 && P == 4) LAUNCH(1, 4);
  else return MDQE_EINVAL;
#undef LAUNCH
  return mdqe_launch_status();
}
