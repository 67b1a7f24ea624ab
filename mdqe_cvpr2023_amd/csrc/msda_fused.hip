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

extern "C" int mdqe_msda_fused_f32(const float* value, long ldv, long v_brows, const int* vidx, const float* offs, long ldo,
                                   const float* logits, long ldl, const float* ref, long ref_bstride, int ref_dim,
                                   int mode, const float* grid, const int* lvH_host, const int* lvW_host,
                                   const int* lvStart_host, int B, int M, int D, int G, int L, int Q, int P, float scale,
                                   float* out, long ldout, void* stream) {
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
