// Shared parameter block of the GEMM kernels (gemm.hip, gemm_f16x3.hip).
#pragma once
struct GemmParams {
  const float* A; const float* W; float* C;
  int M, N, K;
  long lda, ldc;
  // conv mode (KH > 0): A is NHWC [NI, H, W_, Cin]
  int conv; int H, Wd, Cin, OH, OW, KH, KW, stride, pad;
  // epilogue
  const float* bias;        // [N] or null
  const float* residual;    // [res_rows, ldr] or null, added after activation
  long ldr; int res_mod;    // row index = res_mod > 0 ? m % res_mod : m
  int res_first;            // 1: add residual BEFORE the activation (ResNet bottleneck), 0: after
  long img_stride;          // conv mode: floats between consecutive images of X
  const unsigned char* rowmask; int mask_cols;   // C[m, n < mask_cols] = 0 where rowmask[m] != 0
  int act; int act_cols;    // activation on columns < act_cols (act_cols <= 0: all)
  unsigned a_bytes, w_bytes;
  int vec_ok;               // C/bias/residual are 16-B aligned with ld % 4 == 0: float4 epilogue
  int ksplit, kchunk;       // split-K: blockIdx.y = split, K range [y*kchunk, (y+1)*kchunk); raw partials -> ws
  float* ws;                // [ksplit][M][N] partial sums (deterministic two-pass reduction)
  unsigned long long* stamps;   // debug: per-block s_memtime stamps (mdqe_debug_gemm_stamps), null in production
  const void* Wh; const void* Wl;   // pre-split f16 planes of W ([N][K] each; gemm_f16x3w.hip) or null
  const float* ln_g; const float* ln_b; float ln_eps;   // tile 6: LayerNorm over the 256 columns in the epilogue
  // ... and optionally a SECOND LayerNorm of that result into C2 (the decoder's `decoder_norm(norm3(..))`, transformer_dec.py:492-495):
  // C2 = LN(C) * ln2_g + ln2_b, statistics by the same reduction tree (= layernorm_kernel on C, bit for bit)
  const float* ln2_g; const float* ln2_b; float* C2; long ldc2;
  // cat mode (A2 != null, K-step-16 kernel, plain tiles): k < K1 comes from A, k >= K1 from row (img, oh*stride, ow*stride) of the NHWC
  // tensor A2 [*, H, Wd, lda2] -- output row m = (img, oh, ow) on the OH x OW grid (H, Wd, OH, OW, stride as in conv mode)
  const float* A2; long lda2; int K1; unsigned a2_bytes;
  // rank-4 side term (K-step-16 kernel, plain tiles; everywhere else a second pass adds it): C[m, n] += side[m, 0:4] . side_w[n, 0:4]
  // for n < side_cols, before the activation.  The decoder's `(x + pos) W^T` with pos = Linear(2 -> C)(box centre) is
  // `x W^T + box (W P)^T`: the position embedding is never materialised (mdqe_gemm_nt_side_f32)
  const float* side; const float* side_w; int side_cols;
  // Swin window gather (K-step-16 kernel, plain tiles; swin_ws > 0): A is an NHWC map [B, swin_H, swin_W, lda] and output row m is the
  // m-th row of its window partition after zero padding to multiples of swin_ws and the cyclic shift (swin_transformer_v2.py:236-262):
  // the row's source pixel is computed once per block, a padded position reads zeros -- the partitioned copy is never materialised
  int swin_ws, swin_shift, swin_H, swin_W;
  int fast_epi;             // K-step-16 kernel: 1 (default) = interior tiles of plain products take the few-instruction epilogue
  int stagger;              // K-step-16 kernel: the first resident round of blocks starts (slot on the CU) x stagger 10-ns ticks late (0: off)
};

#define OOB_OFF 0xFFFFFFF0u

// defined in gemm_f16x3.hip
int mdqe_launch_gemm_f16x3(const GemmParams& p, int tile, hipStream_t st);
// defined in gemm_k16.hip (tile 1/2/3 as gemm.hip; the caller launches the split-K reduce pass)
int mdqe_launch_gemm_k16(const GemmParams& p, int tile, hipStream_t st);
// defined in gemm_f16x3w.hip (bn = 256 or 128)
int mdqe_launch_gemm_f16x3w(const GemmParams& p, int bn, hipStream_t st);
int mdqe_launch_gemm_f16w(const GemmParams& p, int bn, hipStream_t st);    // mode 2: one f16 pass, p.Wh = the round-to-nearest plane
