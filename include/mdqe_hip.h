/* mdqe_hip.h -- C ABI of libmdqe_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the MDQE eval-only hot path.  Plain pointers and sizes
 * only; every pointer is a DEVICE pointer unless marked host; `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  All entry points are asynchronous on `stream`, never allocate,
 * never synchronise, borrow their inputs and fully overwrite their outputs (graph-capturable).
 * Return value: 0 on success, otherwise one of the MDQE_E* codes; kernel-launch failures are
 * reported through the return value (the reference only printf-s them,
 * mdqe/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:948-952).
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   mdqe_msda_forward_f32      <- ms_deform_attn_forward  (mdqe/models/ops/src/vision.cpp:13-16,
 *                                 src/ms_deform_attn.h:20-39, src/cuda/ms_deform_attn_cuda.cu:20-80),
 *                                 kernel ms_deformable_im2col_gpu_kernel (ms_deform_im2col_cuda.cuh:237-299)
 *   everything else            <- ATen/cuDNN/cuBLAS kernels the reference reaches through PyTorch on
 *                                 the same path (SURVEY.md §2a "Fused ops / ATen kernels"); each
 *                                 prototype names the reference call site it serves.
 */
#ifndef MDQE_HIP_H
#define MDQE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDQE_OK 0
#define MDQE_EINVAL 1   /* bad size / unsupported shape */
#define MDQE_ELAUNCH 2  /* hipGetLastError() != hipSuccess after launch */
#define MDQE_ENULL 3    /* null pointer */
#define MDQE_ESTATE 4   /* the object was left inconsistent by an earlier failed call (tracker: a device error in the middle of an update) and refuses further use */

/* activation codes for fused epilogues */
#define MDQE_ACT_NONE 0
#define MDQE_ACT_RELU 1
#define MDQE_ACT_GELU 2 /* exact erf GELU (nn.GELU default) */
#define MDQE_ACT_SIGMOID 3
#define MDQE_ACT_TANH 4

/* ABI version of this header: bumped whenever an existing entry point changes what it reads or WRITES under an unchanged signature.
 *   6: mdqe_f16x3_split_f32 writes THREE planes (6*n bytes; versions < 6 wrote two, 4*n bytes) -- a caller that sized its buffer for an
 *      older header must not call a newer library.  mdqe_abi_version() returns the library's value; a binding compares it with the
 *      MDQE_ABI_VERSION it was written against before the first call (mdqe_cvpr2023_amd/_lib.py does, and refuses a mismatch). */
#define MDQE_ABI_VERSION 6
int mdqe_abi_version(void);
int mdqe_version(void);
const char* mdqe_strerror(int code);

/* ---- a9: multi-scale deformable attention sampling -------------------------------------------
 * out[b,q,m,:] = sum_l sum_p attn[b,q,m,l,p] * bilinear_zero_pad(value_l[b,:,m,:], loc[b,q,m,l,p])
 * value [B,S,M,D] f32; shapes [L,2] int64 (H,W); level_start [L] int64; loc [B,Q,M,L,P,2] (x,y in
 * [0,1] image-normalised, pixel = loc*size - 0.5); attn [B,Q,M,L,P]; out [B,Q,M*D].
 * Same argument meaning as ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:20-80); im2col_step
 * is a batching artefact of the reference launcher and has no equivalent here. */
int mdqe_msda_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                          const float* loc, const float* attn,
                          int B, int S, int M, int D, int L, int Q, int P,
                          float* out, void* stream);

/* Backward of the same op (ms_deform_attn_cuda_backward, src/cuda/ms_deform_attn_cuda.cu:83-153; kernels
 * ms_deform_im2col_cuda.cuh:87-234,301-920): grad_out [B,Q,M*D] -> grad_value [B,S,M,D] (zeroed here, then float
 * atomics), grad_loc [B,Q,M,L,P,2], grad_attn [B,Q,M,L,P].  Not on the eval path; completes the extension's two exports. */
int mdqe_msda_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                           const float* attn, const float* grad_out, int B, int S, int M, int D, int L, int Q, int P,
                           float* grad_value, float* grad_loc, float* grad_attn, void* stream);

/* The same two exports in float64 -- the reference dispatches float and double (AT_DISPATCH_FLOATING_TYPES,
 * src/cuda/ms_deform_attn_cuda.cu:64 forward, :134 backward) and its own test script checks the double forward against the
 * PyTorch core and the backward by gradcheck in double (mdqe/models/ops/test.py:32-44, :63-86).  csrc/msda_f64.hip. */
int mdqe_msda_forward_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                          const double* attn, int B, int S, int M, int D, int L, int Q, int P, double* out, void* stream);
int mdqe_msda_backward_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                           const double* attn, const double* grad_out, int B, int S, int M, int D, int L, int Q, int P,
                           double* grad_value, double* grad_loc, double* grad_attn, void* stream);

/* Grouped form used for temporal_clip_forward (mdqe/models/ops/modules/ms_deform_attn.py:219-236):
 * the reference issues G (= #spatial levels) separate native calls that share loc/attn and averages
 * them.  Here: shapes/level_start are [G*L] tables into ONE value buffer, out = scale * sum_g (...). */
int mdqe_msda_forward_grouped_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                  const float* loc, const float* attn,
                                  int B, int S, int M, int D, int G, int L, int Q, int P,
                                  float scale, float* out, void* stream);

/* Fused module core (ops/modules/ms_deform_attn.py:141-170 / 198-235): consumes the projection GEMM's raw
 * outputs.  value rows are ldv floats apart, batch b starts at row (vidx ? vidx[b] : b)*v_brows (vidx: device
 * int32[B], lets a batch of overlapping clips address one per-frame value cache); offs row t=(b*Q+q) holds
 * M*L*P*2 floats, logits row t holds M*L*P.  mode 0 (encoder): loc = ref_xy + off/8.  mode 1 (decoder):
 * loc = ref_xy + (grid*0.5*wh + clamp(off, +-8*wh))/8 with ref = (cx,cy,w,h) and grid [M,L,P,2] (device).
 * ref row = b*ref_bstride + q*ref_dim (ref_bstride 0 broadcasts one table over the batch).
 * Level tables (HOST int[G*L]): H, W, start row.  out row t, ldout floats apart; out = scale * sum_g(...).
 * value_rows = number of rows of the value buffer (> 0 enables the bounds-checked buffer-load kernel; 0 = unknown). */
int mdqe_msda_fused_f32(const float* value, long ldv, long v_brows, const int* vidx, const float* offs, long ldo,
                        const float* logits, long ldl, const float* ref, long ref_bstride, int ref_dim,
                        int mode, const float* grid, const int* lvH_host, const int* lvW_host,
                        const int* lvStart_host, int B, int M, int D, int G, int L, int Q, int P, float scale,
                        float* out, long ldout, long value_rows, void* stream);

/* ---- fp32 NT GEMM with fused epilogue (v_mfma_f32_32x32x2_f32, exact fp32) ----------------------
 * C[m,n] = mask( act(sum_k A[m*lda+k] * W[n*K+k] + bias[n]) + residual[(res_mod? m%res_mod : m)*ldr + n] )
 * (res_first != 0: the residual is added before the activation instead, as in a ResNet bottleneck)
 * Serves nn.Linear / 1x1 conv call sites of the path: value_proj/output_proj/sampling_offsets/
 * attention_weights (ops/modules/ms_deform_attn.py:136-171), FFNs (transformer_enc.py:106,
 * transformer_dec.py:356,406), MLP heads (models/misc.py:6-18), MHA projections
 * (transformer_dec.py:350,399), input_proj 1x1 (models/mdqe.py:34-37), dynamic mask product
 * (mdqe/mdqe.py:384).  K % 4 == 0, lda % 4 == 0, A/W 16-B aligned.  act applies to columns
 * < act_cols (<=0: all); rowmask (u8, 1 = zero the row) applies to columns < mask_cols
 * (the masked_fill of ms_deform_attn.py:137-138).  tile: 0 auto, 1 128x128, 2 128x64, 3 64x64
 * (4 64x256 and 5 128x256 -- whole 256-wide rows per block -- are measured alternatives, not picked by auto).
 * ksplit > 1: K is cut into ksplit chunks over blockIdx.y, partial tiles go to splitk_ws (>= ksplit*M*N floats)
 * and a second pass sums them in fixed order and applies the epilogue (deterministic; for skinny large-K products). */
int mdqe_gemm_nt_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc,
                     int M, int N, int K, int act, int act_cols, const float* residual, long ldr, int res_mod,
                     int res_first, const unsigned char* rowmask, int mask_cols, int tile, int ksplit,
                     float* splitk_ws, const void* w_split, void* stream);

/* The last 1x1 conv of a ResNet bottleneck and its projection shortcut as ONE product (detectron2 BottleneckBlock.forward:
 * out = relu(conv3(y) + shortcut(x)), both with folded FrozenBN; the reference reaches them through cuDNN):
 * C = act([A1 | A2'] W^T + bias) with W = [W3 | Ws] along K and bias = b3 + bs, so the shortcut's output is never written and read
 * back.  A1 [M = NI*OH*OW, K1] rows of pitch lda1; A2 the block's input, NHWC [NI, H2, W2, lda2 >= K2], read at pixel
 * (oh*stride, ow*stride) for output row (img, oh, ow).  K1, K2 multiples of 16.  Exact fp32 MFMA. */
int mdqe_gemm_nt_cat2_f32(const float* A1, long lda1, int K1, const float* A2, long lda2, int K2, int NI, int OH, int OW,
                          int H2, int W2, int stride, const float* W, const float* bias, float* C, long ldc, int N, int act,
                          void* stream);

/* Linear + residual + LayerNorm in one kernel, for the encoder / decoder pattern  x = norm(x + dropout(linear(..)))
 * (transformer_enc.py:100-110, transformer_dec.py:352-358,404-409; nn.LayerNorm over d_model = 256):
 * C = LN(A W^T + bias + residual) * gamma + beta, N must be 256; C may alias the residual.  Exact fp32 MFMA. */
int mdqe_gemm_ln_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K,
                     const float* residual, long ldr, const float* gamma, const float* beta, float eps, void* stream);
/* ... followed by a second LayerNorm of the result in the same epilogue: C2 = LN(C) * gamma2 + beta2 (the decoder's shared
 * `decoder_norm` behind `norm3`, transformer_dec.py:492-495); C2 equals mdqe_layernorm_f32(C) bit for bit. */
int mdqe_gemm_ln2_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K,
                      const float* residual, long ldr, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                      float* C2, long ldc2, float eps, void* stream);

/* A projection of `x + pos` whose pos is a linear function of four numbers per row -- the decoder's query position embedding
 * point2pos_proj(box centre) (mdqe/models/transformer_dec.py:42,469,480,495,503), consumed as `x + pos` by the self-attention q/k
 * projections (:348-353, :397-402) and by the sampling-offset / attention-weight projections of cross_attn / temp_attn_inst
 * (ops/modules/ms_deform_attn.py:141-147,198-203):  C = A W^T + bias;  C[:, n < side_cols] += side [M,4] . side_w[n, 0:4]
 * with side_w = W P and the bias carrying W b_P (folded once on the host).  The [M, C] position tensor is never materialised and
 * the q, k (with position) and v (without) projections of one x are ONE product.  side_cols % 4 == 0. */
int mdqe_gemm_nt_side_f32(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K,
                          const float* side, const float* side_w, int side_cols, const void* w_split, void* stream);

/* GEMM arithmetic of the 128x128 tile (process-wide): 0 = exact fp32 MFMA (default), 1 = "f16x3": operands split
 * in-kernel into f16 hi + scaled f16 lo, three f16 MFMAs with fp32 accumulation (~1e-6 relative to fp32; |x| < 32752);
 * 2 = "f16" (round 5): ONE f16 MFMA pass -- both operands rounded to nearest f16, fp32 accumulation, fp32 result -- on the
 * products whose constant weight has planes (mdqe_f16x3_split_f32) and that fill the 128-row tile, exact fp32 everywhere
 * else: the arithmetic of the reference's fp16-autocast regions on a GPU (train_net.py:207) with an fp32 instead of an fp16
 * result (operand error 2^-11 relative; |x| < 65504).  Never the default and never the headline mode. */
int mdqe_set_gemm_precision(int mode);
int mdqe_get_gemm_precision(void);   /* the mode the CALLING thread's launches use (its override if set, else the process-wide one) */
/* The calling host thread's override of the mode: 0 / 1 / 2, or -1 = none (follow the process-wide mode).  Lets one region of the model --
 * the regions the reference's harness runs under fp16 autocast (train_net.py:207) -- take the split-precision kernels
 * while another host thread's launches (the sharded schedule's tracker replay) keep theirs. */
int mdqe_set_gemm_precision_thread(int mode);

/* One-time split of a CONSTANT weight tensor W (n = N*K fp32 values, row-major [N][K]) into the two f16 planes the
 * f16x3 mode consumes + the one mode 2 consumes: planes = { hi[n], lo[n], rn[n] } (6*n bytes), hi = f16_rtz(w),
 * lo = f16_rtz((w - hi) * 2048), rn = f16_rne(w).
 * Passing the planes as `w_split` to mdqe_gemm_nt_f32 / mdqe_conv2d_nhwc_f32 (NULL = none) lets mode 1 stream them
 * straight into LDS (gemm_f16x3w.hip: 128 x 256 tiles, no in-kernel weight conversion); requires K % 32 == 0 and
 * max|w| < 32752 (f16 range, caller-checked).  Ignored in mode 0.  Weights are constants of the eval path (nn.Linear / conv
 * parameters, FrozenBatchNorm folded), so this replaces no reference call. */
int mdqe_f16x3_split_f32(const float* w, long n, void* planes, void* stream);

/* tools/ only: while buf != NULL the f16x3w GEMM kernel writes 4 s_memtime stamps per block (start, prologue done, K loop done,
 * end) to buf[block*4 ..]; NULL (default) disables. */
int mdqe_debug_gemm_stamps(void* buf);
/* tools/ only: fp32 GEMM kernel form, 0 = K-step 32 (gemm.hip), 1 = K-step 16 (gemm_k16.hip), 2 = chosen by shape (default). */
int mdqe_debug_gemm_variant(int v);
int mdqe_debug_gemm_tile_rule(int v); /* auto tile rule variants (tools/ A/B); 0 = default */
int mdqe_debug_gemm_rows_dot(int v); /* products with N <= 8 columns: 1 (default) = rows_dot_kernel, 0 = MFMA tiles */
int mdqe_debug_gemm_fast_epilogue(int v); /* K-step-16 kernel (tools/ A/B): 1 (default) = interior tiles of plain products take the few-instruction epilogue, 0 = the general one */
int mdqe_debug_gemm_stagger(int v);  /* K-step-16 kernel: start offset between the blocks of a CU's first round, 10-ns ticks; 0 = off */
int mdqe_debug_gemm_stages(int v);   /* K-step-16 kernel, 64x64 and smaller tiles: LDS stages 2 / 4; 0 = by grid size (default) */
int mdqe_debug_msda_dec_stage_kb(int kb);   /* tools/ only: LDS staging budget of the decoder's box-level deformable launch (default 72: two blocks per CU) */
int mdqe_debug_trk_fast(int on);            /* tests / tools: 1 (default; env MDQE_TRK_FAST) = a tracker update takes its counts through mdqe_trk_siou_host_f32, 0 = memset + kernel + copy + synchronize */
int mdqe_debug_trk_spin_us(int us);         /* how long the tracker polls the counts' flag before falling back to a stream synchronize (default 2000; env MDQE_TRK_SPIN_US) */
int mdqe_debug_trk_times(double* out5, int on);   /* tools: host seconds of the tracker updates since the last call -- counts launch, counts wait, decision, accumulate launch, updates -- then reset; on != 0 keeps timing */
int mdqe_debug_trk_siou_blocks(int blocks);   /* tools/ only: blocks the tracker's sign-intersection launch aims at (0 = default 512) */
int mdqe_debug_gemm_lds_pad(int bytes);   /* tools/ only: extra dynamic LDS per block of the K-step-16 GEMM launches (caps the blocks per CU) */
/* tools/ only: window attention kernel form, 1 = MFMA where it applies (default), 0 = scalar everywhere. */
int mdqe_debug_window_attn_variant(int v);
int mdqe_debug_msda_xcd_order(int v); /* fused MSDA: 1 = XCD-aware block order (default), 0 = plain */
int mdqe_debug_msda_dec_staged(int v); /* fused MSDA, decoder box-level launch: 1 (default) = LDS-staged kernel, 0 = v2 */
int mdqe_debug_msda_dec_wpe8(int v);   /* fused MSDA, the decoder's 832-thread launches: 1 = the 8-waves-per-SIMD build (tools/ A/B), 0 = natural allocation */
int mdqe_debug_msda_patch(int v);      /* fused MSDA, encoder launch: 1 = a block iteration takes an 8 x 16 patch of queries, 0 (default; MDQE_MSDA_PATCH) = a run of 128 tokens */
int mdqe_debug_msda_stage_kb(int v);   /* fused MSDA (encoder / box level): LDS budget in KB of the staged coarse levels (default 150); tools/ A/B */
int mdqe_debug_msda_tp_staged(int v);  /* fused MSDA, decoder temporal launch: 1 (default) = frame-by-frame LDS staging (msda_fused_tp_kernel), 0 = v2 */
int mdqe_debug_msda_op_staged(int v); /* native MSDA op: 1 (default) = coarse levels staged in LDS (msda_fwd_v3_kernel), 0 = v2 */
int mdqe_debug_msda_variant(int v);   /* fused MSDA (tools/ only), bit field: block-to-query map 0 / 1 / 2 of the gather form; +4: its 8-waves-per-SIMD
                                        * hint; +8: coarse levels staged in LDS; (v >> 4) & 7: queries per block 32 << (k - 1); +128: 512-thread blocks;
                                        * +256: no 8-waves-per-SIMD build of the staged encoder kernel; +1024: first staged level as a compile-time
                                        * constant (a level's 16 corner loads in flight); -1 = default (by shape) */
int mdqe_debug_mha_variant(int v);   /* same for the 196-token decoder self-attention */

/* ---- NHWC convolution as implicit GEMM on the same kernel ---------------------------------------
 * X [NI,H,W,Cin] (Cin % 32 == 0; images x_img_stride floats apart, <=0: dense), Wt [Cout,KH,KW,Cin], Y [NI*OH*OW, ldy] ; zero padding; fused bias,
 * activation and residual; ksplit > 1: deterministic split-K as in mdqe_gemm_nt_f32 (workspace >= ksplit*NI*OH*OW*Cout floats)
 * (ResNet bottlenecks -- detectron2 build_resnet_backbone, call site
 * mdqe/mdqe.py:27,33; input_proj 3x3 s2 models/mdqe.py:40-43; MaskHead 3x3 segmentation.py:42-57). */
int mdqe_conv2d_nhwc_f32(const float* X, long x_img_stride, const float* Wt, const float* bias, float* Y, long ldy,
                         int NI, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                         int act, const float* residual, long ldr, int res_first, int tile, const void* w_split,
                         int ksplit, float* splitk_ws, void* stream);

/* ---- eval-time frame resize (ResizeShortestEdgeClip -> ResizeTransform -> PIL Image.resize(BILINEAR) on uint8 frames,
 * mdqe/data/augmentation.py:364-389, dataset_mapper.py:252-258), Pillow's two-pass fixed-point resampling bit for bit.
 * in: NI images [C,H,W] u8, in_img_stride bytes apart; out [NI,C,oh,ow] u8.  Coefficient tables (DEVICE int32), built on the
 * host: per output column X taps xmin[X] .. xmin[X]+xcnt[X]-1 with 22-bit weights xk[X*kxs + t]; same for rows. */
int mdqe_resize_pil_bilinear_u8(const unsigned char* in, long in_img_stride, int NI, int C, int H, int W, int oh, int ow,
                                const int* xmin, const int* xcnt, const int* xk, int kxs, const int* ymin, const int* ycnt,
                                const int* yk, int kys, unsigned char* out, void* stream);

/* ---- LayerNorm over the last dim: y = LN(x + res) * gamma + beta (res may be NULL) -----------------
 * nn.LayerNorm call sites transformer_enc.py:103-108,136; transformer_dec.py:345-358,394-408,466,492. */
int mdqe_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                       long rows, int C, float eps, void* stream);

/* ---- GroupNorm on NHWC [NI, HW, C] (+ fused activation); workspace >= mdqe_groupnorm_workspace_bytes --
 * nn.GroupNorm call sites models/mdqe.py:36,42; segmentation.py:21-26,104-105,112. */
long mdqe_groupnorm_workspace_bytes(int NI, int G);
int mdqe_groupnorm_nhwc_f32(const float* x, long ldx, long x_img_stride, float* y, long ldy, long y_img_stride,
                            int NI, int HW, int C, int G,
                            const float* gamma, const float* beta, float eps, int act, void* workspace, void* stream);

/* ---- stem: (x-mean)/std (mdqe/mdqe.py:473-484) + zero pad to /32 (ImageList.from_tensors, mdqe.py:318)
 * + 7x7/s2/p3 im2col -> [NI*Hp/2*Wp/2, 160] (K=147 zero-padded) for the stem GEMM.
 * frames: NI CHW images (u8 or f32) h x w, frame_stride elements apart; mean/std are HOST float[3]. */
int mdqe_stem_im2col_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                         const float* mean3_host, const float* std3_host, float* out, void* stream);

/* ---- the same stem as ONE kernel: normalise + zero pad + 7x7/s2/p3 conv (3->64) + bias (folded FrozenBN) + ReLU, frames ->
 * NHWC activation [NI, Hp/2, Wp/2, 64]; no im2col buffer (mdqe/mdqe.py:473-484,318 + detectron2 BasicStem used by
 * mdqe/models/backbone.py).  wk: DEVICE float[154*64], k-major with k = kh*22 + kw*3 + c and entry kh*22+21 zero;
 * bias: DEVICE float[64]; mean/std HOST float[3]. */
int mdqe_stem_conv_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                       const float* mean3_host, const float* std3_host, const float* wk, const float* bias, float* out,
                       void* stream);

/* ---- 3x3/s2/p1 max pool NHWC (ResNet stem) */
int mdqe_maxpool3x3s2_nhwc_f32(const float* x, float* y, int NI, int H, int W, int C, void* stream);

/* ---- y = a + nearest_upsample(b)  (segmentation.py:47-55) */
int mdqe_upsample_nearest_add_nhwc_f32(const float* a, const float* b, float* y, int NI, int H, int W, int Hb, int Wb,
                                       int C, void* stream);

/* ---- depthwise 5x5 conv NHWC, weights [25][C]; up2 != 0 applies it to the virtual output of the depthwise
 * ConvTranspose2d(k=1,s=2,output_padding=1) (weights tw,tb [C]) of an input stored at (H/2)x(W/2)
 * (segmentation.py:28-31,59,92-98). H,W are output sizes. */
int mdqe_dwconv5x5_nhwc_f32(const float* x, const float* wt, const float* bias, float* y, int NI, int H, int W, int C,
                            int up2, const float* tw, const float* tb, void* stream);

/* The same two layers for C == 256 (one wave = the 64 float4 channel groups of a pixel; filter taps in registers):
 * mdqe_dwconv5x5_c256_f32 = the plain depthwise 5x5 on x [NI,H,W,256]; mdqe_dwconv5x5_up2_c256_f32 = ConvTranspose2d(k=1,s=2,
 * output_padding=1,groups=C) -> depthwise 5x5 in one pass, x [NI,Hs,Ws,256] -> y [NI,2Hs,2Ws,256], evaluated per 2x2
 * output quad (segmentation.py:28-29,59,92-98 of the mask-feature head); wt [25,256] tap-major. */
int mdqe_dwconv5x5_c256_f32(const float* x, const float* wt, const float* bias, float* y, int NI, int H, int W, void* stream);
int mdqe_dwconv5x5_up2_c256_f32(const float* x, const float* wt, const float* bias, const float* tw, const float* tb, float* y,
                                int NI, int Hs, int Ws, void* stream);

/* ---- tracker, device half (mdqe/tracking/OverTracker.py) ------------------------------------------
 * siou: out3[i,j,:] = (|A_i & B_j|, |A_i|, |B_j|) with A_i = saved[i*saved_stride + k] > 0, B_j = inp[j*inp_stride + k] > 0,
 * k < n (the overlapping frames of both are contiguous); feeds OverTracker._get_siou (:92-113).
 * accumulate: sum[r[k]*sum_stride + e] += src[c[k]*src_stride + e] (e < n), cnt[r[k]*cnt_stride + f] += 1 (f < nf);
 * r/c are HOST int arrays of any length  (OverTracker._update_memory :65-76). */
int mdqe_trk_siou_f32(const float* saved, long saved_stride, int n_saved, const float* inp, long inp_stride,
                      int n_in, long n, float* out3, void* stream);
int mdqe_trk_accumulate_f32(float* sum, long sum_stride, float* cnt, long cnt_stride, const float* src,
                            long src_stride, long n, int nf, const int* r_host, const int* c_host, int count,
                            void* stream);
/* siou_host (round 5): the same counts delivered to the HOST by the kernel itself -- one launch instead of memset + kernel +
 * device->host copy + synchronize on the tracker's per-clip critical path (the `.cpu()` of OverTracker.py:159).  acc: device scratch
 * of >= n_saved*n_in*3 floats, ZERO on entry and left zero; ticket: one device unsigned, zero on entry and left zero; out_host:
 * >= n_saved*n_in*3 64-bit words of host-coherent pinned memory (hipHostMallocCoherent).  The block that finishes last stores word k =
 * (seq << 32 | float bits of count k), seq != 0.  wait_counts: polls until every word carries seq (at most spin_us microseconds, then a
 * stream synchronize: kernel completion makes the stores visible in any case) and unpacks the counts; MDQE_ELAUNCH if they never arrive. */
int mdqe_trk_siou_host_f32(const float* saved, long saved_stride, int n_saved, const float* inp, long inp_stride,
                           int n_in, long n, float* acc, unsigned* ticket, unsigned long long* out_host,
                           unsigned seq, void* stream);
int mdqe_trk_wait_counts(const unsigned long long* out_host, int n_out, unsigned seq, int spin_us, float* counts, void* stream);

/* window flush, device half (OverTracker.get_result :195-225): window_mean: out[i,f,:] = sum[i,f0+f,:] / max(cnt[i,f0+f],1)
 * for i < n, f < nf (rows of `out` are nf frames apart).  carry: frames [src0, src0+k) of rows < n become frames [0,k) as
 * their mean logits with count 1 where seen; the rest of those rows is zeroed (`carry`: device scratch, n*k*hw floats). */
int mdqe_trk_window_mean_f32(const float* sum, const float* cnt, long hw, int mem_len, int n, int nf, long f0,
                             float* out, void* stream);
int mdqe_trk_carry_f32(float* sum, float* cnt, long hw, int mem_len, int n, int k, long src0, float* carry, void* stream);

/* ---- tracker, host half + drivers (csrc/tracker_native.hip) -----------------------------------------
 * The HOST state and decisions of OverTracker (mdqe/tracking/OverTracker.py:16-63 __init__/_init_memory, :65-90
 * _update_memory, :115-193 update, :195-225 get_result, :228-242 get_ctt_similarity) as an opaque object.  These
 * entry points are the exception to "never synchronise": mdqe_tracker_update* wait for one small device->host copy per
 * clip (the assignment needs the intersection counts), as the reference's update() does (`.cpu()` at :159).
 * All array arguments below are HOST pointers unless named bank_*, masks, counts_dev, out_masks, carry.
 *
 * mdqe_lsap_f64: scipy.optimize.linear_sum_assignment (the reference's call at OverTracker.py:159; scipy==1.8.1,
 *   requirements.txt:3) restated: cost [nr,nc] row-major -> min(nr,nc) (row, col) pairs in ascending row order.
 * create: thr = MODEL.MDQE.APPLY_CLS_THRES; E = embedding width; K = classes.
 * overlap -> (ni saved instances, bank frame s0, clip frame a, nf frames) of the frames of clip [f0, f0+n_frames) that the
 *   window already holds; decide: the update given counts3 [ni, n_in, 3] = (|A&B|, |A|, |B|) over those frames (NULL: no
 *   overlap) -> pairs (bank row r_idx[k] <- clip instance c_idx[k]) and the frame range (s0, a, nf) of the memory write;
 *   result: window class scores out_cls [n,K], n, ln (frames emitted) and, unless is_last, re-bases the host state
 *   (carry_valid [n, mem_len-win] optional).  A stand-in bank (CPU tests) drives these three.
 * update: overlap + counts (mdqe_trk_siou_host_f32 into buffers the object owns, then the flag wait; with MDQE_TRK_FAST=0 /
 *   mdqe_debug_trk_fast(0): siou kernel + D2H + sync through counts_dev / counts_host) + decide + accumulate kernel for one clip.  bank_sum [max_inst, win+T, hw],
 *   bank_cnt [max_inst, win+T]; masks [n_in, frames, hw] rows inst_stride floats apart; counts_dev / counts_host (pinned):
 *   scratch of max_inst*n_in*3 floats.  update_many: the same for a run of clips (clip i: host rows row0[i].. of
 *   scores/cls_probs/embeds, masks[i] = HOST array of device pointers).
 * get_result: result + window_mean into out_masks [n, ln, hw] + (unless is_last) carry. */
int mdqe_lsap_f64(const double* cost, int nr, int nc, int maximize, int* rows_out, int* cols_out, int* n_out);
int mdqe_tracker_create(int max_inst, int T, int win, int stride, int K, int E, float thr, void** handle);
int mdqe_tracker_destroy(void* handle);
int mdqe_tracker_state(void* handle, int* num_inst, int* num_clip, int* start_frame);
int mdqe_tracker_overlap(void* handle, int f0, int n_frames, int* ni, int* s0, int* a, int* nf);
int mdqe_tracker_decide(void* handle, int f0, int n_frames, int n_in, const float* scores, const float* cls_probs,
                        const float* embeds, const float* counts3, int* r_idx, int* c_idx, int* n_pairs,
                        int* s0, int* a, int* nf);
int mdqe_tracker_result(void* handle, int is_last, float* out_cls, int* n, int* ln, unsigned char* carry_valid);
int mdqe_tracker_update(void* handle, float* bank_sum, float* bank_cnt, long hw, int f0, int n_frames, int n_in,
                        const float* scores, const float* cls_probs, const float* embeds, const float* masks,
                        long inst_stride, float* counts_dev, float* counts_host, void* stream);
int mdqe_tracker_update_many(void* handle, float* bank_sum, float* bank_cnt, long hw, int n_clips, const int* f0,
                             const int* n_frames, const int* n_in, const int* row0, const float* scores,
                             const float* cls_probs, const float* embeds, const float* const* masks,
                             const long* inst_stride, float* counts_dev, float* counts_host, void* stream);
int mdqe_tracker_get_result(void* handle, int is_last, float* bank_sum, float* bank_cnt, long hw, float* out_masks,
                            float* carry, float* out_cls, int* n, int* ln, void* stream);

/* ---- per-clip stages as batch kernels (csrc/clip_ops.hip) --------------------------------------------
 * clip_assoc: inter-frame query association (transformer_dec.py:111-145): emb [frames, Q, E] (E in {16,32,64}), fidx [Bc, T]
 *   (device int32: cache frame of (clip, t)) -> idx [Bc, T, Q] = arg-max_q of e[f(b,t),q].e[f(b,ct),k] over the cells within
 *   +-wdw*|t-ct| of k's (Q = nb*nb grid cells).
 * clip_gather_init (:142-143,462,470-471): x [Bc*T*Q, C] = content[f(b,t), idx], ref [Bc*T*Q, 4] = (coords[..], 0.1, 0.1),
 *   xinst [Bc*Q, C] = x[b, ct]; idx NULL = identity.
 * box_refine (:473-480,492-503; util/misc.py:478-482; util/box_ops.py:8-19): boxes = sigmoid(delta + inverse_sigmoid(prev))
 *   [Bc*T*Q, 4]; ibox [Bc*Q, 4] = cxcywh of (min clamped top-left, max clamped bottom-right) over frames [t0, t1).
 * add_rows: out = a + b on row-strided operands.  time_fuse (:374-376): out[b,q,:] = sum_t softmax_t(w[b,t,q]) x[b,t,q,:]
 *   (+ pos into out_plus_pos when given). */
int mdqe_clip_assoc_f32(const float* emb, int Q, int E, const int* fidx, int Bc, int T, int ct, float wdw, int nb,
                        int* idx_out, void* stream);
int mdqe_clip_gather_init_f32(const float* content, const float* coords, const int* fidx, const int* idx, int Bc, int T,
                              int Q, int C, int ct, float* x, float* ref, float* xinst, void* stream);
int mdqe_box_refine_f32(const float* delta, const float* prev, int Bc, int T, int Q, int t0, int t1, float* boxes,
                        float* ibox, void* stream);
int mdqe_add_rows_f32(const float* a, long lda, const float* b, long ldb, float* out, long ldo, long rows, int C, void* stream);
int mdqe_time_fuse_f32(const float* w, const float* x, int Bc, int T, int Q, int C, float* out, const float* pos,
                       float* out_plus_pos, void* stream);
/* Round 3, fewer launches between the decoder's GEMMs.  box_head_refine: bbox_embed's last Linear(C -> 4) + the refinement
 * sigmoid(delta + inverse_sigmoid(prev)) + the clip-circumscribed box (transformer_dec.py:473-480,492-503) in one kernel; h [Bc*T*Q, K]
 * is the head's second hidden activation, W [4, K], bias [4].  time_fuse_dot: time_weights Linear(C -> 1) of the frame queries xw, the
 * softmax over the clip's frames and the weighted sum of src (:374-376) in one kernel.  Same arithmetic as the kernels they replace. */
int mdqe_box_head_refine_f32(const float* h, long ldh, const float* W, const float* bias, const float* prev, int Bc, int T, int Q, int K,
                             int t0, int t1, float* boxes, float* ibox, void* stream);
int mdqe_time_fuse_dot_f32(const float* xw, const float* wt, const float* bt, const float* src, int Bc, int T, int Q, int C, float* out,
                           void* stream);

/* inference_clip (mdqe/mdqe.py:368-428) for a batch of B clips.
 * clip_select (:373-379): cls [B,Q,K], emb [B,Q,C] -> kept [B,Q] (query indices of the kept ranks, score order), n_keep [B]:
 *   sort by best class score, keep >= min(thr, best), drop rank q if max_{p<q} cos(e_p, e_q) >= 0.99, cap at max_keep.
 *   Workspaces: order [B,Q] int, n_thr [B] int, inv_norm [B,Q], sim [B,Q,Q].
 * dyn_mask_nms: the fused dynamic-mask kernel + NMS.  HOST int arrays row0 / n / f0 [B] (first instance row, kept count,
 *   first cache frame of the clip).  coef [B,Q,M], feats [frames,H,W,M] channels-last ->
 *   logits [n_rows,T,H,W] = einsum 'qm,mthw->qthw' (:384); stats [n_rows,5] = (any(x>0), sum sigmoid(x)[hard], count(hard),
 *   sum_half sigmoid(x), count_half(hard)), hard = sigmoid(x) > 0.5, half = F.interpolate(scale_factor=0.5) grid of :394-396
 *   (every 2nd frame when T >= 5); mi [n_rows] = max_{p<q} soft-IoU (:398-405): `soft @ hard.t()` as one batched MFMA launch.
 *   Scratch: soft_h / hard_h [n_rows, Ph] (Ph = ceil(T/t_step)*(H/2)*(W/2)), part [mdqe_dyn_mask_workspace_floats()],
 *   gram [mdqe_nms_workspace_floats(max kept per clip)].
 * clip_finalize (:408-419): out [n_rows, 2+K+C] = (score, label, class scores, embedding) of the j-th selected row of clip b at
 *   row row0[b]+j, sel [n_rows] its instance row, n_sel [B].
 * rows_gather: out[i,:] = src[idx[i],:]. */
int mdqe_clip_select_f32(const float* cls, const float* emb, int B, int Q, int K, int C, float thr, int max_keep,
                         int* order_ws, int* n_thr_ws, float* inv_norm_ws, float* sim_ws, int* kept, int* n_keep, void* stream);
long mdqe_dyn_mask_workspace_floats(int n_rows, int T, int H, int W);
int mdqe_dyn_mask_nms_f32(const float* coef, const int* kept, const float* feats, int B, int Q, int M, int T, int H, int W,
                          const int* row0_host, const int* n_host, const int* f0_host, float* logits, float* soft_h,
                          float* hard_h, float* part, float* gram, float* stats, float* mi, void* stream);
long mdqe_nms_workspace_floats(int n_max);
int mdqe_clip_finalize_f32(const float* cls, const float* emb, const int* kept, const float* stats, const float* mi, int B,
                           int Q, int K, int C, float thr, const int* row0_host, const int* n_host, int* sel, int* n_sel,
                           float* out, void* stream);
int mdqe_rows_gather_f32(const float* src, const int* idx_dev, int n, long len, float* out, void* stream);

/* ---- per-row statistics of dynamic mask logits (mdqe/mdqe.py:387-413) in one pass -------------------
 * logits [n, T, H, W].  stats[r] = (any(x>0), sum sigmoid(x)[x>0], count[x>0], sum_half sigmoid(x), count_half[x>0]);
 * half = every 2nd pixel in y and x (and every 2nd frame when t_step == 2), i.e. F.interpolate(scale_factor=0.5,
 * nearest) of :394-396; soft_h / hard_h [n, Th*(H/2)*(W/2)] receive sigmoid(x) and [x>0] on that grid. */
int mdqe_mask_row_stats_f32(const float* logits, int n, int T, int H, int W, int t_step, float* stats5,
                            float* soft_h, float* hard_h, void* stream);

/* ---- nn.MultiheadAttention core for short sequences (transformer_dec.py:348-353,397-402) --------------
 * o[b,q,h,:] = softmax_k((q*D^-0.5).k) @ v ; qk rows hold q at col h*D and k at col C+h*D.  Q <= 256, D in {8,16,24,32}. */
int mdqe_mha_small_f32(const float* qk, long ldqk, const float* v, long ldv, float* o, long ldo, int B, int Q,
                       int C, int nh, void* stream);

/* ---- grid-guided query selection (transformer_dec.py:81-109): conf [NI,H,W,K] -> coords [NI,nb*nb,2] (x,y);
 * score_ws: NI*H*W floats. */
int mdqe_query_select_f32(const float* conf, int NI, int H, int W, int K, int nb, float* score_ws, float* coords,
                          void* stream);

/* ---- query content (transformer_dec.py:171-179): mean over levels of border-mode bilinear grid_sample of the
 * channels-last tokens [NI,N,C] at coords [NI,Qn,2]; level tables are HOST int[n_levels]. */
int mdqe_sample_levels_mean_f32(const float* tokens, int NI, long N, int C, const float* coords, int Qn,
                                const int* lvH_host, const int* lvW_host, const int* lvStart_host, int n_levels,
                                float* out, void* stream);

/* ---- final masks (mdqe/mdqe.py:357-358,458-462; util/misc.py:485-507) fused: x`factor` aligned-bilinear ->
 * sigmoid -> crop [:h,:w] -> nearest resize to (Ho,Wo) -> > 0.5.  logits [*,Fw,Hm,Wm] (one tracker window);
 * row k of out (uint8 [n_sel][out_inst_stride]) takes instance inst_idx_dev[k], frames written at f_off.. */
int mdqe_final_masks_u8(const float* logits, int n_sel, const int* inst_idx_dev, int Fw, int Hm, int Wm, int factor,
                        int h, int w, int Ho, int Wo, unsigned char* out, long out_inst_stride, int f_off,
                        void* stream);

/* Same masks, straight to COCO run-length form (the result writer's mask_util.encode, mdqe/data/ytvis_eval.py:307-312 ->
 * cocoapi rleEncode): pos[(k*Fw+f)*cap + i] = i-th column-major pixel index at which mask (k,f) changes value (the value
 * before the first pixel is 0), n_pos[k*Fw+f] = number of changes (may exceed cap: only the first cap are stored).
 * Run lengths are the differences of consecutive positions; the dense mask is never written. */
int mdqe_final_masks_rle(const float* logits, int n_sel, const int* inst_idx_dev, int Fw, int Hm, int Wm, int factor,
                         int h, int w, int Ho, int Wo, int cap, int* pos, int* n_pos, void* stream);

/* ---- COCO single-image branch, after the decoder (MDQE.inference_image, mdqe/mdqe.py:486-556), on the centre frame's
 * low-resolution logits [n,Hm,Wm]; aligned_bilinear x`factor` in closed form, crop [:h,:w].
 * stats[k] = {sum(sigmoid*[sigmoid>0.5]), count(sigmoid>0.5), xmin, ymin, xmax, ymax of (logit > 0)} (:512-516, :526;
 * BitMasks.get_bounding_boxes = [xmin, ymin, xmax+1, ymax+1]; xmin > xmax for an empty mask).
 * final masks: F.interpolate(bilinear, align_corners=False) of the cropped up-sampled logits to (Ho,Wo), > 0 (:545-547);
 * row k of out (uint8 [n_sel,Ho,Wo]) takes mask idx_dev[k]. */
int mdqe_image_mask_stats_f32(const float* logits, int n, int Hm, int Wm, int factor, int h, int w, float* stats, void* stream);
int mdqe_image_final_masks_u8(const float* logits, int n_sel, const int* idx_dev, int Hm, int Wm, int factor, int h, int w,
                              int Ho, int Wo, unsigned char* out, void* stream);

/* ---- SwinV2 backbone (mdqe/backbone/swin_transformer_v2.py) --------------------------------------------
 * layernorm_post: y = LN(x)*gamma + beta + post (res-post-norm :287-288).
 * patch4_im2col: normalise + zero-pad + 4x4/s4 im2col, k = c*16+kh*4+kw -> [NI*Hp/4*Wp/4, 48] (PatchEmbed :466-479).
 * swin_window: mode 0 gathers the zero-padded, cyclically shifted map into window-ordered rows [B*nW*ws*ws, C];
 *              mode 1 writes dst[b,y,x,:] = shortcut[b,y,x,:] + rows (window reverse + un-shift + crop, :252-288).
 * window_attn: cosine attention of WindowAttention.forward (:147-186) on qkv rows [n_windows*N, 3C]; scale [nh] =
 *              exp(clamp(logit_scale)), bias [nh,N,N] = 16*sigmoid(cpb), mask [nW,N,N] or NULL (device arrays).
 * patch_merge_gather: [B,H,W,C] -> [B*ceil(H/2)*ceil(W/2), 4C] in the order x0|x1|x2|x3 of PatchMerging (:311-335). */
int mdqe_layernorm_post_f32(const float* x, const float* gamma, const float* beta, const float* post, float* y,
                            long rows, int C, float eps, void* stream);
/* Round 4: the window partition and its reverse without their copies (SwinTransformerBlock.forward, swin_transformer_v2.py:236-288).
 * mdqe_gemm_nt_swin_f32: C = window_partition(roll(pad(X))) W^T + bias -- the qkv product (WindowAttention.forward :153-155) reads its A
 *   rows from the NHWC map X [B, H, Wd, lda] through the window order; C [B*Hp*Wp, ldc], Hp / Wp = H / Wd rounded up to multiples of ws;
 *   padded positions read zeros.  Exact-fp32 mode only (MDQE_EINVAL in the split-precision mode: partition first, then mdqe_gemm_nt_f32).
 * mdqe_layernorm_swin_scatter_f32: out[b,y,x,:] = shortcut[b,y,x,:] + LN(rows[r,:]) * gamma + beta with r the window-order row of the
 *   pixel: norm1, window reverse, un-shift, crop and the residual add (:273-287) in one pass; out may alias shortcut. */
int mdqe_gemm_nt_swin_f32(const float* X, long lda, const float* W, const float* bias, float* C, long ldc, int B, int H, int Wd,
                          int ws, int shift, int N, int K, void* stream);
int mdqe_layernorm_swin_scatter_f32(const float* rows, const float* gamma, const float* beta, const float* shortcut, float* out,
                                    int B, int H, int W, int C, int ws, int shift, float eps, void* stream);
int mdqe_patch4_im2col_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                           const float* mean3_host, const float* std3_host, float* out, void* stream);
int mdqe_swin_window_f32(const float* src, const float* shortcut, float* dst, int B, int H, int W, int C, int ws,
                         int shift, int mode, void* stream);
int mdqe_window_attn_f32(const float* qkv, long ld, float* o, long ldo, int n_windows, int N, int C, int nh,
                         const float* scale, const float* bias, const float* mask, int nW, void* stream);
int mdqe_patch_merge_gather_f32(const float* x, float* out, int B, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
