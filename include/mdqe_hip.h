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

/* activation codes for fused epilogues */
#define MDQE_ACT_NONE 0
#define MDQE_ACT_RELU 1
#define MDQE_ACT_GELU 2 /* exact erf GELU (nn.GELU default) */
#define MDQE_ACT_SIGMOID 3
#define MDQE_ACT_TANH 4

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

/* Grouped form used for temporal_clip_forward (mdqe/models/ops/modules/ms_deform_attn.py:219-236):
 * the reference issues G (= #spatial levels) separate native calls that share loc/attn and averages
 * them.  Here: shapes/level_start are [G*L] tables into ONE value buffer, out = scale * sum_g (...). */
int mdqe_msda_forward_grouped_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                  const float* loc, const float* attn,
                                  int B, int S, int M, int D, int G, int L, int Q, int P,
                                  float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
