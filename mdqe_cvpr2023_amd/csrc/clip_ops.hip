// Per-clip stages of the MDQE eval path as a handful of gfx950 kernels (SURVEY.md §8 a11, a12 glue, a15):
// inter-frame query association, decoder initialisation gathers, iterative box refinement + clip boxes,
// time-softmax fusion, and `inference_clip` (mdqe/mdqe.py:368-428) for a whole BATCH of clips: score sort /
// threshold / near-duplicate removal, the fused dynamic-mask kernel (einsum 'qm,mthw->qthw' :384 with the blank test,
// mask-quality sums and the half-resolution soft / hard maps of the NMS in its epilogue), the soft-IoU NMS matrix and
// the final per-clip top-k.  Every reduction runs in a fixed order: results do not depend on the launch schedule.
#include "common.h"

#include <math.h>

// ------------------------------------------------------------------------------------------------
// a11: inter-frame query association (transformer_dec.py:111-145).  For clip b, frame t, centre-frame query k:
//   idx[b,t,k] = argmax_q  e[f(b,t), q] . e[f(b,ct), k]   over the q whose grid cell lies within +-w*|t-ct| cells of k's
// (softmax is monotone: its arg-max is the arg-max of the masked similarity; first maximum wins).
// One block per (b, t, 64 centre queries); the frame's embeddings [Q, E] go through LDS, the block's centre rows are
// register-resident; every lane reads the same LDS row -> broadcast reads.
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ void __launch_bounds__(64)
clip_assoc_kernel(const float* __restrict__ emb, const int* __restrict__ fidx, int T, int Q, int ct, float wdw, int nb,
                  int* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float sE[];          // [Q][E]
  const int bt = blockIdx.y, b = bt / T, t = bt % T;
  const int k = blockIdx.x * 64 + threadIdx.x;
  const float* et = emb + (long)fidx[b * T + t] * Q * E;
  const float* ec = emb + (long)fidx[b * T + ct] * Q * E;
  for (int i = threadIdx.x; i < Q * (E / 4); i += 64)
    *reinterpret_cast<f32x4*>(sE + i * 4) = *reinterpret_cast<const f32x4*>(et + i * 4);
  __syncthreads();
  if (k >= Q) return;
  float c[E];
#pragma unroll
  for (int e = 0; e < E; e += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ec + (long)k * E + e);
    c[e] = v[0]; c[e + 1] = v[1]; c[e + 2] = v[2]; c[e + 3] = v[3];
  }
  const float w = wdw * (float)abs(t - ct);
  const int ki = k / nb, kj = k % nb;
  // The reference takes `masked_fill(-inf).softmax(-2).argmax(-2)` (transformer_dec.py:142-143): the FIRST index whose fp32
  // softmax value equals the column's largest.  Two roundings make cells below the maximum tie with it: (i) exp(s - max) rounds
  // to 1.0f for every s with s - max > -2^-25; (ii) the division by the column sum: 1.0f / sum and (1 - 2^-24) / sum are half an
  // ulp to one ulp of the quotient apart and can round to the same float (numerators from 1 - 2^-23 down are >= 1 ulp away and
  // cannot).  Pass 1 finds the maximum (first index among exact ties) and the largest value strictly below it.  Only when that
  // runner-up lies within 2^-23 of the maximum -- possible only for |max| < 1, where fp32 similarities are spaced that finely --
  // the slow path evaluates what the reference evaluates: sum = sum over the admissible cells of exp(s - max) (index order),
  // p(q) = exp(s_q - max) / sum in fp32 (exp correctly rounded), and the first admissible q < argmax with p(q) == 1.0f / sum wins.
  // What stays open: torch's vectorised exp and its summation order (lanes of 8 / 16 partial sums) differ from expf and this
  // loop by an ulp, as does the 64-term dot product itself (torch's GEMM vs this loop), so a near-tie of an ulp or two can still
  // resolve differently; a NaN similarity makes the reference's whole column NaN (argmax -> index 0) while this loop keeps the
  // first admissible cell.  tools/fuzz_inference_clip.py injects near-ties at 1-3 ulps and bounds how often.
  constexpr float BAND = -1.1920928955078125e-7f;                       // -2^-23
  auto sim = [&](int q) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sE + q * E + e);
      s += v[0] * c[e]; s += v[1] * c[e + 1]; s += v[2] * c[e + 2]; s += v[3] * c[e + 3];
    }
    return s;
  };
  auto admissible = [&](int q) {
    const int qi = q / nb, qj = q % nb;
    return !((float)abs(qi - ki) > w || (float)abs(qj - kj) > w);
  };
  float best = -INFINITY, second = -INFINITY;
  int bi = 0;
  bool have = false;
  for (int q = 0; q < Q; ++q) {
    if (!admissible(q)) continue;
    const float s = sim(q);
    if (!have || s > best) { if (have) second = best; best = s; bi = q; have = true; }
    else if (s < best && s > second) second = s;
  }
  if (have && second - best > BAND) {                                   // rare: an earlier cell may tie with the maximum in fp32 softmax
    // (exp through double: the hardware expf is an ulp or two off, and whether exp(-3.7e-8) rounds to 1 or to 1 - 2^-24 is the question)
    auto exp32 = [](float d) { return (float)exp((double)d); };
    float sum = 0.f;
    for (int q = 0; q < Q; ++q)
      if (admissible(q)) sum += exp32(sim(q) - best);
    const float top = 1.0f / sum;
    for (int q = 0; q < bi; ++q) {
      if (!admissible(q)) continue;
      const float d = sim(q) - best;
      if (d > BAND && exp32(d) / sum == top) { bi = q; break; }
    }
  }
  idx[((long)b * T + t) * Q + k] = bi;
}

extern "C" int mdqe_clip_assoc_f32(const float* emb, int Q, int E, const int* fidx, int Bc, int T, int ct, float wdw, int nb,
                                   int* idx_out, void* stream) {
  MDQE_REQUIRE(Q > 0 && Bc >= 0 && T > 0 && ct >= 0 && ct < T && nb > 0 && nb * nb == Q);
  MDQE_REQUIRE(E == 16 || E == 32 || E == 64);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(emb); MDQE_CHECK_PTR(fidx); MDQE_CHECK_PTR(idx_out);
  MDQE_REQUIRE((size_t)Q * E * 4 <= 150 * 1024);
  mdqe_clear_error();
  const dim3 grid((Q + 63) / 64, Bc * T);
  const size_t sm = (size_t)Q * E * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (E == 64) hipLaunchKernelGGL(clip_assoc_kernel<64>, grid, dim3(64), sm, st, emb, fidx, T, Q, ct, wdw, nb, idx_out);
  else if (E == 32) hipLaunchKernelGGL(clip_assoc_kernel<32>, grid, dim3(64), sm, st, emb, fidx, T, Q, ct, wdw, nb, idx_out);
  else hipLaunchKernelGGL(clip_assoc_kernel<16>, grid, dim3(64), sm, st, emb, fidx, T, Q, ct, wdw, nb, idx_out);
  return mdqe_launch_status();
}

// Decoder inputs of a batch of clips from the per-frame cache (transformer_dec.py:142-143,462,470-471):
//   x[b,t,k,:] = content[f(b,t), idx[b,t,k], :]; ref[b,t,k] = (coords[f(b,t), idx[b,t,k]], 0.1, 0.1); x_inst[b,k] = x[b,ct,k]
// idx == NULL: identity (single-frame clips).  One wave per row of C floats.
__global__ void __launch_bounds__(256)
clip_gather_init_kernel(const float* __restrict__ content, const float* __restrict__ coords, const int* __restrict__ fidx,
                        const int* __restrict__ idx, int T, int Q, int C, int ct, long rows, float* __restrict__ x,
                        float* __restrict__ ref, float* __restrict__ xinst) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const int k = (int)(r % Q);
  const long bt = r / Q;
  const int t = (int)(bt % T);
  const long b = bt / T;
  const int q = idx ? idx[r] : k;
  const long src = (long)fidx[bt] * Q + q;
  const float* s = content + src * C;
  float* d = x + r * C;
  float* di = (t == ct) ? xinst + (b * Q + k) * C : nullptr;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(s + c);
    *reinterpret_cast<f32x4*>(d + c) = v;
    if (di) *reinterpret_cast<f32x4*>(di + c) = v;
  }
  if (lane == 0) *reinterpret_cast<f32x4*>(ref + r * 4) = f32x4{coords[src * 2], coords[src * 2 + 1], 0.1f, 0.1f};
}

extern "C" int mdqe_clip_gather_init_f32(const float* content, const float* coords, const int* fidx, const int* idx, int Bc, int T,
                                         int Q, int C, int ct, float* x, float* ref, float* xinst, void* stream) {
  MDQE_REQUIRE(Bc >= 0 && T > 0 && Q > 0 && C > 0 && C % 4 == 0 && ct >= 0 && ct < T);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(content); MDQE_CHECK_PTR(coords); MDQE_CHECK_PTR(fidx); MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(ref); MDQE_CHECK_PTR(xinst);
  mdqe_clear_error();
  const long rows = (long)Bc * T * Q;
  hipLaunchKernelGGL(clip_gather_init_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, content, coords, fidx,
                     idx, T, Q, C, ct, rows, x, ref, xinst);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// a12: iterative box refinement + clip-circumscribed box (transformer_dec.py:473-480,492-503; util/misc.py:478-482;
// util/box_ops.py:8-19).  One thread per (clip, query): for every frame t
//   box[b,t,q] = sigmoid(delta[b,t,q] + inverse_sigmoid(prev[b,t,q]))
// and over the frames [t0, t1): the box spanned by the min of the clamped top-left and the max of the clamped bottom-right corners.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float a = fmaxf(x, 1e-5f), b = fmaxf(1.f - x, 1e-5f);
  return logf(a / b);
}

__global__ void __launch_bounds__(256)
box_refine_kernel(const float* __restrict__ delta, const float* __restrict__ prev, int T, int Q, int t0, int t1, long n,
                  float* __restrict__ boxes, float* __restrict__ ibox) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (b, q)
  if (i >= n) return;
  const long b = i / Q;
  const int q = (int)(i % Q);
  float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
  for (int t = 0; t < T; ++t) {
    const long r = (b * T + t) * Q + q;
    const f32x4 d = *reinterpret_cast<const f32x4*>(delta + r * 4);
    const f32x4 p = *reinterpret_cast<const f32x4*>(prev + r * 4);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = 1.0f / (1.0f + expf(-(d[c] + inv_sigmoid(p[c]))));
    *reinterpret_cast<f32x4*>(boxes + r * 4) = o;
    if (t >= t0 && t < t1) {
      const float ax = fminf(fmaxf(o[0] - 0.5f * o[2], 0.f), 1.f), ay = fminf(fmaxf(o[1] - 0.5f * o[3], 0.f), 1.f);
      const float bx = fminf(fmaxf(o[0] + 0.5f * o[2], 0.f), 1.f), by = fminf(fmaxf(o[1] + 0.5f * o[3], 0.f), 1.f);
      x0 = fminf(x0, ax); y0 = fminf(y0, ay); x1 = fmaxf(x1, bx); y1 = fmaxf(y1, by);
    }
  }
  *reinterpret_cast<f32x4*>(ibox + i * 4) = f32x4{(x0 + x1) / 2.f, (y0 + y1) / 2.f, x1 - x0, y1 - y0};
}

extern "C" int mdqe_box_refine_f32(const float* delta, const float* prev, int Bc, int T, int Q, int t0, int t1, float* boxes,
                                   float* ibox, void* stream) {
  MDQE_REQUIRE(Bc >= 0 && T > 0 && Q > 0 && t0 >= 0 && t0 < t1);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(delta); MDQE_CHECK_PTR(prev); MDQE_CHECK_PTR(boxes); MDQE_CHECK_PTR(ibox);
  mdqe_clear_error();
  const long n = (long)Bc * Q;
  hipLaunchKernelGGL(box_refine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, prev, T, Q, t0,
                     t1 < T ? t1 : T, n, boxes, ibox);
  return mdqe_launch_status();
}

// out = a + b (row-strided a, b, out; C % 4 == 0)
__global__ void __launch_bounds__(256)
add_rows_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb, float* __restrict__ o, long ldo,
                long rows, int C4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C4) return;
  const long r = i / C4;
  const int c = (int)(i % C4) * 4;
  const f32x4 x = *reinterpret_cast<const f32x4*>(a + r * lda + c);
  const f32x4 y = *reinterpret_cast<const f32x4*>(b + r * ldb + c);
  *reinterpret_cast<f32x4*>(o + r * ldo + c) = x + y;
}

extern "C" int mdqe_add_rows_f32(const float* a, long lda, const float* b, long ldb, float* out, long ldo, long rows, int C,
                                 void* stream) {
  MDQE_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0);
  if (rows == 0) return MDQE_OK;
  MDQE_CHECK_PTR(a); MDQE_CHECK_PTR(b); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  const long n = rows * (C / 4);
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo,
                     rows, C / 4);
  return mdqe_launch_status();
}

// Instance query from the frame-level queries (transformer_dec.py:374-376):
//   out[b,q,:] = sum_t softmax_t(w[b,t,q]) * x[b,t,q,:]      (+ pos[b,q,:] into out2 when given)
__global__ void __launch_bounds__(256)
time_fuse_kernel(const float* __restrict__ w, const float* __restrict__ x, int T, int Q, int C4, long n, float* __restrict__ out,
                 const float* __restrict__ pos, float* __restrict__ out2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C4) * 4;
  const long bq = i / C4;
  const int q = (int)(bq % Q);
  const long b = bq / Q;
  float m = -INFINITY;
  for (int t = 0; t < T; ++t) m = fmaxf(m, w[(b * T + t) * Q + q]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(w[(b * T + t) * Q + q] - m);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const float p = expf(w[(b * T + t) * Q + q] - m) / den;
    acc += p * *reinterpret_cast<const f32x4*>(x + ((b * T + t) * Q + q) * (long)(C4 * 4) + c);
  }
  *reinterpret_cast<f32x4*>(out + bq * (long)(C4 * 4) + c) = acc;
  if (out2) *reinterpret_cast<f32x4*>(out2 + bq * (long)(C4 * 4) + c) = acc + *reinterpret_cast<const f32x4*>(pos + bq * (long)(C4 * 4) + c);
}

extern "C" int mdqe_time_fuse_f32(const float* w, const float* x, int Bc, int T, int Q, int C, float* out, const float* pos,
                                  float* out_plus_pos, void* stream) {
  MDQE_REQUIRE(Bc >= 0 && T > 0 && Q > 0 && C > 0 && C % 4 == 0);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(w); MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(out);
  if (out_plus_pos) MDQE_CHECK_PTR(pos);
  mdqe_clear_error();
  const long n = (long)Bc * Q * (C / 4);
  hipLaunchKernelGGL(time_fuse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, x, T, Q, C / 4, n, out,
                     pos, out_plus_pos);
  return mdqe_launch_status();
}

// ---- fused forms (round 3): fewer, fatter launches between the decoder's GEMMs -------------------------------------------------
// box_head_refine: the LAST layer of the box head (Linear(C -> 4), transformer_dec.py:492-493 `bbox_embed`) + the iterative
// refinement sigmoid(delta + inverse_sigmoid(prev)) of every frame of a (clip, query) + the clip-circumscribed box (:473-480,
// :496-503) in ONE kernel: one wave per (clip, query) walks its T rows of the head's hidden activation; a lane takes 4 consecutive k
// of a row, the four weight rows come from L1, xor-shuffle sums -- the arithmetic of rows_dot_kernel<4> (gemm.hip) followed by the
// arithmetic of box_refine_kernel, bit for bit, so the fused and the two-kernel forms are interchangeable.
__global__ void __launch_bounds__(256)
box_head_refine_kernel(const float* __restrict__ h, long ldh, const float* __restrict__ W, const float* __restrict__ bias,
                       const float* __restrict__ prev, int T, int Q, int K, int t0, int t1, long n, float* __restrict__ boxes,
                       float* __restrict__ ibox) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);             // (b, q)
  if (i >= n) return;
  const long b = i / Q;
  const int q = (int)(i % Q);
  float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
  for (int t = 0; t < T; ++t) {
    const long r = (b * T + t) * Q + q;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 256) {
      const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h + r * ldh + k0 + lane * 4));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)c * K + k0 + lane * 4);
        acc[c] = fmaf(a[3], w[3], fmaf(a[2], w[2], fmaf(a[1], w[1], fmaf(a[0], w[0], acc[c]))));
      }
    }
    f32x4 d;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = acc[c];
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
      d[c] = v + bias[c];
    }
    const f32x4 p = *reinterpret_cast<const f32x4*>(prev + r * 4);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = 1.0f / (1.0f + expf(-(d[c] + inv_sigmoid(p[c]))));
    if (lane == 0) *reinterpret_cast<f32x4*>(boxes + r * 4) = o;
    if (t >= t0 && t < t1) {
      const float ax = fminf(fmaxf(o[0] - 0.5f * o[2], 0.f), 1.f), ay = fminf(fmaxf(o[1] - 0.5f * o[3], 0.f), 1.f);
      const float bx = fminf(fmaxf(o[0] + 0.5f * o[2], 0.f), 1.f), by = fminf(fmaxf(o[1] + 0.5f * o[3], 0.f), 1.f);
      x0 = fminf(x0, ax); y0 = fminf(y0, ay); x1 = fmaxf(x1, bx); y1 = fmaxf(y1, by);
    }
  }
  if (lane == 0) *reinterpret_cast<f32x4*>(ibox + i * 4) = f32x4{(x0 + x1) / 2.f, (y0 + y1) / 2.f, x1 - x0, y1 - y0};
}

extern "C" int mdqe_box_head_refine_f32(const float* h, long ldh, const float* W, const float* bias, const float* prev, int Bc, int T,
                                        int Q, int K, int t0, int t1, float* boxes, float* ibox, void* stream) {
  MDQE_REQUIRE(Bc >= 0 && T > 0 && Q > 0 && K > 0 && K % 256 == 0 && ldh >= K && ldh % 4 == 0 && t0 >= 0 && t0 < t1);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(h); MDQE_CHECK_PTR(W); MDQE_CHECK_PTR(bias); MDQE_CHECK_PTR(prev); MDQE_CHECK_PTR(boxes); MDQE_CHECK_PTR(ibox);
  MDQE_REQUIRE((((uintptr_t)h | (uintptr_t)W | (uintptr_t)prev | (uintptr_t)boxes | (uintptr_t)ibox) & 15) == 0);
  mdqe_clear_error();
  const long n = (long)Bc * Q;
  hipLaunchKernelGGL(box_head_refine_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, ldh, W, bias, prev, T, Q, K,
                     t0, t1 < T ? t1 : T, n, boxes, ibox);
  return mdqe_launch_status();
}

// time_fuse_dot: the time weights Linear(C -> 1) (transformer_dec.py:374 `time_weights`) of the T frame queries of a (clip, query), their
// softmax over t, and the weighted sum of the frame queries `src` (:375-376) in ONE kernel: one wave per (clip, query); the dot product
// is rows_dot_kernel<1>'s arithmetic, the softmax / sum time_fuse_kernel's -- bit for bit the two-kernel form.  C == 256.
__global__ void __launch_bounds__(256)
time_fuse_dot_kernel(const float* __restrict__ xw, const float* __restrict__ wt, const float* __restrict__ bt,
                     const float* __restrict__ src, int T, int Q, long n, float* __restrict__ out) {
  constexpr int C = 256, TMAX = 8;
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);             // (b, q)
  if (i >= n) return;
  const long b = i / Q;
  const int q = (int)(i % Q);
  const f32x4 w = *reinterpret_cast<const f32x4*>(wt + lane * 4);
  float tw[TMAX];
  float m = -INFINITY;
  for (int t = 0; t < T; ++t) {
    const long r = (b * T + t) * Q + q;
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xw + r * C + lane * 4));
    float v = fmaf(a[3], w[3], fmaf(a[2], w[2], fmaf(a[1], w[1], fmaf(a[0], w[0], 0.f))));
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
    tw[t] = v + bt[0];
    m = fmaxf(m, tw[t]);
  }
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(tw[t] - m);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const float p = expf(tw[t] - m) / den;
    acc += p * *reinterpret_cast<const f32x4*>(src + ((b * T + t) * Q + q) * (long)C + lane * 4);
  }
  *reinterpret_cast<f32x4*>(out + i * (long)C + lane * 4) = acc;
}

extern "C" int mdqe_time_fuse_dot_f32(const float* xw, const float* wt, const float* bt, const float* src, int Bc, int T, int Q, int C,
                                      float* out, void* stream) {
  MDQE_REQUIRE(Bc >= 0 && T > 0 && T <= 8 && Q > 0 && C == 256);
  if (Bc == 0) return MDQE_OK;
  MDQE_CHECK_PTR(xw); MDQE_CHECK_PTR(wt); MDQE_CHECK_PTR(bt); MDQE_CHECK_PTR(src); MDQE_CHECK_PTR(out);
  MDQE_REQUIRE((((uintptr_t)xw | (uintptr_t)wt | (uintptr_t)src | (uintptr_t)out) & 15) == 0);
  mdqe_clear_error();
  const long n = (long)Bc * Q;
  hipLaunchKernelGGL(time_fuse_dot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xw, wt, bt, src, T, Q, n, out);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// a15, step 1 (mdqe/mdqe.py:373-374): per clip, sort the queries by their best class score (descending; equal scores by
// query index) and keep those >= min(thr, best).  One block of 256 threads per clip (Q <= 256): bitonic sort in LDS.
// Outputs: order[b, r] = query at rank r, n_thr[b] = how many pass, inv_norm[b, r] = 1 / max(|embed|, 1e-12) of that query
// (F.normalize, :376) for the near-duplicate test.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia < ib); }

__device__ __forceinline__ void bitonic256(float* key, int* val, int tid) {
  for (int k = 2; k <= 256; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const bool up = (tid & k) == 0;
        const float ka = key[tid], kb = key[p];
        const int ia = val[tid], ib = val[p];
        const bool a_first = before(ka, ia, kb, ib);
        if (up != a_first) { key[tid] = kb; key[p] = ka; val[tid] = ib; val[p] = ia; }
      }
      __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
clip_rank_kernel(const float* __restrict__ cls, const float* __restrict__ emb, int Q, int K, int C, float thr, int* __restrict__ order,
                 int* __restrict__ n_thr, float* __restrict__ inv_norm) {
  __shared__ float key[256];
  __shared__ int val[256];
  __shared__ int cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = -INFINITY;
  if (tid < Q) {
    const float* c = cls + ((long)b * Q + tid) * K;
    s = c[0];
    for (int k = 1; k < K; ++k) s = fmaxf(s, c[k]);
  }
  key[tid] = s; val[tid] = tid;
  if (tid == 0) cnt = 0;
  __syncthreads();
  bitonic256(key, val, tid);
  const float top = key[0];
  const float lim = fminf(top, thr);
  if (tid < Q) {
    order[(long)b * Q + tid] = val[tid];
    if (key[tid] >= lim) atomicAdd(&cnt, 1);
  }
  __syncthreads();
  if (tid == 0) n_thr[b] = cnt;
  // norms of the ranked rows: one wave per row
  const int lane = tid & 63;
  for (int r = tid >> 6; r < Q; r += 4) {
    const float* e = emb + ((long)b * Q + val[r]) * C;
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += e[c] * e[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) inv_norm[(long)b * Q + r] = 1.0f / fmaxf(sqrtf(a), 1e-12f);
  }
}

// Gram matrix of the ranked embeddings of every clip: sim[b, p, q] = e[b, order[b,p]] . e[b, order[b,q]] for p, q < n_thr[b]
// (only the tiles that hold a pair p < q are computed).  64x64 tile per block, 4x4 outputs per thread, K through LDS.
__global__ void __launch_bounds__(256)
clip_gram_kernel(const float* __restrict__ emb, const int* __restrict__ order, const int* __restrict__ n_thr, int Q, int C,
                 float* __restrict__ sim) {
  __shared__ float sA[16][65], sB[16][65];
  const int b = blockIdx.z, tp = blockIdx.y, tq = blockIdx.x;
  const int n = n_thr[b];
  if (tp * 64 >= n || tq * 64 >= n || tq < tp) return;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float acc[4][4] = {};
  const int lr = tid >> 2, lc = (tid & 3) * 4;                // load: row lr (0..63), 4 columns at lc
  const int pa = tp * 64 + lr, pb = tq * 64 + lr;
  const float* ra = pa < n ? emb + ((long)b * Q + order[(long)b * Q + pa]) * C : nullptr;
  const float* rb = pb < n ? emb + ((long)b * Q + order[(long)b * Q + pb]) * C : nullptr;
  for (int k0 = 0; k0 < C; k0 += 16) {
    f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
    if (ra && k0 + lc < C) va = *reinterpret_cast<const f32x4*>(ra + k0 + lc);
    if (rb && k0 + lc < C) vb = *reinterpret_cast<const f32x4*>(rb + k0 + lc);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { sA[lc + e][lr] = va[e]; sB[lc + e][lr] = vb[e]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sA[k][ty * 4 + i]; bb[i] = sB[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = tp * 64 + ty * 4 + i, q = tq * 64 + tx * 4 + j;
      if (p < n && q < n) sim[((long)b * Q + p) * Q + q] = acc[i][j];
    }
}

// a15, step 2 (:375-379): drop rank q if max_{p<q, p passed the threshold} cos(e_p, e_q) >= 0.99 (skipped for a single
// query), cap at max_keep.  Output: kept[b, j] = query index of the j-th kept rank (score order), n_keep[b].
__global__ void __launch_bounds__(256)
clip_dedup_kernel(const float* __restrict__ sim, const int* __restrict__ order, const int* __restrict__ n_thr,
                  const float* __restrict__ inv_norm, int Q, int max_keep, int* __restrict__ kept, int* __restrict__ n_keep) {
  __shared__ int flag[256];
  const int b = blockIdx.x, q = threadIdx.x;
  const int n = n_thr[b];
  int keep = 0;
  if (q < n) {
    keep = 1;
    if (n > 1) {
      float ms = 0.f;                                         // triu(...).max(0): the zeros below the diagonal take part
      const float nq = inv_norm[(long)b * Q + q];
      for (int p = 0; p < q; ++p) ms = fmaxf(ms, sim[((long)b * Q + p) * Q + q] * inv_norm[(long)b * Q + p] * nq);
      keep = ms < 0.99f;
    }
  }
  flag[q] = keep;
  __syncthreads();
  if (q == 0) {
    int c = 0;
    for (int r = 0; r < n; ++r)
      if (flag[r] && c < max_keep) kept[(long)b * Q + c++] = order[(long)b * Q + r];
    n_keep[b] = c;
  }
}

extern "C" int mdqe_clip_select_f32(const float* cls, const float* emb, int B, int Q, int K, int C, float thr, int max_keep,
                                    int* order_ws, int* n_thr_ws, float* inv_norm_ws, float* sim_ws, int* kept, int* n_keep,
                                    void* stream) {
  MDQE_REQUIRE(B >= 0 && Q > 0 && Q <= 256 && K > 0 && C > 0 && C % 4 == 0 && max_keep > 0);
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(cls); MDQE_CHECK_PTR(emb); MDQE_CHECK_PTR(order_ws); MDQE_CHECK_PTR(n_thr_ws); MDQE_CHECK_PTR(inv_norm_ws);
  MDQE_CHECK_PTR(sim_ws); MDQE_CHECK_PTR(kept); MDQE_CHECK_PTR(n_keep);
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(clip_rank_kernel, dim3(B), dim3(256), 0, st, cls, emb, Q, K, C, thr, order_ws, n_thr_ws, inv_norm_ws);
  const int nt = (Q + 63) / 64;
  hipLaunchKernelGGL(clip_gram_kernel, dim3(nt, nt, B), dim3(256), 0, st, emb, order_ws, n_thr_ws, Q, C, sim_ws);
  hipLaunchKernelGGL(clip_dedup_kernel, dim3(B), dim3(256), 0, st, sim_ws, order_ws, n_thr_ws, inv_norm_ws, Q, max_keep, kept, n_keep);
  return mdqe_launch_status();
}

// ------------------------------------------------------------------------------------------------
// a15, step 3: the fused dynamic-mask kernel.  For every kept instance r of clip b (rows row0[b] .. row0[b]+n[b]):
//   logits[r, t, y, x] = sum_m coef[b, kept[b, r-row0], m] * feats[f0[b]+t, y, x, m]          (einsum 'qm,mthw->qthw', :384)
// feats are the channels-last mask features of the frame cache, so a thread owns ONE pixel: its M channels are loaded once
// (M*4 contiguous bytes) and every instance of the clip is a dot product against coefficients broadcast from LDS; stores are
// coalesced along the pixels.  Epilogue, per instance (mdqe/mdqe.py:387-413): blank test any(x > 0); mask quality
// sum(sigmoid(x)[hard]) and count(hard), hard = sigmoid(x) > 0.5; on the half-resolution grid of the NMS (every 2nd pixel in
// y and x, every t_step-th frame: F.interpolate(scale_factor=0.5) of :394-396) sigmoid(x) -> soft_h[r, .] and the hard bits
// of ALL instances of the clip at that pixel -> hard_t[b, pixel, word] (bit r-row0), plus their sums.  Counts come from
// wave ballots, float sums from a fixed-order butterfly; per-(instance, block) partials go to part[r, tile, 5] and are added
// in tile order by mask_stats_reduce_kernel.
// ------------------------------------------------------------------------------------------------
struct ClipMeta { int row0[64]; int n[64]; int f0[64]; };

template <int M>
__global__ void __launch_bounds__(256)
dyn_mask_kernel(const float* __restrict__ coef, const int* __restrict__ kept, const float* __restrict__ feats, int Q, int T, int H,
                int W, int t_step, int c0, ClipMeta meta, float* __restrict__ logits, float* __restrict__ soft_h,
                float* __restrict__ hard_h, float* __restrict__ part, int n_tiles, int Mreal) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int bl = blockIdx.y;                       // clip within this launch
  const int b = c0 + bl;
  const int n = meta.n[bl], row0 = meta.row0[bl];
  if (n == 0) return;
  float* sC = sm;                                  // [n][M]
  float* sP = sm + n * M;                          // [n][4 waves][5]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n * M; i += 256) {
    const int r = i / M, m = i % M;
    sC[i] = m < Mreal ? coef[((long)b * Q + kept[(long)b * Q + r]) * Mreal + m] : 0.f;
  }
  __syncthreads();
  const long P = (long)T * H * W;
  const int Hh = H / 2, Wh = W / 2;
  const long pix = (long)blockIdx.x * 256 + tid;
  const bool in = pix < P;
  float f[M];
#pragma unroll
  for (int m = 0; m < M; ++m) f[m] = 0.f;
  if (in) {
    const float* fp = feats + ((long)meta.f0[bl] * H * W + pix) * Mreal;      // the clip's T frames are consecutive in the cache
#pragma unroll
    for (int m = 0; m < M; m += 4)
      if (m < Mreal) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(fp + m);
        f[m] = v[0]; f[m + 1] = v[1]; f[m + 2] = v[2]; f[m + 3] = v[3];
      }
  }
  const int pix_i = (int)pix;                      // T*H*W < 2^31 (checked by the launcher): 32-bit divisions
  const int xx = pix_i % W;
  const int t2 = pix_i / W;
  const int yy = t2 % H, tt = t2 / H;
  const bool grid_px = in && ((xx | yy) & 1) == 0 && (xx >> 1) < Wh && (yy >> 1) < Hh && (tt % t_step) == 0;
  const long Ph = (long)((T + t_step - 1) / t_step) * Hh * Wh;
  const long ho = ((long)(tt / t_step) * Hh + (yy >> 1)) * Wh + (xx >> 1);      // index on the half-resolution grid
  for (int r = 0; r < n; ++r) {
    const float* c = sC + r * M;
    float v = 0.f;
#pragma unroll
    for (int m = 0; m < M; m += 4) {
      const f32x4 cc = *reinterpret_cast<const f32x4*>(c + m);
      v += cc[0] * f[m]; v += cc[1] * f[m + 1]; v += cc[2] * f[m + 2]; v += cc[3] * f[m + 3];
    }
    if (in) logits[(long)(row0 + r) * P + pix] = v;
    const float s = 1.0f / (1.0f + expf(-v));
    const bool hard = in && s > 0.5f;              // the reference thresholds the sigmoid (:412)
    const bool hg = hard && grid_px;
    const unsigned long long any_m = __ballot(in && v > 0.f), hard_m = __ballot(hard), hg_m = __ballot(hg);
    float qn = hard ? s : 0.f, ss = grid_px ? s : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qn += __shfl_xor(qn, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (lane == 0) {
      float* p = sP + (r * 4 + wave) * 5;
      p[0] = any_m ? 1.f : 0.f; p[1] = qn; p[2] = (float)__popcll(hard_m); p[3] = ss; p[4] = (float)__popcll(hg_m);
    }
    if (grid_px) { soft_h[(long)(row0 + r) * Ph + ho] = s; hard_h[(long)(row0 + r) * Ph + ho] = hg ? 1.f : 0.f; }
  }
  __syncthreads();
  for (int i = tid; i < n * 5; i += 256) {
    const int r = i / 5, k = i % 5;
    const float* p = sP + r * 20 + k;
    const float a = (k == 0) ? fmaxf(fmaxf(p[0], p[5]), fmaxf(p[10], p[15])) : ((p[0] + p[5]) + p[10]) + p[15];
    part[((long)(row0 + r) * n_tiles + blockIdx.x) * 5 + k] = a;
  }
}

// stats[r, k] = partials of instance r added in tile order (k = 0: any -> max)
__global__ void __launch_bounds__(256)
mask_stats_reduce_kernel(const float* __restrict__ part, int n_tiles, long n_rows, float* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * 5) return;
  const long r = i / 5;
  const int k = i % 5;
  const float* p = part + r * n_tiles * 5 + k;
  float a = p[0];
  for (int t = 1; t < n_tiles; ++t) a = (k == 0) ? fmaxf(a, p[(long)t * 5]) : a + p[(long)t * 5];
  stats[i] = a;
}

// Soft-IoU NMS inside each clip (mdqe/mdqe.py:398-408), rows in score order: num[p,q] = soft_h[p,:] . hard_h[q,:] over the
// half-resolution pixels is the reference's own `soft @ hard.t()` -- a skinny NT GEMM per clip (n x n outputs, K = Ph ~ 15k).
// One launch for all clips: block = one wave = one 32x32 tile (ti <= tj) of one clip over one of KS slices of K, on
// v_mfma_f32_32x32x2_f32; the two k values of an instruction come from the two halves of the slice (the pairing of k's is
// free in a dot product), so every lane streams its own row contiguously with 16-B loads.  Partial tiles go to
// part[clip][tile][slice][32x32] and are added in slice order by mask_iou_kernel: deterministic.
__global__ void __launch_bounds__(64)
mask_gram_kernel(const float* __restrict__ soft_h, const float* __restrict__ hard_h, int Ph, int KS, ClipMeta meta,
                 int max_tiles, float* __restrict__ part) {
  const int bl = blockIdx.z;
  const int n = meta.n[bl], row0 = meta.row0[bl];
  const int nt = (n + 31) >> 5;
  int tile = blockIdx.y;
  if (n < 2 || tile >= nt * (nt + 1) / 2) return;
  int ti = 0;
  while (tile >= nt - ti) { tile -= nt - ti; ++ti; }
  const int tj = ti + tile;
  const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
  const int ks = blockIdx.x;
  const int per = ((Ph + KS - 1) / KS + 7) & ~7;             // slice length, multiple of 8 (two halves of 4-wide loads)
  const int k0 = ks * per + lh * (per / 2);
  const int kend = min(Ph, ks * per + (lh + 1) * (per / 2));
  const int ra = min(ti * 32 + lr, n - 1), rb = min(tj * 32 + lr, n - 1);
  const float* pa = soft_h + (long)(row0 + ra) * Ph;
  const float* pb = hard_h + (long)(row0 + rb) * Ph;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = k0; k < k0 + per / 2; k += 4) {                 // both halves run the same trip count (MFMA is wave-wide)
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (k + 3 < kend && (Ph & 3) == 0) {
      a = *reinterpret_cast<const f32x4*>(pa + k);
      b = *reinterpret_cast<const f32x4*>(pb + k);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < kend) { a[e] = pa[k + e]; b[e] = pb[k + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
  float* o = part + (((long)(bl * max_tiles + blockIdx.y) * KS + ks) << 10);
#pragma unroll
  for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[r];     // C(row, col = lr) of the tile
}

// iou[p,q] = num / (sum soft_h[p] + sum hard_h[q] - num + 1) for p < q, blank rows p excluded; mi[q] = max_p iou (>= 0).
// Block per clip; the (p, q) pairs are spread over the threads, the slices of a tile are added in order, the per-q maximum goes
// through an integer atomicMax in LDS (non-negative floats order like their bit patterns: order-independent).
__global__ void __launch_bounds__(256)
mask_iou_kernel(const float* __restrict__ part, int KS, int max_tiles, const float* __restrict__ stats, ClipMeta meta,
                float* __restrict__ mi) {
  __shared__ unsigned best[256];
  const int bl = blockIdx.x;
  const int n = meta.n[bl], row0 = meta.row0[bl];
  best[threadIdx.x] = 0u;
  __syncthreads();
  const int nt = (n + 31) >> 5;
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
    const int p = idx / n, q = idx - p * n;
    if (p >= q || !(stats[(long)(row0 + p) * 5] > 0.f)) continue;          // blank p takes no part (:387-390)
    const int ti = p >> 5, tj = q >> 5;
    const int tile = ti * nt - ti * (ti - 1) / 2 + (tj - ti);
    const float* t = part + (((long)(bl * max_tiles + tile) * KS) << 10) + (p & 31) * 32 + (q & 31);
    float num = 0.f;
    for (int s = 0; s < KS; ++s) num += t[(long)s << 10];
    const float den = stats[(long)(row0 + p) * 5 + 3] + stats[(long)(row0 + q) * 5 + 4] - num;
    atomicMax(&best[q], __float_as_uint(num / (den + 1.f)));
  }
  __syncthreads();
  if ((int)threadIdx.x < n) mi[row0 + threadIdx.x] = __uint_as_float(best[threadIdx.x]);
}

// a15, last step (:408-419) per clip: class scores x (1 - max IoU) x mask quality, best class, drop blank / suppressed rows,
// keep the max(#(score > thr), 1) best of the rest (score order, equal scores by rank).  Writes, for the j-th selected row of
// clip b, its instance row into sel[row0 + j] and (score, label, class scores [K], embedding [C]) into out[row0 + j]; n_sel[b].
__global__ void __launch_bounds__(256)
clip_finalize_kernel(const float* __restrict__ cls, const float* __restrict__ emb, const int* __restrict__ kept,
                     const float* __restrict__ stats, const float* __restrict__ mi, int Q, int K, int C, float thr, int c0,
                     ClipMeta meta, int* __restrict__ sel, int* __restrict__ n_sel, float* __restrict__ out) {
  __shared__ float key[256];
  __shared__ int val[256];
  __shared__ int n_alive, n_above;
  const int bl = blockIdx.x, b = c0 + bl, tid = threadIdx.x;
  const int n = meta.n[bl], row0 = meta.row0[bl];
  if (tid == 0) { n_alive = 0; n_above = 0; }
  __syncthreads();
  float sc = -1.f, f1 = 0.f, f2 = 0.f;
  int lab = 0;
  bool alive = false;
  if (tid < n) {
    const long row = row0 + tid;
    const float m = mi[row];
    const float quality = stats[row * 5 + 1] / (stats[row * 5 + 2] + 1e-6f);
    f1 = 1.f - m; f2 = quality;
    const float* c = cls + ((long)b * Q + kept[(long)b * Q + tid]) * K;
    float best = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const float v = (c[k] * f1) * f2;
      if (v > best) { best = v; lab = k; }
    }
    alive = stats[row * 5] > 0.f && m < 0.5f;
    if (alive) { atomicAdd(&n_alive, 1); if (best > thr) atomicAdd(&n_above, 1); }
    sc = alive ? best : -1.f;
  }
  key[tid] = tid < n ? sc : -INFINITY;
  val[tid] = tid;
  __syncthreads();
  bitonic256(key, val, tid);
  int k_sel = n_above > 1 ? n_above : 1;
  if (k_sel > n_alive) k_sel = n_alive;
  if (tid == 0) n_sel[b] = k_sel;
  // row `tid` learns its output position
  __shared__ int posn[256];
  posn[val[tid]] = tid;
  __syncthreads();
  if (tid < n) {
    const int j = posn[tid];
    if (j < k_sel) {
      sel[row0 + j] = row0 + tid;
      float* o = out + (long)(row0 + j) * (2 + K + C);
      o[0] = sc; o[1] = (float)lab;
      const long src = (long)b * Q + kept[(long)b * Q + tid];
      for (int k = 0; k < K; ++k) o[2 + k] = (cls[src * K + k] * f1) * f2;
      for (int c = 0; c < C; ++c) o[2 + K + c] = emb[src * C + c];
    }
  }
}

#define MDQE_NMS_KS 16
// floats of the NMS partial-tile workspace for a batch whose largest clip keeps n_max instances (<= 64 clips per launch)
extern "C" long mdqe_nms_workspace_floats(int n_max) {
  const long nt = (n_max + 31) / 32;
  return 64L * (nt * (nt + 1) / 2) * MDQE_NMS_KS * 1024;
}

extern "C" long mdqe_dyn_mask_workspace_floats(int n_rows, int T, int H, int W) {
  const long P = (long)T * H * W;
  return (long)n_rows * ((P + 255) / 256) * 5;
}

// The batch form of inference_clip's mask part.  Host arrays (B entries): row0 (first instance row of the clip), n_keep,
// f0 (first frame of the clip in `feats`).  coef [B, Q, M]; kept [B, Q] (device, from mdqe_clip_select_f32); feats
// [frames, H, W, M]; logits [n_rows, T, H, W]; soft_h [n_rows, Ph]; hard_t: B * Ph * ceil(Q/32) words; part: workspace of
// mdqe_dyn_mask_workspace_floats(n_rows, ..) floats; stats [n_rows, 5]; mi [n_rows].  Ph = ceil(T/t_step)*(H/2)*(W/2),
// t_step = 2 when T >= 5.
extern "C" int mdqe_dyn_mask_nms_f32(const float* coef, const int* kept, const float* feats, int B, int Q, int M, int T, int H,
                                     int W, const int* row0_host, const int* n_host, const int* f0_host, float* logits,
                                     float* soft_h, float* hard_h, float* part, float* gram, float* stats, float* mi, void* stream) {
  MDQE_REQUIRE(B >= 0 && Q > 0 && Q <= 256 && M > 0 && M <= 32 && M % 4 == 0 && T > 0 && H > 1 && W > 1 && W / 2 <= 1024);
  MDQE_REQUIRE((long)T * H * W < (1L << 30));
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(row0_host); MDQE_CHECK_PTR(n_host); MDQE_CHECK_PTR(f0_host);
  long n_rows = 0;
  for (int b = 0; b < B; ++b) { MDQE_REQUIRE(n_host[b] >= 0 && n_host[b] <= Q && row0_host[b] == n_rows); n_rows += n_host[b]; }
  if (n_rows == 0) return MDQE_OK;
  MDQE_CHECK_PTR(coef); MDQE_CHECK_PTR(kept); MDQE_CHECK_PTR(feats); MDQE_CHECK_PTR(logits); MDQE_CHECK_PTR(soft_h); MDQE_CHECK_PTR(hard_h); MDQE_CHECK_PTR(gram);
  MDQE_CHECK_PTR(part); MDQE_CHECK_PTR(stats); MDQE_CHECK_PTR(mi);
  hipStream_t st = (hipStream_t)stream;
  mdqe_clear_error();
  const int t_step = T >= 5 ? 2 : 1;
  const long P = (long)T * H * W;
  const long Ph = (long)((T + t_step - 1) / t_step) * (H / 2) * (W / 2);
  const int n_tiles = (int)((P + 255) / 256);
  if (hipMemsetAsync(mi, 0, (size_t)n_rows * sizeof(float), st) != hipSuccess) return MDQE_ELAUNCH;
  for (int c0 = 0; c0 < B; c0 += 64) {
    const int nc = B - c0 < 64 ? B - c0 : 64;
    ClipMeta meta;
    int nmax = 0;
    for (int i = 0; i < 64; ++i) {
      meta.row0[i] = i < nc ? row0_host[c0 + i] : 0; meta.n[i] = i < nc ? n_host[c0 + i] : 0; meta.f0[i] = i < nc ? f0_host[c0 + i] : 0;
      if (meta.n[i] > nmax) nmax = meta.n[i];
    }
    if (nmax == 0) continue;
    const int Mp = M == 24 ? 24 : 32;
    const size_t sm = ((size_t)nmax * Mp + (size_t)nmax * 20) * sizeof(float);
    if (M == 32)
      hipLaunchKernelGGL(dyn_mask_kernel<32>, dim3(n_tiles, nc), dim3(256), sm, st, coef, kept, feats, Q, T, H, W, t_step, c0, meta,
                         logits, soft_h, hard_h, part, n_tiles, M);
    else if (M == 24)
      hipLaunchKernelGGL(dyn_mask_kernel<24>, dim3(n_tiles, nc), dim3(256), sm, st, coef, kept, feats, Q, T, H, W, t_step, c0, meta,
                         logits, soft_h, hard_h, part, n_tiles, M);
    else
      hipLaunchKernelGGL(dyn_mask_kernel<32>, dim3(n_tiles, nc), dim3(256), sm, st, coef, kept, feats, Q, T, H, W, t_step, c0, meta,
                         logits, soft_h, hard_h, part, n_tiles, M);
  }
  hipLaunchKernelGGL(mask_stats_reduce_kernel, dim3((unsigned)((n_rows * 5 + 255) / 256)), dim3(256), 0, st, part, n_tiles, n_rows, stats);
  for (int c0 = 0; c0 < B; c0 += 64) {
    const int nc = B - c0 < 64 ? B - c0 : 64;
    ClipMeta meta;
    int rows = 0;
    for (int i = 0; i < 64; ++i) {
      meta.row0[i] = i < nc ? row0_host[c0 + i] : 0; meta.n[i] = i < nc ? n_host[c0 + i] : 0; meta.f0[i] = 0;
      rows += meta.n[i];
    }
    if (rows == 0) continue;
    int nmax = 0;
    for (int i = 0; i < nc; ++i) if (meta.n[i] > nmax) nmax = meta.n[i];
    if (nmax < 2) continue;
    const int ntm = (nmax + 31) / 32, max_tiles = ntm * (ntm + 1) / 2;
    hipLaunchKernelGGL(mask_gram_kernel, dim3(MDQE_NMS_KS, max_tiles, nc), dim3(64), 0, st, soft_h, hard_h, (int)Ph, MDQE_NMS_KS, meta,
                       max_tiles, gram);
    hipLaunchKernelGGL(mask_iou_kernel, dim3(nc), dim3(256), 0, st, gram, MDQE_NMS_KS, max_tiles, stats, meta, mi);
  }
  return mdqe_launch_status();
}

// out [n_rows, 2+K+C] (only the first n_sel[b] rows of each clip are written), sel [n_rows], n_sel [B]
extern "C" int mdqe_clip_finalize_f32(const float* cls, const float* emb, const int* kept, const float* stats, const float* mi, int B,
                                      int Q, int K, int C, float thr, const int* row0_host, const int* n_host, int* sel, int* n_sel,
                                      float* out, void* stream) {
  MDQE_REQUIRE(B >= 0 && Q > 0 && Q <= 256 && K > 0 && C > 0);
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(cls); MDQE_CHECK_PTR(emb); MDQE_CHECK_PTR(kept); MDQE_CHECK_PTR(row0_host); MDQE_CHECK_PTR(n_host); MDQE_CHECK_PTR(sel);
  MDQE_CHECK_PTR(n_sel); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  for (int c0 = 0; c0 < B; c0 += 64) {
    const int nc = B - c0 < 64 ? B - c0 : 64;
    ClipMeta meta;
    for (int i = 0; i < 64; ++i) { meta.row0[i] = i < nc ? row0_host[c0 + i] : 0; meta.n[i] = i < nc ? n_host[c0 + i] : 0; meta.f0[i] = 0; }
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(nc), dim3(256), 0, (hipStream_t)stream, cls, emb, kept, stats, mi, Q, K, C, thr, c0, meta,
                       sel, n_sel, out);
  }
  return mdqe_launch_status();
}

// out[i, :] = src[idx[i], :]  (rows of `len` floats, len % 4 == 0)
__global__ void __launch_bounds__(256)
rows_gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, long len, float* __restrict__ out) {
  const float* s = src + (long)idx[blockIdx.y] * len;
  float* d = out + (long)blockIdx.y * len;
  for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < len; e += (long)gridDim.x * blockDim.x * 4)
    *reinterpret_cast<f32x4*>(d + e) = *reinterpret_cast<const f32x4*>(s + e);
}

extern "C" int mdqe_rows_gather_f32(const float* src, const int* idx_dev, int n, long len, float* out, void* stream) {
  MDQE_REQUIRE(n >= 0 && len >= 0 && len % 4 == 0);
  if (n == 0 || len == 0) return MDQE_OK;
  MDQE_CHECK_PTR(src); MDQE_CHECK_PTR(idx_dev); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  long bx = (len / 4 + 255) / 256; if (bx > 32) bx = 32;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)bx, n), dim3(256), 0, (hipStream_t)stream, src, idx_dev, len, out);
  return mdqe_launch_status();
}
