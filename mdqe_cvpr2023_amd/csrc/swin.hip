// SwinV2 backbone kernels for gfx950 (SURVEY.md §8 a4'; mdqe/backbone/swin_transformer_v2.py).
// The dense work (qkv / proj / MLP / patch-merge reduction / patch embedding) runs on the GEMM kernels; this
// file holds the data-movement and attention pieces, all channels-last with 16-B lane accesses.
#include "common.h"

// ---- patch embedding input: (x-mean)/std, zero pad to (Hp,Wp), 4x4/s4 im2col with k = c*16 + kh*4 + kw ------------
// (PatchEmbed.forward :466-479 after MDQE.preprocess_image / ImageList padding).  out [NI*Hp/4*Wp/4, 48].
template <typename T>
__global__ void __launch_bounds__(256)
patch4_im2col_kernel(const T* __restrict__ frames, long frame_stride, int NI, int h, int w, int OH, int OW, float m0, float m1,
                     float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const long total = (long)NI * OH * OW * 12;            // 12 float4 per row: (c, kh) pairs, 4 kw each
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % 12);
    const long m = i / 12;
    const int ow = (int)(m % OW); const long t = m / OW; const int oh = (int)(t % OH); const int img = (int)(t / OH);
    const int c = q / 4, kh = q % 4;
    const int ih = oh * 4 + kh;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ih < h) {
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int iw = ow * 4 + kw;
        if (iw < w) v[kw] = ((float)frames[(long)img * frame_stride + ((long)c * h + ih) * w + iw] - mean) / sd;
      }
    }
    *reinterpret_cast<f32x4*>(out + m * 48 + q * 4) = v;
  }
}

extern "C" int mdqe_patch4_im2col_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                                      const float* mean3_host, const float* std3_host, float* out, void* stream) {
  MDQE_REQUIRE(NI >= 0 && h > 0 && w > 0 && Hp >= h && Wp >= w && Hp % 4 == 0 && Wp % 4 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(frames); MDQE_CHECK_PTR(out); MDQE_CHECK_PTR(mean3_host); MDQE_CHECK_PTR(std3_host);
  mdqe_clear_error();
  const int OH = Hp / 4, OW = Wp / 4;
  const long total = (long)NI * OH * OW * 12;
  long nb = (total + 255) / 256; if (nb > 256 * 16) nb = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
  if (is_u8)
    hipLaunchKernelGGL((patch4_im2col_kernel<unsigned char>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned char*)frames,
                       frame_stride, NI, h, w, OH, OW, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                       std3_host[2], out);
  else
    hipLaunchKernelGGL((patch4_im2col_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)frames, frame_stride, NI,
                       h, w, OH, OW, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], out);
  return mdqe_launch_status();
}

// ---- shifted-window partition / reverse (SwinTransformerBlock.forward :252-282) --------------------------------------
// Window row r = ((b*nWy + wy)*nWx + wx)*ws*ws + iy*ws + ix  <->  pixel ((wy*ws+iy+shift) % Hp, (wx*ws+ix+shift) % Wp)
// of the map zero-padded to (Hp,Wp) (pad first, then roll by -shift).
// mode 0 (gather): win[r,:] = inside ? x[b,y,x,:] : 0
// mode 1 (scatter): y[b,y,x,:] = shortcut[b,y,x,:] + win[r,:]   for inside pixels
__global__ void __launch_bounds__(256)
swin_window_kernel(const float* __restrict__ src, const float* __restrict__ shortcut, float* __restrict__ dst, int B, int H, int W,
                   int C, int ws, int shift, int mode) {
  const int c4n = C / 4;
  const int nWy = (H + ws - 1) / ws, nWx = (W + ws - 1) / ws;
  const int Hp = nWy * ws, Wp = nWx * ws;
  const long total = (long)B * Hp * Wp * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long r = i / c4n;                                       // window-ordered row
    const int ix = (int)(r % ws); long t = r / ws;
    const int iy = (int)(t % ws); t /= ws;
    const int wx = (int)(t % nWx); t /= nWx;
    const int wy = (int)(t % nWy); const int b = (int)(t / nWy);
    const int y = (wy * ws + iy + shift) % Hp, x = (wx * ws + ix + shift) % Wp;
    const bool inside = y < H && x < W;
    const long pix = (((long)b * H + y) * W + x) * C + c4 * 4;
    if (mode == 0) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (inside) v = *reinterpret_cast<const f32x4*>(src + pix);
      *reinterpret_cast<f32x4*>(dst + r * C + c4 * 4) = v;
    } else if (inside) {
      *reinterpret_cast<f32x4*>(dst + pix) = *reinterpret_cast<const f32x4*>(shortcut + pix) + *reinterpret_cast<const f32x4*>(src + r * C + c4 * 4);
    }
  }
}

extern "C" int mdqe_swin_window_f32(const float* src, const float* shortcut, float* dst, int B, int H, int W, int C, int ws,
                                    int shift, int mode, void* stream) {
  MDQE_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ws > 0 && shift >= 0 && shift < ws && (mode == 0 || mode == 1));
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(src); MDQE_CHECK_PTR(dst);
  if (mode == 1) MDQE_CHECK_PTR(shortcut);
  mdqe_clear_error();
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  const long total = (long)B * Hp * Wp * (C / 4);
  long nb = (total + 255) / 256; if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(swin_window_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, shortcut, dst, B, H, W, C, ws,
                     shift, mode);
  return mdqe_launch_status();
}

// ---- cosine window attention (WindowAttention.forward :147-186) ------------------------------------------------------
// qkv [nWin*N, 3C] (q | k | v, heads of D inside).  attn = normalize(q).normalize(k)^T * scale[h] + bias[h,i,j]
// (+ mask[win % nW, i, j]) -> softmax -> @ v.  One block per (window, head); K/V in LDS (K rows pre-normalised),
// one thread per query row, online softmax.  N <= 256, D in {8,16,24,32}.
template <int D>
__global__ void __launch_bounds__(256)
window_attn_kernel(const float* __restrict__ qkv, long ld, float* __restrict__ o, long ldo, int N, int C, int nh,
                   const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ mask, int nW) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sK = sm;
  float* sV = sm + N * D;
  const int win = blockIdx.x / nh, h = blockIdx.x % nh;
  const long row0 = (long)win * N;
  const int r = threadIdx.x;
  if (r < N) {
    const float* kp = qkv + (row0 + r) * ld + C + h * D;
    const float* vp = qkv + (row0 + r) * ld + 2 * C + h * D;
    float kk[D], ss = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(kp + c);
      kk[c] = t[0]; kk[c + 1] = t[1]; kk[c + 2] = t[2]; kk[c + 3] = t[3];
      ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
      *reinterpret_cast<f32x4*>(sV + r * D + c) = *reinterpret_cast<const f32x4*>(vp + c);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);             // F.normalize eps
#pragma unroll
    for (int c = 0; c < D; ++c) sK[r * D + c] = kk[c] * inv;
  }
  __syncthreads();
  if (r >= N) return;
  float q[D], acc[D];
  {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ld + h * D + c);
      q[c] = t[0]; q[c + 1] = t[1]; q[c + 2] = t[2]; q[c + 3] = t[3];
      ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
      acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.f;
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] *= inv;
  }
  const float sc = scale[h];
  const float* bp = bias + ((long)h * N + r) * N;
  const float* mp = mask != nullptr ? mask + ((long)(win % nW) * N + r) * N : nullptr;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < N; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 kk = *reinterpret_cast<const f32x4*>(sK + j * D + c);
      s += q[c] * kk[0] + q[c + 1] * kk[1] + q[c + 2] * kk[2] + q[c + 3] * kk[3];
    }
    s = s * sc + bp[j];
    if (mp != nullptr) s += mp[j];
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn);
    const float p = expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + j * D + c);
      acc[c] = acc[c] * corr + p * vv[0]; acc[c + 1] = acc[c + 1] * corr + p * vv[1];
      acc[c + 2] = acc[c + 2] * corr + p * vv[2]; acc[c + 3] = acc[c + 3] * corr + p * vv[3];
    }
    m = mn;
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < D; c += 4)
    *reinterpret_cast<f32x4*>(o + (row0 + r) * ldo + h * D + c) = f32x4{acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv};
}

// MFMA form (head dim 32, N <= 192): one block of three waves per (window, head); normalised K and V of the window in LDS
// (rows padded to 36 floats); wave w owns the 16-query row tiles w, w+3, ...  Both products run on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32 like the scalar form):
//   S^T tile (16 keys x 16 queries) = Kn_tile . Qn_tile^T, the head dimension walked as d = 8*(lane>>4) + t, t = 0..7;
//   the accumulator then holds, for query lane&15, the keys j0 + 4*(lane>>4) + r -- exactly the A-operand layout of the
//   second product if ITS k index is taken as that same key order (B = V[key][d] is read from LDS in any order), so the
//   probabilities never leave their registers;  O tile (16 queries x 16 d) += P . V.
// Softmax statistics per query: 36 scores in-lane, then two cross-lane steps over the four 16-lane groups.
template <int NWV>
__global__ void __launch_bounds__(64 * NWV)
window_attn_mfma_kernel(const float* __restrict__ qkv, long ld, float* __restrict__ o, long ldo, int N, int C, int nh,
                        const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ mask, int nW) {
  constexpr int D = 32, LDK = 36, MAXT = 12;         // up to 12 key tiles (N <= 192)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NT = (N + 15) / 16, NP = NT * 16;
  float* sK = sm;                                    // [NP][36] normalised keys (rows >= N zero)
  float* sV = sm + NP * LDK;                         // [NP][36]
  const int win = blockIdx.x / nh, h = blockIdx.x % nh;
  const long row0 = (long)win * N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, g = lane >> 4;
  for (int r = tid; r < NP; r += 64 * NWV) {
    f32x4 kk[8], vv[8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      kk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[c] = kk[c];
      if (r < N) {
        kk[c] = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ld + C + h * D + c * 4);
        vv[c] = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ld + 2 * C + h * D + c * 4);
      }
      ss += kk[c][0] * kk[c][0] + kk[c][1] * kk[c][1] + kk[c][2] * kk[c][2] + kk[c][3] * kk[c][3];
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);              // F.normalize eps
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      *reinterpret_cast<f32x4*>(sK + r * LDK + c * 4) = kk[c] * inv;
      *reinterpret_cast<f32x4*>(sV + r * LDK + c * 4) = vv[c];
    }
  }
  __syncthreads();
  const float L2E = 1.4426950408889634f;
  const float sc = scale[h] * L2E;
  for (int rt = wave; rt < NT; rt += NWV) {
    const int i0 = rt * 16;
    const int qi = min(i0 + lc, N - 1);                              // padded query rows repeat the last one (never stored)
    // Q fragment: d = 8g .. 8g+7 of query qi, normalised and pre-scaled by logit_scale * log2(e)
    float q[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qkv + (row0 + qi) * ld + h * D + 8 * g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(qkv + (row0 + qi) * ld + h * D + 8 * g + 4);
      float ss = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
      ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
      const float inv = sc / fmaxf(sqrtf(ss), 1e-12f);
      q[0] = a[0] * inv; q[1] = a[1] * inv; q[2] = a[2] * inv; q[3] = a[3] * inv;
      q[4] = b[0] * inv; q[5] = b[1] * inv; q[6] = b[2] * inv; q[7] = b[3] * inv;
    }
    const float* bp = bias + ((long)h * N + qi) * N;
    const float* mp = mask != nullptr ? mask + ((long)(win % nW) * N + qi) * N : nullptr;
    f32x4 sT[MAXT];
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < MAXT; ++jt) {
      if (jt < NT) {
        const float* kr = sK + (jt * 16 + lc) * LDK + 8 * g;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(kr), kb = *reinterpret_cast<const f32x4*>(kr + 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], q[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[t], q[4 + t], acc, 0, 0, 0);
        const int j = jt * 16 + 4 * g;                               // this lane's four keys j .. j+3 (N % 4 == 0)
        if (j < N) {
          f32x4 bb = *reinterpret_cast<const f32x4*>(bp + j);
          if (mp != nullptr) bb += *reinterpret_cast<const f32x4*>(mp + j);
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc[r] = __builtin_fmaf(bb[r], L2E, acc[r]); mx = fmaxf(mx, acc[r]); }
        } else {
          acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        sT[jt] = acc;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    float lsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MAXT; ++jt)
      if (jt < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float pv = __builtin_amdgcn_exp2f(sT[jt][r] - mx); sT[jt][r] = pv; lsum += pv; }
      }
    lsum += __shfl_xor(lsum, 16); lsum += __shfl_xor(lsum, 32);
    const float linv = 1.f / lsum;                                   // of query lc, replicated over the four lane groups
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0, p0 = o0, p1 = o0;      // O[query 4g+r][d = lc] and [d = 16 + lc]; even / odd key
#pragma unroll                                                       // tiles accumulate apart: four independent MFMA chains
    for (int jt = 0; jt < MAXT; ++jt)
      if (jt < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vr = sV + (jt * 16 + 4 * g + r) * LDK + lc;
          if (jt & 1) {
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], vr[0], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], vr[16], p1, 0, 0, 0);
          } else {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], vr[0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[jt][r], vr[16], o1, 0, 0, 0);
          }
        }
      }
    o0 += p0; o1 += p1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float li = __shfl(linv, 4 * g + r);                      // 1/sum of query 4g+r lives in lanes with lc == 4g+r
      const int qrow = i0 + 4 * g + r;
      if (qrow < N) {
        float* op = o + (row0 + qrow) * ldo + h * D + lc;
        op[0] = o0[r] * li;
        op[16] = o1[r] * li;
      }
    }
  }
}

static int g_window_attn_variant = 1;   // 1: MFMA form where it applies, 0: scalar form everywhere (tools/ A/B)
extern "C" int mdqe_debug_window_attn_variant(int v) { g_window_attn_variant = v; return MDQE_OK; }

extern "C" int mdqe_window_attn_f32(const float* qkv, long ld, float* o, long ldo, int n_windows, int N, int C, int nh,
                                    const float* scale, const float* bias, const float* mask, int nW, void* stream) {
  MDQE_REQUIRE(n_windows >= 0 && N > 0 && N <= 256 && nh > 0 && C % nh == 0 && ld % 4 == 0 && ldo % 4 == 0 && nW > 0);
  const int D = C / nh;
  MDQE_REQUIRE(D == 32 || D == 24 || D == 16 || D == 8);
  if (n_windows == 0) return MDQE_OK;
  MDQE_CHECK_PTR(qkv); MDQE_CHECK_PTR(o); MDQE_CHECK_PTR(scale); MDQE_CHECK_PTR(bias);
  mdqe_clear_error();
  hipStream_t st = (hipStream_t)stream;
  if (D == 32 && N <= 192 && N % 4 == 0 && g_window_attn_variant != 0) {
    const size_t smem2 = (size_t)2 * ((N + 15) / 16 * 16) * 36 * sizeof(float);
    // 9 row tiles of 16 queries (N = 144) over NWV waves; g_window_attn_variant: 1 = the measured best, 2 -> 3 waves, 3 -> 5, 4 -> 9
#define LW(NW_) hipLaunchKernelGGL((window_attn_mfma_kernel<NW_>), dim3(n_windows * nh), dim3(64 * NW_), smem2, st, qkv, ld, o, ldo, N, C, nh, scale, bias, mask, nW)
    const int nwv = g_window_attn_variant == 2 ? 3 : g_window_attn_variant == 3 ? 5 : g_window_attn_variant == 4 ? 9 : 3;
    if (nwv == 5) LW(5); else if (nwv == 9) LW(9); else LW(3);
#undef LW
    return mdqe_launch_status();
  }
  const size_t smem = (size_t)2 * N * D * sizeof(float);
#define L(DD) do { (void)hipFuncSetAttribute((const void*)window_attn_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((window_attn_kernel<DD>), dim3(n_windows * nh), dim3(256), smem, st, qkv, ld, o, ldo, N, C, nh, scale, bias, mask, nW); } while (0)
  if (D == 32) L(32); else if (D == 24) L(24); else if (D == 16) L(16); else L(8);
#undef L
  return mdqe_launch_status();
}

// ---- patch merging gather (PatchMerging.forward :311-335): [B,H,W,C] -> [B*H2*W2, 4C] = (x0|x1|x2|x3) ----------------
__global__ void __launch_bounds__(256)
patch_merge_gather_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C) {
  const int c4n = C / 4;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long total = (long)B * H2 * W2 * 4 * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n); long t = i / c4n;
    const int part = (int)(t % 4); t /= 4;                   // 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1)  (dy, dx)
    const int x2 = (int)(t % W2); t /= W2; const int y2 = (int)(t % H2); const int b = (int)(t / H2);
    const int y = 2 * y2 + (part & 1), xx = 2 * x2 + (part >> 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H && xx < W) v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + y) * W + xx) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(out + (((long)b * H2 + y2) * W2 + x2) * 4 * C + part * C + c4 * 4) = v;
  }
}

extern "C" int mdqe_patch_merge_gather_f32(const float* x, float* out, int B, int H, int W, int C, void* stream) {
  MDQE_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  if (B == 0) return MDQE_OK;
  MDQE_CHECK_PTR(x); MDQE_CHECK_PTR(out);
  mdqe_clear_error();
  const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
  long nb = (total + 255) / 256; if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(patch_merge_gather_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C);
  return mdqe_launch_status();
}
