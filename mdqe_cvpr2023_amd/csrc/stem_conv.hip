// ResNet stem as ONE kernel: normalise + zero-pad + 7x7/s2 conv (3 -> 64) + folded FrozenBN bias + ReLU, straight from the
// uint8 / fp32 CHW frames to the NHWC activation (mdqe/mdqe.py:176-178,480 + detectron2 BasicStem).  Replaces the im2col
// buffer (39 MB per 360p frame, written and read back) of mdqe_stem_im2col_f32 + GEMM.
//
// Persistent blocks of 4 waves; a tile = 8 x 16 output pixels x 64 channels.  The 21 x 37 x 3 input patch of a tile is
// normalised once into LDS ([row][col*3+c], pitch 112 words: the 4 lane groups of an A read land on disjoint banks) and
// the whole weight matrix ([k][64], k = kh*22 + kw*3 + c, row entry 21 zero) stays in LDS for the life of the block, so
// every MFMA operand is one ds_read_b32 with an immediate offset.  v_mfma_f32_32x32x2_f32: exact fp32 products.
// A wave owns 2 x 16 pixels and all 64 channels (2 accumulator tiles); 77 K-steps of 2.
#include "common.h"

namespace {
constexpr int TH = 8, TW = 16;               // output pixels per tile
constexpr int PR = 2 * (TH - 1) + 7;         // 21 patch rows
constexpr int PCOL = 2 * (TW - 1) + 7;       // 37 patch columns
constexpr int PITCH = 112;                   // words per patch row (>= 37*3; 2*PITCH = 32 mod 64)
constexpr int KROW = 22;                     // k entries per filter row (21 real + 1 zero)
constexpr int KTOT = 7 * KROW;               // 154
constexpr int W_WORDS = KTOT * 64;           // 9856 words = 39424 B
constexpr int P_WORDS = PR * PITCH;          // 2352 words = 9408 B
}  // namespace

template <typename T>
__global__ void __launch_bounds__(256, 3)
stem_conv_kernel(const T* __restrict__ frames, long frame_stride, int h, int w, int OH, int OW, int tiles_x, int tiles_y,
                 long ntiles, float m0, float m1, float m2, float s0, float s1, float s2, const float* __restrict__ wk,
                 const float* __restrict__ bias, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* sW = lds;
  float* sP = lds + W_WORDS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  for (int i = tid; i < W_WORDS / 4; i += 256)
    reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(wk)[i];
  for (int i = tid; i < P_WORDS; i += 256) sP[i] = 0.f;   // word 111 of a row is read (against a zero weight) but never written
  const float b0 = bias[lr], b1 = bias[32 + lr];

  // this lane's A row: pixel (wave*2 + (lr>>4), lr&15) of the tile, k parity lh
  const float* aP = sP + (wave * 2 + (lr >> 4)) * 2 * PITCH + (lr & 15) * 6 + lh;
  const float* bP = sW + lh * 64 + lr;

  // The patch of the NEXT tile is fetched into registers (raw values, 10 per thread) before the MFMA phase of the current one
  // and normalised into LDS after it: the global-load latency hides behind the 154 MFMAs instead of sitting between two
  // barriers (r01 PMC: matrix pipe busy 0.58).  A second LDS patch would cost the third block per CU.
  constexpr int NE = (3 * PR * PCOL + 255) / 256;    // 10
  T raw[NE];
  unsigned ok = 0;
  auto fetch = [&](long tile) __attribute__((always_inline)) {
    const int tx = (int)(tile % tiles_x);
    const long tt = tile / tiles_x;
    const int ty = (int)(tt % tiles_y), img = (int)(tt / tiles_y);
    const int iy0 = ty * TH * 2 - 3, ix0 = tx * TW * 2 - 3;
    const T* src = frames + (long)img * frame_stride;
    ok = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + i * 256;
      const int cc = e % PCOL, t = e / PCOL, r = t % PR, c = t / PR;
      const int iy = iy0 + r, ix = ix0 + cc;
      raw[i] = (T)0;
      if (e < 3 * PR * PCOL && iy >= 0 && iy < h && ix >= 0 && ix < w) {   // the canvas beyond the real image is zero in normalised space
        raw[i] = src[((long)c * h + iy) * w + ix];
        ok |= 1u << i;
      }
    }
  };
  if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = (int)(tile % tiles_x);
    const long tt = tile / tiles_x;
    const int ty = (int)(tt % tiles_y), img = (int)(tt / tiles_y);
    __syncthreads();                                   // previous tile's patch is no longer read (and sW is complete)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + i * 256;
      if (e < 3 * PR * PCOL) {
        const int cc = e % PCOL, t = e / PCOL, r = t % PR, c = t / PR;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        sP[r * PITCH + cc * 3 + c] = (ok >> i) & 1u ? ((float)raw[i] - mean) / sd : 0.f;
      }
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);      // in flight during this tile's MFMAs

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int j = 0; j < KROW; j += 2) {
        const float a = aP[kh * PITCH + j];
        const float wb0 = bP[(kh * KROW + j) * 64];
        const float wb1 = bP[(kh * KROW + j) * 64 + 32];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb1, acc1, 0, 0, 0);
      }
    }

    // C layout: register r holds pixel row (r&3) + 8*(r>>2) + 4*lh of the wave's 32, channel lr (+32 for acc1):
    // 32 lanes write one 128-B line of a pixel
    const int oy0 = ty * TH + wave * 2, ox0 = tx * TW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
      if (oy < OH && ox < OW) {
        float* o = out + (((long)img * OH + oy) * OW + ox) * 64;
        o[lr] = fmaxf(acc0[r] + b0, 0.f);
        o[32 + lr] = fmaxf(acc1[r] + b1, 0.f);
      }
    }
  }
}

// frames: NI images CHW (uint8 or fp32), h x w; canvas Hp x Wp (even).  wk: device [154*64] (k-major, k = kh*22 + kw*3 + c,
// entry kh*22+21 zero); bias: device [64]; out: [NI, Hp/2, Wp/2, 64] fp32.
extern "C" int mdqe_stem_conv_f32(const void* frames, int is_u8, long frame_stride, int NI, int h, int w, int Hp, int Wp,
                                  const float* mean3_host, const float* std3_host, const float* wk, const float* bias,
                                  float* out, void* stream) {
  MDQE_REQUIRE(NI >= 0 && h > 0 && w > 0 && Hp >= h && Wp >= w && Hp % 2 == 0 && Wp % 2 == 0);
  if (NI == 0) return MDQE_OK;
  MDQE_CHECK_PTR(frames); MDQE_CHECK_PTR(out); MDQE_CHECK_PTR(mean3_host); MDQE_CHECK_PTR(std3_host);
  MDQE_CHECK_PTR(wk); MDQE_CHECK_PTR(bias);
  mdqe_clear_error();
  const int OH = Hp / 2, OW = Wp / 2;
  const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
  const long ntiles = (long)NI * tiles_x * tiles_y;
  const long nb = ntiles < 768 ? ntiles : 768;         // 3 blocks per CU, each walks its share of the tiles
  const size_t smem = (size_t)(W_WORDS + P_WORDS) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (is_u8)
    hipLaunchKernelGGL((stem_conv_kernel<unsigned char>), dim3((unsigned)nb), dim3(256), smem, st, (const unsigned char*)frames,
                       frame_stride, h, w, OH, OW, tiles_x, tiles_y, ntiles, mean3_host[0], mean3_host[1], mean3_host[2],
                       std3_host[0], std3_host[1], std3_host[2], wk, bias, out);
  else
    hipLaunchKernelGGL((stem_conv_kernel<float>), dim3((unsigned)nb), dim3(256), smem, st, (const float*)frames, frame_stride,
                       h, w, OH, OW, tiles_x, tiles_y, ntiles, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0],
                       std3_host[1], std3_host[2], wk, bias, out);
  return mdqe_launch_status();
}
