// Shared helpers for the gfx950 kernels of libmdqe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mdqe_hip.h"

#define MDQE_CHECK_PTR(p) do { if ((p) == nullptr) return MDQE_ENULL; } while (0)
#define MDQE_REQUIRE(c) do { if (!(c)) return MDQE_EINVAL; } while (0)

// hipGetLastError() is per-thread and sticky across *other* libraries' calls (torch's event queries
// leave hipErrorNotReady behind), so entry points clear it before launching and read it after.
static inline void mdqe_clear_error() { (void)hipGetLastError(); }
static inline int mdqe_launch_status() {
  return hipGetLastError() == hipSuccess ? MDQE_OK : MDQE_ELAUNCH;
}

// Dynamic LDS beyond 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize on EVERY kernel function that asks for it; the
// attribute is per function, so the "already done" record is keyed by the function pointer (template instantiations of one
// kernel share a pointer TYPE, not a pointer) and guarded: entry points are called from several host threads.
#include <mutex>
static inline hipError_t mdqe_allow_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static const void* fn[64];
  static int cap[64];
  static int n = 0;
  std::lock_guard<std::mutex> g(mu);
  int i = 0;
  for (; i < n; ++i) if (fn[i] == kern) break;
  if (i < n && cap[i] >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && i < 64) { fn[i] = kern; cap[i] = bytes; if (i == n) ++n; }
  return e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// erf-GELU without the libm erff branchy path: 0.5*(1 - erf(z)) = exp(-z^2) * t*(a1 + t*(a2 + ..)) / 2, t = 1/(1 + p z)
// (Abramowitz-Stegun 7.1.26, |err(erf)| <= 1.5e-7), written so that the negative side is a plain product (no
// cancellation).  Max |error| against the exact function 3.4e-7 over [-12, 12] -- below torch's own fp32 GELU rounding.
__device__ __forceinline__ float mdqe_gelu(float x) {
  // No contraction inside: inlined into two differently shaped epilogues of ONE launch (gemm_k16.hip: interior tiles / edge tiles), the
  // optimiser fused `x - x * h` into an fma in one of them and not in the other -- 1 ulp, and a frame's bits then depend on which tile
  // of a pass it falls into.  Every fma below is written out.
#pragma clang fp contract(off)
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float q = __builtin_fmaf(t, 0.5307027145f, -0.7265760135f);
  q = __builtin_fmaf(t, q, 0.7107068705f);
  q = __builtin_fmaf(t, q, -0.142248368f);
  q = __builtin_fmaf(t, q, 0.127414796f);
  const float h = q * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);   // = (1 - erf(z)) / 2
  const float xh = x * h;
  return x >= 0.f ? x - xh : xh;
}

// The activation of a float4 of a GEMM / conv epilogue (defined below mdqe_act).  ReLU and GELU -- every activation on the per-frame path -- are their own inlined
// loops; the heads' sigmoid / tanh go through ONE rolled loop.  (Round 4: `mdqe_act`'s four-way switch, tanhf included, used to be
// inlined per ELEMENT into every unrolled copy of the epilogue: the 128x128 GEMM kernel was 28 000 instructions long and even a ReLU
// epilogue cost 12 % of the launch -- 107 against 120 TF on the FFN1 shape -- in instruction fetch.)  Same functions, same bits.
__device__ __forceinline__ float mdqe_act(float x, int act) {
  switch (act) {
    case MDQE_ACT_RELU: return x > 0.f ? x : 0.f;
    case MDQE_ACT_GELU: return mdqe_gelu(x);
    case MDQE_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
    case MDQE_ACT_TANH: return tanhf(x);
    default: return x;
  }
}

template <typename V4, typename ColOk>
__device__ __forceinline__ void mdqe_act4(V4& v, int act, ColOk col_ok) {
  if (act == MDQE_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col_ok(e)) v[e] = v[e] > 0.f ? v[e] : 0.f;
  } else if (act == MDQE_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col_ok(e)) v[e] = mdqe_gelu(v[e]);
  } else if (act != MDQE_ACT_NONE) {
#pragma unroll 1
    for (int e = 0; e < 4; ++e) {                      // (rolled; the vector rotates through the loop: no dynamic register index)
      float x = v[0];
      if (col_ok(e)) x = mdqe_act(x, act);
      v = V4{v[1], v[2], v[3], x};
    }
  }
}
