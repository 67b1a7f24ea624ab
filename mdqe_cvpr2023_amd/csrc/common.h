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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float mdqe_act(float x, int act) {
  switch (act) {
    case MDQE_ACT_RELU: return x > 0.f ? x : 0.f;
    case MDQE_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case MDQE_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
    case MDQE_ACT_TANH: return tanhf(x);
    default: return x;
  }
}
