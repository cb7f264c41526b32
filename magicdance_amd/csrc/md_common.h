// Internal helpers shared by the gfx950 kernels of libmagicdance_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/magicdance_hip.h"

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MD_WAVE 64

namespace md {

extern int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return MD_ERR_HIP;
}

#define MD_HIP_CHECK(expr)                        \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return md::hip_fail(_e); \
  } while (0)

// ---- per-family profiling (md_prof_*) ---------------------------------------------------------------
struct ProfScope {
  int family;
  hipStream_t stream;
  hipEvent_t start;
  bool active;
  ProfScope(int family, hipStream_t stream, double flops, double bytes, const char* tag = nullptr);
  ~ProfScope();
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace md
