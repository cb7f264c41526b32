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
// exact-erf GELU (F.gelu default, ldm/modules/attention.py:50-60).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e.
// <= 1e-7 |x| on the GELU -- four orders below the fp16 rounding of the result): one v_rcp + one v_exp + 8 FMAs instead of the
// branchy ocml erff (both of its range branches execute on a diverged wave).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erf_abs = fmaf(-p * t, e, 1.0f);   // erf(|x| / sqrt 2)
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace md
