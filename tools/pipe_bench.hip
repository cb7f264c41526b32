// Pipe-overlap microbenchmark (gfx950): what one SIMD does with the instruction mix of the d = 40 attention loop.
//   * issue cost of v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 / v_exp_f32 / v_cvt_pk_f16_f32 / v_max3_f32 alone,
//   * the SAME wave interleaving MFMAs with exponentials (the pinned order of attn_kernel_v3's block A),
//   * TWO / THREE waves on one SIMD with different roles (one MFMA-only, the others exp-only): do the matrix pipe and the
//     transcendental unit overlap ACROSS waves of a SIMD, or do they serialise?
// Every workgroup is 256 * W threads (W waves per SIMD); role of a wave = wave / 4 (waves 0-3 land on SIMDs 0-3, 4-7 again, ...).
// Times are s_memtime ticks per loop iteration, per wave role, from one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pipe_bench tools/pipe_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// 256 * exp2 of two scores on the PLAIN vector ALU, packed fp16: round-to-integer by a magic add, cubic on the fraction, the
// integer part shifted into the exponent field of a scale factor (scores below -23 flush to zero)
__device__ __forceinline__ unsigned exp2_pair(float s0, float s1) {
  h2 t;
  {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(s0), "v"(s1));
    const unsigned lo = 0xcdc0cdc0u;   // -23.0 | -23.0
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(r), "v"(lo));
    __builtin_memcpy(&t, &r, 4);
  }
  const h2 M = {(_Float16)1559.0f, (_Float16)1559.0f};
  h2 r = t + M;
  asm volatile("" : "+v"(r));
  const h2 nf = r - M;
  const h2 f = t - nf;
  const h2 c3 = {(_Float16)0.05518f, (_Float16)0.05518f}, c2 = {(_Float16)0.2426f, (_Float16)0.2426f},
           c1 = {(_Float16)0.6934f, (_Float16)0.6934f}, c0 = {(_Float16)1.0f, (_Float16)1.0f};
  h2 p = __builtin_elementwise_fma(c3, f, c2);
  p = __builtin_elementwise_fma(p, f, c1);
  p = __builtin_elementwise_fma(p, f, c0);
  us2 rb;
  __builtin_memcpy(&rb, &r, 4);
  rb = rb << (unsigned short)10;
  h2 sc;
  __builtin_memcpy(&sc, &rb, 4);
  p = p * sc;
  unsigned out;
  __builtin_memcpy(&out, &p, 4);
  return out;
}

// ROLE codes: 8 = 16 packed-fp16 polynomial exp2 pairs (plain VALU only), 9 = block A with them instead of v_exp_f32, 10 = whole
// tile with them, 11 = block A with half the pairs on v_exp_f32 and half on the polynomial; 0 idle, 1 = 16 MFMA 16x16x32 per iteration, 2 = 32 v_exp_f32 + 16 cvt_pk per iteration, 3 = block A (16 MFMA
// interleaved with 32 exp + 16 cvt_pk, as attn_kernel_v3 pins them), 4 = 16 MFMA 16x16x16, 5 = 32 v_exp_f32 only,
// 6 = 16 v_max3 + 16 cvt_pk (plain VALU), 7 = block A + block B (12 MFMA + 16 max3): a whole tile without memory
template <int ROLE>
__device__ __forceinline__ void body(int iters, f4 (&acc)[8], h8 (&a)[2], h8 (&b)[2], float (&x)[32], unsigned (&pk)[16], f16v (&big)[4]) {
  for (int it = 0; it < iters; ++it) {
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[(i >> 1) & 1], acc[i & 7], 0, 0, 0);
    } else if constexpr (ROLE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const h4 a4 = {a[i & 1][0], a[i & 1][1], a[i & 1][2], a[i & 1][3]};
        const h4 b4 = {b[(i >> 1) & 1][0], b[(i >> 1) & 1][1], b[(i >> 1) & 1][2], b[(i >> 1) & 1][3]};
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i & 7], 0, 0, 0);
      }
    } else if constexpr (ROLE == 2 || ROLE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p0 = __builtin_amdgcn_exp2f(x[2 * i]), p1 = __builtin_amdgcn_exp2f(x[2 * i + 1]);
        if constexpr (ROLE == 2) {
          unsigned r;
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(p0), "v"(p1));
          pk[i] = r;
          x[2 * i] = p0 * 0.5f - 1.0f;   // (keeps the chain alive and bounded; 2 extra plain VALU per pair)
          x[2 * i + 1] = p1 * 0.5f - 1.0f;
        } else {
          x[2 * i] = p0;
          x[2 * i + 1] = p1;
          asm volatile("" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]));
        }
      }
    } else if constexpr (ROLE == 6) {
      float m = x[0];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x[2 * i]), "v"(m));
        pk[i] = r;
      }
      x[0] = m * 0.999f;
    } else if constexpr (ROLE == 12) {
#pragma unroll
      for (int i = 0; i < 8; ++i) big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], big[i & 3], 0, 0, 0);
    } else if constexpr (ROLE == 13 || ROLE == 14) {
      // the same work as block A / the tile on the 32 x 32 x 16 MFMA (32 cycles each): 4 exp2 + 2 cvt_pk behind every MFMA
      constexpr int NM = ROLE == 13 ? 8 : 6;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], big[i & 3], 0, 0, 0);
        constexpr int PER = 32 / NM;   // exps per MFMA (5 + remainder for 6 MFMAs)
#pragma unroll
        for (int e = i * PER; e < (i == NM - 1 ? 32 : (i + 1) * PER); e += 2) {
          const float p0 = __builtin_amdgcn_exp2f(x[e]), p1 = __builtin_amdgcn_exp2f(x[e + 1]);
          unsigned r;
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(p0), "v"(p1));
          pk[e >> 1] = r;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (ROLE == 14) {
        float m = -1e30f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          h8 pb;
          __builtin_memcpy(&pb, &pk[(i & 3) * 4], 16);
          big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], pb, big[i & 3], 0, 0, 0);
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x[4 * i]), "v"(x[4 * i + 1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x[4 * i + 2]), "v"(x[4 * i + 3]));
          __builtin_amdgcn_sched_barrier(0);
        }
        x[0] = fminf(m, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[16 * i + r] = -fabsf(big[i][r]) * 1e-3f;
      }
    } else if constexpr (ROLE == 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        pk[i] = exp2_pair(x[2 * i], x[2 * i + 1]);
        x[2 * i] = x[2 * i] * 0.999f;
      }
    } else if constexpr (ROLE == 3 || ROLE == 7 || ROLE == 9 || ROLE == 10 || ROLE == 11) {
      // block A: MFMA i followed by its share (2 scores -> exp2, exp2, cvt_pk) -- exps read the PREVIOUS iteration's accumulators
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int s = i & 7;
        const float e0 = x[2 * i], e1 = x[2 * i + 1];
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[(i >> 1) & 1], acc[s], 0, 0, 0);
        if constexpr (ROLE == 9 || ROLE == 10 || ROLE == 11) {
          if (ROLE != 11 || (i & 1)) {
            pk[i] = exp2_pair(e0, e1);
          } else {
            const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
            unsigned r;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(p0), "v"(p1));
            pk[i] = r;
          }
        } else {
        const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(p0), "v"(p1));
        pk[i] = r;
        }
        if constexpr (ROLE == 3 || ROLE == 7) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (ROLE == 7 || ROLE == 10) {
        float m = -1e30f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          h8 pb;
          __builtin_memcpy(&pb, &pk[(i & 3) * 4], 16);
          acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], pb, acc[i & 7], 0, 0, 0);
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
          if (i < 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x[24 + 2 * i]), "v"(x[25 + 2 * i]));
          __builtin_amdgcn_sched_barrier(0);
        }
        x[0] = fminf(m, 0.f);
      }
      // next iteration's "scores": small negative numbers derived from the accumulators (2 plain VALU per 4 scores)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * i + r] = -fabsf(acc[i][r]) * 1e-3f;
      }
    }
  }
}

template <int R0, int R1, int R2>
__global__ void pipe_kernel(int iters, long long* ticks, float* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int role_slot = wave >> 2;   // 0: first four waves (one per SIMD), 1: next four, ...
  const int lane = threadIdx.x & 63;
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a[2], b[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (_Float16)(0.01f * ((lane + e + i) & 7));
      b[i][e] = (_Float16)(0.02f * ((lane * 3 + e + i) & 7) - 0.05f);
    }
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = -0.01f * ((lane + i) & 15);
  unsigned pk[16];
  for (int i = 0; i < 16; ++i) pk[i] = 0x3c003c00u;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  f16v big[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
  if (role_slot == 0) body<R0>(iters, acc, a, b, x, pk, big);
  else if (role_slot == 1) body<R1>(iters, acc, a, b, x, pk, big);
  else body<R2>(iters, acc, a, b, x, pk, big);
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 32; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += big[i][0] + big[i][5] + big[i][15];
  for (int i = 0; i < 16; ++i) s += (float)pk[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int R0, int R1, int R2>
void run(const char* label, int waves_per_simd, int iters, long long* d_ticks, float* d_sink) {
  const int nblk = 256, nthr = 256 * waves_per_simd;
  CHECK(hipMemset(d_ticks, 0, nblk * 16 * sizeof(long long)));
  hipLaunchKernelGGL((pipe_kernel<R0, R1, R2>), dim3(nblk), dim3(nthr), 0, 0, 100, d_ticks, d_sink);   // warm-up
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((pipe_kernel<R0, R1, R2>), dim3(nblk), dim3(nthr), 0, 0, iters, d_ticks, d_sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> t(nblk * 16);
  CHECK(hipMemcpy(t.data(), d_ticks, t.size() * sizeof(long long), hipMemcpyDeviceToHost));
  printf("%-58s", label);
  for (int slot = 0; slot < waves_per_simd; ++slot) {
    double sum = 0;
    for (int b = 0; b < nblk; ++b)
      for (int w = 0; w < 4; ++w) sum += (double)t[b * 16 + slot * 4 + w];
    printf("  slot%d %8.1f", slot, sum / (nblk * 4) / iters);
  }
  printf("   ticks / iteration;  kernel %7.1f ns / iteration\n", ms * 1e6 / iters);
  fflush(stdout);
}

int main() {
  long long* d_ticks;
  float* d_sink;
  CHECK(hipMalloc(&d_ticks, 256 * 16 * sizeof(long long)));
  CHECK(hipMalloc(&d_sink, 64));
  const int N = 4000;
  // s_memtime counts at a fixed 100 MHz on gfx9 parts (not the shader clock): report the ratio to the MFMA-only loop, whose
  // iteration is 16 MFMAs x 16 shader cycles by the counters of profiles/round2_attention_pmc_d40.txt
  run<1, 0, 0>("1 wave/SIMD: 16 x mfma 16x16x32", 1, N, d_ticks, d_sink);
  run<4, 0, 0>("1 wave/SIMD: 16 x mfma 16x16x16", 1, N, d_ticks, d_sink);
  run<5, 0, 0>("1 wave/SIMD: 32 x v_exp_f32", 1, N, d_ticks, d_sink);
  run<2, 0, 0>("1 wave/SIMD: 32 exp + 16 cvt_pk + 32 fma", 1, N, d_ticks, d_sink);
  run<6, 0, 0>("1 wave/SIMD: 16 max3 + 16 cvt_pk", 1, N, d_ticks, d_sink);
  run<3, 0, 0>("1 wave/SIMD: block A (16 mfma | 32 exp, 16 cvt)", 1, N, d_ticks, d_sink);
  run<7, 0, 0>("1 wave/SIMD: block A + block B (28 mfma)", 1, N, d_ticks, d_sink);
  run<1, 1, 0>("2 waves/SIMD: mfma + mfma", 2, N, d_ticks, d_sink);
  run<1, 5, 0>("2 waves/SIMD: mfma + 32 exp", 2, N, d_ticks, d_sink);
  run<1, 2, 0>("2 waves/SIMD: mfma + (32 exp, 16 cvt, 32 fma)", 2, N, d_ticks, d_sink);
  run<1, 6, 0>("2 waves/SIMD: mfma + (16 max3, 16 cvt)", 2, N, d_ticks, d_sink);
  run<5, 5, 0>("2 waves/SIMD: 32 exp + 32 exp", 2, N, d_ticks, d_sink);
  run<3, 3, 0>("2 waves/SIMD: block A + block A", 2, N, d_ticks, d_sink);
  run<7, 7, 0>("2 waves/SIMD: tile + tile", 2, N, d_ticks, d_sink);
  run<7, 7, 7>("3 waves/SIMD: tile + tile + tile", 3, N, d_ticks, d_sink);
  run<1, 5, 5>("3 waves/SIMD: mfma + 32 exp + 32 exp", 3, N, d_ticks, d_sink);
  run<3, 3, 3>("3 waves/SIMD: block A x 3", 3, N, d_ticks, d_sink);
  run<12, 0, 0>("1 wave/SIMD: 8 x mfma 32x32x16", 1, N, d_ticks, d_sink);
  run<13, 0, 0>("1 wave/SIMD: block A on 32x32x16 (8 mfma | 32 exp, 16 cvt)", 1, N, d_ticks, d_sink);
  run<14, 0, 0>("1 wave/SIMD: tile on 32x32x16 (14 mfma)", 1, N, d_ticks, d_sink);
  run<13, 13, 0>("2 waves/SIMD: block A on 32x32x16 x 2", 2, N, d_ticks, d_sink);
  run<14, 14, 0>("2 waves/SIMD: tile on 32x32x16 x 2", 2, N, d_ticks, d_sink);
  run<14, 14, 14>("3 waves/SIMD: tile on 32x32x16 x 3", 3, N, d_ticks, d_sink);
  run<12, 5, 0>("2 waves/SIMD: mfma 32x32x16 + 32 exp", 2, N, d_ticks, d_sink);
  run<12, 2, 0>("2 waves/SIMD: mfma 32x32x16 + (32 exp, 16 cvt, 32 fma)", 2, N, d_ticks, d_sink);
  run<8, 0, 0>("1 wave/SIMD: 16 polynomial exp2 pairs (plain VALU)", 1, N, d_ticks, d_sink);
  run<9, 0, 0>("1 wave/SIMD: block A, polynomial exp2", 1, N, d_ticks, d_sink);
  run<11, 0, 0>("1 wave/SIMD: block A, half v_exp half polynomial", 1, N, d_ticks, d_sink);
  run<10, 0, 0>("1 wave/SIMD: tile, polynomial exp2", 1, N, d_ticks, d_sink);
  run<1, 8, 0>("2 waves/SIMD: mfma + polynomial pairs", 2, N, d_ticks, d_sink);
  run<9, 9, 0>("2 waves/SIMD: block A poly x 2", 2, N, d_ticks, d_sink);
  run<10, 10, 0>("2 waves/SIMD: tile poly x 2", 2, N, d_ticks, d_sink);
  run<10, 10, 10>("3 waves/SIMD: tile poly x 3", 3, N, d_ticks, d_sink);
  run<11, 11, 11>("3 waves/SIMD: block A half/half x 3", 3, N, d_ticks, d_sink);
  run<9, 9, 9>("3 waves/SIMD: block A poly x 3", 3, N, d_ticks, d_sink);
  return 0;
}
