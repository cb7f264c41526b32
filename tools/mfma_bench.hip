// MFMA issue-rate microbenchmark (gfx950): v_mfma_f32_16x16x32_f16 throughput per CU with W waves per SIMD, with and
// without the LDS operand reads of a 64x64 (4x4 fragment) wave tile, plus the occupancy the runtime reports for a
// 256-thread block with 64 KB of dynamic LDS.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_bench tools/mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (operands in registers); MODE 1: per k-step 8 ds_read_b128 + 16 MFMA (the igemm compute_tile shape)
template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 7);
  __syncthreads();
  f4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const h8*>(smem + (i * 16 + lr) * 128 + (lg << 4));
    b[i] = *reinterpret_cast<const h8*>(smem + 8192 + (i * 16 + lr) * 128 + (lg << 4));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (MODE == 1) {
        const int chunk = ks * 4 + lg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (wave & 1) * 64 + i * 16 + lr;
          a[i] = *reinterpret_cast<const h8*>(smem + ((it & 1) << 15) + row * 128 + ((chunk ^ (row & 7)) << 4));
          const int row2 = (wave >> 1) * 64 + i * 16 + lr;
          b[i] = *reinterpret_cast<const h8*>(smem + ((it & 1) << 15) + 16384 + row2 * 128 + ((chunk ^ (row2 & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i], a[j], acc[i][j], 0, 0, 0);
    }
  }
  f4 s = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = s[0];
}

// 32x32x16 f16 chains with NACC independent accumulators per wave, operands in registers: the shape the guide's 2.38-2.50 PFLOP/s
// ceiling was measured with (MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA").  Separates what the matrix pipe can sustain from what
// the 16x16x32 loops above reach.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_loop(int iters, float* out) {
  const int lane = threadIdx.x & 63;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * ((lane + e) & 7));
    b[e] = (_Float16)(0.002f * ((lane * 3 + e) & 7));
  }
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run32(int blocks_per_cu, float* out) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  auto k = mfma32_loop<NACC>;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 10, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 * iters * 8.0 * NACC * 32768.0;
  printf("v_mfma_f32_32x32x16_f16, %d independent accumulators/wave, blocks/CU=%d (waves/SIMD=%d): %7.1f TFLOP/s\n", NACC, blocks_per_cu,
         blocks_per_cu, flops / (ms * 1e-3) / 1e12);
}

template <int MODE>
void run(int blocks_per_cu, float* out, const char* label) {
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  auto k = mfma_loop<MODE>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, 10, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 * iters * 32 * 16384.0;
  const double clk_per_tile = ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu);
  printf("%-34s blocks/CU=%d: %7.1f TFLOP/s, %6.0f clk per 128x128x64 tile per CU (ideal 512)\n", label, blocks_per_cu, flops / (ms * 1e-3) / 1e12,
         clk_per_tile);
}

int main() {
  int dev = 0, v = 0;
  CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
  printf("MaxSharedMemoryPerMultiprocessor = %d\n", v);
  CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
  printf("MaxSharedMemoryPerBlock = %d\n", v);
  CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, dev));
  printf("ClockRate kHz = %d\n", v);
  int nb = 0;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfma_loop<1>, 256, 65536));
  printf("occupancy(256 thr, 64 KB dyn LDS) = %d blocks/CU\n", nb);
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfma_loop<1>, 256, 32768));
  printf("occupancy(256 thr, 32 KB dyn LDS) = %d blocks/CU\n", nb);
  float* out;
  CHECK(hipMalloc(&out, 4));
  for (int b : {1, 2}) {
    run<0>(b, out, "MFMA only (regs)");
    run<1>(b, out, "8 ds_read_b128 + 16 MFMA per k-step");
  }
  for (int b : {1, 2}) {
    run32<1>(b, out);
    run32<2>(b, out);
    run32<4>(b, out);
  }
  return 0;
}
