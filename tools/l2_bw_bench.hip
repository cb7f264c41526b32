// L2 -> CU streaming-rate microbenchmark (gfx950): how many bytes per clock can ONE CU pull from an L2-resident
// region, as a function of the load form (buffer_load..lds = LDS-DMA, or global_load to VGPRs), waves per CU and loads
// in flight per wave.  Sets the per-CU ceiling the implicit-GEMM k-loop runs against (igemm roofline note).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_bw_bench tools/l2_bw_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// MODE 0: buffer_load_dwordx4 ... lds; MODE 1: global_load_dwordx4 -> VGPR
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void stream(const char* __restrict__ base, long long region_bytes, int shared, int iters,
                                                unsigned* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, nw = blockDim.x >> 6;
  const char* my = base + (shared ? 0 : (long long)blockIdx.x * region_bytes);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(my), 0, (int)region_bytes, 0x00020000);
  const unsigned chunk = 1024u * nw * DEPTH;  // bytes per block per round
  u4 acc = {0, 0, 0, 0};
  unsigned off = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned o = off + (unsigned)(d * nw + wave) * 1024u + (unsigned)(tid & 63) * 16u;
      if constexpr (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((d * nw + wave) & 63) * 1024), 16, o, 0, 0, 0);
      } else {
        const u4 v = *reinterpret_cast<const u4*>(my + o);
        acc ^= v;
      }
    }
    if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off += chunk;
    if (off + chunk > (unsigned)region_bytes) off = 0;
  }
  if constexpr (MODE == 0) {
    __syncthreads();
    acc[0] = *reinterpret_cast<unsigned*>(smem + (tid & 255) * 4);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
#endif
}

template <int MODE, int DEPTH>
void run(const char* buf, long long region, int shared, int waves, int blocks, unsigned* sink, const char* label) {
  const int iters = 2000;
  if (1024LL * waves * DEPTH > region) return;  // one round must fit the region
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto k = stream<MODE, DEPTH>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 65536, 0, buf, region, shared, 50, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 65536, 0, buf, region, shared, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * iters * 1024.0 * waves * DEPTH;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-28s waves=%2d depth=%2d blocks=%3d %s: %8.1f GB/s total, %6.1f GB/s per CU (%5.1f B/clk @2.4GHz)\n", label, waves, DEPTH,
         blocks, shared ? "shared " : "private", gbs, gbs / blocks, gbs / blocks / 2.4);
}

int main() {
  const int blocks = 256;
  const long long region = 64 << 10;  // per-CU region: 64 KB x 32 CUs = 2 MB per XCD -> L2 resident (4 MB / XCD)
  char* buf;
  unsigned* sink;
  CHECK(hipMalloc(&buf, region * blocks));
  CHECK(hipMemset(buf, 1, region * blocks));
  CHECK(hipMalloc(&sink, 4));
  for (int shared = 0; shared < 2; ++shared) {
    for (int waves : {4, 8, 16}) {
      run<0, 2>(buf, region, shared, waves, blocks, sink, "buffer_load_dwordx4 lds");
      run<0, 4>(buf, region, shared, waves, blocks, sink, "buffer_load_dwordx4 lds");
      run<0, 8>(buf, region, shared, waves, blocks, sink, "buffer_load_dwordx4 lds");
      run<1, 4>(buf, region, shared, waves, blocks, sink, "global_load_dwordx4 vgpr");
      run<1, 8>(buf, region, shared, waves, blocks, sink, "global_load_dwordx4 vgpr");
    }
  }
  // one block only: no contention from the other CUs of the XCD
  run<0, 8>(buf, region, 0, 8, 1, sink, "lds, single CU");
  run<1, 8>(buf, region, 0, 8, 1, sink, "vgpr, single CU");
  run<0, 8>(buf, region, 0, 8, 32, sink, "lds, 32 blocks");
  run<0, 8>(buf, region, 0, 8, 128, sink, "lds, 128 blocks");
  return 0;
}
