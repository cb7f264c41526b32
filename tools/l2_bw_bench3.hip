// L2 -> LDS rate of the igemm LOOP SHAPE (no MFMA): per iteration every wave issues DEPTH buffer_load..lds (1 KiB each),
// then s_waitcnt vmcnt(0) [+ s_barrier], like the 2-stage k-loop; BPC workgroups of 4 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_bw_bench3 tools/l2_bw_bench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH, int BARRIER, int STAGES>
__global__ __launch_bounds__(256) void loop(const char* __restrict__ base, unsigned region_bytes, int iters, unsigned* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region_bytes, 0x00020000);
  unsigned off = (blockIdx.x * 7919u) % 512u * 4096u;
  constexpr unsigned TILE = 4096u * DEPTH;  // bytes per block per iteration
  auto fetch = [&](int stage) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned o = off + (unsigned)(d * 4 + wave) * 1024u + (unsigned)lane * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * TILE + (d * 4 + wave) * 1024), 16, o, 0, 0, 0);
    }
    off += TILE;
    if (off + TILE > region_bytes) off = 0;
  };
  if constexpr (STAGES == 2) {
    fetch(0);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (BARRIER) __builtin_amdgcn_s_barrier();
      fetch((it + 1) & 1);
      // stand-in for the compute phase: read the landed tile from LDS (16 ds_read_b128 per wave for DEPTH 4)
      unsigned acc = 0;
#pragma unroll
      for (int d = 0; d < DEPTH * 2; ++d) acc ^= *reinterpret_cast<const unsigned*>(smem + (it & 1) * TILE + ((d * 256 + tid) * 16) % TILE);
      if (acc == 0x12345678u) sink[1] = acc;
    }
  } else {  // 3 stages: two tiles in flight, counted wait
    fetch(0);
    fetch(1);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
      if (BARRIER) __builtin_amdgcn_s_barrier();
      fetch((it + 2) % 3);
      unsigned acc = 0;
#pragma unroll
      for (int d = 0; d < DEPTH * 2; ++d) acc ^= *reinterpret_cast<const unsigned*>(smem + (it % 3) * TILE + ((d * 256 + tid) * 16) % TILE);
      if (acc == 0x12345678u) sink[1] = acc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink[0] == 0x7fffffffu) sink[2] = *reinterpret_cast<unsigned*>(smem + tid * 4);
#endif
}

template <int DEPTH, int BARRIER, int STAGES>
void run(const char* buf, unsigned region, int bpc, unsigned* sink) {
  const int iters = 3000, blocks = 256 * bpc;
  auto k = loop<DEPTH, BARRIER, STAGES>;
  const int lds = STAGES * 4096 * DEPTH;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, buf, region, 50, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, buf, region, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double gbs = (double)blocks * iters * 4096.0 * DEPTH / (ms * 1e-3) / 1e9;
  printf("tile %2d KB/iter stages=%d barrier=%d blocks/CU=%d: %6.1f GB/s per CU (%5.1f B/clk), %5.0f clk per iteration per block\n", 4 * DEPTH, STAGES,
         BARRIER, bpc, gbs / 256, gbs / 256 / 2.4, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
  const unsigned region = 2u << 20;
  char* buf;
  unsigned* sink;
  CHECK(hipMalloc(&buf, region));
  CHECK(hipMemset(buf, 1, region));
  CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(sink, 0, 16));
  for (int bpc : {1, 2, 3, 4}) {
    run<4, 0, 2>(buf, region, bpc, sink);   // 64x64 tile: 16 KB per iteration
    run<4, 1, 2>(buf, region, bpc, sink);
    run<4, 1, 3>(buf, region, bpc, sink);
  }
  for (int bpc : {1, 2}) {
    run<8, 0, 2>(buf, region, bpc, sink);   // 128x128 tile: 32 KB per iteration
    run<8, 1, 2>(buf, region, bpc, sink);
    run<8, 1, 3>(buf, region, bpc, sink);
  }
  return 0;
}
