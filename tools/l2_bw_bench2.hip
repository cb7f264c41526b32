// L2 -> LDS (buffer_load_dwordx4 ... lds) rate per CU as a function of the ACCESS SHAPE of one wave-instruction:
// ROWB contiguous bytes per row (ROWB/16 lanes per row, 1024/ROWB rows per instruction), rows STRIDE bytes apart,
// optional XOR swizzle of the 16-byte chunk order inside a 128-byte row (what the igemm loader does).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_bw_bench2 tools/l2_bw_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ base, unsigned region_bytes, int rowb, unsigned stride, int swz,
                                               int iters, unsigned* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region_bytes, 0x00020000);
  const int lanes_per_row = rowb / 16, rows_per_instr = 1024 / rowb;
  const int r_in = lane / lanes_per_row;
  int c_in = lane % lanes_per_row;
  if (swz) c_in = (c_in & ~7) | ((c_in & 7) ^ (r_in & 7));
  // each block starts at a different row so that blocks do not all hit the same lines at once
  unsigned row = (blockIdx.x * 37u) % 64u * (unsigned)rows_per_instr * nw * DEPTH;
  const unsigned total_rows = region_bytes / stride;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      unsigned r = row + (unsigned)((d * nw + wave) * rows_per_instr + r_in);
      if (r >= total_rows) r -= total_rows;
      const unsigned o = r * stride + (unsigned)c_in * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((d * nw + wave) & 63) * 1024), 16, o, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    row += (unsigned)rows_per_instr * nw * DEPTH;
    if (row >= total_rows) row -= total_rows;
  }
  __syncthreads();
  acc = *reinterpret_cast<unsigned*>(smem + (tid & 255) * 4);
  if (acc == 0x12345678u) sink[0] = 1;
#endif
}

template <int DEPTH>
void run(const char* buf, unsigned region, int rowb, unsigned stride, int swz, int waves, unsigned* sink) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto k = stream<DEPTH>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 65536, 0, buf, region, rowb, stride, swz, 50, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 65536, 0, buf, region, rowb, stride, swz, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double gbs = (double)blocks * iters * 1024.0 * waves * DEPTH / (ms * 1e-3) / 1e9;
  printf("rowB=%4d stride=%6u swz=%d waves=%d depth=%d: %8.1f GB/s total, %6.1f GB/s per CU (%5.1f B/clk @2.4GHz)\n", rowb, stride, swz, waves,
         DEPTH, gbs, gbs / blocks, gbs / blocks / 2.4);
}

int main() {
  const unsigned region = 2u << 20;  // shared by all CUs, L2 resident in every XCD
  char* buf;
  unsigned* sink;
  CHECK(hipMalloc(&buf, region));
  CHECK(hipMemset(buf, 1, region));
  CHECK(hipMalloc(&sink, 4));
  for (int waves : {4, 8}) {
    run<8>(buf, region, 1024, 1024, 0, waves, sink);   // fully contiguous
    run<8>(buf, region, 128, 128, 0, waves, sink);     // 128-B rows, back to back (same bytes, row-shaped lanes)
    run<8>(buf, region, 128, 128, 1, waves, sink);     // + chunk swizzle
    run<8>(buf, region, 128, 640, 0, waves, sink);     // 128-B rows of a c=320 NHWC tensor
    run<8>(buf, region, 128, 1280, 0, waves, sink);    // c=640
    run<8>(buf, region, 128, 1280, 1, waves, sink);
    run<8>(buf, region, 128, 2560, 1, waves, sink);    // c=1280
    run<8>(buf, region, 128, 11520, 1, waves, sink);   // weight rows, K=5760
    run<8>(buf, region, 256, 1280, 0, waves, sink);    // 256-B rows (BK=128)
    run<8>(buf, region, 256, 11520, 0, waves, sink);
    run<8>(buf, region, 512, 1280, 0, waves, sink);    // 512-B rows (BK=256)
    run<8>(buf, region, 512, 11520, 0, waves, sink);
    run<4>(buf, region, 128, 1280, 1, waves, sink);
    run<4>(buf, region, 256, 1280, 0, waves, sink);
  }
  return 0;
}
