// Does v_pk_fma_f32 with a crossing op_sel (low result half reads the HIGH register of a source pair) mis-execute on gfx950?
// Round 5 found the compiler's packed form of the folded-LayerNorm transform giving run-to-run different results inside md_igemm
// (DESIGN.md section 2); this probe isolates the instruction: checker waves execute the packed instruction (inline asm) and the two
// scalar fmas it stands for on the same registers, millions of times, and count disagreements -- alone, beside waves that hammer the
// matrix pipe of the same SIMDs, with vector loads in flight, and behind v_rsq / v_cndmask producers.  GPU box only:
//   hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_probe.hip -o tools/bin/pk_hazard_probe && tools/bin/pk_hazard_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// FORM 0: op_sel:[0,1,0] + neg (the transform's form)   1: plain packed fma (no op_sel)   2: op_sel_hi:[1,0,1] (broadcast LOW)
//      3: v_pk_mul_f32 with an SGPR-pair source and op_sel:[1,0] (how the transform scales its row sums)
__device__ __forceinline__ float uni(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

template <int FORM>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, float* __restrict__ sink, int iters,
                                             int mfma_waves, int loads, int producers) {
  const int wave = threadIdx.x >> 6;
  if (wave < mfma_waves) {   // neighbours: keep the matrix pipe of this SIMD busy
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = (_Float16)(0.01f * (threadIdx.x + e));
      b[e] = (_Float16)(0.02f * (threadIdx.x - e));
    }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters * 2; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
    return;
  }
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f2 s1 = {in[(t * 6 + 0) & 0xfffff], in[(t * 6 + 1) & 0xfffff]};
  f2 mu = {in[(t * 6 + 2) & 0xfffff], in[(t * 6 + 3) & 0xfffff]};
  f2 acc = {in[(t * 6 + 4) & 0xfffff], in[(t * 6 + 5) & 0xfffff]};
  unsigned cnt = 0;
  float keep = 0.f;
  for (int i = 0; i < iters; ++i) {
    float ld = 0.f;
    if (loads) ld = in[(t * 7 + i * 64) & 0xfffff];           // a vector load in flight around the packed instruction
    if (producers) {                                           // the transform's producers: v_rsq + v_cndmask feeding the pair
      const float v = fabsf(mu.x) + 1e-5f;
      const float r = __builtin_amdgcn_rsqf(v);
      mu.x = v < 1e-3f ? r * 0.5f : r;
      mu.y = mu.y * 1.0000001f + 1e-9f;
    }
    f2 r2;
    if (FORM == 0)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r2) : "v"(s1), "v"(mu), "v"(acc));
    else if (FORM == 1)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r2) : "v"(s1), "v"(mu), "v"(acc));
    else if (FORM == 2)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r2) : "v"(s1), "v"(mu), "v"(acc));
    else {   // the transform's scalar-pair multiply: both halves take the HIGH register of an SGPR pair (eps | 1 / K in the kernel)
      f2 sp = {uni(1e-5f + 0.f * (float)i), uni(0.003125f)};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r2) : "s"(sp), "v"(acc));
    }
    float e0, e1;
    if (FORM == 0) {
      e0 = __builtin_fmaf(-s1.x, mu.y, acc.x);
      e1 = __builtin_fmaf(-s1.y, mu.y, acc.y);
    } else if (FORM == 1) {
      e0 = __builtin_fmaf(-s1.x, mu.x, acc.x);
      e1 = __builtin_fmaf(-s1.y, mu.y, acc.y);
    } else if (FORM == 2) {
      e0 = __builtin_fmaf(-s1.x, mu.x, acc.x);
      e1 = __builtin_fmaf(-s1.y, mu.x, acc.y);
    } else {
      e0 = 0.003125f * acc.x;
      e1 = 0.003125f * acc.y;
    }
    cnt += (r2.x != e0) || (r2.y != e1);
    acc.x += 0.25f;                                            // (new operands every trip)
    acc.y -= 0.125f;
    s1.x = s1.x * 1.0000002f;
    keep += ld + r2.x;
  }
  if (keep == 12345.f) sink[1] = keep;
  if (cnt) atomicAdd(bad, cnt);
}

int main() {
  const int N = 1 << 20;
  std::vector<float> h(N);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) & 0xffff) / 65536.0f * 4.f - 2.f;
  }
  float *din, *sink;
  unsigned* bad;
  hipMalloc(&din, N * 4);
  hipMalloc(&sink, 64);
  hipMalloc(&bad, 4);
  hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
  const int iters = 4000, grid = 2048;
  const char* names[4] = {"op_sel:[0,1,0] (low half reads the HIGH register)", "plain packed fma", "op_sel_hi:[1,0,1] (high half reads the LOW register)",
                          "v_pk_mul_f32 with an SGPR pair, op_sel:[1,0]"};
  for (int form = 0; form < 4; ++form)
    for (int mw = 0; mw <= 4; mw += 4)
      for (int loads = 0; loads < 2; ++loads)
        for (int prod = 0; prod < 2; ++prod) {
          hipMemset(bad, 0, 4);
          if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, din, bad, sink, iters, mw, loads, prod);
          if (form == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, din, bad, sink, iters, mw, loads, prod);
          if (form == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, din, bad, sink, iters, mw, loads, prod);
          if (form == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, din, bad, sink, iters, mw, loads, prod);
          hipDeviceSynchronize();
          unsigned b = 0;
          hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
          const double total = (double)grid * (512 - 64 * mw) * iters;
          printf("PKPROBE %-52s | MFMA neighbour waves %d | loads in flight %d | rsq/cndmask producers %d | %u of %.3g packed results differ from the scalar fmas\n",
                 names[form], mw, loads, prod, b, total);
        }
  return 0;
}
