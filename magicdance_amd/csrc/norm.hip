// GroupNorm(+SiLU) and LayerNorm for NHWC fp16 activations on gfx950: HBM/L2-bound streaming kernels, fp32
// statistics, 16-byte vector access, deterministic reductions (no atomics).
//
// GroupNorm = two launches:
//   gn_stats : grid (pixel chunks, groups, batch); a block reads its group's channel slice of 1024 pixels
//              (two-source channel concat resolved per channel pair) and writes one (sum, sumsq) partial.
//   gn_apply : grid (item chunks, batch); a block folds the partials of its sample into per-channel
//              scale/shift tables in LDS (y = x*scale[c] + shift[c]), then streams 16-byte chunks, applies SiLU.
// Reference arithmetic replaced: GroupNorm32 (util.py:252-254) + SiLU in ResBlock in_layers/out_layers and the
// UNet head (openaimodel.py:221-225,245-252,746-750); Normalize eps 1e-6 (attention.py:89-90); nn.LayerNorm
// (attention.py:270-272).
#include "md_common.h"

namespace {

constexpr int GN_PIX = 1024;     // pixels per stats block
constexpr int GN_ITEMS = 2048;   // 16-byte chunks per apply block

struct GnArgs {
  const half_t* x0;
  const half_t* x1;
  int c0, c1, c, cpg, groups;
  int batch, hw, nchunks;
  float eps;
  const float* gamma;
  const float* beta;
  int silu;
  half_t* out;
  float* ws;  // [batch][groups][nchunks][2]
};

__global__ __launch_bounds__(256) void gn_stats(const GnArgs g) {
  const int chunk = blockIdx.x, grp = blockIdx.y, b = blockIdx.z;
  const int p_begin = chunk * GN_PIX, p_end = min(g.hw, p_begin + GN_PIX);
  const int cbeg = grp * g.cpg;
  const int pairs = g.cpg >> 1;
  float s = 0.f, ss = 0.f;
  for (int p = p_begin + threadIdx.x; p < p_end; p += 256) {
    const long long pix = (long long)b * g.hw + p;
    for (int i = 0; i < pairs; ++i) {
      const int c = cbeg + 2 * i;
      const h2 v = (c < g.c0) ? *reinterpret_cast<const h2*>(g.x0 + pix * g.c0 + c)
                              : *reinterpret_cast<const h2*>(g.x1 + pix * g.c1 + (c - g.c0));
      const float a = (float)v[0], d = (float)v[1];
      s += a + d;
      ss += a * a + d * d;
    }
  }
  s = md::wave_sum(s);
  ss = md::wave_sum(ss);
  __shared__ float red[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave] = s;
    red[4 + wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = g.ws + (((long long)b * g.groups + grp) * g.nchunks + chunk) * 2;
    o[0] = red[0] + red[1] + red[2] + red[3];
    o[1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(256) void gn_apply(const GnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // scale[c], shift[c], then mean/rstd[groups]
  float* scale = tab;
  float* shift = tab + g.c;
  float* mean = tab + 2 * g.c;
  float* rstd = mean + g.groups;
  const int b = blockIdx.y;
  if (threadIdx.x < g.groups) {
    const float* pw = g.ws + ((long long)b * g.groups + threadIdx.x) * g.nchunks * 2;
    float s = 0.f, ss = 0.f;
    for (int i = 0; i < g.nchunks; ++i) {
      s += pw[2 * i];
      ss += pw[2 * i + 1];
    }
    const float inv_n = 1.0f / ((float)g.hw * (float)g.cpg);
    const float mu = s * inv_n;
    const float var = fmaxf(ss * inv_n - mu * mu, 0.f);
    mean[threadIdx.x] = mu;
    rstd[threadIdx.x] = rsqrtf(var + g.eps);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < g.c; c += 256) {
    const int grp = c / g.cpg;
    const float sc = rstd[grp] * g.gamma[c];
    scale[c] = sc;
    shift[c] = g.beta[c] - mean[grp] * sc;
  }
  __syncthreads();
  const int ch8 = g.c >> 3;
  const long long total = (long long)g.hw * ch8;
  const long long begin = (long long)blockIdx.x * GN_ITEMS;
  const long long end = begin + GN_ITEMS < total ? begin + GN_ITEMS : total;
  for (long long it = begin + threadIdx.x; it < end; it += 256) {
    const int p = (int)(it / ch8);
    const int c = (int)(it - (long long)p * ch8) * 8;
    const long long pix = (long long)b * g.hw + p;
    const h8 v = (c < g.c0) ? *reinterpret_cast<const h8*>(g.x0 + pix * g.c0 + c)
                            : *reinterpret_cast<const h8*>(g.x1 + pix * g.c1 + (c - g.c0));
    const f4 s0 = *reinterpret_cast<const f4*>(scale + c), s1 = *reinterpret_cast<const f4*>(scale + c + 4);
    const f4 h0 = *reinterpret_cast<const f4*>(shift + c), h1 = *reinterpret_cast<const f4*>(shift + c + 4);
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sc = i < 4 ? s0[i] : s1[i - 4], sh = i < 4 ? h0[i] : h1[i - 4];
      float y = (float)v[i] * sc + sh;
      if (g.silu) y = md::silu_f(y);
      o[i] = (half_t)y;
    }
    *reinterpret_cast<h8*>(g.out + pix * g.c + c) = o;
  }
}

// LayerNorm: one wave per row, row kept in registers (c <= 2048), two-pass statistics in fp32.
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half_t* __restrict__ out,
                                                        int rows, int c, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int ch8 = c >> 3;
  const half_t* xr = x + (long long)row * c;
  h8 v[4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = lane + 64 * j;
    if (i < ch8) {
      v[j] = *reinterpret_cast<const h8*>(xr + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[j][e];
    }
  }
  const float mu = md::wave_sum(s) / (float)c;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = lane + 64 * j;
    if (i < ch8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[j][e] - mu;
        ss += d * d;
      }
    }
  }
  const float rs = rsqrtf(md::wave_sum(ss) / (float)c + eps);
  half_t* orow = out + (long long)row * c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = lane + 64 * j;
    if (i < ch8) {
      const f4 g0 = *reinterpret_cast<const f4*>(gamma + i * 8), g1 = *reinterpret_cast<const f4*>(gamma + i * 8 + 4);
      const f4 b0 = *reinterpret_cast<const f4*>(beta + i * 8), b1 = *reinterpret_cast<const f4*>(beta + i * 8 + 4);
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gm = e < 4 ? g0[e] : g1[e - 4], bt = e < 4 ? b0[e] : b1[e - 4];
        o[e] = (half_t)(((float)v[j][e] - mu) * rs * gm + bt);
      }
      *reinterpret_cast<h8*>(orow + i * 8) = o;
    }
  }
}

}  // namespace

extern "C" int64_t md_groupnorm_workspace_bytes(int32_t batch, int32_t hw, int32_t groups) {
  const int64_t nchunks = (hw + GN_PIX - 1) / GN_PIX;
  return (int64_t)batch * groups * nchunks * 2 * sizeof(float);
}

extern "C" int md_groupnorm(const md_groupnorm_params* p, void* stream) {
  if (!p || !p->x0 || !p->gamma || !p->beta || !p->out || !p->ws) return MD_ERR_BAD_ARG;
  const int c = p->c0 + p->c1;
  if (p->c0 <= 0 || (p->c0 & 7) || p->c1 < 0 || (p->c1 & 7) || ((p->c1 > 0) != (p->x1 != nullptr))) return MD_ERR_BAD_ARG;
  if (p->groups <= 0 || p->groups > 256 || c % p->groups || ((c / p->groups) & 1)) return MD_ERR_UNSUPPORTED;
  if (p->batch <= 0 || p->hw <= 0) return MD_ERR_BAD_ARG;
  if (p->ws_bytes < md_groupnorm_workspace_bytes(p->batch, p->hw, p->groups)) return MD_ERR_WORKSPACE;
  GnArgs g;
  g.x0 = (const half_t*)p->x0;
  g.x1 = (const half_t*)p->x1;
  g.c0 = p->c0;
  g.c1 = p->c1;
  g.c = c;
  g.groups = p->groups;
  g.cpg = c / p->groups;
  g.batch = p->batch;
  g.hw = p->hw;
  g.nchunks = (p->hw + GN_PIX - 1) / GN_PIX;
  g.eps = p->eps;
  g.gamma = p->gamma;
  g.beta = p->beta;
  g.silu = p->silu;
  g.out = (half_t*)p->out;
  g.ws = (float*)p->ws;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_NORM, s, 0.0, (double)p->batch * p->hw * c * 2.0 * 3.0);
  hipLaunchKernelGGL(gn_stats, dim3(g.nchunks, g.groups, g.batch), dim3(256), 0, s, g);
  MD_HIP_CHECK(hipGetLastError());
  const long long items = (long long)p->hw * (c >> 3);
  const size_t lds = (2 * (size_t)c + 2 * (size_t)p->groups) * sizeof(float);
  hipLaunchKernelGGL(gn_apply, dim3((unsigned)((items + GN_ITEMS - 1) / GN_ITEMS), g.batch), dim3(256), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_layernorm(const void* x, const float* gamma, const float* beta, void* out, int32_t rows, int32_t c,
                            float eps, void* stream) {
  if (!x || !gamma || !beta || !out || rows <= 0) return MD_ERR_BAD_ARG;
  if (c <= 0 || (c & 7) || c > 2048) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_NORM, s, 0.0, (double)rows * c * 4.0);
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, (const half_t*)x, gamma, beta,
                     (half_t*)out, rows, c, eps);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}
