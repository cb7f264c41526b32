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
#include "gn_small.h"

namespace {
using namespace mdgn;

constexpr int GN_ITEMS = 512;    // 16-byte chunks per apply block (2 per thread: the loop is latency-bound, not table-bound)
constexpr int GN_MAX_TY = 8;

// stats: fully coalesced -- a block streams [pix_per_chunk][C] once with 16-byte loads, per-channel fp32 partials stay in
// registers, then a fixed-order reduction over the block's pixel lanes and the channels of each group (deterministic).
__global__ __launch_bounds__(512) void gn_stats(const GnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][ty][C]
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int tx = threadIdx.x % g.ch8, ty = threadIdx.x / g.ch8;
  const int p_begin = chunk * g.pix_per_chunk, p_end = min(g.hw, p_begin + g.pix_per_chunk);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  const int c = tx * 8;
  if (ty < g.ty) {
    // 16 pixels (16-byte vectors) per round trip: a thread owns <= 16 pixels of a 64x64 slice, so its loads all go out at once
    // (four at a time the loop was a chain of four memory round trips)
#ifndef MD_GN_NV
#define MD_GN_NV 16
#endif
    constexpr int NV = MD_GN_NV;
    const bool first = c < g.c0;
    const half_t* src = first ? g.x0 + c : g.x1 + (c - g.c0);
    const long long cs = first ? g.c0 : g.c1;
    for (int p0 = p_begin + ty; p0 < p_end; p0 += NV * g.ty) {
      h8 v[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int p = p0 + u * g.ty;
        if (p < p_end) v[u] = *reinterpret_cast<const h8*>(src + ((long long)b * g.hw + p) * cs);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        if (p0 + u * g.ty < p_end) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[u][e];
            s[e] += f;
            ss[e] += f * f;
          }
        }
      }
    }
    float* r0 = red + ty * g.c + c;
    float* r1 = red + (g.ty + ty) * g.c + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r0[e] = s[e];
      r1[e] = ss[e];
    }
  }
  __syncthreads();
  // 2 * groups (sum | sumsq) reductions, one per thread, fixed order
  if ((int)threadIdx.x < 2 * g.groups) {
    const int which = threadIdx.x / g.groups, grp = threadIdx.x % g.groups;
    float acc = 0.f;
    for (int t = 0; t < g.ty; ++t) {
      const float* r = red + (which * g.ty + t) * g.c + grp * g.cpg;
      for (int i = 0; i < g.cpg; ++i) acc += r[i];
    }
    g.ws[(((long long)b * g.nchunks + chunk) * g.groups + grp) * 2 + which] = acc;
  }
}

// apply: every global load this thread needs (its <= 2 input chunks, the partials it folds, gamma/beta of the table
// entries it builds) is issued up front so their latencies overlap; the only serial chain left is LDS + barriers.
__global__ __launch_bounds__(256) void gn_apply(const GnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // scale[c], shift[c], mean/rstd[groups], partial[8][2*groups]
  float* scale = tab;
  float* shift = tab + g.c;
  float* mean = tab + 2 * g.c;
  float* rstd = mean + g.groups;
  float* part = rstd + g.groups;
  const int b = blockIdx.y;
  const int ch8 = g.ch8;
  const int total = g.hw * ch8;
  const int begin = blockIdx.x * GN_ITEMS;
  const int end = min(begin + GN_ITEMS, total);
  constexpr int NI = GN_ITEMS / 256;
  // (1) this thread's input chunks
  h8 v[NI];
  int pc[NI], cc8[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int it = begin + threadIdx.x + 256 * u;
    const int p = it / ch8, cc = it - p * ch8;
    pc[u] = p;
    cc8[u] = cc * 8;
    if (it < end) {
      const long long pix = (long long)b * g.hw + p;
      v[u] = (cc8[u] < g.c0) ? *reinterpret_cast<const h8*>(g.x0 + pix * g.c0 + cc8[u])
                             : *reinterpret_cast<const h8*>(g.x1 + pix * g.c1 + (cc8[u] - g.c0));
    }
  }
  // (2) partial sums: 8 lanes per (group, which), <= 32 chunks per sample
  const int ng2 = 2 * g.groups;
  for (int i = threadIdx.x; i < 8 * ng2; i += 256) {
    const int lane8 = i / ng2, gw = i % ng2;  // gw = grp*2 + which
    float w4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = lane8 + 8 * u;
      if (ch < g.nchunks) w4[u] = g.ws[((long long)b * g.nchunks + ch) * ng2 + gw];
    }
    part[lane8 * ng2 + gw] = (w4[0] + w4[1]) + (w4[2] + w4[3]);
  }
  // (3) gamma / beta of the table entries this thread builds (C <= 4096 -> at most 16 per thread)
  float gm[16], bt[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < g.c) {
      gm[u] = (b >= g.batch2 ? g.gamma2 : g.gamma)[c];
      bt[u] = (b >= g.batch2 ? g.beta2 : g.beta)[c];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < g.groups) {
    float sm = 0.f, sq = 0.f;
    for (int l = 0; l < 8; ++l) {
      sm += part[l * ng2 + threadIdx.x * 2];
      sq += part[l * ng2 + threadIdx.x * 2 + 1];
    }
    const float inv_n = 1.0f / ((float)g.hw * (float)g.cpg);
    const float mu = sm * inv_n;
    const float var = fmaxf(sq * inv_n - mu * mu, 0.f);
    mean[threadIdx.x] = mu;
    rstd[threadIdx.x] = rsqrtf(var + g.eps);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < g.c) {
      const int grp = (int)(((unsigned)c * g.cpg_magic) >> 20);
      const float sc = rstd[grp] * gm[u];
      scale[c] = sc;
      shift[c] = bt[u] - mean[grp] * sc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int it = begin + threadIdx.x + 256 * u;
    if (it >= end) continue;
    const int c = cc8[u];
    const f4 s0 = *reinterpret_cast<const f4*>(scale + c), s1 = *reinterpret_cast<const f4*>(scale + c + 4);
    const f4 h0 = *reinterpret_cast<const f4*>(shift + c), h1 = *reinterpret_cast<const f4*>(shift + c + 4);
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sc = i < 4 ? s0[i] : s1[i - 4], sh = i < 4 ? h0[i] : h1[i - 4];
      float y = (float)v[u][i] * sc + sh;
      if (g.silu) y = md::silu_f(y);
      o[i] = (half_t)y;
    }
    *reinterpret_cast<h8*>(g.out + ((long long)b * g.hw + pc[u]) * g.c + c) = o;
  }
}


// Single-launch GroupNorm for SMALL slices (the 32x32 and lower levels of the step are launch-bound: two launches cost
// ~12 us for a few hundred KB): a block owns `gper` whole groups (cw = gper * cpg channels, a multiple of 8) of one sample
// and ALL its pixels, so statistics and normalisation need no cross-block exchange: pass 1 reads the slice (fp32 sums,
// fixed-order reductions -> deterministic); the slice stays in registers (measured +0.6 % end to end over re-reading it from
// L2), is normalised, SiLU-ed and stored.
__global__ __launch_bounds__(GN_SMALL_THREADS) void gn_small(const GnArgs g, int gper, int cw8) {
  __shared__ float red[GN_SMALL_THREADS / 64][2 * GN_GPER_MAX];
  __shared__ float stat[2 * GN_GPER_MAX];  // mean[gper], rstd[gper]
  const GnSmallThread t(g, gper, cw8);
  const bool first = t.c < g.c0;
  const half_t* src = first ? g.x0 + t.c : g.x1 + (t.c - g.c0);
  const long long cs = first ? g.c0 : g.c1;
  const long long pix0 = (long long)t.b * g.hw;
  // the whole slice of this thread (<= GN_SMALL_MAXV vectors, launcher-checked) is loaded up front and stays in registers:
  // one trip to L2, all loads in flight together
  h8 x[GN_SMALL_MAXV];
#pragma unroll
  for (int i = 0; i < GN_SMALL_MAXV; ++i) {
    const int p = t.pl + i * t.ps;
    if (t.active && p < g.hw) x[i] = *reinterpret_cast<const h8*>(src + (pix0 + p) * cs);
  }
  f4 gb[4];
  gn_small_affine(g, t, gb);
  gn_small_finish<GN_SMALL_MAXV>(g, gper, t, x, gb, red, stat);
}

// GroupNorm from PRODUCER partials (round 3; the ResBlock's "conv + GroupNorm + SiLU" fusion): the md_igemm that wrote x also
// wrote, per 64-row granule and channel, the sum and the sum of squares of the fp16 values it stored (md_igemm_params.gn_part), so
// the statistics pass over x disappears.  gn_finalize folds the partials of one (sample, group) -- hw / 64 granules x cpg
// channels x 2, about 6 % of the tensor's bytes in total, fixed order -> deterministic -- into the (sum, sumsq) pair gn_apply reads
// (its workspace format with ONE chunk per sample); gn_apply then streams x fully coalesced, exactly as after gn_stats.
// (A single-launch form -- every apply block folding the partials of its own groups -- was built first and measured no faster
//  than stats + apply at one frame and 1.6 % slower end to end at 8: its blocks own 80-byte row slices, i.e. poorly coalesced
//  reads, or else re-read the whole partial table per block.)
// Two-source concat: channel ca < c0 reads part0, the rest part1 (a group may straddle the two).
__global__ __launch_bounds__(256) void gn_finalize(const GnArgs g, const float* __restrict__ part0, const float* __restrict__ part1) {
  __shared__ float red[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, b = blockIdx.y;
  const int P = g.hw >> 6;
  const int lanes = 256 / g.cpg;                 // granule lanes (cpg <= 128, launcher-checked)
  const int cl = tid % g.cpg, pg = tid / g.cpg;
  float s = 0.f, q = 0.f;
  if (pg < lanes) {
    const int ca = grp * g.cpg + cl;
    const bool f0 = ca < g.c0;
    const float* pt = f0 ? part0 + ca : part1 + (ca - g.c0);
    const long long pc = f0 ? g.c0 : g.c1;
    for (int gr = pg; gr < P; gr += lanes) {
      const float* row = pt + ((long long)b * P + gr) * 2 * pc;
      s += row[0];
      q += row[pc];
    }
  }
  s = md::wave_sum(s);
  q = md::wave_sum(q);
  if (lane == 0) {
    red[wave][0] = s;
    red[wave][1] = q;
  }
  __syncthreads();
  if (tid < 2) g.ws[((long long)b * g.groups + grp) * 2 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// producer partials usable: whole 64-row granules per sample, a group's channels fit one finalize block
inline bool gn_part_ok(int hw, int cpg) { return (hw & 63) == 0 && cpg <= 128; }

// stats grid geometry shared by the launcher and the workspace query
inline void gn_geometry(int batch, int hw, int c, int* ty, int* pix_per_chunk, int* nchunks) {
  const int ch8 = c >> 3;
  int t = 512 / ch8;
  if (t < 1) t = 1;
  if (t > GN_MAX_TY) t = GN_MAX_TY;
  // at most 32 partials per sample (every apply block folds them), at least ty pixels per block
  (void)batch;
  int ppc = (hw + 31) / 32;
  if (ppc < t) ppc = t;
  if (ppc > hw) ppc = hw;
  *ty = t;
  *pix_per_chunk = ppc;
  *nchunks = (hw + ppc - 1) / ppc;
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
  // upper bound over channel counts: the stats grid never uses more than hw chunks per sample
  int64_t nchunks = 32;
  return (int64_t)batch * groups * nchunks * 2 * sizeof(float);
}

extern "C" int md_groupnorm_wants_partials(int32_t batch, int32_t hw, int32_t c, int32_t groups) {
  (void)batch;
  if (groups <= 0 || c <= 0 || c % groups) return 0;
  int gper = 1, cw8 = 1;
  if (gn_small_ok(hw, c / groups, groups, &gper, &cw8)) return 0;
  return gn_part_ok(hw, c / groups) ? 1 : 0;
}

extern "C" int md_groupnorm(const md_groupnorm_params* p, void* stream) {
  if (!p || !p->ws) return MD_ERR_BAD_ARG;
  GnArgs g;
  if (const int rc = gn_fill_common(p, g)) return rc;
  const int c = g.c;
  gn_geometry(p->batch, p->hw, c, &g.ty, &g.pix_per_chunk, &g.nchunks);
  if ((int64_t)p->batch * g.nchunks * p->groups * 2 * (int64_t)sizeof(float) > p->ws_bytes) return MD_ERR_WORKSPACE;
  // exact c / cpg by multiply-shift, verified for this (C, cpg)
  g.cpg_magic = ((1u << 20) + (unsigned)g.cpg - 1u) / (unsigned)g.cpg;
  for (int ch = 0; ch < c; ++ch)
    if ((int)(((unsigned)ch * g.cpg_magic) >> 20) != ch / g.cpg) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_NORM, s, 0.0, (double)p->batch * p->hw * c * 2.0 * 3.0);
  {
    int gper = 1, cw8 = 1;
    if (gn_small_ok(p->hw, g.cpg, p->groups, &gper, &cw8)) {
      hipLaunchKernelGGL(gn_small, dim3(p->groups / gper, g.batch), dim3(GN_SMALL_THREADS), 0, s, g, gper, cw8);
      MD_HIP_CHECK(hipGetLastError());
      return MD_OK;
    }
  }
  if (p->part0 && (p->c1 == 0 || p->part1) && gn_part_ok(p->hw, g.cpg)) {
    g.nchunks = 1;   // one (sum, sumsq) pair per (sample, group), folded from the producers' partials
    hipLaunchKernelGGL(gn_finalize, dim3(p->groups, g.batch), dim3(256), 0, s, g, p->part0, p->part1 ? p->part1 : p->part0);
    MD_HIP_CHECK(hipGetLastError());
  } else {
    const int threads = g.ch8 * g.ty;
    const size_t lds1 = 2 * (size_t)g.ty * c * sizeof(float);
    hipLaunchKernelGGL(gn_stats, dim3(g.nchunks, g.batch), dim3(threads < 64 ? 64 : threads), lds1, s, g);
    MD_HIP_CHECK(hipGetLastError());
  }
  const long long items = (long long)p->hw * g.ch8;
  const size_t lds2 = (2 * (size_t)c + 2 * (size_t)p->groups + 16 * (size_t)p->groups) * sizeof(float);
  hipLaunchKernelGGL(gn_apply, dim3((unsigned)((items + GN_ITEMS - 1) / GN_ITEMS), g.batch), dim3(256), lds2, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

// Row softmax, fp32 scores -> fp16 probabilities: one 256-thread block per row, the row held in registers
// (cols <= 16384), max and sum reduced wave-wise then across the four waves through LDS.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long long ld_s,
                                                           half_t* __restrict__ p, long long ld_p, int cols,
                                                           float scale_log2e) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sr = s + (long long)blockIdx.x * ld_s;
  half_t* pr = p + (long long)blockIdx.x * ld_p;
  const int c4 = cols >> 2;
  f4 v[16];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = tid + 256 * j;
    if (i < c4) {
      v[j] = *reinterpret_cast<const f4*>(sr + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[j][e]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = tid + 256 * j;
    if (i < c4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[j][e] = exp2f(v[j][e] * scale_log2e - mx);
        sum += v[j][e];
      }
    }
  }
  sum = md::wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = tid + 256 * j;
    if (i < c4) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(v[j][e] * inv);
      *reinterpret_cast<h4*>(pr + i * 4) = o;
    }
  }
}

extern "C" int md_softmax_rows(const float* s, int64_t ld_s, void* p, int64_t ld_p, int32_t rows, int32_t cols, float scale,
                               void* stream) {
  if (!s || !p || rows <= 0 || cols <= 0 || ld_s < cols || ld_p < cols) return MD_ERR_BAD_ARG;
  if ((cols & 3) || (ld_s & 3) || (ld_p & 3) || cols > 16384 || scale <= 0.f) return MD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_NORM, st, 0.0, (double)rows * cols * 6.0);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, (long long)ld_s, (half_t*)p,
                     (long long)ld_p, cols, scale * 1.44269504088896340736f);
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
