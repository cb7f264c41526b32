// Single-launch GroupNorm for SMALL slices, shared by md_groupnorm (norm.hip: gn_small) and by the split-K reduction of md_igemm
// that normalises its own output (igemm.hip: igemm_splitk_reduce_gn).  The 32x32 and lower levels of a step are launch-bound (two
// launches cost ~12 us for a few hundred KB): a block owns `gper` whole groups (cw = gper * cpg channels, a multiple of 8) of one
// sample and ALL its pixels, so statistics and normalisation need no cross-block exchange: the slice is produced once (loaded, or
// summed from the split-K slabs) and stays in registers (measured +0.6 % end to end over re-reading it from L2), fp32 sums with
// fixed-order reductions -> deterministic; then it is normalised, SiLU-ed and stored.  Both kernels run THE SAME statistics /
// normalisation code on the same thread mapping, so the fused form is bit-identical to the two launches it replaces.
#pragma once
#include "md_common.h"

namespace mdgn {

struct GnArgs {
  const half_t* x0;
  const half_t* x1;
  int c0, c1, c, cpg, groups;
  int batch, hw, nchunks, pix_per_chunk;
  int ch8, ty;            // stats block = ch8 x ty threads: thread (tx, ty) owns 8 channels tx*8.. of pixels ty, ty+TY, ..
  unsigned cpg_magic;     // c / cpg == (c * cpg_magic) >> 20 for every c < C (checked on the host)
  float eps;
  const float* gamma;
  const float* beta;
  const float* gamma2;   // samples b >= batch2 (second network of a merged batch)
  const float* beta2;
  int batch2;
  int silu;
  half_t* out;
  float* ws;  // [batch][nchunks][groups][2]
};

constexpr int GN_SMALL_THREADS = 512, GN_GPER_MAX = 8, GN_SMALL_MAXV = 16;

// whole groups per block such that the block's channel slice is a multiple of 8 channels (16-byte vectors)
inline bool gn_group_block(int cpg, int groups, int* gper, int* cw8) {
  int gp = 1;
  while (gp <= GN_GPER_MAX && ((gp * cpg) & 7)) gp <<= 1;
  if (gp > GN_GPER_MAX || groups % gp) return false;
  *gper = gp;
  *cw8 = gp * cpg / 8;
  return *cw8 >= 1 && *cw8 <= GN_SMALL_THREADS;
}
constexpr long long g_gn_small_bytes = 96LL << 10;
// small slices: one launch, a block owns whole groups of one sample and all its pixels (gn_small)
inline bool gn_small_ok(int hw, int cpg, int groups, int* gper, int* cw8) {
  return gn_group_block(cpg, groups, gper, cw8) && (long long)hw * *gper * cpg * 2 <= g_gn_small_bytes &&
         hw <= GN_SMALL_MAXV * (GN_SMALL_THREADS / *cw8);
}

// the fields of GnArgs every GroupNorm kernel reads (md_groupnorm adds its statistics-grid geometry); MD_OK or the error
inline int gn_fill_common(const md_groupnorm_params* p, GnArgs& g) {
  if (!p || !p->x0 || !p->gamma || !p->beta || !p->out) return MD_ERR_BAD_ARG;
  const int c = p->c0 + p->c1;
  if (p->c0 <= 0 || (p->c0 & 7) || p->c1 < 0 || (p->c1 & 7) || ((p->c1 > 0) != (p->x1 != nullptr))) return MD_ERR_BAD_ARG;
  if (p->groups <= 0 || p->groups > 128 || c % p->groups || c > 4096) return MD_ERR_UNSUPPORTED;
  if (p->batch <= 0 || p->hw <= 0) return MD_ERR_BAD_ARG;
  g.x0 = (const half_t*)p->x0;
  g.x1 = (const half_t*)p->x1;
  g.c0 = p->c0;
  g.c1 = p->c1;
  g.c = c;
  g.groups = p->groups;
  g.cpg = c / p->groups;
  g.batch = p->batch;
  g.hw = p->hw;
  g.ch8 = c >> 3;
  if (g.ch8 > 512) return MD_ERR_UNSUPPORTED;
  g.eps = p->eps;
  g.gamma = p->gamma;
  g.beta = p->beta;
  const bool dual = p->gamma2 && p->beta2 && p->batch2 > 0;
  g.gamma2 = dual ? p->gamma2 : p->gamma;
  g.beta2 = dual ? p->beta2 : p->beta;
  g.batch2 = dual ? p->batch2 : 0x7fffffff;
  g.silu = p->silu;
  g.out = (half_t*)p->out;
  g.ws = (float*)p->ws;
  g.nchunks = 1;
  g.pix_per_chunk = p->hw;
  g.ty = 1;
  g.cpg_magic = 0;
  return MD_OK;
}

// thread mapping of a small-slice block: fixed vector column v (8 channels c..c+7), pixel lane pl, pixel stride ps
struct GnSmallThread {
  int tid, b, v, pl, ps, c;
  bool active;
  int lg[8];  // local group of each of this thread's 8 channels
  __device__ __forceinline__ GnSmallThread(const GnArgs& g, int gper, int cw8) {
    tid = threadIdx.x;
    b = blockIdx.y;
    v = tid % cw8;
    pl = tid / cw8;
    ps = GN_SMALL_THREADS / cw8;
    active = pl < ps;
    c = blockIdx.x * gper * g.cpg + v * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) lg[e] = (v * 8 + e) / g.cpg;
  }
};

// gamma / beta of this thread's 8 channels: issued with the slice (they do not depend on the statistics; behind the barriers they
// were a second round trip)
__device__ __forceinline__ void gn_small_affine(const GnArgs& g, const GnSmallThread& t, f4 (&gb)[4]) {
  gb[0] = gb[1] = gb[2] = gb[3] = f4{0.f, 0.f, 0.f, 0.f};
  if (t.active) {
    const float* gamma = (t.b >= g.batch2 ? g.gamma2 : g.gamma) + t.c;
    const float* beta = (t.b >= g.batch2 ? g.beta2 : g.beta) + t.c;
    gb[0] = *reinterpret_cast<const f4*>(gamma);
    gb[1] = *reinterpret_cast<const f4*>(gamma + 4);
    gb[2] = *reinterpret_cast<const f4*>(beta);
    gb[3] = *reinterpret_cast<const f4*>(beta + 4);
  }
}

// statistics of the block's groups from the register-resident slice x[] (vector i = pixel pl + i ps), normalise, SiLU, store
template <int MAXV>
__device__ __forceinline__ void gn_small_finish(const GnArgs& g, int gper, const GnSmallThread& t, const h8 (&x)[MAXV], const f4 (&gb)[4],
                                                float (*red)[2 * GN_GPER_MAX], float* stat) {
  // every multiply-add below is spelled out (explicit fma, contraction off): this function is instantiated in two kernels whose
  // results must agree bit for bit, and left to the compiler one instantiation fuses a multiply-add that the other keeps apart
#pragma clang fp contract(off)
  const int tid = t.tid, lane = tid & 63, wave = tid >> 6;
  const long long pix0 = (long long)t.b * g.hw;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int p = t.pl + i * t.ps;
    if (t.active && p < g.hw) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)x[i][e];
        s[e] += f;
        q[e] = __builtin_fmaf(f, f, q[e]);
      }
    }
  }
  for (int k = 0; k < gper; ++k) {  // per local group: this thread's share, then wave, then block (fixed order)
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a += t.lg[e] == k ? s[e] : 0.f;
      bq += t.lg[e] == k ? q[e] : 0.f;
    }
    a = md::wave_sum(a);
    bq = md::wave_sum(bq);
    if (lane == 0) {
      red[wave][2 * k] = a;
      red[wave][2 * k + 1] = bq;
    }
  }
  __syncthreads();
  if (tid < gper) {
    float sm = 0.f, sq = 0.f;
    for (int w = 0; w < GN_SMALL_THREADS / 64; ++w) {
      sm += red[w][2 * tid];
      sq += red[w][2 * tid + 1];
    }
    const float inv_n = 1.0f / ((float)g.hw * (float)g.cpg);
    const float mu = sm * inv_n;
    stat[tid] = mu;
    stat[GN_GPER_MAX + tid] = rsqrtf(fmaxf(__builtin_fmaf(-mu, mu, sq * inv_n), 0.f) + g.eps);
  }
  __syncthreads();
  if (!t.active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = stat[GN_GPER_MAX + t.lg[e]] * (e < 4 ? gb[0][e] : gb[1][e - 4]);
    sh[e] = __builtin_fmaf(-stat[t.lg[e]], sc[e], e < 4 ? gb[2][e] : gb[3][e - 4]);
  }
  half_t* dst = g.out + t.c;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int p = t.pl + i * t.ps;
    if (p < g.hw) {
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = __builtin_fmaf((float)x[i][e], sc[e], sh[e]);
        if (g.silu) y = md::silu_f(y);
        o[e] = (half_t)y;
      }
      *reinterpret_cast<h8*>(dst + (pix0 + p) * g.c) = o;
    }
  }
}

}  // namespace mdgn
