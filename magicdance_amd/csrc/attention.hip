// md_attention: flash-style attention for gfx950 with two K/V segments (self tokens, then appearance-bank tokens).
//
// Per (batch, head, 64*QF query rows) workgroup of 4 waves; each wave owns QF fragments of 16 query rows.
//   S^T[kv][q]  = K[kv][:] . Q[q][:]          v_mfma_f32_16x16x32_f16, A = K tile rows (LDS), B = Q (registers)
//   softmax over kv: a lane holds 4 kv values per 16-kv fragment for ONE query column -> row max / row sum are
//                    in-lane plus two cross-lane steps (xor 16, xor 32); running max / sum in fp32 (online softmax)
//   O^T[d][q]  += V^T[d][kv] . P^T[kv][q]     A = V^T tile rows (LDS, token-contiguous), B = P^T straight from the
//                    S^T accumulator registers (k-slot j of lane group g <-> kv = 16*(j>>2) + 4*g + (j&3) on both operands)
// V is consumed transposed ([H*d][tokens]); the projection GEMM writes it that way (md_igemm n_tr_begin), so K and V^T
// tiles are plain 16-byte copies global -> registers -> LDS, prefetched one tile ahead of the MFMAs.
// The concat [self ; bank] of the reference (attention.py:305-311) is never materialised: tiles walk segment 0 then 1.
#include <cstdio>
#include <cstdlib>

#include "md_common.h"

namespace {

struct AttnArgs {
  const half_t* q;
  long long q_bs;
  int ld_q;
  const half_t* k0;
  long long k0_bs;
  int ld_k0;
  const half_t* vt0;
  long long vt0_bs;
  int ld_vt0;
  int n0;
  const half_t* k1;
  long long k1_bs;
  int ld_k1;
  const half_t* vt1;
  long long vt1_bs;
  int ld_vt1;
  int n1, n1_batches;
  half_t* out;
  long long out_bs;
  int ld_out;
  int batch, heads, nq;
  float c;  // scale * log2(e)
};

template <int D, int QF>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs g) {
  constexpr int DK = (D + 31) / 32 * 32;  // QK^T contraction length (zero padded)
  constexpr int KSTEPS = DK / 32;
  constexpr int DF = (D + 15) / 16;       // 16-row fragments of O^T
  constexpr int DV = DF * 16;
  constexpr int CPR = DK / 8;             // 16-byte chunks per K row
  constexpr int KROW = (DK + 16) * 2;     // LDS bytes per K row: stride = 32 (mod 64) bytes makes the 16-row ds_read_b128 conflict-free
  constexpr int VROW = 144;               // LDS bytes per V^T row: 64 kv fp16 + 16 pad
  constexpr int KJ = (64 * CPR + 255) / 256;
  constexpr int VJ = (DV * 8 + 255) / 256;
  constexpr int BQ = 64 * QF;

  __shared__ __attribute__((aligned(16))) char smem[64 * KROW + DV * VROW];
  char* Ks = smem;
  char* Vs = smem + 64 * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qbase = blockIdx.x * BQ + wave * (16 * QF);

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = lr][d = ks*32 + lg*8 .. +8] -------------
  h8 qf[QF][KSTEPS];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int row = qbase + f * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      const int d = ks * 32 + lg * 8;
      if (row < g.nq && d < D) v = *reinterpret_cast<const h8*>(g.q + b * g.q_bs + (long long)row * g.ld_q + h * D + d);
      qf[f][ks] = v;
    }
  }

  const int t0 = (g.n0 + 63) >> 6;
  const int t1 = (g.k1 != nullptr && b < g.n1_batches) ? ((g.n1 + 63) >> 6) : 0;
  const int ntiles = t0 + t1;

  h8 rk[KJ], rv[VJ];
  auto load_tile = [&](int t) {
    const bool s1 = t >= t0;
    const int kv0 = (s1 ? t - t0 : t) << 6;
    const int nseg = s1 ? g.n1 : g.n0;
    const half_t* kp = s1 ? g.k1 + b * g.k1_bs : g.k0 + b * g.k0_bs;
    const half_t* vp = s1 ? g.vt1 + b * g.vt1_bs : g.vt0 + b * g.vt0_bs;
    const int ldk = s1 ? g.ld_k1 : g.ld_k0;
    const int ldv = s1 ? g.ld_vt1 : g.ld_vt0;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int i = tid + 256 * j;
      const int row = i / CPR, col = i - row * CPR;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (i < 64 * CPR && kv0 + row < nseg && col * 8 < D)
        v = *reinterpret_cast<const h8*>(kp + (long long)(kv0 + row) * ldk + h * D + col * 8);
      rk[j] = v;
    }
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int i = tid + 256 * j;
      const int row = i >> 3, col = i & 7;
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (i < DV * 8 && row < D && kv0 + col * 8 < nseg)
        v = *reinterpret_cast<const h8*>(vp + (long long)(h * D + row) * ldv + kv0 + col * 8);
      rv[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int i = tid + 256 * j;
      const int row = i / CPR, col = i - row * CPR;
      if (i < 64 * CPR) *reinterpret_cast<h8*>(Ks + row * KROW + col * 16) = rk[j];
    }
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int i = tid + 256 * j;
      const int row = i >> 3, col = i & 7;
      if (i < DV * 8) *reinterpret_cast<h8*>(Vs + row * VROW + col * 16) = rv[j];
    }
  };

  f4 o[DF][QF];
#pragma unroll
  for (int i = 0; i < DF; ++i)
#pragma unroll
    for (int f = 0; f < QF; ++f) o[i][f] = f4{0.f, 0.f, 0.f, 0.f};
  float m_run[QF], l_run[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    m_run[f] = -INFINITY;
    l_run[f] = 0.f;
  }

  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // every wave is done reading the previous tile from LDS
    store_tile();
    __syncthreads();
    if (t + 1 < ntiles) load_tile(t + 1);  // in flight under the MFMAs below

    const bool s1 = t >= t0;
    const int kv0 = (s1 ? t - t0 : t) << 6;
    const int nseg = s1 ? g.n1 : g.n0;

    // ---- S^T = K Q^T ---------------------------------------------------------------------------------
    f4 st[QF][4];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) st[f][kf] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const h8 kfrag = *reinterpret_cast<const h8*>(Ks + (kf * 16 + lr) * KROW + (ks * 32 + lg * 8) * 2);
#pragma unroll
        for (int f = 0; f < QF; ++f)
          st[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfrag, qf[f][ks], st[f][kf], 0, 0, 0);
      }
    }
    if (__builtin_amdgcn_readfirstlane(kv0 + 64 - nseg) > 0) {  // tail tile of a segment (wave-uniform, rare): mask kv >= nseg
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = kv0 + kf * 16 + lg * 4 + r >= nseg;
#pragma unroll
          for (int f = 0; f < QF; ++f) st[f][kf][r] = dead ? -INFINITY : st[f][kf][r];
        }
    }

    // ---- online softmax (per query column = per lane, fp32) ---------------------------------------------
    h8 pf[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      float mx = st[f][0][0];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[f][kf][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[f], mx * g.c);
      // the running max stops moving after the first few tiles: rescale O / l only when some lane's max grew
      const bool grew = __builtin_amdgcn_ballot_w64(m_new > m_run[f]) != 0;
      const float alpha = grew ? __builtin_amdgcn_exp2f(m_run[f] - m_new) : 1.0f;
      m_run[f] = m_new;
      float ps = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[f][kf][r] * g.c - m_new);
          ps += p;
          pf[f][kf >> 1][(kf & 1) * 4 + r] = (half_t)p;
        }
      if (grew) {
        l_run[f] *= alpha;
#pragma unroll
        for (int i = 0; i < DF; ++i) o[i][f] *= alpha;
      }
      l_run[f] += ps;
    }

    // ---- O^T += V^T P^T ---------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < DF; ++i) {
#pragma unroll
      for (int pk = 0; pk < 2; ++pk) {
        const char* vrow = Vs + (i * 16 + lr) * VROW;
        const h4 lo = *reinterpret_cast<const h4*>(vrow + ((2 * pk) * 16 + lg * 4) * 2);
        const h4 hi = *reinterpret_cast<const h4*>(vrow + ((2 * pk + 1) * 16 + lg * 4) * 2);
        const h8 vfrag = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int f = 0; f < QF; ++f)
          o[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf[f][pk], o[i][f], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane owns O[q = lr][d = i*16 + lg*4 .. +4] -------------------------------------
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    float l = l_run[f];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = qbase + f * 16 + lr;
    if (row >= g.nq) continue;
    half_t* op = g.out + b * g.out_bs + (long long)row * g.ld_out + h * D;
#pragma unroll
    for (int i = 0; i < DF; ++i) {
      const int d = i * 16 + lg * 4;
      if (d < D) {
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[i][f][r] * inv);
        *reinterpret_cast<h4*>(op + d) = ov;
      }
    }
  }
}

template <int D, int QF>
int launch(const AttnArgs& g, hipStream_t s) {
  constexpr int BQ = 64 * QF;
  dim3 grid((g.nq + BQ - 1) / BQ, g.heads, g.batch);
  hipLaunchKernelGGL((attn_kernel<D, QF>), grid, dim3(256), 0, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

}  // namespace

extern "C" int md_attention(const md_attention_params* p, void* stream) {
  if (!p || !p->q || !p->k0 || !p->vt0 || !p->out) return MD_ERR_BAD_ARG;
  if (p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->n0 <= 0) return MD_ERR_BAD_ARG;
  if ((p->ld_q & 7) || (p->ld_k0 & 7) || (p->ld_vt0 & 7) || (p->ld_out & 3)) return MD_ERR_BAD_ARG;
  if (p->ld_vt0 < ((p->n0 + 7) & ~7)) return MD_ERR_BAD_ARG;
  if ((p->q_batch_stride & 7) || (p->k0_batch_stride & 7) || (p->vt0_batch_stride & 7) || (p->out_batch_stride & 3))
    return MD_ERR_BAD_ARG;
  if (p->k1) {
    if (!p->vt1 || p->n1 <= 0 || (p->ld_k1 & 7) || (p->ld_vt1 & 7) || p->ld_vt1 < ((p->n1 + 7) & ~7)) return MD_ERR_BAD_ARG;
    if ((p->k1_batch_stride & 7) || (p->vt1_batch_stride & 7)) return MD_ERR_BAD_ARG;
  }
  AttnArgs g;
  g.q = (const half_t*)p->q;
  g.q_bs = p->q_batch_stride;
  g.ld_q = p->ld_q;
  g.k0 = (const half_t*)p->k0;
  g.k0_bs = p->k0_batch_stride;
  g.ld_k0 = p->ld_k0;
  g.vt0 = (const half_t*)p->vt0;
  g.vt0_bs = p->vt0_batch_stride;
  g.ld_vt0 = p->ld_vt0;
  g.n0 = p->n0;
  g.k1 = (const half_t*)p->k1;
  g.k1_bs = p->k1_batch_stride;
  g.ld_k1 = p->ld_k1;
  g.vt1 = (const half_t*)p->vt1;
  g.vt1_bs = p->vt1_batch_stride;
  g.ld_vt1 = p->ld_vt1;
  g.n1 = p->k1 ? p->n1 : 0;
  g.n1_batches = p->k1 ? p->n1_batches : 0;
  g.out = (half_t*)p->out;
  g.out_bs = p->out_batch_stride;
  g.ld_out = p->ld_out;
  g.batch = p->batch;
  g.heads = p->heads;
  g.nq = p->nq;
  g.c = p->scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  const double nkv = (double)p->n0 + (double)g.n1 * ((double)(g.n1_batches < p->batch ? g.n1_batches : p->batch) / p->batch);
  char tag[96];
  snprintf(tag, sizeof(tag), "B=%d H=%d nq=%d n0=%d n1=%d n1b=%d d=%d", p->batch, p->heads, p->nq, p->n0, g.n1, g.n1_batches,
           p->d);
  md::ProfScope prof(MD_FAM_ATTENTION, s, 4.0 * p->batch * p->heads * (double)p->nq * nkv * p->d,
                     2.0 * p->batch * p->heads * p->d * (2.0 * p->nq + 2.0 * nkv), tag);
  // 128-row query blocks (QF 2) halve the K/V traffic per MFMA but need >= ~2 workgroups per CU to hide the per-tile
  // latency chain; below that 64-row blocks win (measured: d=80 72->49 us, d=40 B=1 104->94 us, d=40 B=2 unchanged)
  static const int qf_force = [] {
    const char* e = getenv("MD_ATTN_QF");
    return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0;
  }();
  const long long wg128 = (long long)((p->nq + 127) / 128) * p->heads * p->batch;
  const int qf = qf_force ? qf_force : (wg128 >= 512 ? 2 : 1);
  switch (p->d) {
    case 40: return qf == 1 ? launch<40, 1>(g, s) : launch<40, 2>(g, s);
    case 80: return qf == 1 ? launch<80, 1>(g, s) : launch<80, 2>(g, s);
    case 160: return launch<160, 1>(g, s);
    case 8: return launch<8, 1>(g, s);    // small-geometry test nets (model_channels 64, 8 heads)
    case 16: return launch<16, 1>(g, s);
    case 32: return launch<32, 1>(g, s);
    case 64: return launch<64, 2>(g, s);
    case 128: return launch<128, 1>(g, s);
    default: return MD_ERR_UNSUPPORTED;
  }
}
