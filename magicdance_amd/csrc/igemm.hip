// md_igemm: implicit-GEMM convolution / linear layer for gfx950 (MI355X), fp16 operands, fp32 MFMA accumulate.
//
//   D[n][m] = sum_k W[n][k] * A(m,k)        (operands swapped so a lane owns 4 consecutive n of one output row m:
//                                            NHWC epilogue = one 8-byte store per fragment, bias/residual vector loads)
//   m = (b, oy, ox)   k = tap*(c0+c1) + c   A gathered on the fly from one or two NHWC sources (channel concat),
//                                            with zero padding, stride 2 and nearest x2 upsample folded in the gather.
//
// Tile: BM x BN x 64, 256 threads = 4 waves (WAVES_M x WAVES_N), v_mfma_f32_16x16x32_f16.  LDS holds two stages of
// [BM][64] + [BN][64] fp16 with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) so the fragment ds_read_b128 of 16
// consecutive rows is at most 2-way conflicted.  Global->register->LDS staging (the gather needs per-lane predication),
// next tile's global loads are issued before the current tile's MFMAs, one barrier per k-tile.
// Small-M layers (8x8 / 16x16 latents) are weight-streaming bound: split-K over blockIdx.z with fp32 slabs and a
// deterministic reduce kernel that applies the same epilogue.
//
// Reference arithmetic replaced: see include/magicdance_hip.h (md_igemm).
#include "md_common.h"

namespace {

struct IgemmArgs {
  const half_t* a0;
  const half_t* a1;
  int c0, c1, cin;
  int batch, hin, win, hout, wout, tokens;
  int ksize, stride, ups, pad;
  const half_t* w;
  int M, N, K;
  int nk, splitk, tiles_per_split;
  int tiles_m, tiles_n;
  // epilogue
  const float* bias;
  long long bias_bs;
  const half_t* res;
  int ld_res;
  int act;
  void* out;
  int ld_out;
  int out_f32;
  half_t* out_t;
  int n_tr_begin;
  int ld_t;
  float* ws;
};

// Store 4 consecutive output columns n..n+3 of row m (b = m / tokens precomputed).
__device__ __forceinline__ void epi_store4(const IgemmArgs& g, int m, int b, int n, f4 v) {
  if (g.bias) {
    const f4 bv = *reinterpret_cast<const f4*>(g.bias + (long long)b * g.bias_bs + n);
    v += bv;
  }
  if (g.act == MD_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = md::silu_f(v[i]);
  }
  if (n >= g.n_tr_begin) {
    // transposed store (V^T): [b][n - n_tr][tok]
    const int tok = m - b * g.tokens;
    const int ntr = g.N - g.n_tr_begin;
    half_t* o = g.out_t + ((long long)b * ntr + (n - g.n_tr_begin)) * g.ld_t + tok;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[(long long)i * g.ld_t] = (half_t)v[i];
    return;
  }
  if (g.res) {
    const h4 rv = *reinterpret_cast<const h4*>(g.res + (long long)m * g.ld_res + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
  }
  if (g.out_f32) {
    *reinterpret_cast<f4*>(reinterpret_cast<float*>(g.out) + (long long)m * g.ld_out + n) = v;
  } else {
    h4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
    *reinterpret_cast<h4*>(reinterpret_cast<half_t*>(g.out) + (long long)m * g.ld_out + n) = o;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs g) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int AJ = BM / 32, WJ = BN / 32;  // 16-byte chunks per thread per k-tile
  constexpr int STAGE_BYTES = (BM + BN) * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;

  // XCD-aware bijective remap: consecutive logical tiles (same weight panel) share one XCD's L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical % g.tiles_m, tile_n = logical / g.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz = blockIdx.z;
  const int kt_begin = kz * g.tiles_per_split;
  const int kt_end = min(g.nk, kt_begin + g.tiles_per_split);

  // ---- loader role: chunk column lc (16 B of k), rows lrow + 32 j -------------------------------------
  const int lc = tid & 7, lrow = tid >> 3;
  int a_iy0[AJ], a_ix0[AJ], a_b[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + lrow + 32 * j;
    if (m < g.M) {
      const int b = m / g.tokens;
      const int rem = m - b * g.tokens;
      const int oy = rem / g.wout;
      const int ox = rem - oy * g.wout;
      a_b[j] = b;
      a_iy0[j] = oy * g.stride - g.pad;
      a_ix0[j] = ox * g.stride - g.pad;
    } else {
      a_b[j] = -1;
      a_iy0[j] = 0;
      a_ix0[j] = 0;
    }
  }
  const int vh = g.ups ? 2 * g.hin : g.hin, vw = g.ups ? 2 * g.win : g.win;

  h8 ra[AJ], rw[WJ];
  auto load_tile = [&](int kt) {
    const int k = kt * 64 + lc * 8;
    const bool kvalid = k < g.K;
    int tap = 0, cc = k;
    if (g.ksize == 3) {
      tap = k / g.cin;
      cc = k - tap * g.cin;
    }
    const int dy = tap / 3, dx = tap - dy * 3;
    const half_t* src = g.a0;
    int cs = g.c0, ccc = cc;
    if (cc >= g.c0) {
      src = g.a1;
      cs = g.c1;
      ccc = cc - g.c0;
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      if (kvalid && a_b[j] >= 0 && iy >= 0 && iy < vh && ix >= 0 && ix < vw) {
        const int sy = g.ups ? (iy >> 1) : iy, sx = g.ups ? (ix >> 1) : ix;
        const long long off = ((long long)(a_b[j] * g.hin + sy) * g.win + sx) * cs + ccc;
        v = *reinterpret_cast<const h8*>(src + off);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      const int n = n0 + lrow + 32 * j;
      if (kvalid && n < g.N) v = *reinterpret_cast<const h8*>(g.w + (long long)n * g.K + k);
      rw[j] = v;
    }
  };
  auto store_tile = [&](int stage) {
    char* As = smem + stage * STAGE_BYTES;
    char* Ws = As + BM * 128;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int row = lrow + 32 * j;
      *reinterpret_cast<h8*>(As + row * 128 + ((lc ^ (row & 7)) << 4)) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int row = lrow + 32 * j;
      *reinterpret_cast<h8*>(Ws + row * 128 + ((lc ^ (row & 7)) << 4)) = rw[j];
    }
  };

  f4 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int stage = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) load_tile(kt + 1);
    const char* As = smem + stage * STAGE_BYTES;
    const char* Ws = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 af[MF], wf[NF];
      const int chunk = ks * 4 + lg;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int row = wm * WTM + i * 16 + lr;
        af[i] = *reinterpret_cast<const h8*>(As + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int row = wn * WTN + i * 16 + lr;
        wf[i] = *reinterpret_cast<const h8*>(Ws + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(stage ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  if (g.splitk > 1) {
#pragma unroll
    for (int j = 0; j < MF; ++j) {
      const int m = m0 + wm * WTM + j * 16 + lr;
      if (m >= g.M) continue;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int n = n0 + wn * WTN + i * 16 + lg * 4;
        if (n >= g.N) continue;
        *reinterpret_cast<f4*>(g.ws + ((long long)kz * g.M + m) * g.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (g.act == MD_ACT_GEGLU) {
    if constexpr (NF % 2 == 0) {
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        const int m = m0 + wm * WTM + j * 16 + lr;
        if (m >= g.M) continue;
#pragma unroll
        for (int i = 0; i < NF; i += 2) {
          const int np = n0 + wn * WTN + i * 16 + lg * 4;  // packed row of the "a" half; gate rows are +16
          if (np + 16 >= g.N) continue;
          f4 av = acc[i][j], gv = acc[i + 1][j];
          if (g.bias) {
            av += *reinterpret_cast<const f4*>(g.bias + np);
            gv += *reinterpret_cast<const f4*>(g.bias + np + 16);
          }
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(av[r] * md::gelu_erf_f(gv[r]));
          const int oc = (n0 + wn * WTN) / 2 + (i / 2) * 16 + lg * 4;
          *reinterpret_cast<h4*>(reinterpret_cast<half_t*>(g.out) + (long long)m * g.ld_out + oc) = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int m = m0 + wm * WTM + j * 16 + lr;
    if (m >= g.M) continue;
    const int b = m / g.tokens;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int n = n0 + wn * WTN + i * 16 + lg * 4;
      if (n >= g.N) continue;
      epi_store4(g, m, b, n, acc[i][j]);
    }
  }
}

// Deterministic split-K reduction + epilogue: one thread per (m, 4 columns).
__global__ __launch_bounds__(256) void igemm_splitk_reduce(const IgemmArgs g) {
  const int n4 = g.N >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)g.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long long)m * n4) * 4;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < g.splitk; ++z) s += *reinterpret_cast<const f4*>(g.ws + ((long long)z * g.M + m) * g.N + n);
  epi_store4(g, m, m / g.tokens, n, s);
}

struct TileCfg {
  int bm, bn;
  float eff;
};
const TileCfg kCfgs[] = {{128, 128, 1.00f}, {128, 64, 0.85f}, {64, 128, 0.85f}, {64, 64, 0.70f}};
constexpr int kNumCfgs = 4;

template <int BM, int BN, int WMv, int WNv>
int launch_cfg(const IgemmArgs& g, hipStream_t s) {
  const size_t lds = 2 * (BM + BN) * 128;
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.splitk);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WMv, WNv>), grid, dim3(256), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

int validate(const md_igemm_params* p) {
  if (!p || !p->a0 || !p->w || !p->out) return MD_ERR_BAD_ARG;
  if (p->ksize != 1 && p->ksize != 3) return MD_ERR_UNSUPPORTED;
  if (p->stride != 1 && p->stride != 2) return MD_ERR_UNSUPPORTED;
  if (p->ups && (p->ksize != 3 || p->stride != 1)) return MD_ERR_UNSUPPORTED;
  if (p->c0 <= 0 || (p->c0 & 7) || p->c1 < 0 || (p->c1 & 7)) return MD_ERR_BAD_ARG;
  if ((p->c1 > 0) != (p->a1 != nullptr)) return MD_ERR_BAD_ARG;
  if (p->n <= 0 || (p->n & 3) || (p->ld_out & 3)) return MD_ERR_BAD_ARG;
  if (p->batch <= 0 || p->hin <= 0 || p->win <= 0 || p->hout <= 0 || p->wout <= 0) return MD_ERR_BAD_ARG;
  if (p->res && (p->ld_res & 3)) return MD_ERR_BAD_ARG;
  if (p->n_tr_begin < 0 || p->n_tr_begin > p->n || (p->n_tr_begin & 15)) return MD_ERR_BAD_ARG;
  if (p->n_tr_begin < p->n && (!p->out_t || p->ld_t <= 0)) return MD_ERR_BAD_ARG;
  if (p->bias_batch_stride & 3) return MD_ERR_BAD_ARG;
  if (p->act == MD_ACT_GEGLU) {
    if ((p->n & 31) || p->res || p->out_f32 || p->n_tr_begin != p->n || p->bias_batch_stride) return MD_ERR_UNSUPPORTED;
  } else if (p->act != MD_ACT_NONE && p->act != MD_ACT_SILU) {
    return MD_ERR_UNSUPPORTED;
  }
  return MD_OK;
}

// Pick tile config + split-K from a crude time model (tunable through force_cfg / force_splitk).
void choose(const md_igemm_params* p, long long M, int N, int K, long long ws_bytes, int* cfg_out, int* split_out) {
  const int nk = (K + 63) / 64;
  double best = 1e30;
  int bc = 3, bs = 1;
  for (int c = 0; c < kNumCfgs; ++c) {
    if (p->force_cfg >= 0 && c != p->force_cfg) continue;
    const long long tm = (M + kCfgs[c].bm - 1) / kCfgs[c].bm, tn = (N + kCfgs[c].bn - 1) / kCfgs[c].bn;
    const long long blocks = tm * tn;
    const double rate_cu = 2.5e15 / 256.0 * 0.35 * kCfgs[c].eff;  // flop/s per CU we expect from this tile
    for (int s = 1; s <= 32; s = (s < 4 ? s + 1 : s * 2)) {
      if (p->force_splitk > 0 && s != p->force_splitk) continue;
      if (s > 1) {
        if (p->act == MD_ACT_GEGLU) break;
        if ((long long)s * M * N * 4 > ws_bytes) break;
        if (nk / s < 4) break;
      }
      const int tps = (nk + s - 1) / s;
      const long long waves = (blocks * s + 255) / 256;
      double t = (double)waves * (2.0 * kCfgs[c].bm * kCfgs[c].bn * (double)tps * 64.0) / rate_cu + 2e-6;
      if (s > 1) t += (double)M * N * 4.0 * (s + 1) / 3e12 + 3e-6;
      if (t < best) {
        best = t;
        bc = c;
        bs = s;
      }
    }
  }
  if (p->force_cfg >= 0 && best > 1e29) bc = p->force_cfg;
  *cfg_out = bc;
  *split_out = bs;
}

}  // namespace

extern "C" int64_t md_igemm_workspace_bytes(const md_igemm_params* p) {
  if (validate(p) != MD_OK) return 0;
  const long long M = (long long)p->batch * p->hout * p->wout;
  // split-K is only ever worth it for small M; cap at 32 splits
  if (M > 2048) return 0;
  return (int64_t)32 * M * p->n * 4;
}

extern "C" int md_igemm(const md_igemm_params* p, void* stream) {
  const int v = validate(p);
  if (v != MD_OK) return v;
  IgemmArgs g;
  g.a0 = (const half_t*)p->a0;
  g.a1 = (const half_t*)p->a1;
  g.c0 = p->c0;
  g.c1 = p->c1;
  g.cin = p->c0 + p->c1;
  g.batch = p->batch;
  g.hin = p->hin;
  g.win = p->win;
  g.hout = p->hout;
  g.wout = p->wout;
  g.tokens = p->hout * p->wout;
  g.ksize = p->ksize;
  g.stride = p->stride;
  g.ups = p->ups;
  g.pad = p->ksize / 2;
  g.w = (const half_t*)p->w;
  const long long M = (long long)p->batch * g.tokens;
  if (M > 0x7fffffffLL) return MD_ERR_BAD_ARG;
  g.M = (int)M;
  g.N = p->n;
  g.K = p->ksize * p->ksize * g.cin;
  g.nk = (g.K + 63) / 64;
  g.bias = p->bias;
  g.bias_bs = p->bias_batch_stride;
  g.res = (const half_t*)p->res;
  g.ld_res = p->ld_res;
  g.act = p->act;
  g.out = p->out;
  g.ld_out = p->ld_out;
  g.out_f32 = p->out_f32;
  g.out_t = (half_t*)p->out_t;
  g.n_tr_begin = p->n_tr_begin;
  g.ld_t = p->ld_t;
  g.ws = (float*)p->ws;
  int cfg, split;
  choose(p, M, g.N, g.K, p->ws ? p->ws_bytes : 0, &cfg, &split);
  if (split > 1 && (!p->ws || (long long)split * M * g.N * 4 > p->ws_bytes)) return MD_ERR_WORKSPACE;
  if (p->act == MD_ACT_GEGLU && split > 1) return MD_ERR_UNSUPPORTED;
  g.splitk = split;
  g.tiles_per_split = (g.nk + split - 1) / split;
  g.tiles_m = (g.M + kCfgs[cfg].bm - 1) / kCfgs[cfg].bm;
  g.tiles_n = (g.N + kCfgs[cfg].bn - 1) / kCfgs[cfg].bn;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_IGEMM, s, 2.0 * (double)M * g.N * g.K,
                     (double)M * g.cin * 2.0 + (double)g.N * g.K * 2.0 + (double)M * g.N * 2.0);
  int rc;
  switch (cfg) {
    case 0: rc = launch_cfg<128, 128, 2, 2>(g, s); break;
    case 1: rc = launch_cfg<128, 64, 2, 2>(g, s); break;
    case 2: rc = launch_cfg<64, 128, 2, 2>(g, s); break;
    default: rc = launch_cfg<64, 64, 2, 2>(g, s); break;
  }
  if (rc != MD_OK) return rc;
  if (split > 1) {
    const long long work = M * (g.N >> 2);
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, g);
    MD_HIP_CHECK(hipGetLastError());
  }
  return MD_OK;
}
