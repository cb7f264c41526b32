// md_igemm: implicit-GEMM convolution / linear layer for gfx950 (MI355X), fp16 operands, fp32 MFMA accumulate.
//
//   D[n][m] = sum_k W[n][k] * A(m,k)        (operands swapped so a lane owns 4 consecutive n of one output row m:
//                                            NHWC epilogue = one 8-byte store per fragment, bias/residual vector loads)
//   m = (b, oy, ox)   k = tap*(c0+c1) + c   A gathered on the fly from one or two NHWC sources (channel concat),
//                                            with zero padding, stride 2 and nearest x2 upsample folded in the gather.
//
// Tile: BM x BN x 64 per k-group of 4 waves (WAVES_M x WAVES_N), v_mfma_f32_16x16x32_f16.  LDS holds, per k-group, two stages of
// KT x ([BM][64] + [BN][64]) fp16 filled by LDS-DMA, with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) applied to the SOURCE
// chunk so the fragment ds_read_b128 of 16 consecutive rows is conflict-free.  One vmcnt(0) + barrier per stage.
// Small-M layers (8x8 / 16x16 / 32x32 latents of a one-frame step) have too few output tiles for 256 CUs: their K dimension is
// split (a) across the 1 / 2 / 4 k-groups of one workgroup, combined through LDS, and (b) beyond that over blockIdx.z with fp32
// slabs and a deterministic reduce kernel that applies the same epilogue.
//
// Reference arithmetic replaced: see include/magicdance_hip.h (md_igemm).
#include <cstdio>
#include <cstdlib>

#include "md_common.h"

// 256 zero bytes: the source of out-of-image / out-of-range 16-byte chunks for the direct-to-LDS loader
__device__ __attribute__((aligned(256))) unsigned char md_zero_page[256];

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
  int tiles_m, tiles_n, group_m;
  unsigned div_tok_mul, div_tok_sh, div_w_mul, div_w_sh;  // exact n / tokens and n / wout for n < 2^24 (fast_div)
  // epilogue
  const float* bias;
  long long bias_bs;
  const half_t* res;
  const half_t* res_lo;
  int ld_res;
  int act;
  void* out;
  half_t* out_lo;
  float col_scale;
  int col_scale_end;
  unsigned char* k8;      // columns [k8_begin, k8_end) go here as e4m3 bytes ([M][ld_k8]) instead of to `out`
  int k8_begin, k8_end, ld_k8;
  int vt_fp8;             // the transposed columns (n >= n_tr_begin) are stored as e4m3 bytes
  int ld_out;
  int out_f32;
  half_t* out_t;
  int n_tr_begin;
  int ld_t;
  float* ws;
  // LayerNorm folded into the GEMM (A rows are normalised on the fly): out = rstd_m (acc - mu_m s1[n]) + s0[n]
  const float* ln_s1;
  const float* ln_s0;
  float ln_eps, ln_inv_k;
  // second parameter set: rows m >= m_split use w2 / bias2 / ln2_* and form their own m-tiles, tile_m >= tiles_m1 (INT_MAX: one set)
  const half_t* w2;
  const float* bias2;
  const float* ln2_s1;
  const float* ln2_s0;
  int m_split, tiles_m1;
  float* part;   // GroupNorm partial statistics [M / 64][2][N] (sum | sum of squares of the stored fp16 values), or nullptr
  int w_tiled;   // 1: W is stored [N / 16][k-tile in consumption order][16][64] (md_igemm_params.w_tiled), 0: row-major [N][K]
  int epi_stage; // 1: the fp16 epilogue goes through LDS and leaves as whole-row 16-byte stores (host: alignment / shape checks)
};

// n / d for n < 2^24: q = (n * mul) >> sh with mul = floor(2^sh / d) + 1, sh = 24 + ceil(log2 d) (host side below)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(((unsigned long long)(unsigned)n * mul) >> sh);
}

// Epilogue-only parameters.  The k-loop kernel reads them from the kernarg segment AFTER the loop (scalar loads behind an opaque
// pointer) into plain locals: as by-value kernel arguments they would be loaded at entry and held in SGPRs across the loop, which
// pushed the kernel over the SGPR budget (spill to scratch at entry, scratch reload at the head of every wave's epilogue); as a
// struct the compiler keeps them in scratch, hence the long-hand parameter list.
#define MD_LATE_PARAMS                                                                                                          \
  float e_col_scale, int e_col_scale_end, unsigned char *e_k8, int e_k8_begin, int e_k8_end, int e_ld_k8, int e_vt_fp8,        \
      const half_t *e_res_lo, half_t *e_out_lo, half_t *e_out_t, int e_n_tr_begin, int e_ld_t
#define MD_LATE_ARGS e_col_scale, e_col_scale_end, e_k8, e_k8_begin, e_k8_end, e_ld_k8, e_vt_fp8, e_res_lo, e_out_lo, e_out_t, e_n_tr_begin, e_ld_t
#define MD_LATE_LOAD(src)                                                                                                       \
  const float e_col_scale = (src).col_scale;                                                                                    \
  const int e_col_scale_end = (src).col_scale_end;                                                                              \
  unsigned char* const e_k8 = (src).k8;                                                                                         \
  const int e_k8_begin = (src).k8_begin, e_k8_end = (src).k8_end, e_ld_k8 = (src).ld_k8, e_vt_fp8 = (src).vt_fp8;               \
  const half_t* const e_res_lo = (src).res_lo;                                                                                  \
  half_t* const e_out_lo = (src).out_lo;                                                                                        \
  half_t* const e_out_t = (src).out_t;                                                                                          \
  const int e_n_tr_begin = (src).n_tr_begin, e_ld_t = (src).ld_t;

// (1) every global load the epilogue of (m, n..n+3) needs -- bias, residual, second residual term -- ISSUED together ...
__device__ __forceinline__ void epi_load(const IgemmArgs& g, MD_LATE_PARAMS, int m, int b, int n, f4& bv, h4& rv, h4& rl) {
  const float* bias = m >= g.m_split ? g.bias2 : g.bias;
  const bool row_major = n < e_n_tr_begin && !(e_k8 && n >= e_k8_begin && n < e_k8_end);   // lands in `out` (not V^T / e4m3 K)
  bv = f4{0.f, 0.f, 0.f, 0.f};
  rv = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  rl = rv;
  if (bias) bv = *reinterpret_cast<const f4*>(bias + (long long)b * g.bias_bs + n);
  if (g.res && row_major) {
    rv = *reinterpret_cast<const h4*>(g.res + (long long)m * g.ld_res + n);
    if (e_res_lo) rl = *reinterpret_cast<const h4*>(e_res_lo + (long long)m * g.ld_res + n);
  }
}

// (2) ... and consumed here: bias, column scale, activation, residual, store(s).
__device__ __forceinline__ void epi_finish(const IgemmArgs& g, MD_LATE_PARAMS, int m, int b, int n, f4 v, f4 bv, h4 rv, h4 rl) {
  v += bv;
  if (n < e_col_scale_end) v *= e_col_scale;   // attention scale folded into the q columns (before the fp16 rounding)
  if (g.act == MD_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = md::silu_f(v[i]);
  }
  if (n >= e_n_tr_begin) {
    // transposed store (V^T): [b][n - n_tr][tok]
    const int tok = m - b * g.tokens;
    const int ntr = g.N - e_n_tr_begin;
    if (e_vt_fp8) {   // e4m3 bytes (fp8 attention path)
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
      unsigned char* o8 = reinterpret_cast<unsigned char*>(e_out_t) + ((long long)b * ntr + (n - e_n_tr_begin)) * e_ld_t + tok;
#pragma unroll
      for (int i = 0; i < 4; ++i) o8[(long long)i * e_ld_t] = (unsigned char)((unsigned)w >> (8 * i));
      return;
    }
    half_t* o = e_out_t + ((long long)b * ntr + (n - e_n_tr_begin)) * e_ld_t + tok;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[(long long)i * e_ld_t] = (half_t)v[i];
    return;
  }
  if (e_k8 && n >= e_k8_begin && n < e_k8_end) {   // K columns of a fused q|k|v projection as e4m3 bytes
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    *reinterpret_cast<int*>(e_k8 + (long long)m * e_ld_k8 + (n - e_k8_begin)) = w;
    return;
  }
  if (g.res) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
    if (e_res_lo) {  // second term of the two-term residual stream: the chain value is res + res_lo
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)rl[i];
    }
  }
  if (g.out_f32) {
    *reinterpret_cast<f4*>(reinterpret_cast<float*>(g.out) + (long long)m * g.ld_out + n) = v;
  } else {
    h4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
    *reinterpret_cast<h4*>(reinterpret_cast<half_t*>(g.out) + (long long)m * g.ld_out + n) = o;
    if (e_out_lo) {  // what the fp16 store dropped, for the next link of the residual chain
      h4 l;
#pragma unroll
      for (int i = 0; i < 4; ++i) l[i] = (half_t)(v[i] - (float)o[i]);
      *reinterpret_cast<h4*>(e_out_lo + (long long)m * g.ld_out + n) = l;
    }
  }
}

// Store 4 consecutive output columns n..n+3 of row m (b = m / tokens precomputed): one memory round trip, then the stores.
__device__ __forceinline__ void epi_store4(const IgemmArgs& g, MD_LATE_PARAMS, int m, int b, int n, f4 v) {
  f4 bv;
  h4 rv, rl;
  epi_load(g, MD_LATE_ARGS, m, b, n, bv, rv, rl);
  epi_finish(g, MD_LATE_ARGS, m, b, n, v, bv, rv, rl);
}

// byte offset of W row n (k-tile 0): row-major [N][K], or the tiled form [N / 16][K / 64][16][64] -- 16 rows x 128 bytes of one
// k-tile are one contiguous 2 KiB block and a 16-row panel's k-tiles follow each other IN THE ORDER THE KERNEL CONSUMES THEM (3x3:
// channel block outer, tap inner), so a workgroup's weight stream is BN / 16 sequential streams instead of BN x 128-byte pieces
// K * 2 bytes apart (one DRAM page each)
__device__ __forceinline__ unsigned w_row_offset(int n, const IgemmArgs& g) {
  return g.w_tiled ? (unsigned)(n >> 4) * (unsigned)g.nk * 2048u + (unsigned)(n & 15) * 128u : (unsigned)n * (unsigned)g.K * 2u;
}

// 16-lane (one DPP row = the 16 lr lanes that share lg) sum, fixed order -> deterministic: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror.  Every lane of the row ends up with the row's sum.
__device__ __forceinline__ float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
#endif
  return v;
}

// Operands go global -> LDS directly (LDS-DMA, 1 KiB = 8 rows x 128 B per wave instruction, no VGPR round trip, no ds_write); the
// LDS image is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk instead (lane (r, c) fetches global chunk
// c ^ (r & 7)).
// LOADER 1 (ragged channel counts: the stem, the hint encoder, the VAE's 3 / 4-channel ends): global_load_lds with per-lane 64-bit
// pointers, padding / k-tails read md_zero_page.  Address generation is branch-free: per row a 9-bit tap-validity mask and the
// (y, x) of tap (0,0) are computed once; per k-tile a thread derives (tap, channel) of ITS fixed 16-byte k-chunk incrementally
// and selects pointer-or-zero.
// LOADER 2 (every layer whose sources are multiples of 64 channels): raw buffer loads -- three wave-uniform descriptors (a0, a1,
// w), per-lane 32-bit byte offsets, "out of image / past K" expressed as an out-of-range offset, which the hardware turns into
// zeros in LDS; tap / source / channel base of a 64-channel k-tile are wave-uniform scalars.
// KT = 64-deep k-tiles per pipeline stage (two stages): one vmcnt(0) + barrier round trip fetches KT k-tiles -- the k-loop of a
// small GEMM on cold weights is one HBM round trip per iteration.
// KS = 1: the launcher guarantees a 1x1 / linear layer on ONE source (ksize 1, stride 1, no upsample, c1 == 0): the per-tile
// LDS-DMA issue needs no tap / source / validity arithmetic at all -- row offsets are constants, the k offset is 128 bytes per tile.
// KS = 3 (round 3): a 3x3 conv on ONE source without upsample: no source select, no upsample form, no uniform branches between the
// barrier and the loads -- per tile two scalar multiplies for the tap offset and and / cmp / cndmask per row for the padding taps.
// KS = 4: two sources (channel concat), 3x3 or 1x1, no upsample: the same with one uniform source select.
// KG = k-groups per workgroup (round 3): the workgroup has 4 KG waves; group kg = wave / 4 runs the 2-stage loop above on the
// k-tiles kt_begin + kg, + kg + KG, ... of THE SAME output tile in its own LDS stages (one barrier serves all groups), and the
// groups' fp32 accumulators are summed through LDS in fixed order before the epilogue.  This is split-K without slabs, reduce
// kernel or a second launch: a one-frame layer has too few output tiles to fill 256 CUs with more than one 4-wave workgroup
// each, so its k-loop was a chain of L2 / HBM round trips with 4 waves per CU to hide them -- with KG = 4 a CU has four k-tiles
// in flight and four waves per SIMD issuing MFMAs for the same tile.
// GroupNorm partial statistics (g.part != nullptr, common fp16 epilogue only): per 64-row granule and output column the sum and
// the sum of squares of the fp16 values just stored, so that the GroupNorm that consumes this tensor needs no statistics pass.
template <int BM, int BN, int WAVES_M, int WAVES_N, int LOADER, bool LN = false, int KT = 1, int KS = 0, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm_kernel(const IgemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)  // device pass only: the buffer-descriptor builtins do not exist for the host stub
  constexpr bool BUF = LOADER == 2;    // buffer_load ... lds with hardware out-of-range -> 0 and 32-bit offsets
  static_assert(LOADER == 1 || LOADER == 2, "LDS-DMA loaders only");
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per k-group");
  static_assert(!LN || BUF, "LayerNorm folding is instantiated for the buffer loader");
  static_assert(KG == 1 || BUF, "k-groups exist for the buffer loader");
  static_assert(KS == 0 || BUF, "the specialised issue paths belong to the buffer loader");
  static_assert(KS == 0 || KS == 1 || KS == 3 || KS == 4, "issue path: generic, 1x1 on one source, 3x3 on one source, two sources");
  constexpr bool KS1 = KS == 1;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int AJ = BM / 32, WJ = (BN + 31) / 32;  // 16-byte chunks per thread per k-tile
  // BN % 32 == 16 (the 80-wide SD tiles): the last W pass covers 16 rows = waves 0 and 1 only (a wave serves 8 rows)
  constexpr bool W_TAIL = (BN % 32) != 0;
  static_assert(!W_TAIL || BUF, "ragged BN is implemented for the buffer loader only");
  static_assert(BM % 32 == 0 && BN % 16 == 0 && WTM <= 64, "tile shape");
  constexpr int STAGE_BYTES = (BM + BN) * 128;   // bytes of ONE k-tile slot
  constexpr int GROUP_BYTES = 2 * KT * STAGE_BYTES;
  static_assert(KT == 1 || BUF, "multi-tile stages exist for the buffer loader");
  static_assert((KG - 1) * NF * MF * 4096 + (LN ? (KG - 1) * MF * 2 * 1024 : 0) <= KG * GROUP_BYTES, "cross-group reduction fits the stage memory");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x & 255;   // thread within its k-group
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave within the group
  const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // k-group of this wave
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  char* const gsm = smem + kg * GROUP_BYTES;   // this group's stages

  // XCD-aware bijective remap: consecutive logical tiles (same weight panel) share one XCD's L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // grouped raster inside the XCD's contiguous range: group_m m-tiles x all n-tiles are consecutive, so the ~64 workgroups
  // resident on one XCD at a time share group_m A tiles and tiles_n W panels (both re-read from that XCD's L2)
  const int per_group = g.group_m * g.tiles_n;
  const int grp = logical / per_group, in_grp = logical - grp * per_group;
  const int first_m = grp * g.group_m;
  const int gsz = min(g.tiles_m - first_m, g.group_m);
  const int tile_n = in_grp / gsz, tile_m = first_m + (in_grp - tile_n * gsz);
  // two parameter sets = two GEMMs stacked along M: the second one's tiles start AT m_split (not at a multiple of BM), the
  // first one's rows end there
  const bool set2 = tile_m >= g.tiles_m1;
  const int m0 = set2 ? g.m_split + (tile_m - g.tiles_m1) * BM : tile_m * BM;
  const int Mlim = set2 ? g.M : min(g.M, g.m_split);
  const int n0 = tile_n * BN;
  const int kz = blockIdx.z;
  const int kt_begin = kz * g.tiles_per_split;
  const int kt_end = min(g.nk, kt_begin + g.tiles_per_split);
  // k-tiles of this group: kt_first, kt_first + KG, ... < kt_end; every group runs the loop of group 0 (the longest)
  const int kt_first = kt_begin + kg;
  const int n_mine = kt_first < kt_end ? (kt_end - kt_first + KG - 1) / KG : 0;
  const int n_max = kt_begin < kt_end ? (kt_end - kt_begin + KG - 1) / KG : 0;
  // parameter set of this tile
  const half_t* const gw = set2 ? g.w2 : g.w;
  [[maybe_unused]] const float* const gbias = set2 ? g.bias2 : g.bias;
  [[maybe_unused]] const float* const gln_s1 = set2 ? g.ln2_s1 : g.ln_s1;
  [[maybe_unused]] const float* const gln_s0 = set2 ? g.ln2_s0 : g.ln_s0;

  // ---- loader role: LDS slot (row, lc) for rows lrow + 32 j; fixed global k-chunk gc -------------------------
  const int lc = tid & 7, lrow = tid >> 3;
  // source chunk held at LDS position lc of this thread's rows (lrow + 32 j: the swizzle key is the same for all of them)
  const int gc = lc ^ (lrow & 7);
  const int ups = g.ups;           // 0 / 1: source coordinate = virtual coordinate >> ups
  // buffer loader: the W part of the FIRST k-tile depends on nothing computed below, so its LDS-DMA is issued before the per-row
  // im2col setup (a few hundred VALU): the HBM latency of the layer's cold weights overlaps it
  [[maybe_unused]] bool skip_w_once = false;
  if constexpr (BUF) {
    if (n_mine > 0) {
      int tap0 = 0, cc0 = kt_first * 64;
      if (g.ksize == 3) {
        const int cb = kt_first / 9;
        tap0 = kt_first - cb * 9;
        cc0 = cb * 64;
      }
      const unsigned ksoff0 = g.w_tiled ? (unsigned)kt_first * 2048u : (unsigned)(tap0 * g.cin + cc0) * 2u;
      const __amdgpu_buffer_rsrc_t rs_w0 =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(gw), 0, g.N * g.K * 2, 0x00020000);
      char* Ws0 = gsm + BM * 128;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        if (W_TAIL && j == WJ - 1 && wave >= 2) break;
        const unsigned wo = w_row_offset(min(n0 + lrow + 32 * j, g.N - 1), g) + (unsigned)gc * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w0, (__attribute__((address_space(3))) void*)(Ws0 + (32 * j + 8 * wave) * 128),
                                                 16, wo, ksoff0, 0, 0);
      }
      skip_w_once = true;
    }
  }
  const int vh = g.hin << ups, vw = g.win << ups;
  int a_y[AJ], a_x[AJ], a_pix[AJ], a_mask[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = min(m0 + lrow + 32 * j, Mlim - 1);  // rows past the end are computed on a clamped row and never stored
    const int b = fast_div(m, g.div_tok_mul, g.div_tok_sh);
    const int rem = m - b * g.tokens;
    const int oy = fast_div(rem, g.div_w_mul, g.div_w_sh);
    const int ox = rem - oy * g.wout;
    a_y[j] = oy * g.stride - g.pad;
    a_x[j] = ox * g.stride - g.pad;
    a_pix[j] = b * g.hin * g.win;
    int mask = 1;
    if (g.ksize == 3) {  // bit t = tap (t/3, t%3) inside the (virtual) image: 3 column bits replicated per valid row
      int cx = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) cx |= ((unsigned)(a_x[j] + d) < (unsigned)vw) ? (1 << d) : 0;
      mask = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) mask |= ((unsigned)(a_y[j] + d) < (unsigned)vh) ? (cx << (3 * d)) : 0;
    }
    a_mask[j] = mask;
  }
  [[maybe_unused]] const half_t* w_ptr[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) w_ptr[j] = gw + (long long)min(n0 + lrow + 32 * j, g.N - 1) * g.K;
  [[maybe_unused]] const half_t* zero = reinterpret_cast<const half_t*>(md_zero_page);

  // LOADER 1: (tap, channel) of this thread's k-chunk, advanced by 64 channels per k-tile
  [[maybe_unused]] int k_cur = kt_begin * 64 + gc * 8;
  [[maybe_unused]] int tap = 0, cc = k_cur;
  if (!BUF && g.ksize == 3) {
    tap = k_cur / g.cin;
    cc = k_cur - tap * g.cin;
  }
  // ---- BUF loader state: per-row byte offsets of the tap-centre pixel (+ this thread's chunk) in either source, relative
  // to descriptors whose base is shifted back by pad*(win+1) pixels so that every tap offset is >= 0.  Per k-tile the
  // address of row j is  rowbase[j] (VGPR, constant)  +  soffset (SGPR: tap, channel base)  -> no per-tile address VALU
  // beyond the validity select (mask bit ? rowbase : OOB).
  [[maybe_unused]] unsigned w_off[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) w_off[j] = w_row_offset(min(n0 + lrow + 32 * j, g.N - 1), g) + (unsigned)gc * 16u;
  [[maybe_unused]] unsigned rowbase0[AJ], rowbase1[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const unsigned centre = (unsigned)(a_pix[j] + (a_y[j] + g.pad) * g.win + (a_x[j] + g.pad));
    rowbase0[j] = centre * (unsigned)g.c0 * 2u + (unsigned)gc * 16u;
    rowbase1[j] = centre * (unsigned)g.c1 * 2u + (unsigned)gc * 16u;
  }
  const int shift_pix = ups ? 0 : g.pad * (g.win + 1);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(g.a0) - (long long)shift_pix * g.c0, 0, (g.batch * g.hin * g.win + shift_pix) * g.c0 * 2, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(g.a1 ? g.a1 : g.a0) - (long long)shift_pix * (g.a1 ? g.c1 : g.c0), 0,
      (g.batch * g.hin * g.win + shift_pix) * (g.a1 ? g.c1 : g.c0) * 2, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(gw), 0, g.N * g.K * 2, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;  // beyond any tensor: the load returns zeros

  // BUF loader: the launcher guarantees cin % 64 == 0 and c0 % 64 == 0, so a 64-channel k-tile lies in ONE tap of ONE
  // source: tap / source / channel base are wave-uniform scalars advanced per tile (no waterfall on the descriptor).
  // k-tile ORDER of the 3x3 convs: channel-block outer, tap inner (k-tile t -> channels 64*(t/9).., tap t%9).  The nine
  // taps of one 64-channel block re-read the same (BM + halo) pixels x 128 B, so the A operand is served from L1/L2 instead
  // of being re-streamed from the Infinity Cache nine times per pass (the tap-outer order had a reuse distance of a whole
  // Cin sweep, ~BM*Cin*2 B per workgroup, x 64 resident workgroups per XCD >> the 4 MB L2).  W's k offset follows.
  // A group's tile sequence advances by KG tiles per fetch.
  int kt_i = kt_first, tap_u = 0, cc_u = kt_first * 64;
  if (g.ksize == 3) {
    const int cb = kt_first / 9;
    tap_u = kt_first - cb * 9;
    cc_u = cb * 64;
  }
  auto fetch_tile = [&](int slot) {  // loads this group's next tile (if it has one) into ``slot``, then advances
    char* As = gsm + slot * STAGE_BYTES;
    char* Ws = As + BM * 128;
    if constexpr (BUF) {
      const bool kvalid = kt_i < kt_end;  // uniform (K % 64 == 0 here)
      if (kvalid) {
        if constexpr (KS1) {
          const unsigned soff1 = (unsigned)kt_i * 128u;   // k-tile kt_i = channels 64 kt_i .. of the only tap of the only source
          const unsigned soffw = g.w_tiled ? (unsigned)kt_i * 2048u : soff1;
#pragma unroll
          for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128), 16,
                                                     rowbase0[j], soff1, 0, 0);
          if (skip_w_once) {
            skip_w_once = false;
          } else {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
              if (W_TAIL && j == WJ - 1 && wave >= 2) break;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Ws + (32 * j + 8 * wave) * 128), 16,
                                                       w_off[j], soffw, 0, 0);
            }
          }
        } else if constexpr (KS == 3) {
          const int dy = (tap_u * 11) >> 5, dx = tap_u - dy * 3;
          const int tapbit = 1 << tap_u;
          const unsigned soff = ((unsigned)(dy * g.win + dx) * (unsigned)g.c0 + (unsigned)cc_u) * 2u;
#pragma unroll
          for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128), 16,
                                                     (a_mask[j] & tapbit) ? rowbase0[j] : OOB, soff, 0, 0);
          const unsigned ksoff = g.w_tiled ? (unsigned)kt_i * 2048u : (unsigned)(tap_u * g.cin + cc_u) * 2u;
          if (skip_w_once) {
            skip_w_once = false;
          } else {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
              if (W_TAIL && j == WJ - 1 && wave >= 2) break;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Ws + (32 * j + 8 * wave) * 128), 16,
                                                       w_off[j], ksoff, 0, 0);
            }
          }
        } else if constexpr (KS == 4) {   // channel concat of two sources (skip connections), 3x3 or 1x1 (tap_u stays 0), no upsample
          const int dy = (tap_u * 11) >> 5, dx = tap_u - dy * 3;
          const int tapbit = 1 << tap_u;
          const bool second = cc_u >= g.c0;
          const unsigned cs = second ? g.c1 : g.c0;
          const unsigned soff = ((unsigned)(dy * g.win + dx) * cs + (unsigned)(second ? cc_u - g.c0 : cc_u)) * 2u;
          if (second) {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128),
                                                       16, (a_mask[j] & tapbit) ? rowbase1[j] : OOB, soff, 0, 0);
          } else {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128),
                                                       16, (a_mask[j] & tapbit) ? rowbase0[j] : OOB, soff, 0, 0);
          }
          const unsigned ksoff = g.w_tiled ? (unsigned)kt_i * 2048u : (unsigned)(tap_u * g.cin + cc_u) * 2u;
          if (skip_w_once) {
            skip_w_once = false;
          } else {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
              if (W_TAIL && j == WJ - 1 && wave >= 2) break;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Ws + (32 * j + 8 * wave) * 128), 16,
                                                       w_off[j], ksoff, 0, 0);
            }
          }
        } else {
          const int dy = (tap_u * 11) >> 5, dx = tap_u - dy * 3;
          const bool second = cc_u >= g.c0;
          const unsigned cs = second ? g.c1 : g.c0;
          const unsigned cbase = (unsigned)(second ? cc_u - g.c0 : cc_u);
          const int tapbit = 1 << tap_u;
          unsigned voff[AJ];
          unsigned soff = 0;
          if (ups) {  // nearest x2 upsample: the source pixel is not affine in the tap -> full per-lane offset
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
              const int sy = (a_y[j] + dy) >> 1, sx = (a_x[j] + dx) >> 1;
              const unsigned pix = (unsigned)(a_pix[j] + (int)__mul24(sy, g.win) + sx);
              const unsigned v = (__umul24(pix, cs) + cbase + (unsigned)gc * 8u) * 2u;
              voff[j] = (a_mask[j] & tapbit) ? v : OOB;
            }
          } else {
            soff = ((unsigned)(dy * g.win + dx) * cs + cbase) * 2u;
#pragma unroll
            for (int j = 0; j < AJ; ++j) voff[j] = (a_mask[j] & tapbit) ? (second ? rowbase1[j] : rowbase0[j]) : OOB;
          }
          if (second) {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128),
                                                       16, voff[j], soff, 0, 0);
          } else {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128),
                                                       16, voff[j], soff, 0, 0);
          }
          // W column of (tap, channel block); ksize 1: tap_u == 0.  Tiled W: the k-tiles are stored in consumption order
          const unsigned ksoff = g.w_tiled ? (unsigned)kt_i * 2048u : (unsigned)(tap_u * g.cin + cc_u) * 2u;
          if (skip_w_once) {
            skip_w_once = false;  // the first tile's W part is already in flight (issued ahead of the row setup)
          } else {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
              if (W_TAIL && j == WJ - 1 && wave >= 2) break;  // wave-uniform: rows 32 j + 8 wave .. + 7 lie beyond BN
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Ws + (32 * j + 8 * wave) * 128),
                                                       16, w_off[j], ksoff, 0, 0);
            }
          }
        }
      }
      kt_i += KG;
      if (KS == 3 || (KS != 1 && g.ksize == 3)) {
        tap_u += KG;
        if (tap_u >= 9) {
          tap_u -= 9;
          cc_u += 64;
        }
      } else {
        cc_u += 64 * KG;
      }
      return;
    } else {
      const bool kvalid = kt_i < kt_end && k_cur < g.K;
      ++kt_i;
      const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
      const bool second = cc >= g.c0;
      const int cs = second ? g.c1 : g.c0;
      const int ccc = second ? cc - g.c0 : cc;
      {
        const half_t* src = second ? g.a1 : g.a0;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const int sy = (a_y[j] + dy) >> ups, sx = (a_x[j] + dx) >> ups;
          // branch-free pointer-or-zero-page: mask the element offset, select the base
          const long long okmask = -(long long)((kvalid ? (a_mask[j] >> tap) : 0) & 1);
          const long long off = ((long long)(a_pix[j] + sy * g.win + sx) * cs + ccc) & okmask;
          const half_t* p = (okmask ? src : zero) + off;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                           (__attribute__((address_space(3))) void*)(As + (32 * j + 8 * wave) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          const half_t* p = (kvalid ? w_ptr[j] : zero) + (kvalid ? k_cur : 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                           (__attribute__((address_space(3))) void*)(Ws + (32 * j + 8 * wave) * 128), 16, 0, 0);
        }
      }
      k_cur += 64;
      cc += 64;
      if (g.ksize == 3) {
        while (cc >= g.cin) {
          cc -= g.cin;
          ++tap;
        }
      }
    }
  };

  f4 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // LN: per-lane partial sum / sum of squares of the A rows this lane reads as MFMA operands (row i*16+lr, k-chunks lg and
  // 4+lg of every 64-wide tile = a quarter of K); the four lg lanes (and the k-groups) are combined after the k-loop.
  [[maybe_unused]] float ln_sum[MF], ln_sq[MF];
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < MF; ++i) ln_sum[i] = ln_sq[i] = 0.f;
  }

  auto no_mid = [] {};
  auto compute_tile = [&](int slot, auto&& mid) {
    const char* As = gsm + slot * STAGE_BYTES;
    const char* Ws = As + BM * 128;
    // Both 32-deep k-steps of the tile: ALL operand fragments are requested from LDS before the first MFMA (the compiler, left
    // alone, keeps one ds_read_b128 in flight per pair of MFMAs -- ~70 idle MFMA cycles per pair at one or two waves per SIMD),
    // and ``mid`` (the next tile's LDS-DMA issue: address VALU + buffer_load..lds) runs while the first fragments are in flight.
    h8 af[2][MF], wf[2][NF];
    auto load_frags = [&](int ks) {
      const int chunk = ks * 4 + lg;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int row = wm * WTM + i * 16 + lr;
        af[ks][i] = *reinterpret_cast<const h8*>(As + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int row = wn * WTN + i * 16 + lr;
        wf[ks][i] = *reinterpret_cast<const h8*>(Ws + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
    };
    load_frags(0);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (LN) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const h2v ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int i = 0; i < MF; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const h2v p = {af[ks][i][2 * e], af[ks][i][2 * e + 1]};
            ln_sum[i] = __builtin_amdgcn_fdot2(p, ones, ln_sum[i], false);
            ln_sq[i] = __builtin_amdgcn_fdot2(p, p, ln_sq[i], false);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- k-loop: two stages of KT tiles per group; iteration ``it`` computes the group's tiles it*KT .. it*KT + KT - 1 ----------
  if (n_max > 0) {
#pragma unroll
    for (int u = 0; u < KT; ++u) fetch_tile(u);
  }
  const int iters = (n_max + KT - 1) / KT;
  for (int it = 0; it < iters; ++it) {
    const int stage = it & 1;
    // explicit drain of this wave's LDS-DMA: the compiler's own wait before a barrier is not reliable for
    // buffer_load..lds (see attention.hip), and a tile read before it has landed is a silent, rare corruption
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the stage's tiles have landed for every wave; every wave is done with the other stage
    // the refill of the other stage is issued from INSIDE the first tile's compute, behind its first fragment reads
    auto refill = [&] {
      if (it + 1 < iters) {
#pragma unroll
        for (int u = 0; u < KT; ++u) fetch_tile((stage ^ 1) * KT + u);
      }
    };
    if (it * KT < n_mine)
      compute_tile(stage * KT, refill);
    else
      refill();
#pragma unroll
    for (int u = 1; u < KT; ++u)
      if (it * KT + u < n_mine) compute_tile(stage * KT + u, no_mid);
  }

  // ---- k-groups: fixed-order sum of the groups' accumulators (and LayerNorm row sums) through LDS ------------------------------
  if constexpr (KG > 1) {
    __syncthreads();   // every group is done reading its stages
    f4* const red = reinterpret_cast<f4*>(smem);
    float* const red_ln = reinterpret_cast<float*>(smem + (KG - 1) * NF * MF * 4096);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) red[((kg - 1) * NF * MF + i * MF + j) * 256 + tid] = acc[i][j];
      if constexpr (LN) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
          red_ln[((kg - 1) * MF * 2 + 2 * j) * 256 + tid] = ln_sum[j];
          red_ln[((kg - 1) * MF * 2 + 2 * j + 1) * 256 + tid] = ln_sq[j];
        }
      }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int q = 1; q < KG; ++q) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int j = 0; j < MF; ++j) acc[i][j] += red[((q - 1) * NF * MF + i * MF + j) * 256 + tid];
        if constexpr (LN) {
#pragma unroll
          for (int j = 0; j < MF; ++j) {
            ln_sum[j] += red_ln[((q - 1) * MF * 2 + 2 * j) * 256 + tid];
            ln_sq[j] += red_ln[((q - 1) * MF * 2 + 2 * j + 1) * 256 + tid];
          }
        }
      }
    }
  }
  const bool epi = kg == 0;   // the epilogue belongs to group 0; the other groups only keep the (uniform) barriers below company

  // ---- epilogue ---------------------------------------------------------------------------------------
  const __attribute__((address_space(4))) IgemmArgs* gp = (const __attribute__((address_space(4))) IgemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(gp));   // opaque: the loads below cannot move above the k-loop
  MD_LATE_LOAD(*gp)
  float* const e_part = gp->part;
  if (!epi && !e_part && !g.epi_stage) return;
  if constexpr (LN) {
    // row statistics over the full K (the launcher forbids split-K here), then acc <- rstd (acc - mu s1[n]) + s0[n]:
    // LayerNorm(x) W^T + b with gamma folded into W, s1[n] = sum_k gamma_k W[n][k], s0[n] = sum_k beta_k W[n][k] + b[n]
    if (epi) {
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        float sm = ln_sum[j], sq = ln_sq[j];
        sm += __shfl_xor(sm, 16, 64);
        sq += __shfl_xor(sq, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float mu = sm * g.ln_inv_k;
        const float rstd = rsqrtf(fmaxf(sq * g.ln_inv_k - mu * mu, 0.f) + g.ln_eps);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int n = min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 4);
          const f4 s1 = *reinterpret_cast<const f4*>(gln_s1 + n), s0 = *reinterpret_cast<const f4*>(gln_s0 + n);
          acc[i][j] = rstd * (acc[i][j] - mu * s1) + s0;
        }
      }
    }
  }
  if (g.splitk > 1) {
    if (!epi) return;
#pragma unroll
    for (int j = 0; j < MF; ++j) {
      const int m = m0 + wm * WTM + j * 16 + lr;
      if (m >= Mlim) continue;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int n = n0 + wn * WTN + i * 16 + lg * 4;
        if (n >= g.N) continue;
        *reinterpret_cast<f4*>(g.ws + ((long long)kz * g.M + m) * g.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (g.act == MD_ACT_GEGLU && g.epi_stage) {
    // GEGLU through LDS (see the staged epilogue below): a * gelu(gate) as fp16 in the fragment layout -> LDS -> whole-row 16-byte
    // stores ([M][N / 2] output, BN / 2 columns per tile)
    if constexpr (NF % 2 == 0) {
      constexpr int SROWH = BN / 2 + 8;   // fp16 row stride (16-byte aligned rows)
      constexpr int CHG = BN / 16;        // 16-byte pieces per tile row
      constexpr int NT = 256 * KG;
      constexpr int U = (BM * CHG + NT - 1) / NT;
      static_assert(BM * SROWH * 2 <= KG * GROUP_BYTES, "staged GEGLU tile fits the stage memory");
      half_t* const stg = reinterpret_cast<half_t*>(smem);
      __syncthreads();
      if (epi) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
#pragma unroll
          for (int i = 0; i < NF; i += 2) {
            const int np = min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 20);
            f4 av = acc[i][j], gv = acc[i + 1][j];
            if (gbias) {
              av += *reinterpret_cast<const f4*>(gbias + np);
              gv += *reinterpret_cast<const f4*>(gbias + np + 16);
            }
            h4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)(av[r] * md::gelu_erf_f(gv[r]));
            *reinterpret_cast<h4*>(stg + (wm * WTM + j * 16 + lr) * SROWH + (wn * WTN) / 2 + (i / 2) * 16 + lg * 4) = o;
          }
        }
      }
      __syncthreads();
      half_t* __restrict__ const outp = reinterpret_cast<half_t*>(g.out);
      const int nh = g.N >> 1;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = (int)threadIdx.x + NT * u;
        const int r = c / CHG, cc = c - r * CHG;
        const int m = m0 + r, oc = (n0 >> 1) + cc * 8;
        if (c >= BM * CHG || m >= Mlim || oc >= nh) continue;
        *reinterpret_cast<h8*>(outp + (long long)m * g.ld_out + oc) = *reinterpret_cast<const h8*>(stg + r * SROWH + cc * 8);
      }
    }
    return;
  }
  if (g.act == MD_ACT_GEGLU) {
    if (!epi) return;
    if constexpr (NF % 2 == 0) {
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        const int m = m0 + wm * WTM + j * 16 + lr;
        if (m >= Mlim) continue;
#pragma unroll
        for (int i = 0; i < NF; i += 2) {
          const int np = n0 + wn * WTN + i * 16 + lg * 4;  // packed row of the "a" half; gate rows are +16
          if (np + 16 >= g.N) continue;
          f4 av = acc[i][j], gv = acc[i + 1][j];
          if (gbias) {
            av += *reinterpret_cast<const f4*>(gbias + np);
            gv += *reinterpret_cast<const f4*>(gbias + np + 16);
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
  if (g.epi_stage) {
    // LDS-staged epilogue (round 3).  In the fragment layout a lane owns 4 consecutive columns of 16 DIFFERENT rows: every
    // residual load / output store instruction of a wave is 64 x 8 bytes spread over 16 rows (32-byte pieces of 16 cache lines),
    // and a 128 x 160 tile with the two-term residual stream needs 80 such instructions per lane -- on the short-K GEMMs (a
    // 5-tile k-loop per output tile) the vector-memory ISSUE of the epilogue, not the MFMAs, the L2 or HBM, bounded the kernel.
    // Here phase 1 parks bias + activation (fp32, fragment layout) in the stage memory, phase 2 walks the tile row-major: a
    // thread owns 8 consecutive columns of one row = ONE 16-byte load per residual term and ONE 16-byte store per output term,
    // consecutive lanes on consecutive 16-byte pieces of a row (whole 128-byte lines per instruction).
    constexpr int LDS_TOTAL = KG * GROUP_BYTES;
    constexpr int SROW = BN + 4;                  // fp32 row stride: +4 keeps the 16 lanes of a DPP row on distinct banks
    constexpr int ROUNDS = (BM * SROW * 4 <= LDS_TOTAL) ? 1 : 2;
    static_assert(ROUNDS == 1 || (WAVES_M % 2 == 0 && (BM / 2) * SROW * 4 <= LDS_TOTAL), "staged epilogue fits the stage memory");
    constexpr int RR = BM / ROUNDS;               // tile rows per round
    constexpr int WPR = WAVES_M / ROUNDS;         // wave rows per round
    constexpr int CH = BN / 8;                    // 16-byte output pieces per tile row
    constexpr int NT = 256 * KG;
    constexpr int U = (RR * CH + NT - 1) / NT;
    float* const stg = reinterpret_cast<float*>(smem);
    const float* __restrict__ const bp = gbias;
    const half_t* __restrict__ const resp = g.res;
    const half_t* __restrict__ const rlp = e_res_lo;
    half_t* __restrict__ const outp = reinterpret_cast<half_t*>(g.out);
    half_t* __restrict__ const olp = e_out_lo;
    const int tall = (int)threadIdx.x;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      __syncthreads();   // the stage memory is free: k-loop reads (rd = 0) / the previous round's phase 2 are done
      if (epi && wm / WPR == rd) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
          const int m = m0 + wm * WTM + j * 16 + lr;
          const int b = fast_div(min(m, Mlim - 1), g.div_tok_mul, g.div_tok_sh);
          f4 bv[NF];
#pragma unroll
          for (int i = 0; i < NF; ++i) bv[i] = f4{0.f, 0.f, 0.f, 0.f};
          if (bp) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
              bv[i] = *reinterpret_cast<const f4*>(bp + (long long)b * g.bias_bs + min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 4));
          }
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            f4 v = acc[i][j] + bv[i];
            if (n0 + wn * WTN + i * 16 + lg * 4 < e_col_scale_end) v *= e_col_scale;
            if (g.act == MD_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = md::silu_f(v[e]);
            }
            *reinterpret_cast<f4*>(stg + ((wm % WPR) * WTM + j * 16 + lr) * SROW + wn * WTN + i * 16 + lg * 4) = v;
          }
        }
      }
      // phase 2 in batches of UB pieces per thread (register budget: all of a 128 x 160 tile's 5 pieces in flight cost a wave
      // per SIMD); the first batch's residual loads do not depend on phase 1 and are in flight across the barrier
      constexpr int UB = U < 3 ? U : 3;
#pragma unroll
      for (int u0 = 0; u0 < U; u0 += UB) {
        h8 rv[UB], rl[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int c = tall + NT * (u0 + u);
          const int r = c / CH, cc = c - r * CH;
          const int mc = min(m0 + rd * RR + r, Mlim - 1), nc = min(n0 + cc * 8, g.N - 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[u][e] = (half_t)0.f;
          rl[u] = rv[u];
          if (u0 + u < U && resp) {
            rv[u] = *reinterpret_cast<const h8*>(resp + (long long)mc * g.ld_res + nc);
            if (rlp) rl[u] = *reinterpret_cast<const h8*>(rlp + (long long)mc * g.ld_res + nc);
          }
        }
        if (u0 == 0) __syncthreads();
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (u0 + u >= U) continue;
          const int c = tall + NT * (u0 + u);
          const int r = c / CH, cc = c - r * CH;
          const int m = m0 + rd * RR + r, n = n0 + cc * 8;
          if (c >= RR * CH || m >= Mlim || n >= g.N) continue;
          const float* sp = stg + r * SROW + cc * 8;
          const f4 v0 = *reinterpret_cast<const f4*>(sp), v1 = *reinterpret_cast<const f4*>(sp + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          h8 o, l;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] += (float)rv[u][e];
            v[e] += (float)rl[u][e];
            o[e] = (half_t)v[e];
            l[e] = (half_t)(v[e] - (float)o[e]);
          }
          *reinterpret_cast<h8*>(outp + (long long)m * g.ld_out + n) = o;
          if (olp) *reinterpret_cast<h8*>(olp + (long long)m * g.ld_out + n) = l;
        }
      }
    }
    return;
  }
  if (!g.out_f32 && e_n_tr_begin >= g.N && !e_k8) {
    // Common case (fp16 row-major output: every conv, the attention / feed-forward output projections, proj_in / proj_out): the
    // loads of a whole column of fragments -- bias, residual, second residual term -- are issued together, on clamped addresses,
    // before any is consumed.  (Fragment by fragment through epi_store4 the compiler emits load -> s_waitcnt vmcnt(0) -> store
    // chains: 3 x NF x MF serial memory round trips per wave, more than the k-loop of a 5-tile GEMM.)
    const bool has_bias = gbias != nullptr;
    // restrict: `res` / `res_lo` are either disjoint from `out` / `out_lo` or IDENTICAL to them (in-place residual add: a lane
    // reads exactly the elements it then writes, ordered by the data dependence) -- never partially overlapping (header contract).
    // (Measured and dropped, same-box A/B in profiles/round2_igemm_epilogue_ab.txt: two columns in flight with unconditional
    //  zero-page loads for absent operands -- 3-8 % slower on the epilogue-dominated GEMMs, end to end -0.4 %.)
    const float* __restrict__ const bp = gbias;
    const half_t* __restrict__ const resp = g.res;
    const half_t* __restrict__ const rlp = e_res_lo;
    half_t* __restrict__ const outp = reinterpret_cast<half_t*>(g.out);
    half_t* __restrict__ const olp = e_out_lo;
    // GroupNorm partials: per 16-row fragment the lanes' values are summed over the fragment's rows at once (DPP, fixed order) and
    // parked in LDS [wave row][fragment][column][sum | sumsq] -- nothing is carried in registers across fragments (accumulating in
    // registers cost 40 VGPRs and a wave per SIMD of occupancy on the 160-wide tiles, whether or not partials were requested)
    float* const pred = reinterpret_cast<float*>(smem);
    if (e_part) __syncthreads();   // every wave is done with the k-loop's (and the k-group reduction's) LDS reads
    if (epi) {
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        const int m = m0 + wm * WTM + j * 16 + lr;
        const int mc = min(m, Mlim - 1);
        const int b = fast_div(mc, g.div_tok_mul, g.div_tok_sh);
        f4 bv[NF];
        h4 rv[NF], rl[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          bv[i] = f4{0.f, 0.f, 0.f, 0.f};
          rv[i] = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
          rl[i] = rv[i];
        }
        if (has_bias) {   // (bias_bs == 0: the same vector for every row -- served by L1)
#pragma unroll
          for (int i = 0; i < NF; ++i)
            bv[i] = *reinterpret_cast<const f4*>(bp + (long long)b * g.bias_bs + min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 4));
        }
        if (resp) {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            rv[i] = *reinterpret_cast<const h4*>(resp + (long long)mc * g.ld_res + min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 4));
          if (rlp) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
              rl[i] = *reinterpret_cast<const h4*>(rlp + (long long)mc * g.ld_res + min(n0 + wn * WTN + i * 16 + lg * 4, g.N - 4));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool row_ok = m < Mlim;
        if (!row_ok && !e_part) continue;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int n = n0 + wn * WTN + i * 16 + lg * 4;
          if (n >= g.N) continue;   // (uniform over the 16 lanes of a DPP row: they share lg)
          f4 v = acc[i][j];
          v += bv[i];
          if (n < e_col_scale_end) v *= e_col_scale;
          if (g.act == MD_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = md::silu_f(v[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[i][e];   // (zeros when there is no residual / no second term)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rl[i][e];
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
          if (row_ok) {
            *reinterpret_cast<h4*>(outp + (long long)m * g.ld_out + n) = o;
            if (olp) {
              h4 l;
#pragma unroll
              for (int e = 0; e < 4; ++e) l[e] = (half_t)(v[e] - (float)o[e]);
              *reinterpret_cast<h4*>(olp + (long long)m * g.ld_out + n) = l;
            }
          }
          if (e_part) {   // statistics of the value the GroupNorm will read: the rounded fp16 hi term (rows past the end: 0)
            float sv[4], qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f = row_ok ? (float)o[e] : 0.f;
              sv[e] = row16_sum(f);
              qv[e] = row16_sum(f * f);
            }
            if (lr == 0) {
              float* d = pred + (((wm * MF + j) * BN) + wn * WTN + i * 16 + lg * 4) * 2;
              *reinterpret_cast<f4*>(d) = f4{sv[0], qv[0], sv[1], qv[1]};
              *reinterpret_cast<f4*>(d + 4) = f4{sv[2], qv[2], sv[3], qv[3]};
            }
          }
        }
      }
    }
    if (e_part) {
      __syncthreads();
      if (epi && tid < BN && n0 + tid < g.N) {   // per 64-row granule and column: wave rows, then fragments, in fixed order
        constexpr int WPG = 64 / WTM > WAVES_M ? WAVES_M : 64 / WTM;   // wave rows per granule
#pragma unroll
        for (int gi = 0; gi < (BM + 63) / 64; ++gi) {
          if (m0 + gi * 64 >= Mlim) break;
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WPG; ++w2)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
              s += pred[(((gi * WPG + w2) * MF + j) * BN + tid) * 2];
              q += pred[(((gi * WPG + w2) * MF + j) * BN + tid) * 2 + 1];
            }
          float* dst = e_part + ((long long)(m0 / 64 + gi) * 2) * g.N + n0 + tid;
          dst[0] = s;
          dst[g.N] = q;
        }
      }
    }
    return;
  }
  if (!epi) return;
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int m = m0 + wm * WTM + j * 16 + lr;
    if (m >= Mlim) continue;
    const int b = fast_div(m, g.div_tok_mul, g.div_tok_sh);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int n = n0 + wn * WTN + i * 16 + lg * 4;
      if (n >= g.N) continue;
      epi_store4(g, MD_LATE_ARGS, m, b, n, acc[i][j]);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// Deterministic split-K reduction + epilogue: one thread per (m, 4 columns).
__global__ __launch_bounds__(256) void igemm_splitk_reduce(const IgemmArgs g) {
  const int n4 = g.N >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)g.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long long)m * n4) * 4;
  // all slab loads of a group of 8 are issued before any is consumed (the loop is latency-, not bandwidth-bound); the
  // summation ORDER stays z = 0, 1, 2, ... so the result does not depend on the grouping
  MD_LATE_LOAD(g)
  f4 ebv;   // the epilogue's own loads go out with the first slab loads (one round trip for both)
  h4 erv, erl;
#ifndef MD_REDUCE_PRELOAD
#define MD_REDUCE_PRELOAD 1
#endif
  if (MD_REDUCE_PRELOAD) epi_load(g, MD_LATE_ARGS, m, m / g.tokens, n, ebv, erv, erl);
  const float* base = g.ws + (long long)m * g.N + n;
  const long long slab = (long long)g.M * g.N;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= g.splitk; z += 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f4*>(base + (z + u) * slab);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z + 2 <= g.splitk; z += 2) {
    const f4 v0 = *reinterpret_cast<const f4*>(base + z * slab), v1 = *reinterpret_cast<const f4*>(base + (z + 1) * slab);
    s += v0;
    s += v1;
  }
  if (z < g.splitk) s += *reinterpret_cast<const f4*>(base + z * slab);
  if (!MD_REDUCE_PRELOAD) epi_load(g, MD_LATE_ARGS, m, m / g.tokens, n, ebv, erv, erl);
  epi_finish(g, MD_LATE_ARGS, m, m / g.tokens, n, s, ebv, erv, erl);
}

// Tile configs (ids are stable: igemm_tuned.inc refers to them).
//   4..7   LOADER 1 (global_load_lds + zero page; ragged channel counts): 128x128, 128x64, 64x128, 64x64
//   12..15 LOADER 2 (buffer_load..lds, hardware out-of-range zeros):      128x128, 128x64, 64x128, 64x64
//   24..27 SD-shaped tiles of the buffer loader -- 128x80, 128x160, 64x160, 64x80.  Every channel count of SD-1.5 is a multiple
//          of 80 (320 = 4 x 80), so these cover N exactly where the 64 / 128-wide tiles waste up to 17 %, and e.g. M = 8192,
//          N = 320 becomes exactly 256 workgroups of 128x80 (one per CU) moving 33 % fewer L2->LDS bytes than 64x64 tiles.
//   28..31 two k-tiles per stage (KT = 2) for 64x64, 64x80, 128x80, 64x160 -- half the HBM round trips in the k-loop of the small
//          cold-weight GEMMs of a 1-frame step; 32, 33: four k-tiles per stage for 64x64, 64x80 (128 / 147 KB of LDS).
// Every buffer-loader config also exists with 2 (and, LDS / registers permitting, 4) k-groups per workgroup: max_kg().
const float kTileEff[14] = {1.00f, 0.85f, 0.85f, 0.70f, 0.90f, 1.00f, 0.85f, 0.60f, 0.70f, 0.60f, 0.90f, 0.85f, 0.70f, 0.60f};
const int kTileBM[14] = {128, 128, 64, 64, 128, 128, 64, 64, 64, 64, 128, 64, 64, 64};
const int kTileBN[14] = {128, 64, 128, 64, 80, 160, 160, 80, 64, 80, 80, 160, 64, 80};
constexpr int kFirstSdCfg = 24, kNumAllCfgs = 34;
inline bool cfg_exists(int c) { return (c >= 4 && c < 8) || (c >= 12 && c < 16) || (c >= kFirstSdCfg && c < kNumAllCfgs); }
struct TileCfg {
  int bm, bn;
  float eff;
};
inline TileCfg cfg_of(int c) {
  const int t = c >= kFirstSdCfg ? 4 + (c - kFirstSdCfg) : (c & 3);
  return TileCfg{kTileBM[t], kTileBN[t], kTileEff[t]};
}
// tiles whose per-wave fragment count along N is odd cannot host the GEGLU pairing
inline bool cfg_geglu_ok(int c) { return c < kFirstSdCfg || c == 28 || c == 32; }
inline bool cfg_ln_ok(int c) { return (c >= 12 && c < 16) || (c >= kFirstSdCfg && c < kNumAllCfgs); }
// k-groups per workgroup a config is instantiated with: 160 KB of LDS (KG x 2 stages x KT tiles) and, for KG = 4, 128 VGPRs
inline int max_kg(int c) {
  switch (c) {
    case 15: case 27: return 4;
    case 12: case 13: case 14: case 24: case 25: case 26: case 28: case 29: return 2;
    default: return 1;
  }
}

template <int BM, int BN, int WMv, int WNv, int LOADER, bool LN = false, int KT = 1, int KS = 0, int KG = 1>
int launch_cfg(const IgemmArgs& g, hipStream_t s) {
  constexpr size_t lds = (size_t)KG * 2 * KT * (BM + BN) * 128;
  static_assert(lds <= 160 * 1024, "stages do not fit the 160 KB LDS");
  static bool attr_set[64] = {};   // per DEVICE: the attribute belongs to the device's copy of the kernel
  if (lds > 65536) {
    int devi = 0;
    MD_HIP_CHECK(hipGetDevice(&devi));
    if (devi < 0 || devi >= 64 || !attr_set[devi]) {
      MD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WMv, WNv, LOADER, LN, KT, KS, KG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (devi >= 0 && devi < 64) attr_set[devi] = true;
    }
  }
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.splitk);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WMv, WNv, LOADER, LN, KT, KS, KG>), grid, dim3(256 * KG), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

// the buffer-loader tiles exist with and without the folded LayerNorm, with the generic and the 1x1 issue path
template <int BM, int BN, int WMv, int WNv, int KT, int KG>
int launch_buf_kg(const IgemmArgs& g, hipStream_t s) {
  // the folded LayerNorm exists for 1x1 / linear layers only (validate()): always the KS1 issue path; others: KS1 when it applies
  if (g.ln_s1) return launch_cfg<BM, BN, WMv, WNv, 2, true, KT, 1, KG>(g, s);
  if (g.ksize == 1 && g.stride == 1 && !g.ups && g.c1 == 0) return launch_cfg<BM, BN, WMv, WNv, 2, false, KT, 1, KG>(g, s);
  if (g.ksize == 3 && !g.ups && g.c1 == 0) return launch_cfg<BM, BN, WMv, WNv, 2, false, KT, 3, KG>(g, s);
  if (!g.ups && g.c1 > 0) return launch_cfg<BM, BN, WMv, WNv, 2, false, KT, 4, KG>(g, s);
  return launch_cfg<BM, BN, WMv, WNv, 2, false, KT, 0, KG>(g, s);
}
template <int BM, int BN, int WMv, int WNv, int KT = 1, int MAXKG = 1>
int launch_buf2(const IgemmArgs& g, hipStream_t s, int kg) {
  if constexpr (MAXKG >= 4) {
    if (kg == 4) return launch_buf_kg<BM, BN, WMv, WNv, KT, 4>(g, s);
  }
  if constexpr (MAXKG >= 2) {
    if (kg == 2) return launch_buf_kg<BM, BN, WMv, WNv, KT, 2>(g, s);
  }
  if (kg != 1) return MD_ERR_UNSUPPORTED;
  return launch_buf_kg<BM, BN, WMv, WNv, KT, 1>(g, s);
}

void fast_div_magic(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned l = 0;
  while ((1u << l) < d) ++l;
  *sh = 24 + l;
  *mul = (unsigned)(((unsigned long long)1 << (24 + l)) / d) + 1u;
}

int validate(const md_igemm_params* p) {
  if (!p || !p->a0 || !p->w || !p->out) return MD_ERR_BAD_ARG;
  if (p->ksize != 1 && p->ksize != 3) return MD_ERR_UNSUPPORTED;
  if (p->stride != 1 && p->stride != 2) return MD_ERR_UNSUPPORTED;
  if (p->ups && (p->ksize != 3 || p->stride != 1)) return MD_ERR_UNSUPPORTED;
  if (p->asym_pad && (p->ksize != 3 || p->ups)) return MD_ERR_UNSUPPORTED;
  if ((p->ln_s1 != nullptr) != (p->ln_s0 != nullptr)) return MD_ERR_BAD_ARG;
  if (p->ln_s1 && (p->ksize != 1 || p->c1 != 0 || (p->c0 & 63) || p->bias || p->stride != 1 || !(p->ln_eps > 0.f))) return MD_ERR_UNSUPPORTED;
  if (p->c0 <= 0 || (p->c0 & 7) || p->c1 < 0 || (p->c1 & 7)) return MD_ERR_BAD_ARG;
  if ((p->c1 > 0) != (p->a1 != nullptr)) return MD_ERR_BAD_ARG;
  if (p->n <= 0 || (p->n & 3) || (p->ld_out & 3)) return MD_ERR_BAD_ARG;
  if (p->batch <= 0 || p->hin <= 0 || p->win <= 0 || p->hout <= 0 || p->wout <= 0) return MD_ERR_BAD_ARG;
  if (p->res && (p->ld_res & 3)) return MD_ERR_BAD_ARG;
  if (p->res_lo && !p->res) return MD_ERR_BAD_ARG;
  if (p->out_lo && (p->out_f32 || p->act == MD_ACT_GEGLU)) return MD_ERR_UNSUPPORTED;
  if (p->k8 && ((p->k8_begin & 3) || (p->k8_end & 3) || (p->ld_k8 & 3) || p->k8_begin < 0 || p->k8_end > p->n_tr_begin || p->act == MD_ACT_GEGLU))
    return MD_ERR_BAD_ARG;
  if (p->col_scale_end < 0 || (p->col_scale_end & 3) || (p->col_scale_end && p->act == MD_ACT_GEGLU)) return MD_ERR_BAD_ARG;
  if (p->n_tr_begin < 0 || p->n_tr_begin > p->n || (p->n_tr_begin < p->n && (p->n_tr_begin & 15))) return MD_ERR_BAD_ARG;
  if (p->n_tr_begin < p->n && (!p->out_t || p->ld_t <= 0)) return MD_ERR_BAD_ARG;
  if (p->bias_batch_stride & 3) return MD_ERR_BAD_ARG;
  if (p->gn_part) {   // GroupNorm partials: the common fp16 epilogue, whole 64-row granules per sample
    if (p->out_f32 || p->n_tr_begin != p->n || p->k8 || p->act == MD_ACT_GEGLU || ((p->hout * p->wout) & 63)) return MD_ERR_UNSUPPORTED;
  }
  if (p->force_kg != 0 && p->force_kg != 1 && p->force_kg != 2 && p->force_kg != 4) return MD_ERR_BAD_ARG;
  if (p->w2 && p->batch2 > 0) {
    if (p->batch2 >= p->batch || p->bias_batch_stride) return MD_ERR_UNSUPPORTED;
    if ((p->bias != nullptr) != (p->bias2 != nullptr) || (p->ln_s1 != nullptr) != (p->ln2_s1 != nullptr) ||
        (p->ln_s0 != nullptr) != (p->ln2_s0 != nullptr))
      return MD_ERR_BAD_ARG;
  }
  if (p->act == MD_ACT_GEGLU) {
    if ((p->n & 31) || p->res || p->out_f32 || p->n_tr_begin != p->n || p->bias_batch_stride) return MD_ERR_UNSUPPORTED;
  } else if (p->act != MD_ACT_NONE && p->act != MD_ACT_SILU) {
    return MD_ERR_UNSUPPORTED;
  }
  return MD_OK;
}

// Measured-best (config, split-K) per layer shape, generated on an MI355X by tools/tune_igemm.py
struct TunedEntry {
  int m, n, k, ksize, stride, ups, cfg, split, kg;   // kg 0: an entry older than the k-groups (= 1)
};
const TunedEntry kTuned[] = {
#include "igemm_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};
int g_use_tuned = [] {
  const char* e = getenv("MD_IGEMM_TUNED");
  return (e && e[0] == '0') ? 0 : 1;
}();

// Pick tile config + split-K: tuned table first, otherwise a crude time model (overridable: force_cfg / force_splitk).
void choose(const md_igemm_params* p, long long M, int N, int K, long long ws_bytes, int* cfg_out, int* split_out, int* kg_out) {
  *kg_out = 0;
  const int nk = (K + 63) / 64;
  // two parameter sets: a batch of 3F samples (2F UNet + F ControlNet) is not in the table -- the entry of the 2F-sample layer
  // (same N, K, tile economics; 1.5x the workgroups) is the second choice before the model below
  const long long M_alt = (p->w2 && p->batch2 > 0) ? (long long)p->batch2 * p->hout * p->wout : M;
  for (int pass = 0; pass < 2 && g_use_tuned && p->force_cfg < 0 && p->force_splitk <= 0; ++pass) {
    const long long Mq = pass == 0 ? M : M_alt;
    if (pass == 1 && M_alt == M) break;
    for (const TunedEntry* t = kTuned; t->m; ++t) {
      if (t->m == Mq && t->n == N && t->k == K && t->ksize == p->ksize && t->stride == p->stride && t->ups == p->ups) {
        const bool ok_split = t->split == 1 || (p->act != MD_ACT_GEGLU && (long long)t->split * M * N * 4 <= ws_bytes);
        const bool ok_buf = t->cfg < 12 || (((p->c0 + p->c1) % 64 == 0) && (p->c0 % 64 == 0));
        const bool ok_act = cfg_geglu_ok(t->cfg) || p->act != MD_ACT_GEGLU;
        const bool ok_ln = !p->ln_s1 || (t->split == 1 && cfg_ln_ok(t->cfg));
        if (ok_split && ok_buf && ok_act && ok_ln && cfg_exists(t->cfg)) {
          *cfg_out = t->cfg;
          *split_out = t->split;
          *kg_out = t->kg > 0 ? t->kg : 1;   // a measured (config, split): entries older than the k-groups keep their 4-wave form
          return;
        }
      }
    }
  }
  double best = 1e30;
  // the buffer-descriptor loader needs tile-uniform (tap, source): 64-channel k-tiles must not straddle either
  const bool buf_ok = ((p->c0 + p->c1) % 64 == 0) && (p->c0 % 64 == 0);
  const int fam = buf_ok ? 3 : 1;
  int bc = 4 * fam + 3, bs = 1;
  for (int c = 0; c < kNumAllCfgs; ++c) {
    if (!cfg_exists(c)) continue;
    if (p->force_cfg >= 0 && c != p->force_cfg) continue;
    if (p->force_cfg < 0 && (c >= 16 || c / 4 != fam)) continue;
    const long long tm = (M + cfg_of(c).bm - 1) / cfg_of(c).bm, tn = (N + cfg_of(c).bn - 1) / cfg_of(c).bn;
    const long long blocks = tm * tn;
    const double rate_cu = 2.5e15 / 256.0 * 0.35 * cfg_of(c).eff;  // flop/s per CU we expect from this tile
    for (int s = 1; s <= 32; s = (s < 4 ? s + 1 : s * 2)) {
      if (p->force_splitk > 0 && s != p->force_splitk) continue;
      if (s > 1) {
        if (p->act == MD_ACT_GEGLU || p->ln_s1) break;
        if ((long long)s * M * N * 4 > ws_bytes) break;
        if (nk / s < 4) break;
      }
      const int tps = (nk + s - 1) / s;
      const long long waves = (blocks * s + 255) / 256;
      double t = (double)waves * (2.0 * cfg_of(c).bm * cfg_of(c).bn * (double)tps * 64.0) / rate_cu + 2e-6;
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

// k-groups for a shape the tuned table does not hold (the time model above chose config and split): turn global split-K into
// in-workgroup k-groups where the config has them (no slabs, no reduce launch), and give grids that leave CUs without a second
// workgroup more waves per tile.
int default_kg(int cfg, int* split, long long tiles, int nk, bool allow) {
  const int mk = allow ? max_kg(cfg) : 1;
  if (mk == 1) return 1;
  if (*split > 1) {
    int kg = (mk >= 4 && *split % 4 == 0) ? 4 : ((*split % 2 == 0) ? 2 : 1);
    *split /= kg;
    return kg;
  }
  if (tiles <= 256 && nk >= 16 && mk >= 4) return 4;
  if (tiles <= 256 && nk >= 8) return 2;
  return 1;
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
  g.pad = p->asym_pad ? 0 : p->ksize / 2;
  g.w = (const half_t*)p->w;
  const long long M = (long long)p->batch * g.tokens;
  if (M >= (1LL << 24)) return MD_ERR_BAD_ARG;  // fast_div domain
  g.M = (int)M;
  fast_div_magic((unsigned)g.tokens, &g.div_tok_mul, &g.div_tok_sh);
  fast_div_magic((unsigned)g.wout, &g.div_w_mul, &g.div_w_sh);
  g.N = p->n;
  g.K = p->ksize * p->ksize * g.cin;
  g.nk = (g.K + 63) / 64;
  g.bias = p->bias;
  g.bias_bs = p->bias_batch_stride;
  g.res = (const half_t*)p->res;
  g.res_lo = (const half_t*)p->res_lo;
  g.out_lo = (half_t*)p->out_lo;
  g.col_scale = p->col_scale;
  g.col_scale_end = p->col_scale_end;
  g.k8 = (unsigned char*)p->k8;
  g.k8_begin = p->k8_begin;
  g.k8_end = p->k8_end;
  g.ld_k8 = p->ld_k8;
  g.vt_fp8 = p->vt_fp8;
  g.ld_res = p->ld_res;
  g.act = p->act;
  g.out = p->out;
  g.ld_out = p->ld_out;
  g.out_f32 = p->out_f32;
  g.out_t = (half_t*)p->out_t;
  g.n_tr_begin = p->n_tr_begin;
  g.ld_t = p->ld_t;
  g.ws = (float*)p->ws;
  g.ln_s1 = p->ln_s1;
  g.ln_s0 = p->ln_s0;
  g.ln_eps = p->ln_eps;
  const bool dual = p->w2 && p->batch2 > 0;
  g.w2 = dual ? (const half_t*)p->w2 : g.w;
  g.bias2 = dual ? p->bias2 : g.bias;
  g.ln2_s1 = dual ? p->ln2_s1 : g.ln_s1;
  g.ln2_s0 = dual ? p->ln2_s0 : g.ln_s0;
  g.m_split = dual ? p->batch2 * g.tokens : 0x7fffffff;
  g.ln_inv_k = 1.0f / (float)g.K;
  g.part = (float*)p->gn_part;
  g.w_tiled = p->w_tiled;
  g.epi_stage = 0;
  int cfg, split, kg;
  choose(p, M, g.N, g.K, p->ws ? p->ws_bytes : 0, &cfg, &split, &kg);
  if (!cfg_exists(cfg)) return MD_ERR_BAD_ARG;
  if (cfg >= 12 && (g.cin % 64 != 0 || g.c0 % 64 != 0)) return MD_ERR_UNSUPPORTED;  // forced buffer loader on a ragged shape
  if (g.w_tiled && (cfg < 12 || (g.N & 15) != 0 || g.cin % 64 != 0 || g.c0 % 64 != 0)) return MD_ERR_UNSUPPORTED;  // tiled W: buffer loader only
  if (g.part && split > 1) {   // the partials come from the in-kernel epilogue
    if (p->force_splitk > 1) return MD_ERR_UNSUPPORTED;
    split = 1;
    kg = 0;
  }
  if (p->force_kg > 0) {
    kg = p->force_kg;
  } else if (kg <= 0) {
    const long long tl = ((M + cfg_of(cfg).bm - 1) / cfg_of(cfg).bm) * ((g.N + cfg_of(cfg).bn - 1) / cfg_of(cfg).bn) * split;
    kg = default_kg(cfg, &split, tl, g.nk, p->force_splitk <= 0);
  }
  if (kg > max_kg(cfg)) return MD_ERR_UNSUPPORTED;
  if (split > 1 && (!p->ws || (long long)split * M * g.N * 4 > p->ws_bytes)) return MD_ERR_WORKSPACE;
  if (p->act == MD_ACT_GEGLU && split > 1) return MD_ERR_UNSUPPORTED;
  if (p->act == MD_ACT_GEGLU && !cfg_geglu_ok(cfg)) return MD_ERR_UNSUPPORTED;  // odd fragment count per wave
  if (p->ln_s1 && (split > 1 || !cfg_ln_ok(cfg))) return MD_ERR_UNSUPPORTED;
  {  // the LDS-staged epilogue: plain fp16 row-major output in whole 16-byte pieces
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (p->act == MD_ACT_GEGLU)   // [M][n / 2] fp16 output, 8 output columns per piece
      g.epi_stage = (p->n & 15) == 0 && (p->ld_out & 7) == 0 && al16(p->out);
    else
      g.epi_stage = split == 1 && !g.part && !p->out_f32 && !p->k8 && p->n_tr_begin >= p->n &&
                  (p->n & 7) == 0 && (p->ld_out & 7) == 0 && al16(p->out) && al16(p->out_lo) &&
                  (!p->res || ((p->ld_res & 7) == 0 && al16(p->res) && al16(p->res_lo)));
  }
  g.splitk = split;
  g.tiles_per_split = (g.nk + split - 1) / split;
  g.tiles_m = (g.M + cfg_of(cfg).bm - 1) / cfg_of(cfg).bm;
  g.tiles_m1 = 0x7fffffff;
  if (dual) {
    g.tiles_m1 = (g.m_split + cfg_of(cfg).bm - 1) / cfg_of(cfg).bm;
    g.tiles_m = g.tiles_m1 + (g.M - g.m_split + cfg_of(cfg).bm - 1) / cfg_of(cfg).bm;
  }
  g.tiles_n = (g.N + cfg_of(cfg).bn - 1) / cfg_of(cfg).bn;
  {  // ~64 workgroups are resident per XCD: make them a (group_m x tiles_n) block of the tile grid
    int gm = (64 + g.tiles_n - 1) / g.tiles_n;
    if (gm < 1) gm = 1;
    if (gm > g.tiles_m) gm = g.tiles_m;
    g.group_m = gm;
  }
  hipStream_t s = (hipStream_t)stream;
  char tag[128];
  snprintf(tag, sizeof(tag), "M=%lld N=%d K=%d ks=%d st=%d up=%d cfg=%d split=%d kg=%d B=%d h=%d w=%d c0=%d c1=%d act=%d", M, g.N, g.K,
           g.ksize, g.stride, g.ups, cfg, split, kg, g.batch, g.hin, g.win, g.c0, g.c1, g.act);
  md::ProfScope prof(MD_FAM_IGEMM, s, 2.0 * (double)M * g.N * g.K,
                     (double)M * g.cin * 2.0 + (double)g.N * g.K * 2.0 + (double)M * g.N * 2.0, tag);
  int rc;
  switch (cfg) {
    case 24: rc = launch_buf2<128, 80, 4, 1, 1, 2>(g, s, kg); break;
    case 25: rc = launch_buf2<128, 160, 2, 2, 1, 2>(g, s, kg); break;
    case 26: rc = launch_buf2<64, 160, 2, 2, 1, 2>(g, s, kg); break;
    case 27: rc = launch_buf2<64, 80, 4, 1, 1, 4>(g, s, kg); break;
    case 28: rc = launch_buf2<64, 64, 2, 2, 2, 2>(g, s, kg); break;
    case 29: rc = launch_buf2<64, 80, 4, 1, 2, 2>(g, s, kg); break;
    case 30: rc = launch_buf2<128, 80, 4, 1, 2>(g, s, kg); break;
    case 31: rc = launch_buf2<64, 160, 2, 2, 2>(g, s, kg); break;
    case 32: rc = launch_buf2<64, 64, 2, 2, 4>(g, s, kg); break;
    case 33: rc = launch_buf2<64, 80, 4, 1, 4>(g, s, kg); break;
    case 4: rc = launch_cfg<128, 128, 2, 2, 1>(g, s); break;
    case 5: rc = launch_cfg<128, 64, 2, 2, 1>(g, s); break;
    case 6: rc = launch_cfg<64, 128, 2, 2, 1>(g, s); break;
    case 7: rc = launch_cfg<64, 64, 2, 2, 1>(g, s); break;
    case 12: rc = launch_buf2<128, 128, 2, 2, 1, 2>(g, s, kg); break;
    case 13: rc = launch_buf2<128, 64, 2, 2, 1, 2>(g, s, kg); break;
    case 14: rc = launch_buf2<64, 128, 2, 2, 1, 2>(g, s, kg); break;
    case 15: rc = launch_buf2<64, 64, 2, 2, 1, 4>(g, s, kg); break;
    default: return MD_ERR_BAD_ARG;
  }
  if (rc != MD_OK) return rc;
  if (split > 1) {
    const long long work = M * (g.N >> 2);
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, g);
    MD_HIP_CHECK(hipGetLastError());
  }
  return MD_OK;
}
