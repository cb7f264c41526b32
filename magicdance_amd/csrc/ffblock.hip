// md_ff_block: the row-local tail of BasicTransformerBlock as ONE kernel for gfx950 (MI355X).
//
//   t2  = attn2 Wo^T + bo + t                 (optional head: CrossAttention.to_out of attn2 + residual, attention.py:318)
//   out = GEGLU(LayerNorm3(t2)) W2^T + b2 + t2   (FeedForward, attention.py:50-77, 319)
//
// Everything here is local to a row of the token matrix, so a workgroup keeps a BM x C tile of the stream resident in LDS and runs
// the whole chain on it: the [M][4C] GEGLU hidden activation, the normalised rows and (with the head) the intermediate stream t2
// never leave the CU.  Against the three md_igemm launches this replaces (to_out, GEGLU projection, feed-forward output) that is
// 2 M 4C 2 B (hidden write + read) + 4 M C 2 B (t2 hi / lo write + read) of HBM / L2 traffic and two kernel boundaries less.
//
// Structure (8 waves = WM x WN, 2 per SIMD; v_mfma_f32_16x16x32_f16, fp32 accumulate, D[n][m] fragment layout as md_igemm):
//   * A tile [BM][C] fp16 = C / 64 k-tiles of [BM][64], XOR-swizzled 16-byte chunks (the md_igemm stage layout), filled by LDS-DMA.
//   * The weights are consumed as ONE linear stream of 8 KiB "pieces" (64 weight rows x one 64-deep k-tile of the tiled storage
//     form, md_igemm_params.w_tiled): [head: C/64 k-tiles x C/64 pieces of Wo] then per 64-wide hidden chunk j:
//     [5 k-tiles x 2 pieces of the LN-folded GEGLU projection rows 128 j .. 128 j + 127 | C/64 pieces of W2's k-tile j].
//     Every piece is ONE LDS-DMA instruction per wave (8 waves x 8 rows x 128 B) into an R-slot ring; a step waits with a COUNTED
//     s_waitcnt vmcnt(n) (n = this wave's DMA instructions issued after the step's last piece: the stream position is known
//     analytically), passes one raw s_barrier, refills the slots the previous step freed, then computes: R - 2 pieces (40-64 KiB
//     per CU) stay in flight across every barrier.
//   * GEMM 1 of a chunk: S[BM][128] = A W1'[chunk]^T over K = C; wave (wm, wn) owns 16 MF rows x 128 / WN columns.  The folded
//     LayerNorm (row statistics of the fp16 A tile, rank-1 correction with s1 / s0 as md_igemm ln_*) and a * gelu(gate) run on the
//     accumulators; the fp16 result is parked in a [BM][64] LDS tile (k-tile layout) because the waves of a wave row split the
//     hidden columns but each needs all of them as the K dimension of
//   * GEMM 2: acc[BM][C] += h[BM][64] W2[:, chunk]^T; wave (wm, wn) owns the output fragments f with f % WN == wn of its rows, so
//     every W2 piece feeds all eight waves.  acc (80 VGPRs at BM = 128) lives in registers for the whole kernel; with the head it is
//     first loaded with t2 in fp32 -- the residual stream crosses the kernel in fp32, not as hi + lo.
//   * s1 / s0 of a chunk travel in the same stream (two 256-byte LDS-DMA instructions per wave and chunk, wave-private, double
//     buffered): no VGPR-destination load sits between the LDS-DMA instructions of the main loop.
// Parity: per-kernel tests vs fp32 torch in tests/test_gpu_ffblock.py; the unfused md_igemm pair stays the reference form.
#include <cstdio>
#include <type_traits>

#include "md_common.h"

#ifndef MD_FF_ABLATE
#define MD_FF_ABLATE 0   // timing experiments only (tools/ffblock_bench.py): 1 no s1 / s0 DMA, 2 no DMA at all, 4 no MFMAs
#endif

#define FF_MFMA(a, b, c) ((MD_FF_ABLATE & 4) ? (c) : __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0))

namespace {

struct FfArgs {
  const half_t* x;      // [M][C] residual stream (hi term); without the head also the A operand
  const half_t* x_lo;   // second term or nullptr
  const half_t* attn;   // head only: [M][C] attention output (A operand of the to_out GEMM)
  half_t* out;
  half_t* out_lo;
  int M, m_split, tiles_m1;
  float ln_eps;
  // parameter set 1 | 2 (rows >= m_split)
  const half_t* w1[2];   // [8C][C] tiled, LayerNorm-folded, a / gate interleaved in groups of 16 rows
  const float* s1[2];
  const float* s0[2];
  const half_t* w2[2];   // [C][4C] tiled
  const float* b2[2];
  const half_t* wo[2];   // [C][C] tiled (head)
  const float* bo[2];
};

// s_waitcnt vmcnt(n) lgkmcnt(0) [; s_barrier] with a run-time (wave-uniform) n: the count is an immediate.  A smaller count than
// asked for is merely stricter.
template <bool BARRIER>
__device__ __forceinline__ void cnt_wait(int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  n = n < 0 ? 0 : n;
#define MD_FW(k)                                                                          \
  case k:                                                                                 \
    if (BARRIER)                                                                          \
      asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else                                                                                  \
      asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory");                    \
    break;
  switch (n) {
    MD_FW(0) MD_FW(1) MD_FW(2) MD_FW(3) MD_FW(4) MD_FW(5) MD_FW(6) MD_FW(7) MD_FW(8) MD_FW(9)
    MD_FW(10) MD_FW(11) MD_FW(12) MD_FW(13) MD_FW(14) MD_FW(15) MD_FW(16) MD_FW(17) MD_FW(18) MD_FW(19)
    MD_FW(20) MD_FW(21) MD_FW(22) MD_FW(23) MD_FW(24) MD_FW(25) MD_FW(26) MD_FW(27) MD_FW(28) MD_FW(29)
    default:
      if (BARRIER)
        asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)" ::: "memory");
  }
#undef MD_FW
#endif
}

template <int C, int WM, int MF, int R, bool HEAD>
__global__ __launch_bounds__(512) void ff_block_kernel(const FfArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WN = 8 / WM, BM = 16 * MF * WM;
  constexpr int NKC = C / 64;              // k-tiles of a K = C GEMM = 64-row pieces of a C-row weight panel
  constexpr int NCH = 4 * C / 64;          // hidden chunks (64 hidden columns each)
  constexpr int NF1 = 8 / WN;              // S fragments per wave: 128 columns / 16 / WN (pairs: a | gate)
  constexpr int FPP = 4 / WN;              // output fragments per 64-row piece per wave
  constexpr int NF2 = NKC * FPP;           // output fragments per wave
  constexpr int PPC = 3 * NKC;             // pieces per hidden chunk: 2 NKC (projection) + NKC (W2)
  constexpr int HEAD_P = HEAD ? NKC * NKC : 0;
  constexpr int P_TOTAL = HEAD_P + NCH * PPC;
  constexpr int PIECE = 8192;
  constexpr int A_BYTES = NKC * BM * 128, H_BYTES = BM * 128, S_BYTES = 8 * 1024;
  constexpr int H_OFF = A_BYTES, S_OFF = H_OFF + H_BYTES, RING_OFF = S_OFF + S_BYTES;
  static_assert(WM * WN == 8 && (WN == 2 || WN == 4), "8 waves as WM x WN");
  static_assert(C % 64 == 0 && RING_OFF + R * PIECE <= 160 * 1024, "LDS budget");
  static_assert(WN * BM * 8 <= H_BYTES, "the statistics exchange fits the (not yet used) h tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem + RING_OFF;

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wv % WM, wn = wv / WM;
  const int lr = lane & 15, lg = lane >> 4;

  const int tile = blockIdx.x;
  const bool set2 = tile >= g.tiles_m1;
  const int ps = set2 ? 1 : 0;
  const int m0 = set2 ? g.m_split + (tile - g.tiles_m1) * BM : tile * BM;
  const int Mlim = set2 ? g.M : min(g.M, g.m_split);
  // Every workgroup streams the SAME weight bytes.  In lockstep all 32 CUs of an XCD would ask the same one or two L2 channels
  // for the same 8 KiB at the same time (measured: 13-17 GB/s per CU, a step per microsecond); so workgroup i walks the hidden
  // chunks (and the k-tiles of the head GEMM) in an order rotated by i / 8 -- block i runs on XCD i % 8, i / 8 counts the
  // workgroups of one XCD.  The sums are order-independent up to fp32 rounding; the order is a function of the tile index only.
  const int tloc = set2 ? tile - g.tiles_m1 : tile;   // (tile within its parameter set: one launch == two launches on the row ranges)
  const int rot_ch = (tloc >> 3) % NCH, rot_kt = (tloc >> 3) % NKC;

  // ---- loader role ------------------------------------------------------------------------------------------------------------
  const int r8 = lane >> 3, c8 = lane & 7;
  const unsigned gcb = (unsigned)(c8 ^ r8) * 16u;   // the XOR swizzle lives on the source side (md_igemm stage layout)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(HEAD ? g.attn : g.x), 0, g.M * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.w1[ps]), 0, 8 * C * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.w2[ps]), 0, 4 * C * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(HEAD ? g.wo[ps] : g.w2[ps]), 0, (HEAD ? C * C : 4 * C * C) * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.s1[ps]), 0, 8 * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.s0[ps]), 0, 8 * C * 4, 0x00020000);
  // a piece = rows 64 i .. 64 i + 63 of a weight matrix in the tiled form [N / 16][nk][16][64]: wave wv fetches rows 8 wv .. + 7
  // = panel 4 i + (wv >> 1), rows (wv & 1) 8 + r8 of it
  const unsigned in_panel = (unsigned)((wv & 1) * 8 + r8) * 128u + gcb;
  const unsigned voff_kc = (unsigned)(wv >> 1) * (unsigned)(NKC * 2048) + in_panel;   // nk = NKC: W1', Wo
  const unsigned voff_kh = (unsigned)(wv >> 1) * (unsigned)(NCH * 2048) + in_panel;   // nk = NCH: W2
  const int scol = wn * (128 / WN);                                 // this wave's first S column inside a chunk
  const unsigned svoff = (unsigned)(lane & (NF1 * 16 - 1)) * 4u;    // (WN = 4: the upper half of the wave re-reads; in range)

  int tot = 0;   // LDS-DMA instructions this wave has issued (the vmcnt sequence number of the youngest)
  auto dma16 = [&](const __amdgpu_buffer_rsrc_t& rs, char* dst, unsigned voff, unsigned soff) {
    if (MD_FF_ABLATE & 2) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
    ++tot;
  };
  auto dma4 = [&](const __amdgpu_buffer_rsrc_t& rs, char* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 4, voff, soff, 0, 0);
    ++tot;
  };

  // A tile: NKC k-tiles x BM / 8 row pieces, wave wv takes pieces wv, wv + 8, ...
  constexpr int AJ = (BM / 8 + 7) / 8;
#pragma unroll
  for (int t = 0; t < NKC; ++t) {
#pragma unroll
    for (int jj = 0; jj < AJ; ++jj) {
      const int q = wv + 8 * jj;
      if (q < BM / 8) {
        const unsigned row = (unsigned)min(m0 + q * 8 + r8, Mlim - 1);   // rows past the end: clamped, never stored
        dma16(rs_a, smem + t * BM * 128 + q * 1024, row * (unsigned)(C * 2) + gcb, (unsigned)t * 128u);
      }
    }
  }
  const int a_cnt = tot;

  // ---- the piece stream -------------------------------------------------------------------------------------------------------
  auto issue_piece = [&](int p, int slot) {
    char* const dst = ring + slot * PIECE + wv * 1024;
    if (HEAD && p < HEAD_P) {
      int t = p / NKC;
      const int i = p - t * NKC;
      t += rot_kt;
      if (t >= NKC) t -= NKC;
      dma16(rs_wo, dst, voff_kc, (unsigned)((4 * i) * NKC + t) * 2048u);
      return;
    }
    const int pp = p - HEAD_P;
    const int jl = pp / PPC, q = pp - jl * PPC;   // position in the stream -> the hidden chunk this workgroup visits there
    const int j = jl + rot_ch >= NCH ? jl + rot_ch - NCH : jl + rot_ch;
    if (q == 0 && !(MD_FF_ABLATE & 3)) {   // s1 | s0 of this wave's S columns of the chunk (wave-private, double-buffered by stream parity)
      char* const sd = smem + S_OFF + wv * 1024 + (jl & 1) * 512;
      const unsigned so = (unsigned)(j * 128 + scol) * 4u;
      dma4(rs_s1, sd, svoff, so);
      dma4(rs_s0, sd + 256, svoff, so);
    }
    if (q < 2 * NKC) {
      const int t = q >> 1, half = q & 1;
      dma16(rs_w1, dst, voff_kc, (unsigned)((j * 8 + half * 4) * NKC + t) * 2048u);
    } else {
      const int i = q - 2 * NKC;
      dma16(rs_w2, dst, voff_kh, (unsigned)((4 * i) * NCH + j) * 2048u);
    }
  };
  // sequence number of the last DMA instruction of piece p (pieces are one instruction per wave; + the s-loads before it)
  auto seq_of = [&](int p) {
    if (MD_FF_ABLATE & 2) return a_cnt;
    if (MD_FF_ABLATE & 1) return a_cnt + p + 1;
    return a_cnt + p + 1 + ((!HEAD || p >= HEAD_P) ? 2 * ((p - HEAD_P) / PPC + 1) : 0);
  };

  int pi = 0, pf = 0, pc = 0, islot = 0, cslot = 0;   // issued / freed / consumed piece counts, slot of the next issue / consume
  auto issue_avail = [&] {
    while (pi < P_TOTAL && pi < pf + R) {
      issue_piece(pi, islot);
      ++pi;
      islot = islot + 1 == R ? 0 : islot + 1;
    }
  };
  // a step over the next np pieces: they have landed (own DMA counted, then the barrier); every wave is done with the previous
  // step's pieces -> their slots are refilled
  auto step_begin = [&](int np) {
    cnt_wait<true>(tot - seq_of(pc + np - 1));
    pf = pc;
    issue_avail();
  };
  auto step_end = [&](int np) {
    pc += np;
    cslot += np;
    if (cslot >= R) cslot -= R;
  };
  auto slot_ptr = [&](int k) -> const char* {
    int s = cslot + k;
    if (s >= R) s -= R;
    return ring + s * PIECE;
  };
  issue_avail();   // fill the ring behind the A tile

  // ---- fragment reads -----------------------------------------------------------------------------------------------------------
  // operand fragment (16 rows x 32 k) of a [rows][64] k-tile image: lane (lr, lg) holds row lr, k = ks 32 + lg 8 .. + 7
  auto frag = [&](const char* base, int row, int ks) {
    return *reinterpret_cast<const h8*>(base + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
  };
  const int arow = wm * 16 * MF + lr;   // + 16 mf

  f4 acc[NF2][MF];
#pragma unroll
  for (int f = 0; f < NF2; ++f)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[f][j] = f4{0.f, 0.f, 0.f, 0.f};

  // acc += B-operand fragments bf (this wave's rows, K = 64) x the 64-row weight pieces k = 0 .. np - 1 of the current step, which
  // hold output fragments 4 (i0 + k) .. + 3
  auto mma_pieces = [&](const h8 (&bf)[2][MF], int np, auto i0c) {
    constexpr int i0 = decltype(i0c)::value;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k >= np) break;
      const char* const Wp = slot_ptr(k);
#pragma unroll
      for (int u = 0; u < FPP; ++u) {
        const int fr = WN == 2 ? 2 * u + wn : wn;
        const h8 w0 = frag(Wp, fr * 16 + lr, 0), w1 = frag(Wp, fr * 16 + lr, 1);
#pragma unroll
        for (int j = 0; j < MF; ++j) {
          acc[(i0 + k) * FPP + u][j] = FF_MFMA(w0, bf[0][j], acc[(i0 + k) * FPP + u][j]);
          acc[(i0 + k) * FPP + u][j] = FF_MFMA(w1, bf[1][j], acc[(i0 + k) * FPP + u][j]);
        }
      }
    }
  };
  // the NKC pieces of one [C][64] weight k-tile, two per step
  auto gemm_c_rows = [&](const char* bbase) {
    h8 bf[2][MF];
    bool have = false;
    auto one = [&](auto i0c) {
      constexpr int i0 = decltype(i0c)::value;
      constexpr int np = NKC - i0 < 2 ? NKC - i0 : 2;
      step_begin(np);
      if (!have) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < MF; ++j) bf[ks][j] = frag(bbase, arow + 16 * j, ks);
        have = true;
      }
      mma_pieces(bf, np, i0c);
      step_end(np);
    };
    one(std::integral_constant<int, 0>{});
    if constexpr (NKC > 2) one(std::integral_constant<int, 2>{});
    if constexpr (NKC > 4) one(std::integral_constant<int, 4>{});
    if constexpr (NKC > 6) one(std::integral_constant<int, 6>{});
    if constexpr (NKC > 8) one(std::integral_constant<int, 8>{});
    static_assert(NKC <= 10, "piece steps");
  };
  // output column of this lane's 4 values of output fragment f
  auto out_col = [&](int f) {
    const int i = f / FPP, u = f - i * FPP;
    const int fr = WN == 2 ? 2 * u + wn : wn;
    return (4 * i + fr) * 16 + lg * 4;
  };

  // ---- the A tile has landed ----------------------------------------------------------------------------------------------------
  cnt_wait<true>(tot - a_cnt);
  float mu[MF], rstd[MF];
  float* const stat = reinterpret_cast<float*>(smem + H_OFF);   // (the h tile is first written after GEMM 1 of chunk 0)

  if constexpr (HEAD) {
    // t2 = attn Wo^T + bo + x (+ x_lo): K = C over the resident attention tile
#pragma unroll 1
    for (int t = 0; t < NKC; ++t) {
      const int tr = t + rot_kt >= NKC ? t + rot_kt - NKC : t + rot_kt;
      gemm_c_rows(smem + tr * BM * 128);
    }
    const float* const bo = g.bo[ps];
    float sm[MF], sq[MF];
#pragma unroll
    for (int j = 0; j < MF; ++j) sm[j] = sq[j] = 0.f;
    asm volatile("s_barrier" ::: "memory");   // every wave is done reading the attention tile: it becomes the t2 tile
#pragma unroll
    for (int j = 0; j < MF; ++j) {
      const int row = arow + 16 * j;
      const long long m = min(m0 + row, Mlim - 1);
      h4 xv[NF2], xl[NF2];
#pragma unroll
      for (int f = 0; f < NF2; ++f) {
        xv[f] = *reinterpret_cast<const h4*>(g.x + m * C + out_col(f));
        xl[f] = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      }
      if (g.x_lo) {
#pragma unroll
        for (int f = 0; f < NF2; ++f) xl[f] = *reinterpret_cast<const h4*>(g.x_lo + m * C + out_col(f));
      }
#pragma unroll
      for (int f = 0; f < NF2; ++f) {
        const int n = out_col(f);
        const f4 bv = *reinterpret_cast<const f4*>(bo + n);
        f4 v = acc[f][j] + bv;
        h4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += (float)xv[f][e];
          v[e] += (float)xl[f][e];
          hv[e] = (half_t)v[e];
          const float r = (float)hv[e];
          sm[j] += r;
          sq[j] += r * r;
        }
        acc[f][j] = v;   // the residual stream stays in fp32
        *reinterpret_cast<h4*>(smem + (n >> 6) * (BM * 128) + row * 128 + ((((n & 63) >> 3) ^ (row & 7)) << 4) + (n & 7) * 2) = hv;
      }
      sm[j] += __shfl_xor(sm[j], 16, 64);
      sq[j] += __shfl_xor(sq[j], 16, 64);
      sm[j] += __shfl_xor(sm[j], 32, 64);
      sq[j] += __shfl_xor(sq[j], 32, 64);
      if (lg == 0) {
        stat[(wn * BM + row) * 2] = sm[j];
        stat[(wn * BM + row) * 2 + 1] = sq[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < MF; ++j) {
      const int row = arow + 16 * j;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        s += stat[(w * BM + row) * 2];
        q += stat[(w * BM + row) * 2 + 1];
      }
      mu[j] = s * (1.0f / C);
      rstd[j] = rsqrtf(fmaxf(q * (1.0f / C) - mu[j] * mu[j], 0.f) + g.ln_eps);
    }
  } else {
    // row statistics of the fp16 tile: 512 / BM threads per row, 16-byte pieces of the row in any order
    constexpr int TPR = 512 / BM;
    const int row = (int)threadIdx.x / TPR, part = (int)threadIdx.x % TPR;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int idx = 0; idx < (NKC * 8 + TPR - 1) / TPR; ++idx) {
      const int c = part + idx * TPR;
      if (c < NKC * 8) {
        const h8 v = *reinterpret_cast<const h8*>(smem + (c >> 3) * (BM * 128) + row * 128 + (c & 7) * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          s += f;
          q += f * f;
        }
      }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (part == 0) {
      const float m_ = s * (1.0f / C);
      stat[row * 2] = m_;
      stat[row * 2 + 1] = rsqrtf(fmaxf(q * (1.0f / C) - m_ * m_, 0.f) + g.ln_eps);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < MF; ++j) {
      mu[j] = stat[(arow + 16 * j) * 2];
      rstd[j] = stat[(arow + 16 * j) * 2 + 1];
    }
  }
  // (the first step barrier below separates these reads of the statistics from the first h-tile write)

  // ---- the feed-forward: NCH hidden chunks ------------------------------------------------------------------------------------
  const int hp = WN == 2 ? wn : (wn >> 1);           // which of a projection k-tile's two pieces holds this wave's S columns
  const int rbase = WN == 2 ? 0 : (wn & 1) * 32;     // ... and from which row of it
  char* const hbuf = smem + H_OFF;
#pragma unroll 1
  for (int j = 0; j < NCH; ++j) {
    f4 S[NF1][MF];
#pragma unroll
    for (int i = 0; i < NF1; ++i)
#pragma unroll
      for (int jj = 0; jj < MF; ++jj) S[i][jj] = f4{0.f, 0.f, 0.f, 0.f};
    // GEMM 1: K = C, one step per k-tile (two pieces: S columns 0..63 | 64..127)
#pragma unroll 1
    for (int t = 0; t < NKC; ++t) {
      step_begin(2);
      const char* const Wp = slot_ptr(hp);
      const char* const At = smem + t * BM * 128;
      h8 af[2][MF], wf[2][NF1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int jj = 0; jj < MF; ++jj) af[ks][jj] = frag(At, arow + 16 * jj, ks);
#pragma unroll
        for (int i = 0; i < NF1; ++i) wf[ks][i] = frag(Wp, rbase + i * 16 + lr, ks);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < NF1; ++i)
#pragma unroll
          for (int jj = 0; jj < MF; ++jj) S[i][jj] = FF_MFMA(wf[ks][i], af[ks][jj], S[i][jj]);
      step_end(2);
    }
    // folded LayerNorm + GEGLU on the accumulators -> fp16 h tile [BM][64] (k-tile layout)
    {
      const char* const sb = smem + S_OFF + wv * 1024 + (j & 1) * 512;
#pragma unroll
      for (int pr = 0; pr < NF1 / 2; ++pr) {
        const f4 s1a = *reinterpret_cast<const f4*>(sb + ((2 * pr) * 16 + lg * 4) * 4);
        const f4 s1g = *reinterpret_cast<const f4*>(sb + ((2 * pr + 1) * 16 + lg * 4) * 4);
        const f4 s0a = *reinterpret_cast<const f4*>(sb + 256 + ((2 * pr) * 16 + lg * 4) * 4);
        const f4 s0g = *reinterpret_cast<const f4*>(sb + 256 + ((2 * pr + 1) * 16 + lg * 4) * 4);
        const int col = wn * (64 / WN) + pr * 16 + lg * 4;   // hidden column inside the chunk
#pragma unroll
        for (int jj = 0; jj < MF; ++jj) {
          const int row = arow + 16 * jj;
          h4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float va = rstd[jj] * (S[2 * pr][jj][e] - mu[jj] * s1a[e]) + s0a[e];
            const float vg = rstd[jj] * (S[2 * pr + 1][jj][e] - mu[jj] * s1g[e]) + s0g[e];
            hv[e] = (half_t)(va * md::gelu_erf_f(vg));
          }
          *reinterpret_cast<h4*>(hbuf + row * 128 + (((col >> 3) ^ (row & 7)) << 4) + (col & 7) * 2) = hv;
        }
      }
    }
    // GEMM 2: K = the chunk's 64 hidden columns; the first step's barrier publishes the h tile
    gemm_c_rows(hbuf);
  }

  // ---- epilogue: + b2 (+ residual) -> hi / lo -------------------------------------------------------------------------------------
  const float* const b2 = g.b2[ps];
#pragma unroll
  for (int j = 0; j < MF; ++j) {
    const int mrow = m0 + arow + 16 * j;
    const long long m = min(mrow, Mlim - 1);
    h4 xv[NF2], xl[NF2];
#pragma unroll
    for (int f = 0; f < NF2; ++f) {
      xv[f] = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      xl[f] = xv[f];
    }
    if constexpr (!HEAD) {
#pragma unroll
      for (int f = 0; f < NF2; ++f) xv[f] = *reinterpret_cast<const h4*>(g.x + m * C + out_col(f));
      if (g.x_lo) {
#pragma unroll
        for (int f = 0; f < NF2; ++f) xl[f] = *reinterpret_cast<const h4*>(g.x_lo + m * C + out_col(f));
      }
    }
    if (mrow >= Mlim) continue;
#pragma unroll
    for (int f = 0; f < NF2; ++f) {
      const int n = out_col(f);
      f4 v = acc[f][j] + *reinterpret_cast<const f4*>(b2 + n);
      h4 o, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += (float)xv[f][e];
        v[e] += (float)xl[f][e];
        o[e] = (half_t)v[e];
        l[e] = (half_t)(v[e] - (float)o[e]);
      }
      *reinterpret_cast<h4*>(g.out + m * C + n) = o;
      if (g.out_lo) *reinterpret_cast<h4*>(g.out_lo + m * C + n) = l;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int C, int WM, int MF, int R, bool HEAD>
int launch_ff(const FfArgs& g, int tiles, hipStream_t s) {
  constexpr int BM = 16 * MF * WM;
  constexpr int lds = (C / 64) * BM * 128 + BM * 128 + 8 * 1024 + R * 8192;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set[64] = {};   // per DEVICE: the attribute belongs to the device's copy of the kernel
  int devi = 0;
  MD_HIP_CHECK(hipGetDevice(&devi));
  if (devi < 0 || devi >= 64 || !attr_set[devi]) {
    MD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_block_kernel<C, WM, MF, R, HEAD>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (devi >= 0 && devi < 64) attr_set[devi] = true;
  }
  hipLaunchKernelGGL((ff_block_kernel<C, WM, MF, R, HEAD>), dim3(tiles), dim3(512), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

template <int C, bool HEAD>
int launch_ff_bm(const FfArgs& g, int bm, int tiles, hipStream_t s) {
  if constexpr (C == 320) {
    if (bm == 128) return launch_ff<C, 4, 2, 7, HEAD>(g, tiles, s);
    if (bm == 64) return launch_ff<C, 4, 1, 10, HEAD>(g, tiles, s);
    if (bm == 32) return launch_ff<C, 2, 1, 10, HEAD>(g, tiles, s);
  } else {
    if (bm == 64) return launch_ff<C, 4, 1, 8, HEAD>(g, tiles, s);
    if (bm == 32) return launch_ff<C, 2, 1, 10, HEAD>(g, tiles, s);
  }
  return MD_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int md_ff_block_supported(int32_t m, int32_t c) {
  return (c == 320 || c == 640) && m > 0 ? 1 : 0;
}

extern "C" int md_ff_block(const md_ff_block_params* p, void* stream) {
  if (!p || !p->x || !p->w1 || !p->s1 || !p->s0 || !p->w2 || !p->b2 || !p->out) return MD_ERR_BAD_ARG;
  if (p->m <= 0 || !(p->ln_eps > 0.f)) return MD_ERR_BAD_ARG;
  if (p->c != 320 && p->c != 640) return MD_ERR_UNSUPPORTED;
  if ((p->attn != nullptr) != (p->wo != nullptr) || (p->attn != nullptr) != (p->bo != nullptr)) return MD_ERR_BAD_ARG;
  if ((long long)p->m * p->c * 2 >= (1LL << 31)) return MD_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(p->x) || !al16(p->x_lo) || !al16(p->attn) || !al16(p->out) || !al16(p->out_lo) || !al16(p->w1) || !al16(p->w2) ||
      !al16(p->wo) || !al16(p->s1) || !al16(p->s0) || !al16(p->b2) || !al16(p->bo))
    return MD_ERR_BAD_ARG;
  const bool dual = p->w1_2 && p->m_split > 0 && p->m_split < p->m;
  if (dual) {
    if (!p->s1_2 || !p->s0_2 || !p->w2_2 || !p->b2_2 || (p->attn && (!p->wo_2 || !p->bo_2))) return MD_ERR_BAD_ARG;
    if (!al16(p->w1_2) || !al16(p->w2_2) || !al16(p->wo_2) || !al16(p->s1_2) || !al16(p->s0_2) || !al16(p->b2_2) || !al16(p->bo_2))
      return MD_ERR_BAD_ARG;
  }
  const bool head = p->attn != nullptr;
  // rows per workgroup: the largest tile that still gives every CU a workgroup (a workgroup streams ALL the block's weights
  // through its LDS, so fewer, taller tiles move fewer L2 -> LDS bytes; a grid below the CU count leaves CUs idle)
  int bm = p->force_bm;
  if (bm == 0) {
    if (p->c == 320)
      bm = p->m >= 128 * 384 ? 128 : 64;
    else
      bm = 64;
  }
  if (bm != 32 && bm != 64 && bm != 128) return MD_ERR_BAD_ARG;
  FfArgs g;
  g.x = (const half_t*)p->x;
  g.x_lo = (const half_t*)p->x_lo;
  g.attn = (const half_t*)p->attn;
  g.out = (half_t*)p->out;
  g.out_lo = (half_t*)p->out_lo;
  g.M = p->m;
  g.ln_eps = p->ln_eps;
  g.w1[0] = (const half_t*)p->w1;
  g.s1[0] = p->s1;
  g.s0[0] = p->s0;
  g.w2[0] = (const half_t*)p->w2;
  g.b2[0] = p->b2;
  g.wo[0] = (const half_t*)p->wo;
  g.bo[0] = p->bo;
  g.w1[1] = dual ? (const half_t*)p->w1_2 : g.w1[0];
  g.s1[1] = dual ? p->s1_2 : g.s1[0];
  g.s0[1] = dual ? p->s0_2 : g.s0[0];
  g.w2[1] = dual ? (const half_t*)p->w2_2 : g.w2[0];
  g.b2[1] = dual ? p->b2_2 : g.b2[0];
  g.wo[1] = dual ? (const half_t*)p->wo_2 : g.wo[0];
  g.bo[1] = dual ? p->bo_2 : g.bo[0];
  g.m_split = dual ? p->m_split : 0x7fffffff;
  g.tiles_m1 = 0x7fffffff;
  int tiles = (p->m + bm - 1) / bm;
  if (dual) {
    g.tiles_m1 = (p->m_split + bm - 1) / bm;
    tiles = g.tiles_m1 + (p->m - p->m_split + bm - 1) / bm;
  }
  hipStream_t s = (hipStream_t)stream;
  const double C = p->c;
  char tag[96];
  snprintf(tag, sizeof(tag), "ff_block M=%d C=%d bm=%d head=%d dual=%d", p->m, p->c, bm, head ? 1 : 0, dual ? 1 : 0);
  md::ProfScope prof(MD_FAM_IGEMM, s, 2.0 * p->m * (12.0 * C * C + (head ? C * C : 0.0)),
                     (double)p->m * C * 2.0 * (head ? 5.0 : 4.0) + (12.0 + (head ? 1.0 : 0.0)) * C * C * 2.0, tag);
  if (p->c == 320) return head ? launch_ff_bm<320, true>(g, bm, tiles, s) : launch_ff_bm<320, false>(g, bm, tiles, s);
  return head ? launch_ff_bm<640, true>(g, bm, tiles, s) : launch_ff_bm<640, false>(g, bm, tiles, s);
}
