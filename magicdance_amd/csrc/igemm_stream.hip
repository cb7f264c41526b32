// md_igemm, STATIC ring form for the 3x3 convs of the low-resolution levels (round 5; tile configs 65.., see igemm_ring.hip::kRing).
//
// The 8x8 / 16x16 layers of a one-frame DDIM step stream 30-60 MB of cold weights through a few hundred 64-row output tiles: a
// workgroup does 8 MFMAs per wave and k-tile, so its time is the per-k-tile bookkeeping and latency, not arithmetic.  The ring form
// of round 4 (igemm_ring.hip) keeps that bookkeeping at run time -- instruction counters, a queue of per-step sequence numbers, a
// 40-way branch tree around `s_waitcnt vmcnt(n)`, tap / channel-block decoding per issued tile: ~130 scalar instructions per wave
// and step, and the CU's one scalar unit serialises its waves (the same finding as md_ff_block, profiles/round5_ffblock_bench.txt:
// 152 -> 52 us from making that bookkeeping static).  Here everything a step does is a compile-time function of the step:
//   * W ring = 9 slots = the nine taps of ONE 64-channel block: tap t of channel block cb always lives in slot t;
//   * a step = 3 taps (24 MFMAs per wave): wait (immediate count), one raw barrier, refill the three slots the previous step freed
//     with the same taps of the next channel block, compute;
//   * the haloed A block of a channel block (BM + 2 win + 2 pixels x 64 channels, every pixel loaded once for all nine taps --
//     igemm_ring.hip) is double-buffered by channel-block parity and re-issued once per channel block;
//   * every wave issues the same number of LDS-DMA instructions per tile (A blocks are padded to 8 x waves rows with clamped,
//     never-read rows), so the counts below hold for every wave:
//       program order per channel block:  [step 0: A(cb+1), W(cb, taps 6-8)] [step 1: W(cb+1, 0-2)] [step 2: W(cb+1, 3-5)]
//       step 0 needs A(cb), W(cb, 0-2): younger = W(cb, 3-5)                       -> vmcnt(3 WJ)
//       step 1 needs W(cb, 3-5):        younger = A(cb+1), W(cb, 6-8)              -> vmcnt(AJ + 3 WJ)
//       step 2 needs W(cb, 6-8):        younger = W(cb+1, 0-2)                     -> vmcnt(3 WJ)
//     i.e. six taps (48 KiB at BN = 64) + an A block stay in flight across every barrier.
// Same tile mapping, fragment layout, split-K over whole channel blocks and epilogue (igemm_core.h) as the other md_igemm kernels.
#include <type_traits>
#include <utility>

#include "igemm_core.h"

namespace mdig {
namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <int W>
__device__ __forceinline__ void wait_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(W >= 0 && W < 60, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W) : "memory");
#endif
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int AJ>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_stream_kernel(const IgemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int TILE_W = BN * 128;
  constexpr int WJ = BN / 8 / NW;              // LDS-DMA instructions per wave and W tile
  constexpr int AR = AJ * 8 * NW;              // rows of an A block (>= BM + 2 win + 2: launcher)
  constexpr int A_BYTES = AR * 128;
  constexpr int W_BYTES = 9 * TILE_W;
  constexpr int A_OFF = W_BYTES, ZERO_OFF = A_OFF + 2 * A_BYTES, LDS_TOTAL = ZERO_OFF + 128;
  static_assert(BN % (8 * NW) == 0 && BM % 32 == 0 && WTM <= 64 && LDS_TOTAL <= 160 * 1024, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wv % WAVES_M, wn = wv / WAVES_M;

  // ---- tile of this workgroup: same mapping as igemm.hip ------------------------------------------------------------------------
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8g = nwg & 7;
  const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (bid >> 3);
  const int per_group = g.group_m * g.tiles_n;
  const int grp = logical / per_group, in_grp = logical - grp * per_group;
  const int first_m = grp * g.group_m;
  const int gsz = min(g.tiles_m - first_m, g.group_m);
  const int tile_n = in_grp / gsz, tile_m = first_m + (in_grp - tile_n * gsz);
  const bool set2 = tile_m >= g.tiles_m1;
  const int m0 = set2 ? g.m_split + (tile_m - g.tiles_m1) * BM : tile_m * BM;
  const int Mlim = set2 ? g.M : min(g.M, g.m_split);
  const int n0 = tile_n * BN;
  const int kz = blockIdx.z;
  const int cb_begin = (kz * g.tiles_per_split) / 9;                       // (tiles_per_split is a multiple of 9: whole channel blocks)
  const int cb_end = min(g.nk, kz * g.tiles_per_split + g.tiles_per_split) / 9;
  const half_t* const gw = set2 ? g.w2 : g.w;

  // ---- loader role ------------------------------------------------------------------------------------------------------------
  const int r8 = lane >> 3, c8 = lane & 7;
  const unsigned gcb = (unsigned)(c8 ^ r8) * 16u;
  const int mtot = g.batch * g.hin * g.win;   // source pixels (stride 1: = M)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(gw), 0, g.N * g.K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.a0), 0, mtot * g.c0 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.a1 ? g.a1 : g.a0), 0, mtot * (g.a1 ? g.c1 : g.c0) * 2, 0x00020000);
  unsigned w_off[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) w_off[j] = w_row_offset(min(n0 + (wv + NW * j) * 8 + r8, g.N - 1), g) + gcb;
  const unsigned tap_stride = g.w_tiled ? 2048u : (unsigned)g.cin * 2u;      // W: k offset of one tap inside a channel block
  const unsigned cb_stride = g.w_tiled ? 9u * 2048u : 128u;                   // ... and of one channel block
  // A block: pixels p_lo .. p_lo + AR - 1 (clamped into the tensor; rows beyond BM + 2 win + 2 are never read)
  const int p_lo = m0 - (g.win + 1);
  unsigned a_vo0[AJ], a_vo1[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const unsigned p = (unsigned)min(max(p_lo + (wv + NW * j) * 8 + r8, 0), mtot - 1);
    a_vo0[j] = p * (unsigned)g.c0 * 2u + gcb;
    a_vo1[j] = p * (unsigned)g.c1 * 2u + gcb;
  }
  char* const w_dst = smem + wv * 1024;             // + slot * TILE_W + NW * 1024 * j
  char* const a_dst = smem + A_OFF + wv * 1024;     // + parity * A_BYTES + NW * 1024 * j

  auto issue_a = [&](int cb, int parity) {
    const int cc = cb * 64;
    const bool second = cc >= g.c0;
    const unsigned soff = (unsigned)(second ? cc - g.c0 : cc) * 2u;
    char* const d = a_dst + parity * A_BYTES;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      if (second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (__attribute__((address_space(3))) void*)(d + NW * 1024 * j), 16, a_vo1[j], soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(d + NW * 1024 * j), 16, a_vo0[j], soff, 0, 0);
    }
  };
  auto issue_w3 = [&](int cb, auto t0c) {   // taps T0 .. T0 + 2 of channel block cb -> slots T0 .. T0 + 2
    constexpr int t0 = decltype(t0c)::value;
    const unsigned base = (unsigned)cb * cb_stride;
#pragma unroll
    for (int t = t0; t < t0 + 3; ++t) {
      const unsigned soff = base + (unsigned)t * tap_stride;
#pragma unroll
      for (int j = 0; j < WJ; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(w_dst + t * TILE_W + NW * 1024 * j), 16,
                                                 w_off[j], soff, 0, 0);
    }
  };

  // ---- prologue: the zero row, A(cb_begin), W(cb_begin, taps 0-5) -----------------------------------------------------------------
  if (tid < 8) *reinterpret_cast<f4*>(smem + ZERO_OFF + tid * 16) = f4{0.f, 0.f, 0.f, 0.f};
  if (cb_begin < cb_end) {
    issue_a(cb_begin, 0);
    issue_w3(cb_begin, std::integral_constant<int, 0>{});
    issue_w3(cb_begin, std::integral_constant<int, 3>{});
  }

  // ---- compute role -------------------------------------------------------------------------------------------------------------
  // per m-fragment the 9-bit mask of taps that fall inside the image for this lane's output pixel
  int amask[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int m = min(m0 + wm * WTM + i * 16 + lr, Mlim - 1);
    const int b = fast_div(m, g.div_tok_mul, g.div_tok_sh);
    const int rem = m - b * g.tokens;
    const int oy = fast_div(rem, g.div_w_mul, g.div_w_sh);
    const int ox = rem - oy * g.wout;
    int cx = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) cx |= ((unsigned)(ox + d - 1) < (unsigned)g.win) ? (1 << d) : 0;
    int mask = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) mask |= ((unsigned)(oy + d - 1) < (unsigned)g.hin) ? (cx << (3 * d)) : 0;
    amask[i] = mask;
  }
  // block-row byte offset of this lane's row of m-fragment i at tap (0, 0), with the swizzle of k-step 0 applied at read time
  f4 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float ln_sum[MF], ln_sq[MF];

  auto compute_tap = [&](auto tc, int ablk) {
    constexpr int t = decltype(tc)::value;
    constexpr int dy = t / 3, dx = t % 3;
    const int tapoff = dy * g.win + dx;   // block row of this tap = (m - m0) + tapoff
    const char* const Wt = smem + t * TILE_W;
    h8 af[2][MF], wf[2][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int row = wm * WTM + i * 16 + lr + tapoff;
        int off = ablk + row * 128 + ((lg ^ (row & 7)) << 4);
        off = ((amask[i] >> t) & 1) ? off : ZERO_OFF;
        af[ks][i] = *reinterpret_cast<const h8*>(smem + (off ^ (ks << 6)));   // ks 1: chunk 4 + lg of the same row (the zero row is 128 B)
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int row = wn * WTN + i * 16 + lr;
        wf[ks][i] = *reinterpret_cast<const h8*>(Wt + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
  };

  int par = 0;
#pragma unroll 1
  for (int cb = cb_begin; cb < cb_end; ++cb) {
    const int cbn = min(cb + 1, cb_end - 1);   // (past the end: valid addresses, loaded but never read)
    const int ablk = A_OFF + par * A_BYTES;
    // step 0
    wait_barrier<3 * WJ>();
    issue_a(cbn, par ^ 1);
    issue_w3(cb, std::integral_constant<int, 6>{});
    sfor<3>([&](auto tc) { compute_tap(std::integral_constant<int, decltype(tc)::value>{}, ablk); });
    // step 1
    wait_barrier<AJ + 3 * WJ>();
    issue_w3(cbn, std::integral_constant<int, 0>{});
    sfor<3>([&](auto tc) { compute_tap(std::integral_constant<int, 3 + decltype(tc)::value>{}, ablk); });
    // step 2
    wait_barrier<3 * WJ>();
    issue_w3(cbn, std::integral_constant<int, 3>{});
    sfor<3>([&](auto tc) { compute_tap(std::integral_constant<int, 6 + decltype(tc)::value>{}, ablk); });
    par ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads issued past the end of this workgroup's channel blocks

  // the epilogue's view of the launch arguments is read from the kernarg segment HERE (scalar loads behind an opaque pointer): as
  // fields of the by-value argument they would be held in SGPRs across the loop (the first build spilled 45 of them, 52 bytes of
  // scratch per lane -- and scratch traffic is VMEM traffic: it would break the instruction counts of the waits above)
  const __attribute__((address_space(4))) IgemmArgs* gp = (const __attribute__((address_space(4))) IgemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(gp));
  const IgemmArgs ge = *gp;
  igemm_epilogue<BM, BN, WAVES_M, WAVES_N, false, 1, LDS_TOTAL>(ge, smem, acc, ln_sum, ln_sq, tid, 0, wm, wn, m0, n0, Mlim, kz,
                                                                set2 ? ge.bias2 : ge.bias, nullptr, nullptr);
#endif  // __HIP_DEVICE_COMPILE__
}

// The same for 1x1 convs / linear layers (configs 67.., round 5): a ring of R slots, each one k-tile of A ([BM][64]) and of W ([BN][64]);
// a step = TPS k-tiles; the unrolled loop body covers R k-tiles, so slot and wait count are compile-time again:
//   step j of a body issues the k-tiles R ahead of the ones step j - 1 consumed (into that step's slots) -> before a step's wait the
//   wave has issued (R - 2 TPS) k-tiles = (R - 2 TPS) (AJ1 + WJ) instructions after the step's own.
// K = 1280 linears of the 8x8 / 16x16 levels (20 k-tiles, M = 64 .. 768): the 2-stage kernels pay one L2 / HBM round trip per pipeline
// stage (vmcnt(0)); here 6 of 8 slots (96 KiB per CU) are in flight across every barrier.  Folded LayerNorm as in the other kernels.
template <int BM, int BN, int WAVES_M, int WAVES_N, int R, int TPS, bool LN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_stream1_kernel(const IgemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int TILE_A = BM * 128, TILE_W = BN * 128, SLOT = TILE_A + TILE_W;
  constexpr int AJ1 = BM / 8 / NW, WJ = BN / 8 / NW, IPT = AJ1 + WJ;
  constexpr int LDS_TOTAL = R * SLOT;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WTM <= 64 && LDS_TOTAL <= 160 * 1024 && R % TPS == 0 && R >= 2 * TPS, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wv % WAVES_M, wn = wv / WAVES_M;

  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8g = nwg & 7;
  const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (bid >> 3);
  const int per_group = g.group_m * g.tiles_n;
  const int grp = logical / per_group, in_grp = logical - grp * per_group;
  const int first_m = grp * g.group_m;
  const int gsz = min(g.tiles_m - first_m, g.group_m);
  const int tile_n = in_grp / gsz, tile_m = first_m + (in_grp - tile_n * gsz);
  const bool set2 = tile_m >= g.tiles_m1;
  const int m0 = set2 ? g.m_split + (tile_m - g.tiles_m1) * BM : tile_m * BM;
  const int Mlim = set2 ? g.M : min(g.M, g.m_split);
  const int n0 = tile_n * BN;
  const int kz = blockIdx.z;
  const int kt_begin = kz * g.tiles_per_split;
  const int nkt = max(0, min(g.nk, kt_begin + g.tiles_per_split) - kt_begin);
  const half_t* const gw = set2 ? g.w2 : g.w;

  const int r8 = lane >> 3, c8 = lane & 7;
  const unsigned gcb = (unsigned)(c8 ^ r8) * 16u;
  const int mtot = g.batch * g.hin * g.win;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(gw), 0, g.N * g.K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.a0), 0, mtot * g.c0 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.a1 ? g.a1 : g.a0), 0, mtot * (g.a1 ? g.c1 : g.c0) * 2, 0x00020000);
  unsigned w_off[WJ], ra0[AJ1], ra1[AJ1];
#pragma unroll
  for (int j = 0; j < WJ; ++j) w_off[j] = w_row_offset(min(n0 + (wv + NW * j) * 8 + r8, g.N - 1), g) + gcb;
#pragma unroll
  for (int j = 0; j < AJ1; ++j) {
    const unsigned m = (unsigned)min(m0 + (wv + NW * j) * 8 + r8, Mlim - 1);   // rows past the end: clamped, never stored
    ra0[j] = m * (unsigned)g.c0 * 2u + gcb;
    ra1[j] = m * (unsigned)g.c1 * 2u + gcb;
  }
  const unsigned wk_stride = g.w_tiled ? 2048u : 128u;
  char* const dst0 = smem + wv * 1024;
  const int c0v = g.c0;

  // k-tile ``rel`` of this workgroup (clamped to its last one: loads past the end read valid addresses and are never consumed) -> slot
  auto issue_tile = [&](int rel, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    const int kt = kt_begin + min(rel, nkt - 1);
    const int cc = kt * 64;
    const bool second = cc >= c0v;
    const unsigned soff = (unsigned)(second ? cc - c0v : cc) * 2u;
    char* const d = dst0 + slot * SLOT;
#pragma unroll
    for (int j = 0; j < AJ1; ++j) {
      if (second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (__attribute__((address_space(3))) void*)(d + NW * 1024 * j), 16, ra1[j], soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (__attribute__((address_space(3))) void*)(d + NW * 1024 * j), 16, ra0[j], soff, 0, 0);
    }
    const unsigned ksoff = (unsigned)kt * wk_stride;
#pragma unroll
    for (int j = 0; j < WJ; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(d + TILE_A + NW * 1024 * j), 16, w_off[j], ksoff, 0, 0);
  };

  f4 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float ln_sum[MF], ln_sq[MF];
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < MF; ++i) ln_sum[i] = ln_sq[i] = 0.f;
  }

  auto compute_tile = [&](auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    const char* const At = smem + slot * SLOT;
    const char* const Wt = At + TILE_A;
    h8 af[2][MF], wf[2][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int row = wm * WTM + i * 16 + lr;
        af[ks][i] = *reinterpret_cast<const h8*>(At + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int row = wn * WTN + i * 16 + lr;
        wf[ks][i] = *reinterpret_cast<const h8*>(Wt + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (LN) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const h2v ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          if (WAVES_N > 1 && (i % WAVES_N) != wn) continue;   // (the n-waves of a wave row share the statistics work: igemm.hip)
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
        for (int j = 0; j < MF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
    }
  };

  // prologue: k-tiles 0 .. R - TPS - 1 (as if a step of TPS tiles had just run)
  if (nkt > 0) sfor<R - TPS>([&](auto tc) { issue_tile(decltype(tc)::value, tc); });
#pragma unroll 1
  for (int base = 0; base < nkt; base += R) {
    sfor<R / TPS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (base + j * TPS < nkt) {   // (uniform: the steps past this workgroup's last k-tile are skipped by every wave)
        wait_barrier<(R - 2 * TPS) * IPT>();
        // refill the slots of the previous step with the k-tiles R further on
        sfor<TPS>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int slot = ((j + R / TPS - 1) % (R / TPS)) * TPS + k;
          issue_tile(base + (j - 1) * TPS + k + R, std::integral_constant<int, slot>{});
        });
        sfor<TPS>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if (base + j * TPS + k < nkt) compute_tile(std::integral_constant<int, j * TPS + k>{});
        });
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const __attribute__((address_space(4))) IgemmArgs* gp = (const __attribute__((address_space(4))) IgemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(gp));
  const IgemmArgs ge = *gp;
  igemm_epilogue<BM, BN, WAVES_M, WAVES_N, LN, 1, LDS_TOTAL>(ge, smem, acc, ln_sum, ln_sq, tid, 0, wm, wn, m0, n0, Mlim, kz,
                                                             set2 ? ge.bias2 : ge.bias, set2 ? ge.ln2_s1 : ge.ln_s1, set2 ? ge.ln2_s0 : ge.ln_s0);
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WMv, int WNv, int R, int TPS, bool LN>
int launch_stream1_k(const IgemmArgs& g, hipStream_t s) {
  constexpr size_t lds = (size_t)R * (BM + BN) * 128;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set[64] = {};   // per DEVICE: the attribute belongs to the device's copy of the kernel
  int devi = 0;
  MD_HIP_CHECK(hipGetDevice(&devi));
  if (devi < 0 || devi >= 64 || !attr_set[devi]) {
    MD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_stream1_kernel<BM, BN, WMv, WNv, R, TPS, LN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (devi >= 0 && devi < 64) attr_set[devi] = true;
  }
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.splitk);
  hipLaunchKernelGGL((igemm_stream1_kernel<BM, BN, WMv, WNv, R, TPS, LN>), grid, dim3(64 * WMv * WNv), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

template <int BM, int BN, int WMv, int WNv, int AJ>
int launch_stream_k(const IgemmArgs& g, hipStream_t s) {
  constexpr int NW = WMv * WNv;
  constexpr size_t lds = (size_t)9 * BN * 128 + 2 * (size_t)AJ * 8 * NW * 128 + 128;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set[64] = {};   // per DEVICE: the attribute belongs to the device's copy of the kernel
  int devi = 0;
  MD_HIP_CHECK(hipGetDevice(&devi));
  if (devi < 0 || devi >= 64 || !attr_set[devi]) {
    MD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_stream_kernel<BM, BN, WMv, WNv, AJ>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (devi >= 0 && devi < 64) attr_set[devi] = true;
  }
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.splitk);
  hipLaunchKernelGGL((igemm_stream_kernel<BM, BN, WMv, WNv, AJ>), grid, dim3(64 * NW), lds, s, g);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

}  // namespace

// rows of the haloed A block in units of 8 x waves rows; 0: this image width is not served
int stream_aj(int bm, int waves, int win) {
  const int aj = (bm + 2 * win + 2 + 8 * waves - 1) / (8 * waves);
  if (bm == 64 && waves == 4) return (aj >= 3 && aj <= 7) ? aj : (aj < 3 ? 3 : 0);
  if (bm == 128 && waves == 4) return (aj >= 5 && aj <= 9) ? aj : (aj < 5 ? 5 : 0);
  return 0;
}

long long stream_lds_bytes(int bm, int bn, int waves, int win) {
  const int aj = stream_aj(bm, waves, win);
  if (aj == 0) return 1LL << 40;
  return (long long)9 * bn * 128 + 2LL * aj * 8 * waves * 128 + 128;
}

int igemm_stream_launch(const IgemmArgs& g, int bm, int bn, hipStream_t s) {
  if (g.ksize != 3 || g.stride != 1 || g.ups || bn != 64) return MD_ERR_UNSUPPORTED;
  const int aj = stream_aj(bm, 4, g.win);
  if (bm == 64) {
    switch (aj) {
      case 3: return launch_stream_k<64, 64, 2, 2, 3>(g, s);
      case 4: return launch_stream_k<64, 64, 2, 2, 4>(g, s);
      case 5: return launch_stream_k<64, 64, 2, 2, 5>(g, s);
      case 6: return launch_stream_k<64, 64, 2, 2, 6>(g, s);
      case 7: return launch_stream_k<64, 64, 2, 2, 7>(g, s);
      default: return MD_ERR_UNSUPPORTED;
    }
  }
  if (bm == 128) {
    switch (aj) {
      case 5: return launch_stream_k<128, 64, 2, 2, 5>(g, s);
      case 6: return launch_stream_k<128, 64, 2, 2, 6>(g, s);
      case 7: return launch_stream_k<128, 64, 2, 2, 7>(g, s);
      case 8: return launch_stream_k<128, 64, 2, 2, 8>(g, s);
      case 9: return launch_stream_k<128, 64, 2, 2, 9>(g, s);
      default: return MD_ERR_UNSUPPORTED;
    }
  }
  return MD_ERR_UNSUPPORTED;
}

// the 1x1 / linear form: (bm, bn) -> instantiation; LayerNorm folding by the launch arguments
int igemm_stream1_launch(const IgemmArgs& g, int bm, int bn, hipStream_t s) {
  if (g.ksize != 1 || g.stride != 1 || g.ups || bn != 64) return MD_ERR_UNSUPPORTED;
  if (bm == 64) return g.ln_s1 ? launch_stream1_k<64, 64, 2, 2, 8, 2, true>(g, s) : launch_stream1_k<64, 64, 2, 2, 8, 2, false>(g, s);
  if (bm == 128) return g.ln_s1 ? launch_stream1_k<128, 64, 2, 2, 6, 2, true>(g, s) : launch_stream1_k<128, 64, 2, 2, 6, 2, false>(g, s);
  return MD_ERR_UNSUPPORTED;
}

}  // namespace mdig
