// Small streaming / GEMV kernels of the sampling loop (gfx950): layout conversion at the NCHW-fp32 boundary of
// apply_model, pose-residual adds, timestep embedding + embedding MLP GEMVs, per-step table selection for the
// captured step graph, and the fused CFG + DDIM update.  All HBM/L2-bound, vectorised where the layout allows.
#include "md_common.h"

namespace {

__global__ __launch_bounds__(256) void nchw_to_nhwc_f16(const float* __restrict__ x, half_t* __restrict__ out, int c,
                                                        int hw, int cpad, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index over [b][p][cpad]
  if (i >= total) return;
  const int cc = (int)(i % cpad);
  const long long bp = i / cpad;
  const int p = (int)(bp % hw);
  const long long b = bp / hw;
  out[i] = cc < c ? (half_t)x[(b * c + cc) * hw + p] : (half_t)0.f;
}

template <bool F32>
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32(const void* __restrict__ x, float* __restrict__ out, int c,
                                                        int hw, int ld, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index over [b][c][p]
  if (i >= total) return;
  const int p = (int)(i % hw);
  const long long bc = i / hw;
  const int cc = (int)(bc % c);
  const long long b = bc / c;
  const long long src = (b * hw + p) * ld + cc;
  out[i] = F32 ? reinterpret_cast<const float*>(x)[src] : (float)reinterpret_cast<const half_t*>(x)[src];
}

__global__ __launch_bounds__(256) void add_f16(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                               half_t* __restrict__ out, long long n8, long long period8) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n8; i += stride) {
    const h8 va = reinterpret_cast<const h8*>(a)[i];
    const h8 vb = reinterpret_cast<const h8*>(b)[i % period8];
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)va[e] + (float)vb[e]);
    reinterpret_cast<h8*>(out)[i] = o;
  }
}

__global__ void timestep_embedding(const float* __restrict__ t, float* __restrict__ out, int nt, int dim,
                                   float max_period) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= nt * dim) return;
  const int r = i / dim, j = i - r * dim;
  float v = 0.f;
  if (j < 2 * half) {
    const int f = j < half ? j : j - half;
    // util.py:199-205: freqs = exp(-log(max_period) * arange(half) / half); emb = [cos(args), sin(args)]
    const float freq = expf(-logf(max_period) * (float)f / (float)half);
    const float arg = t[r] * freq;
    v = j < half ? cosf(arg) : sinf(arg);
  }
  out[i] = v;
}

// y[r][n] = bias[n] + sum_k act(x[r][k]) w[n][k]; one wave per output column n, up to 8 rows per pass.
template <int R>
__global__ __launch_bounds__(256) void gemv_f32(const float* __restrict__ x, const half_t* __restrict__ w,
                                                const float* __restrict__ bias, float* __restrict__ y, int rows, int k,
                                                int n, int act_in) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 4 + wave;
  if (col >= n) return;
  const int r0 = blockIdx.y * R;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  const half_t* wr = w + (long long)col * k;
  for (int kk = lane * 8; kk < k; kk += 64 * 8) {
    const h8 wv = *reinterpret_cast<const h8*>(wr + kk);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r0 + r < rows) {
        const float* xr = x + (long long)(r0 + r) * k + kk;
        const f4 x0 = *reinterpret_cast<const f4*>(xr), x1 = *reinterpret_cast<const f4*>(xr + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xv = e < 4 ? x0[e] : x1[e - 4];
          if (act_in) xv = md::silu_f(xv);
          acc[r] += xv * (float)wv[e];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float s = md::wave_sum(acc[r]);
    if (lane == 0 && r0 + r < rows) y[(long long)(r0 + r) * n + col] = s + (bias ? bias[col] : 0.f);
  }
}

// The row index comes from a device-side counter that a captured step graph advances on every replay: it is clamped to the
// table ([0, nrows)), so a replay past the last step re-reads the last row instead of running off the end of the table.
__global__ void select_row_f32(const float* __restrict__ table, const int* __restrict__ counter, int row_offset, int nrows,
                               float* __restrict__ dst, int width) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= width) return;
  const int row = min(max((counter ? *counter : 0) + row_offset, 0), nrows - 1);
  dst[i] = table[(long long)row * width + i];
}

// one blockIdx.y per segment; seg = (offset of the segment inside a row block, row length, dst offset) in 16-byte units.
// Table layout: [row block][segment][rows_per_block][row length]: a block of rows_per_block consecutive DDIM rows is one
// contiguous piece of memory (what one batched appearance pass produces, and what one rank contributes to an all-gather).
__global__ __launch_bounds__(256) void gather_rows(const uint4* __restrict__ table, const long long* __restrict__ seg,
                                                   const int* __restrict__ counter, int row_offset, int nrows,
                                                   int rows_per_block, long long block_units, uint4* __restrict__ dst) {
  const long long* sg = seg + 3 * blockIdx.y;
  const long long len = sg[1];
  const int row = min(max((counter ? *counter : 0) + row_offset, 0), nrows - 1);
  const int blk = row / rows_per_block, within = row - blk * rows_per_block;
  const uint4* src = table + (long long)blk * block_units + sg[0] + (long long)within * len;
  uint4* d = dst + sg[2];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long long)gridDim.x * 256) d[i] = src[i];
}

// decoded frames -> image bytes: out[b][p][c] = uint8(clamp(x[b][c][p] * scale + bias, 0, 1) * 255 + 0.5)
__global__ __launch_bounds__(256) void image_to_u8(const float* __restrict__ x, unsigned char* __restrict__ out, int c, int hw,
                                                   float scale, float bias, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, p): one pixel per thread
  if (i >= total) return;
  const long long b = i / hw;
  const int p = (int)(i - b * hw);
  for (int ch = 0; ch < c; ++ch) {
    const float v = fminf(fmaxf(x[(b * c + ch) * hw + p] * scale + bias, 0.f), 1.f);
    out[i * c + ch] = (unsigned char)(v * 255.0f + 0.5f);
  }
}

__global__ void counter_add(int* counter, int delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter += delta;
}

__global__ __launch_bounds__(256) void ddim_update(const float* __restrict__ eps_c, const float* __restrict__ eps_u,
                                                   int ld_eps, const float* __restrict__ x,
                                                   const float* __restrict__ noise, const float* __restrict__ coef,
                                                   float* __restrict__ x_prev, float* __restrict__ pred_x0,
                                                   float* __restrict__ eps_out, int c, int hw, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // NCHW index
  if (i >= total) return;
  const int p = (int)(i % hw);
  const long long bc = i / hw;
  const int cc = (int)(bc % c);
  const long long b = bc / c;
  const long long e_idx = (b * hw + p) * ld_eps + cc;
  const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], s1m = coef[3], scale = coef[4];
  float e = eps_c[e_idx];
  if (eps_u) {
    const float eu = eps_u[e_idx];
    e = eu + scale * (e - eu);  // ddim.py:605
  }
  const float xv = x[i];
  const float px0 = (xv - s1m * e) / sqrtf(a_t);                      // ddim.py:624
  const float dir = sqrtf(1.0f - a_prev - sigma * sigma) * e;         // ddim.py:640
  float xp = sqrtf(a_prev) * px0 + dir;                               // ddim.py:644
  if (noise) xp += sigma * noise[i];
  x_prev[i] = xp;
  if (pred_x0) pred_x0[i] = px0;
  if (eps_out) eps_out[i] = e;
}

}  // namespace

extern "C" int md_nchw_to_nhwc_f16(const float* x, void* out, int32_t batch, int32_t c, int32_t hw, int32_t cpad,
                                   void* stream) {
  if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || cpad < c) return MD_ERR_BAD_ARG;
  const long long total = (long long)batch * hw * cpad;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)total * 2.0 + (double)batch * c * hw * 4.0);
  hipLaunchKernelGGL(nchw_to_nhwc_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (half_t*)out, c, hw,
                     cpad, total);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_nhwc_to_nchw_f32(const void* x, int32_t x_is_f32, float* out, int32_t batch, int32_t c, int32_t hw,
                                   int32_t ld, void* stream) {
  if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || ld < c) return MD_ERR_BAD_ARG;
  const long long total = (long long)batch * hw * c;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)total * 8.0);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (x_is_f32)
    hipLaunchKernelGGL(nhwc_to_nchw_f32<true>, grid, dim3(256), 0, s, x, out, c, hw, ld, total);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_f32<false>, grid, dim3(256), 0, s, x, out, c, hw, ld, total);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_add_f16(const void* a, const void* b, void* out, int64_t n, int64_t b_period, void* stream) {
  if (!a || !b || !out || n <= 0 || (n & 7) || b_period <= 0 || (b_period & 7)) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)n * 6.0);
  const long long n8 = n >> 3;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_f16, dim3((unsigned)blocks), dim3(256), 0, s, (const half_t*)a, (const half_t*)b, (half_t*)out,
                     n8, (long long)(b_period >> 3));
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_timestep_embedding(const float* t, float* out, int32_t nt, int32_t dim, float max_period,
                                     void* stream) {
  if (!t || !out || nt <= 0 || dim <= 0) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)nt * dim * 4.0);
  hipLaunchKernelGGL(timestep_embedding, dim3((nt * dim + 255) / 256), dim3(256), 0, s, t, out, nt, dim, max_period);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_gemv_f32(const float* x, const void* w, const float* bias, float* y, int32_t rows, int32_t k,
                           int32_t n, int32_t act_in, void* stream) {
  if (!x || !w || !y || rows <= 0 || rows > 1024 || k <= 0 || (k & 7) || n <= 0) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 2.0 * rows * (double)k * n, (double)n * k * 2.0);
  hipLaunchKernelGGL(gemv_f32<8>, dim3((n + 3) / 4, (rows + 7) / 8), dim3(256), 0, s, x, (const half_t*)w, bias, y, rows,
                     k, n, act_in);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_select_row_f32(const float* table, const int32_t* row_counter, int32_t row_offset, int32_t nrows,
                                 float* dst, int32_t width, void* stream) {
  if (!table || !dst || width <= 0 || nrows <= 0) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)width * 8.0);
  hipLaunchKernelGGL(select_row_f32, dim3((width + 255) / 256), dim3(256), 0, s, table, row_counter, row_offset, nrows, dst,
                     width);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_gather_rows(const void* table, const int64_t* seg, int32_t nseg, int64_t max_row_units,
                              const int32_t* row_counter, int32_t row_offset, int32_t nrows, int32_t rows_per_block,
                              int64_t block_units, void* dst, void* stream) {
  if (!table || !seg || !dst || nseg <= 0 || max_row_units <= 0 || nrows <= 0 || rows_per_block <= 0 || block_units < 0)
    return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, 0.0);
  long long gx = (max_row_units + 1023) / 1024;  // ~4 units per thread
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(gather_rows, dim3((unsigned)gx, (unsigned)nseg), dim3(256), 0, s, (const uint4*)table,
                     (const long long*)seg, row_counter, row_offset, nrows, rows_per_block, (long long)block_units, (uint4*)dst);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_image_to_u8(const float* x, void* out, int32_t batch, int32_t c, int32_t hw, float scale, float bias,
                              void* stream) {
  if (!x || !out || batch <= 0 || c <= 0 || c > 4 || hw <= 0) return MD_ERR_BAD_ARG;
  const long long total = (long long)batch * hw;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)total * c * 5.0);
  hipLaunchKernelGGL(image_to_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (unsigned char*)out, c, hw, scale,
                     bias, total);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_counter_add(int32_t* counter, int32_t delta, void* stream) {
  if (!counter) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(counter_add, dim3(1), dim3(64), 0, s, counter, delta);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

// ---- temporal overlap sampling (ddim.py:569-594) inside a captured step: windows of frames picked by a device index table ---------
// idx_table [steps][windows][n_idx] int32; the step is the device counter (clamped to the table), the window a launch argument.
namespace {
__global__ __launch_bounds__(256) void gather_frames_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                            const int32_t* __restrict__ idx_table, const int32_t* __restrict__ counter,
                                                            int steps, int windows, int window, int n_idx, long long row_vec) {
  const int st = min(max(counter ? counter[0] : 0, 0), steps - 1);
  const int32_t* idx = idx_table + ((long long)st * windows + window) * n_idx;
  const long long total = (long long)n_idx * row_vec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i / row_vec);
    dst[i] = src[(long long)idx[j] * row_vec + (i - (long long)j * row_vec)];
  }
}
// pred[idx[j]][p][ch] += e_u + scale (e_c - e_u) for the window's n_idx frames (distinct inside a window; windows of a step run one
// after the other on the stream), counts[idx[j]] += 1.  eps: NHWC fp32 [n_idx][hw][ld_eps]; pred: [frames][hw][c]
__global__ __launch_bounds__(256) void cfg_scatter_add_kernel(const float* __restrict__ eps_c, const float* __restrict__ eps_u, int ld_eps,
                                                              const float* __restrict__ coef, const int32_t* __restrict__ idx_table,
                                                              const int32_t* __restrict__ counter, int steps, int windows, int window,
                                                              int n_idx, float* __restrict__ pred, float* __restrict__ counts, int hw, int c) {
  const int st = min(max(counter ? counter[0] : 0, 0), steps - 1);
  const int32_t* idx = idx_table + ((long long)st * windows + window) * n_idx;
  const float scale = coef[4];
  const long long per = (long long)hw * c, total = (long long)n_idx * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i / per);
    const long long r = i - (long long)j * per;
    const long long pix = r / c;
    const int ch = (int)(r - pix * c);
    const long long e = ((long long)j * hw + pix) * ld_eps + ch;
    const float eu = eps_u[e];
    pred[(long long)idx[j] * per + r] += eu + scale * (eps_c[e] - eu);
    if (r == 0) counts[idx[j]] += 1.0f;
  }
}
// eps[f] = pred[f] / counts[f] (every frame is visited by at least one window), then pred and counts are cleared for the next step
__global__ __launch_bounds__(256) void window_mean_kernel(float* __restrict__ pred, float* __restrict__ counts, float* __restrict__ eps,
                                                          long long per) {
  const int f = blockIdx.x;
  const float inv = 1.0f / counts[f];
  for (long long i = threadIdx.x; i < per; i += blockDim.x) {
    eps[(long long)f * per + i] = pred[(long long)f * per + i] * inv;
    pred[(long long)f * per + i] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[f] = 0.f;
}
}  // namespace

extern "C" int md_gather_frames(const void* src, void* dst, const int32_t* idx_table, const int32_t* step_counter, int32_t steps,
                                int32_t windows, int32_t window, int32_t n_idx, int64_t row_bytes, void* stream) {
  if (!src || !dst || !idx_table || steps <= 0 || windows <= 0 || window < 0 || window >= windows || n_idx <= 0 || row_bytes <= 0 ||
      (row_bytes & 15))
    return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, 2.0 * n_idx * (double)row_bytes);
  const long long total = (long long)n_idx * (row_bytes >> 4);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(gather_frames_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst,
                     idx_table, step_counter, steps, windows, window, n_idx, (long long)(row_bytes >> 4));
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_cfg_scatter_add(const float* eps_c, const float* eps_u, int32_t ld_eps, const float* coef, const int32_t* idx_table,
                                  const int32_t* step_counter, int32_t steps, int32_t windows, int32_t window, int32_t n_idx,
                                  float* pred, float* counts, int32_t hw, int32_t c, void* stream) {
  if (!eps_c || !eps_u || !coef || !idx_table || !pred || !counts || steps <= 0 || windows <= 0 || window < 0 || window >= windows ||
      n_idx <= 0 || hw <= 0 || c <= 0 || ld_eps < c)
    return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, 16.0 * n_idx * (double)hw * c);
  const long long total = (long long)n_idx * hw * c;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(cfg_scatter_add_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, eps_c, eps_u, ld_eps, coef,
                     idx_table, step_counter, steps, windows, window, n_idx, pred, counts, hw, c);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_window_mean(float* pred, float* counts, float* eps, int32_t frames, int64_t per_frame, void* stream) {
  if (!pred || !counts || !eps || frames <= 0 || per_frame <= 0) return MD_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, 12.0 * frames * (double)per_frame);
  hipLaunchKernelGGL(window_mean_kernel, dim3(frames), dim3(256), 0, s, pred, counts, eps, (long long)per_frame);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}

extern "C" int md_ddim_update(const float* eps_c, const float* eps_u, int32_t ld_eps, const float* x,
                              const float* noise, const float* coef, float* x_prev, float* pred_x0, float* eps_out,
                              int32_t batch, int32_t c, int32_t hw, void* stream) {
  if (!eps_c || !x || !coef || !x_prev || batch <= 0 || c <= 0 || hw <= 0 || ld_eps < c) return MD_ERR_BAD_ARG;
  const long long total = (long long)batch * c * hw;
  hipStream_t s = (hipStream_t)stream;
  md::ProfScope prof(MD_FAM_ELEMENTWISE, s, 0.0, (double)total * 4.0 * 5.0);
  hipLaunchKernelGGL(ddim_update, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, eps_c, eps_u, ld_eps, x, noise,
                     coef, x_prev, pred_x0, eps_out, c, hw, total);
  MD_HIP_CHECK(hipGetLastError());
  return MD_OK;
}
