// Row-scaled two-term fp16 split ("h3") GEMM: C = A W^T with fp32-level accuracy from THREE fp16
// matrix-core products per k-step (half the passes of gemm_x6.hip, which is limited by the chip's power budget).
//
// Every operand row is scaled by a power of two so that its largest magnitude lies in [2^14, 2^15):
//     x * 2^e = h + l,   h = fp16_rne(x * 2^e),   l = fp16_rne(x * 2^e - h)      (22 mantissa bits)
// l needs no extra scaling: the row scale keeps it a normal fp16 for every element within 2^16 of the row maximum,
// and below that its absolute error is 2^-39 of the row maximum.  a*b = (hh + hl + lh + ll) 2^-(ea+eb); ll is below
// 2^-24 and dropped, the other three accumulate in ONE fp32 accumulator (they have their natural magnitudes) and the
// epilogue multiplies by inv_a[row] * inv_w[col] (powers of two: exact).  CPU emulation of the whole ViT:
// tools/split_fp16_study.py.  In the ViT forward (ANYLOC_GEMM=h3) LayerNorm quantises its own output (the row is in
// registers); the attention output and the FFN hidden activation are written as fp32 and quantised by split_h2_kernel,
// because their rows are produced by different workgroups and the exact row maximum must be known first.
//
// Operand image ("h2"): [k/16][plane 0..1][row][16] fp16, 32 bytes per (k-block, plane, row), 16-byte halves swapped
// when (row >> 3) & 1 -- the x3 image of gemm_x6.hip with two planes; same DMA staging, same fragment reads.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "gemm_h3_kernel.hpp"

namespace anyloc {

namespace {

// (h2_store_chunk / h2_store_rows / ln_rows_tiled: gemm_h3_kernel.hpp -- shared with the LayerNorm lead role of the GEMM kernel)

// wide rows (K > 2048): two passes over the row instead of holding it in registers -- pass 1 finds the maximum,
// pass 2 re-reads the 16 rows (L2-resident: 16 x 4 K floats) and quantises; 8x the occupancy of the register version
__global__ __launch_bounds__(256) void split_h2_stream_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                              unsigned char* __restrict__ out, float* __restrict__ inv,
                                                              int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const f32x4* xr[4];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    xr[q] = reinterpret_cast<const f32x4*>(x + row * ldx);
    float amax = 0.f;
    for (int idx = lane; idx < n4; idx += 64) {
      const f32x4 t = xr[q][idx];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))));
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * 4 + q < rows) inv[row] = iv;
  }
  const int nchunks = (dim + 255) / 256;
  for (int i = 0; i < nchunks; ++i) {
    const int idx = lane + 64 * i;
    if (i > 0) __syncthreads();
    if (idx < n4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o = xr[q][idx];
        o[0] *= scale[q]; o[1] *= scale[q]; o[2] *= scale[q]; o[3] *= scale[q];
        *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
      }
    }
    __syncthreads();
    h2_store_chunk(tile, i, dim, row0, rows, out, R);
  }
}

// ---- very wide rows (retrieval: K = 49 152 VLAD columns; any K % 16 == 0 above 4096) -- two kernels, both with one
// workgroup per small unit of work so that thousands of them stream HBM: (1) one workgroup per row finds the row's largest
// magnitude (-> the power-of-two scale) and its sum of squares (the caller's F.normalize / L2 terms come out of the
// same read); (2) one workgroup per (16 rows, 2048 columns) quantises with the scales of (1).
__global__ __launch_bounds__(256) void row_amax_sq_kernel(const float* __restrict__ x, int64_t ldx, int64_t dim,
                                                          float* __restrict__ inv, float* __restrict__ ss) {
  __shared__ float red_m[4], red_s[4];
  const f32x4* r = reinterpret_cast<const f32x4*>(x + (int64_t)blockIdx.x * ldx);
  const int64_t n4 = dim >> 2;
  float amax = 0.f, sq = 0.f;
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {        // four 16-byte loads in flight per thread
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + 256 * u;
      if (j < n4) t[u] = r[j];
      else t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(t[u][0]), fabsf(t[u][1])), fmaxf(fabsf(t[u][2]), fabsf(t[u][3]))));
      sq += (t[u][0] * t[u][0] + t[u][1] * t[u][1]) + (t[u][2] * t[u][2] + t[u][3] * t[u][3]);
    }
  }
  amax = wave_max(amax);
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) {
    red_m[threadIdx.x >> 6] = amax;
    red_s[threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float iv;
    h2_row_scale(fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3])), iv);
    inv[blockIdx.x] = iv;
    if (ss) ss[blockIdx.x] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
  }
}

constexpr int WIDE_CHUNKS = 8;      // 256-column chunks per workgroup of the quantising kernel

__global__ __launch_bounds__(256) void split_h2_wide_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                            unsigned char* __restrict__ out, const float* __restrict__ inv,
                                                            int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int c0 = blockIdx.y * WIDE_CHUNKS;
  const int nchunks = min(WIDE_CHUNKS, (dim + 255) / 256 - c0);
  const int n4 = dim >> 2;
  const f32x4* xr[4];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    xr[q] = reinterpret_cast<const f32x4*>(x + row * ldx);
    scale[q] = h2_scale_of_inv(inv[row]);
  }
  f32x4 cur[4], nxt[4];
  auto load = [&](int i, f32x4 (&dst)[4]) {
    const int idx = lane + 64 * (c0 + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = idx < n4 ? xr[q][idx] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  load(0, cur);
  for (int i = 0; i < nchunks; ++i) {
    if (i + 1 < nchunks) load(i + 1, nxt);                 // the next chunk's loads fly over this chunk's LDS round trip
    if (i > 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = cur[q];
      o[0] *= scale[q]; o[1] *= scale[q]; o[2] *= scale[q]; o[3] *= scale[q];
      *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
    }
    __syncthreads();
    h2_store_chunk(tile, c0 + i, dim, row0, rows, out, R);
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
}

// The screened retrieval's quantiser (scores_screen.hip): only the LEADING plane of the image is written (the score panels
// read nothing else) and the squared norm of what the leading plane leaves out, sum (x 2^e - hi)^2, is added to resid_sq[row]
// (zeroed by the caller; one atomicAdd per row and workgroup: the sum's rounding varies run to run, which only moves the
// candidate margin by an ulp -- the re-scored result does not depend on it).  Same staging as split_h2_wide_kernel.
__global__ __launch_bounds__(256) void split_h1_wide_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                            unsigned char* __restrict__ out, const float* __restrict__ inv,
                                                            int64_t R, float* __restrict__ resid_sq) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  __shared__ float rpart[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int c0 = blockIdx.y * WIDE_CHUNKS;
  const int nchunks = min(WIDE_CHUNKS, (dim + 255) / 256 - c0);
  const int n4 = dim >> 2;
  const f32x4* xr[4];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    xr[q] = reinterpret_cast<const f32x4*>(x + row * ldx);
    scale[q] = h2_scale_of_inv(inv[row]);
  }
  f32x4 cur[4], nxt[4];
  auto load = [&](int i, f32x4 (&dst)[4]) {
    const int idx = lane + 64 * (c0 + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = idx < n4 ? xr[q][idx] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  float rs = 0.f;                                          // this thread's items all belong to row (tid >> 1) & 15
  load(0, cur);
  for (int i = 0; i < nchunks; ++i) {
    if (i + 1 < nchunks) load(i + 1, nxt);
    if (i > 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = cur[q];
      o[0] *= scale[q]; o[1] *= scale[q]; o[2] *= scale[q]; o[3] *= scale[q];
      *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
    }
    __syncthreads();
    // (k-block, row, half) items of the chunk: 16 rows x 32 = 512 items, two per thread, both of row (tid >> 1) & 15
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = tid + 256 * u;
      const int kbl = item >> 5, r = (item >> 1) & 15, half = item & 1;
      const int k0 = 256 * (c0 + i) + 16 * kbl + 8 * half;
      const int64_t row = row0 + r;
      if (k0 < dim && row < rows) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half + 4]);
        hu32x4 ph;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x2 pr;
          pr[0] = j < 2 ? lo[2 * j] : hi[2 * j - 4];
          pr[1] = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
          const f16x2 h = __builtin_convertvector(pr, f16x2);
          const float d0 = pr[0] - (float)h[0], d1 = pr[1] - (float)h[1];
          rs += d0 * d0 + d1 * d1;
          ph[j] = __builtin_bit_cast(unsigned, h);
        }
        *reinterpret_cast<hu32x4*>(out + (((int64_t)(k0 >> 4) * 2) * R + row) * 32 + ((half ^ (int)((row >> 3) & 1)) << 4)) = ph;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
  // threads of one row: lane bits 0 and 5 within a wave, all four waves
  rs += __shfl_xor(rs, 1, 64);
  rs += __shfl_xor(rs, 32, 64);
  if ((lane & 33) == 0) rpart[wave][(lane >> 1) & 15] = rs;
  __syncthreads();
  if (tid < 16 && row0 + tid < rows) atomicAdd(resid_sq + row0 + tid, (rpart[0][tid] + rpart[1][tid]) + (rpart[2][tid] + rpart[3][tid]));
}

// fp32 row-major [rows, K] -> h2 image + inv[row] = 2^-e
template <int NV>
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                       unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  f32x4 v[4][NV];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ldx);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        v[q][i] = xr[idx];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[q][i][0]), fabsf(v[q][i][1])), fmaxf(fabsf(v[q][i][2]), fabsf(v[q][i][3]))));
      }
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * 4 + q < rows) inv[row] = iv;
  }
  h2_store_rows<NV>(v, scale, tile, dim, row0, rows, out, R);
}

// LayerNorm (torch semantics, biased variance) whose output is quantised straight into the h2 image
// bound_inv != nullptr (LN2 of a block whose FFN activation is quantised in the fc1 epilogue): also writes
// bound_inv[row] = 2^-e, where 2^e scales an UPPER BOUND of |act(fc1(y_row))| into [2^14, 2^15).  Cauchy-Schwarz:
// |fc1_j(y)| <= ||y||_2 max_j ||W_j||_2 + max_j |b_j|, |gelu(t)| <= |t|, |silu(g) v| <= |g| |v|; bound4 = {gate (or fc1)
// row-norm maximum, gate bias maximum, value row-norm maximum, value bias maximum} (value pair 0 / 0: GELU MLP).
// RPW rows per wave (4 waves per block): 4 by default; 1 when there are few rows (one or two images: 530 rows are 34
// blocks of 16 rows on 256 CUs -- 25 us of latency per launch; 133 blocks of 4 rows spread them).  Per-row arithmetic does
// not depend on RPW, so the results are bitwise the same.
template <int NV, int RPW, int NW = 4>
__global__ __launch_bounds__(64 * NW) void layernorm_h2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, int dim, int64_t rows, float eps,
                                                           unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R,
                                                           const f32x4 bound4, float* __restrict__ bound_inv) {
  __shared__ __attribute__((aligned(16))) float tile[NW * RPW][256 + 4];
  ln_rows_tiled<NV, RPW, NW, false>(x, w, b, dim, rows, eps, out, inv, R, (int64_t)blockIdx.x * (NW * RPW), bound4, bound_inv, tile,
                                    null_rsrc());          // (gemm_h3_kernel.hpp)
}


// The same LayerNorm for FEW rows (one or two images): one wave per row, one row per workgroup, and every lane stores its own
// four-column groups straight into the image (8 bytes per plane and group) -- no LDS tile, no barriers.  A 530-row launch
// is 530 single-wave workgroups on 256 CUs, one wave's load -> two reductions -> store chain each, instead of 133 blocks with
// 12 barriers.  Measured: the same 12 us per HIP-event-bracketed launch as the tiled kernel (profiles/r04_b1_plan_sweep_depth.log:
// ~6 us of that is the bracket, the rest launch ramp and one dependent load chain) -- kept because it needs no LDS; at large
// M its scattered 8-byte stores are 2.3 x slower than the tiled kernel (option ln_direct_rows).  Per-row arithmetic as above:
// same bits.
template <int NV>
__global__ __launch_bounds__(64) void layernorm_h2_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, int dim, int64_t rows, float eps,
                                                                 unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R,
                                                                 const f32x4 bound4, float* __restrict__ bound_inv) {
  ln_row_direct<NV, false>(x, w, b, dim, eps, out, inv, R, (int64_t)blockIdx.x, bound4, bound_inv);   // (gemm_h3_kernel.hpp)
}

// FFN-bound telemetry (anyloc_vit_set_telemetry).  The fc1 / w12 epilogue leaves, per block and token row, the largest scaled
// magnitude the row holds in the fc2 operand image (atomicMax of the bit patterns, gemm_h3_kernel.hpp); a bound that is 2^L
// above the row's real maximum shows as 2^(15 - L).  One workgroup per (block, row group): out = max over the group's rows of
// 2^15 / max (a row that left nothing nonzero lies more than 2^39 below its bound, or is exactly zero -- a token row of an
// FFN activation is never that: reported as 2^40); a block none of whose rows left a maximum did not run fused: 0.
__global__ __launch_bounds__(256) void ffn_looseness_kernel(const unsigned* __restrict__ rowmax, int64_t M, int64_t rpg, int groups,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ unsigned any[4];
  const int l = blockIdx.x / groups, g = blockIdx.x % groups;
  const unsigned* rm = rowmax + (int64_t)l * M;
  const int64_t r0 = (int64_t)g * rpg, r1 = min(M, r0 + rpg);
  float loose = 0.f;
  unsigned seen = 0u;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
    const unsigned b = rm[r];                              // (positive floats order as their bit patterns)
    seen |= b;
    loose = fmaxf(loose, b ? 32768.0f / __uint_as_float(b) : 1.099511627776e12f);
  }
  loose = wave_max(loose);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) seen |= (unsigned)__shfl_xor((int)seen, o, 64);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = loose;
    any[threadIdx.x >> 6] = seen;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.x] = (any[0] | any[1] | any[2] | any[3]) ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 0.0f;
}

}  // namespace

int ffn_looseness(const unsigned* rowmax, int nblocks, int64_t M, int64_t rows_per_group, float* out, hipStream_t stream) {
  ANYLOC_CHECK_ARG(rowmax && out && nblocks > 0 && M > 0 && rows_per_group > 0, "ffn_looseness: bad arguments");
  const int groups = (int)((M + rows_per_group - 1) / rows_per_group);
  ProfScope prof("ffn_telemetry", stream, 0.0, 4.0 * nblocks * M);
  hipLaunchKernelGGL(ffn_looseness_kernel, dim3((unsigned)(nblocks * groups)), dim3(256), 0, stream, rowmax, M, rows_per_group,
                     groups, out);
  return launch_status("ffn_looseness_kernel");
}

size_t h2_bytes(int64_t rows, int64_t K) { return (size_t)((K + 15) / 16) * 2 * (size_t)rows * 32; }

int split_h2_wide(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, float* row_sumsq,
                  hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && rows > 0 && K > 0 && ldx >= K, "split_h2_wide: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && ldx % 4 == 0 && K < (1ll << 31) && rows < (1ll << 31) && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                   "split_h2_wide: K must be a multiple of 16, rows 16-byte aligned (K=%lld)", (long long)K);
  ProfScope prof("split_h2_wide", stream, 0.0, 12.0 * rows * K);
  hipLaunchKernelGGL(row_amax_sq_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, K, inv_scale, row_sumsq);
  ANYLOC_TRY(launch_status("row_amax_sq_kernel"));
  const int chunks = (int)((K + 255) / 256);
  const dim3 grid((unsigned)((rows + 15) / 16), (unsigned)((chunks + WIDE_CHUNKS - 1) / WIDE_CHUNKS));
  ANYLOC_CHECK_ARG(grid.y < 65536, "split_h2_wide: K too large");
  hipLaunchKernelGGL(split_h2_wide_kernel, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, static_cast<unsigned char*>(h2),
                     inv_scale, rows);
  return launch_status("split_h2_wide_kernel");
}

// split_h2_wide for the screened retrieval: leading plane only + resid_sq[row] += |x 2^e - hi|^2 (scaled units; the caller
// zeroes resid_sq)
int split_h1_wide(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, float* row_sumsq, float* resid_sq,
                  hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && resid_sq && rows > 0 && K > 0 && ldx >= K, "split_h1_wide: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && ldx % 4 == 0 && K < (1ll << 31) && rows < (1ll << 31) && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                   "split_h1_wide: K must be a multiple of 16, rows 16-byte aligned (K=%lld)", (long long)K);
  ProfScope prof("split_h1_wide", stream, 0.0, 10.0 * rows * K);
  hipLaunchKernelGGL(row_amax_sq_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, K, inv_scale, row_sumsq);
  ANYLOC_TRY(launch_status("row_amax_sq_kernel"));
  const int chunks = (int)((K + 255) / 256);
  const dim3 grid((unsigned)((rows + 15) / 16), (unsigned)((chunks + WIDE_CHUNKS - 1) / WIDE_CHUNKS));
  ANYLOC_CHECK_ARG(grid.y < 65536, "split_h1_wide: K too large");
  hipLaunchKernelGGL(split_h1_wide_kernel, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, static_cast<unsigned char*>(h2),
                     inv_scale, rows, resid_sq);
  return launch_status("split_h1_wide_kernel");
}

int row_scales_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, float* inv_scale, float* row_sumsq, hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && inv_scale && rows > 0 && K > 0 && ldx >= K && K % 4 == 0 && ldx % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && rows < (1ll << 31),
                   "row_scales_h2: bad arguments");
  hipLaunchKernelGGL(row_amax_sq_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, K, inv_scale, row_sumsq);
  return launch_status("row_amax_sq_kernel");
}

int split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && rows > 0 && K > 0 && ldx >= K, "split_h2: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && ldx % 4 == 0, "split_h2: K must be a multiple of 16 (got %lld)", (long long)K);
  if (K > 4096) return split_h2_wide(x, ldx, rows, K, h2, inv_scale, nullptr, stream);
  ProfScope prof("split_h2", stream, 0.0, 8.0 * rows * K);
  const dim3 grid((unsigned)((rows + 15) / 16));
  unsigned char* out = static_cast<unsigned char*>(h2);
  const int nv = (int)((K / 4 + 63) / 64);
#define ANYLOC_SPLIT_H2(NVV) \
  hipLaunchKernelGGL(split_h2_kernel<NVV>, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, out, inv_scale, rows)
  if (nv <= 1) ANYLOC_SPLIT_H2(1);
  else if (nv <= 2) ANYLOC_SPLIT_H2(2);
  else if (nv <= 4) ANYLOC_SPLIT_H2(4);
  else if (nv <= 6) ANYLOC_SPLIT_H2(6);
  else if (nv <= 8) ANYLOC_SPLIT_H2(8);
  else hipLaunchKernelGGL(split_h2_stream_kernel, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, out, inv_scale, rows);
#undef ANYLOC_SPLIT_H2
  return launch_status("split_h2_kernel");
}

int layernorm_h2(const float* x, const float* w, const float* b, int64_t rows, int dim, float eps, void* h2,
                 float* inv_scale, hipStream_t stream, const float* bound, float* bound_inv) {
  ANYLOC_CHECK_ARG(dim % 16 == 0 && dim <= 2048, "layernorm_h2: dim %d (needs a multiple of 16, at most 2048)", dim);
  ProfScope prof("layernorm_h2", stream, 8.0 * rows * dim, 8.0 * rows * dim);
  unsigned char* out = static_cast<unsigned char*>(h2);
  const int nv = (dim / 4 + 63) / 64;
  f32x4 b4;                                                  // bound: HOST array of 4 floats (or null)
  for (int i = 0; i < 4; ++i) b4[i] = bound ? bound[i] : 0.0f;
  // few rows (option ln_small_rows, default 4096 = seven 322 x 322 images): one row per wave, four per block; otherwise two
  // rows per wave (48 instead of 96 data registers: more waves in flight; B=61: 6.48 -> 5.65 ms per step, four rows per wave
  // was the round-2 kernel, one row per wave at this size 11.4 ms -- profiles/r03_ab_attn_kbatch_ln_rpw.log).  Option
  // ln_rows_per_wave (0 = that rule) forces 1, 2 or 4 at every size (A/B; same per-row arithmetic, same bits)
  const int64_t forced = option(OPT_LN_ROWS_PER_WAVE);
  if (forced == 0 && rows < option(OPT_LN_DIRECT_ROWS)) {
    // a few hundred rows: one single-wave workgroup per row, image written straight from registers
    const dim3 grid((unsigned)rows);
#define ANYLOC_LN_H2_D(NVV)                                                                                              \
  hipLaunchKernelGGL((layernorm_h2_direct_kernel<NVV>), grid, dim3(64), 0, stream, x, w, b, dim, rows, eps, out, inv_scale, \
                     rows, b4, bound ? bound_inv : nullptr)
    if (nv <= 1) ANYLOC_LN_H2_D(1);
    else if (nv <= 2) ANYLOC_LN_H2_D(2);
    else if (nv <= 3) ANYLOC_LN_H2_D(3);
    else if (nv <= 4) ANYLOC_LN_H2_D(4);
    else if (nv <= 6) ANYLOC_LN_H2_D(6);
    else ANYLOC_LN_H2_D(8);
#undef ANYLOC_LN_H2_D
    return launch_status("layernorm_h2_direct_kernel");
  }
  const int rpw = (forced == 1 || forced == 2 || forced == 4) ? (int)forced : (rows < option(OPT_LN_SMALL_ROWS) ? 1 : 2);
  // two rows per wave: EIGHT waves per block (option ln_waves = 8, default) -- 16 rows per block, i.e. 512-byte runs per store
  // instruction instead of the 256-byte runs of four waves, at the register count of two rows per wave (B = 61: 5.65 -> 5.55 ms
  // per step; four rows per wave on four waves: 6.4, one row per wave: 11.1, sixteen waves = 32 rows per block: 6.2 -- a bare
  // copy in this pattern takes 80 us per call against LayerNorm's 87-89: tools/micro/ln_store_pattern.hip)
  const int nw = (rpw == 2 && option(OPT_LN_WAVES) == 8) ? 8 : 4;
  const dim3 grid((unsigned)((rows + nw * rpw - 1) / (nw * rpw)));
#define ANYLOC_LN_H2_R(NVV, RPWV, NWV)                                                                                   \
  hipLaunchKernelGGL((layernorm_h2_kernel<NVV, RPWV, NWV>), grid, dim3(64 * NWV), 0, stream, x, w, b, dim, rows, eps, out,  \
                     inv_scale, rows, b4, bound ? bound_inv : nullptr)
#define ANYLOC_LN_H2(NVV)              \
  do {                                 \
    if (rpw == 1) ANYLOC_LN_H2_R(NVV, 1, 4);      \
    else if (rpw == 2 && nw == 8) ANYLOC_LN_H2_R(NVV, 2, 8); \
    else if (rpw == 2) ANYLOC_LN_H2_R(NVV, 2, 4); \
    else ANYLOC_LN_H2_R(NVV, 4, 4);               \
  } while (0)
  if (nv <= 1) ANYLOC_LN_H2(1);
  else if (nv <= 2) ANYLOC_LN_H2(2);
  else if (nv <= 3) ANYLOC_LN_H2(3);
  else if (nv <= 4) ANYLOC_LN_H2(4);
  else if (nv <= 6) ANYLOC_LN_H2(6);
  else ANYLOC_LN_H2(8);
#undef ANYLOC_LN_H2_R
#undef ANYLOC_LN_H2
  return launch_status("layernorm_h2_kernel");
}

// ---- LayerNorm lead role inside a batched launch (gemm_h3_kernel<..., LNL = 2>; tile_order.hpp: LeadPlan) ----
// The plan of a shape is simulated once on the host: every GEMM tile exactly once, every row normalised exactly once, and every
// lead workgroup of a tile row ahead (in workgroup-id order) of every GEMM tile of that row.  grid = 8 x the longest XCD sequence.
struct LeadPlanInfo { bool ok; unsigned grid; };
LeadPlanInfo lead_plan_info(const LeadPlan& lp) {
  static std::mutex mu;
  static std::map<std::array<long long, 5>, LeadPlanInfo> cache;
  const std::array<long long, 5> key{lp.tiles_m, lp.tiles_n, lp.group_m, lp.bm, lp.M};
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  LeadPlanInfo info{false, 0};
  const int nb = lp.tiles_m * lp.tiles_n;
  bool ok = lp.tiles_m >= 8 * lp.group_m && lp.bm % LEAD_ROWS == 0 && lp.M > 0 && (lp.M + lp.bm - 1) / lp.bm == lp.tiles_m;
  if (ok) {
    std::vector<int> tile_seen((size_t)nb, 0);
    std::vector<long long> first_tile_id((size_t)lp.tiles_m, -1), last_lead_id((size_t)lp.tiles_m, -1), rows_done((size_t)lp.tiles_m, 0);
    int longest = 0;
    for (int x = 0; x < 8 && ok; ++x) {
      int len = 0;
      lead_decode(lp, x, 0, &len);
      longest = std::max(longest, len);
      for (int loc = 0; loc < len && ok; ++loc) {
        const LeadRole role = lead_decode(lp, x, loc);
        const long long id = (long long)loc * 8 + x;
        if (role.kind == 1) {
          if (role.tm < 0 || role.tm >= lp.tiles_m || role.tn < 0 || role.tn >= lp.tiles_n || tile_seen[(size_t)role.tm * lp.tiles_n + role.tn]++) ok = false;
          else if (first_tile_id[role.tm] < 0 || id < first_tile_id[role.tm]) first_tile_id[role.tm] = id;
        } else if (role.kind == 2) {
          if (role.row0 < 0 || role.row0 >= lp.M || role.row0 % LEAD_ROWS) { ok = false; break; }
          const int tm = (int)(role.row0 / lp.bm);
          rows_done[tm] += std::min<long long>(LEAD_ROWS, lp.M - role.row0);
          last_lead_id[tm] = std::max(last_lead_id[tm], id);
        } else {
          ok = false;
        }
      }
      if (lead_decode(lp, x, len).kind != 0) ok = false;
    }
    for (int t = 0; t < nb && ok; ++t) ok = tile_seen[t] == 1;
    for (int tm = 0; tm < lp.tiles_m && ok; ++tm)
      ok = rows_done[tm] == std::min<long long>(lp.bm, lp.M - (long long)tm * lp.bm) && last_lead_id[tm] >= 0 && last_lead_id[tm] < first_tile_id[tm];
    info.ok = ok;
    info.grid = (unsigned)(8 * longest);
  }
  cache[key] = info;
  return info;
}

template <int EPI>
constexpr bool lead_compiled() { return EPI == EPI_QKV_PLANES || EPI == EPI_SWIGLU_T_H2 || EPI == EPI_SWIGLU_H2 || EPI == EPI_GELU_H2; }

template <int EPI>
int dispatch_h3(const H3Problem& p, hipStream_t stream) {
  // option h3_cfg (micro-benchmarks): 0 = 128x256 tile, 3-deep ring (default; 128x128 when there are few tiles); 1 = 2-deep;
  // 2-5 = 256x256 tiles (see the switch)
  const int cfg = (int)option(OPT_H3_CFG);
#define ANYLOC_LAUNCH_H3(MI, NI, WM, WN, ST, OCC) ANYLOC_LAUNCH_H3K(MI, NI, WM, WN, ST, OCC, 1)
#define ANYLOC_LAUNCH_H3K(MI, NI, WM, WN, ST, OCC, KB)                                                                \
  do {                                                                                                                \
    using Cfg = H3Cfg<MI, NI, WM, WN, ST, KB>;                                                                        \
    const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);         \
    static DynLds dyn_lds_once; \
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_h3_kernel<MI, NI, WM, WN, ST, OCC, EPI, KB>), (int)(Cfg::LDS)));                                                                                                                 \
    hipLaunchKernelGGL((gemm_h3_kernel<MI, NI, WM, WN, ST, OCC, EPI, KB>), dim3((unsigned)(tiles_m * tiles_n)),        \
                       dim3(64 * WM * WN), Cfg::LDS, stream, p, tiles_m, tiles_n);                                    \
  } while (0)
  const bool small = ((p.M + 127) / 128) * ((p.N + 255) / 256) < 512;
  if (small && cfg == 0) {
    // one or a few images (the reference's scripts call the extractor per image): tile shape, ring depth and split-K
    // factor come from the small-M plan table (gemm_h3s.hip); the unfused epilogues (A/B data flows) keep two fixed shapes
    if constexpr (EPI == EPI_STORE || EPI == EPI_LS_RESID || EPI == EPI_QKV_PLANES || EPI == EPI_GELU_H2 ||
                  EPI == EPI_SWIGLU_H2 || EPI == EPI_SWIGLU_T_H2) {
      return gemm_h3_small(p, EPI, stream);
    } else {
      if (((p.M + 127) / 128) * ((p.N + 127) / 128) < option(OPT_H3_TINY_MAX)) ANYLOC_LAUNCH_H3(1, 2, 2, 1, 3, 2);   // 64x64, 2 waves
      else ANYLOC_LAUNCH_H3(2, 2, 2, 2, 3, 2);                                                                         // 128x128
      return launch_status("gemm_h3_kernel");
    }
  }
  if (p.ln_x) {
    // LayerNorm in front of this GEMM as the lead role of this launch (linear_h3 asked h3_ln_lead_feasible first)
    if constexpr (lead_compiled<EPI>()) {
      using Cfg = H3Cfg<2, 4, 2, 2, 3, 1>;
      const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
      const LeadPlanInfo info = lead_plan_info(LeadPlan{tiles_m, tiles_n, p.group_m, Cfg::BM, (long long)p.M});
      ANYLOC_CHECK_ARG(cfg == 0 && info.ok && p.ln_tickets && p.ln_w && p.ln_b && (!p.ln_has_bound || p.c_inv) && p.ln_dim == 16 * p.K16 &&
                           p.ln_dim <= 1536 && p.ksplit <= 1,
                       "gemm_h3: LayerNorm lead role asked for a launch it does not fit (h3_ln_lead_feasible)");
      static DynLds dyn_lds_lead;
      ANYLOC_TRY(ensure_dyn_lds(dyn_lds_lead, reinterpret_cast<const void*>(&gemm_h3_kernel<2, 4, 2, 2, 3, 2, EPI, 1, 2>), (int)(Cfg::LDS)));
      hipLaunchKernelGGL((gemm_h3_kernel<2, 4, 2, 2, 3, 2, EPI, 1, 2>), dim3(info.grid), dim3(256), Cfg::LDS, stream, p, tiles_m, tiles_n);
      return launch_status("gemm_h3_kernel (LayerNorm lead role)");
    } else {
      ANYLOC_CHECK_ARG(false, "gemm_h3: LayerNorm lead role asked for an epilogue it is not compiled for (h3_ln_lead_feasible)");
    }
  }
  switch (cfg) {
    case 1: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 2, 2); break;
    case 2: ANYLOC_LAUNCH_H3(2, 4, 4, 2, 3, 2); break;     // 256x256, 8 waves (2 per SIMD, one workgroup per CU), 96 KiB ring
    case 3: ANYLOC_LAUNCH_H3(2, 4, 4, 2, 4, 2); break;     // the same, 4-deep ring (128 KiB)
    case 4: ANYLOC_LAUNCH_H3(4, 4, 2, 2, 4, 1); break;     // 256x256, 4 waves of 128x128 (one per SIMD), 4-deep ring
    case 5: ANYLOC_LAUNCH_H3(4, 4, 2, 2, 3, 1); break;     // the same, 3-deep ring
    default: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 3, 2); break;
  }
#undef ANYLOC_LAUNCH_H3
#undef ANYLOC_LAUNCH_H3K
  return launch_status("gemm_h3_kernel");
}

// would gemm_h3 run this GEMM with the LayerNorm lead role (H3Problem::ln_x)?  The caller then skips its LayerNorm launch.
// One image per call: the small-M plans' rule (gemm_h3s.hip, option h3s_ln_lead); batched: option h3_ln_lead, the default tile
// configuration, an epilogue the role is compiled for, rows of at most 1536 columns, and a plan that passes lead_plan_info.
bool h3_ln_lead_feasible(const H3Problem& p, int epilogue) {
  if (option(OPT_H3_CFG) != 0) return false;
  if (((p.M + 127) / 128) * ((p.N + 255) / 256) < 512) return h3s_ln_lead_feasible(p, epilogue);
  if (option(OPT_H3_LN_LEAD) == 0) return false;
  if (epilogue != EPI_QKV_PLANES && epilogue != EPI_SWIGLU_T_H2 && epilogue != EPI_SWIGLU_H2 && epilogue != EPI_GELU_H2) return false;
  if (16 * (int64_t)p.K16 > 1536 || p.ksplit > 1) return false;
  using Cfg = H3Cfg<2, 4, 2, 2, 3, 1>;
  const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
  const int gm = (int)std::max<int64_t>(1, option(OPT_H3_GROUP_M));
  return lead_plan_info(LeadPlan{tiles_m, tiles_n, gm, Cfg::BM, (long long)p.M}).ok;
}
// host-side check of a lead plan (tests, no GPU needed): 1 = the plan passes, 0 = it does not (the caller keeps two launches)
int h3_lead_plan_check(int tiles_m, int tiles_n, int group_m, int64_t M, unsigned* grid) {
  const LeadPlanInfo info = lead_plan_info(LeadPlan{tiles_m, tiles_n, group_m, 128, (long long)M});
  if (grid) *grid = info.grid;
  return info.ok ? 1 : 0;
}

int gemm_h3(const H3Problem& p_in, int epilogue, hipStream_t stream) {
  H3Problem p = p_in;
  ANYLOC_CHECK_ARG(p.A2 && p.a_inv && p.W2 && p.w_inv && (p.C || epilogue >= EPI_QKV_PLANES), "gemm_h3: null operand");
  ANYLOC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K16 > 0 && p.RA >= p.M && p.RW >= p.N, "gemm_h3: bad shape");
  ANYLOC_CHECK_ARG((size_t)p.K16 * 2 * (size_t)p.RA * 32 < (1ull << 31) && (size_t)p.K16 * 2 * (size_t)p.RW * 32 < (1ull << 31),
                   "gemm_h3: operand image exceeds the 2 GiB buffer-addressing range");
  const int64_t K = 16ll * p.K16;
  ProfScope prof(p.tag ? p.tag : "gemm_h3", stream, 2.0 * p.M * p.N * K, 4.0 * (p.M + p.N) * K + 4.0 * p.M * p.N);
  // tile-rows per scheduling group (co-resident workgroups of an XCD share A / W panels through its L2): option
  // h3_group_m, default 8
  p.group_m = (int)std::max<int64_t>(1, option(OPT_H3_GROUP_M));
  // plain-store GEMMs of >= 256 tiles of 256 x 256 run on the 16 x 16 x 32 MFMA kernel (gemm_h3m.hip; option h3_mfma16: -1 =
  // when the contraction is >= 4096 long -- the retrieval panels, +3.6 % -- 0 never, 1 whatever the length)
  const int64_t m16 = option(OPT_H3_MFMA16);
  if (epilogue == EPI_STORE && m16 != 0 && (m16 > 0 || p.K16 >= 256) && ((p.M + 255) / 256) * ((p.N + 255) / 256) >= 256) {
    const int rc = gemm_h3m(p, epilogue, stream);
    if (rc != ANYLOC_ERR_UNSUPPORTED) return rc;
  }
  switch (epilogue) {
    case EPI_STORE: return dispatch_h3<EPI_STORE>(p, stream);
    case EPI_GELU: return dispatch_h3<EPI_GELU>(p, stream);
    case EPI_PATCH:
      ANYLOC_CHECK_ARG(p.pos && p.patches > 0 && p.M % p.patches == 0, "gemm_h3: PATCH needs pos and M = batch * patches");
      return dispatch_h3<EPI_PATCH>(p, stream);
    case EPI_LS_RESID: {
      ANYLOC_CHECK_ARG(p.gamma && p.resid, "gemm_h3: LS_RESID needs gamma and resid");
      H3Problem q = p;
      // option h3_epi_lds = 0: the dword read-modify-write epilogue (also the fallback for unaligned C)
      q.epi_lds = option(OPT_H3_EPI_LDS) != 0 && p.N % 4 == 0 && p.ldc % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
      return dispatch_h3<EPI_LS_RESID>(q, stream);
    }
    case EPI_SWIGLU:
      ANYLOC_CHECK_ARG(p.N % 64 == 0, "gemm_h3: SWIGLU needs N %% 64 == 0");
      return dispatch_h3<EPI_SWIGLU>(p, stream);
    case EPI_QKV_PLANES:
      ANYLOC_CHECK_ARG(p.qkv_planes && p.qkv_inv && p.heads > 0 && p.N == 3ll * p.heads * 64 && (p.heads * 64) % 128 == 0 &&
                           p.groups == (p.M + 31) / 32,
                       "gemm_h3: QKV_PLANES needs N = 3 * heads * 64, D %% 128 == 0 and groups = ceil(M / 32)");
      return dispatch_h3<EPI_QKV_PLANES>(p, stream);
    case EPI_GELU_H2:
      ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M && p.N % 64 == 0, "gemm_h3: GELU_H2 needs an output image, c_inv and N %% 64 == 0");
      return dispatch_h3<EPI_GELU_H2>(p, stream);
    case EPI_SWIGLU_H2:
      ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M && p.N % 128 == 0, "gemm_h3: SWIGLU_H2 needs an output image, c_inv and N %% 128 == 0");
      p.fast_silu = option(OPT_H3_FAST_SILU) != 0;
      return dispatch_h3<EPI_SWIGLU_H2>(p, stream);
    case EPI_SWIGLU_T:
    case EPI_SWIGLU_T_H2:
      ANYLOC_CHECK_ARG(p.N % 128 == 0 && (reinterpret_cast<uintptr_t>(p.w_inv) & 15) == 0 &&
                           (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0),
                       "gemm_h3: SWIGLU_T needs N %% 128 == 0 and 16-byte aligned w_inv / bias");
      if (epilogue == EPI_SWIGLU_T_H2)
        ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M, "gemm_h3: SWIGLU_T_H2 needs an output image and c_inv");
      else
        ANYLOC_CHECK_ARG(p.C && p.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0, "gemm_h3: SWIGLU_T needs an aligned fp32 output");
      p.fast_silu = option(OPT_H3_FAST_SILU) != 0;
      return epilogue == EPI_SWIGLU_T ? dispatch_h3<EPI_SWIGLU_T>(p, stream) : dispatch_h3<EPI_SWIGLU_T_H2>(p, stream);
    default: set_error("gemm_h3: unsupported epilogue %d", epilogue); return ANYLOC_ERR_INVALID_ARG;
  }
}

}  // namespace anyloc

using namespace anyloc;

extern "C" size_t anyloc_h2_bytes(int64_t rows, int64_t K) { return h2_bytes(rows, K); }

extern "C" int anyloc_h3_lead_plan_check(int32_t tiles_m, int32_t tiles_n, int32_t group_m, int64_t M, uint32_t* grid) {
  if (tiles_m <= 0 || tiles_n <= 0 || group_m <= 0 || M <= 0) return 0;
  unsigned g = 0;
  const int ok = anyloc::h3_lead_plan_check(tiles_m, tiles_n, group_m, M, &g);
  if (grid) *grid = g;
  return ok;
}

extern "C" int anyloc_split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, void* stream) {
  return split_h2(x, ldx, rows, K, h2, inv_scale, (hipStream_t)stream);
}

extern "C" int anyloc_gemm_nt_h3(const void* a2, const float* a_inv, const void* w2, const float* w_inv, const float* bias,
                                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 16 == 0 && ldc >= N, "gemm_nt_h3: bad shape");
  H3Problem p{};
  p.A2 = static_cast<const unsigned char*>(a2); p.RA = M; p.a_inv = a_inv;
  p.W2 = static_cast<const unsigned char*>(w2); p.RW = N; p.w_inv = w_inv;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K16 = (int)(K / 16); p.bias = bias;
  return gemm_h3(p, EPI_STORE, (hipStream_t)stream);
}
