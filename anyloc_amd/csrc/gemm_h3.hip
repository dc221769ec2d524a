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
#include <cstdlib>

#include "gemm_h3_kernel.hpp"

namespace anyloc {

namespace {

// ---- quantisers: rows held in registers (NV float4 per lane and row, 4 rows per wave, 16 rows per block) ----
// the scaled values of 16 rows go through a 16 x 256 LDS tile, chunk by chunk, and are stored in IMAGE order
// (thread = (k-block, row, half): whole 512-byte runs per store instruction, 16 bytes per lane and plane)
// store chunk i (256 columns) of 16 rows, already scaled and sitting in the LDS tile, in IMAGE order
// (thread = (k-block, row, half): whole 512-byte runs per store instruction, 16 bytes per lane and plane)
template <int RB = 16, int NT = 256>
__device__ __forceinline__ void h2_store_chunk(float (*tile)[256 + 4], int i, int dim, int64_t row0, int64_t rows,
                                               unsigned char* out, int64_t R) {
  static_assert(RB == 32 || RB == 16 || RB == 8 || RB == 4, "rows per block: a power of two (k-block, row, half) decoding");
  const int tid = threadIdx.x;
  constexpr int ITEMS = RB * 32;                           // (k-block, row, half) triples of one 256-column chunk
#pragma unroll
  for (int u = 0; u < (ITEMS + NT - 1) / NT; ++u) {
    const int item = tid + NT * u;
    const int kbl = item / (2 * RB), r = (item >> 1) & (RB - 1), half = item & 1;
    const int k0 = 256 * i + 16 * kbl + 8 * half;
    const int64_t row = row0 + r;
    if (item < ITEMS && k0 < dim && row < rows) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half]);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half + 4]);
      hu32x4 ph, plo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 pr;
        pr[0] = j < 2 ? lo[2 * j] : hi[2 * j - 4];
        pr[1] = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
        const f16x2 h = __builtin_convertvector(pr, f16x2);
        f32x2 res;
        res[0] = pr[0] - (float)h[0];
        res[1] = pr[1] - (float)h[1];
        const f16x2 l = __builtin_convertvector(res, f16x2);
        ph[j] = __builtin_bit_cast(unsigned, h);
        plo[j] = __builtin_bit_cast(unsigned, l);
      }
      unsigned char* dst = out + (((int64_t)(k0 >> 4) * 2) * R + row) * 32 + ((half ^ (int)((row >> 3) & 1)) << 4);
      *reinterpret_cast<hu32x4*>(dst) = ph;
      *reinterpret_cast<hu32x4*>(dst + R * 32) = plo;
    }
  }
}

// rows held in registers: the scaled values of 16 rows go through the LDS tile chunk by chunk
template <int NV, int RPW = 4, int NW = 4>
__device__ __forceinline__ void h2_store_rows(const f32x4 (&v)[RPW][NV], const float (&scale)[RPW], float (*tile)[256 + 4],
                                              int dim, int64_t row0, int64_t rows, unsigned char* out, int64_t R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (i > 0) __syncthreads();
    if (idx < n4) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[q][i][j] * scale[q];
        *reinterpret_cast<f32x4*>(&tile[wave * RPW + q][4 * lane]) = o;
      }
    }
    __syncthreads();
    h2_store_chunk<NW * RPW, 64 * NW>(tile, i, dim, row0, rows, out, R);
  }
}

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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * (NW * RPW);
  f32x4 v[RPW][NV];
  float scale[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int64_t row = min(row0 + wave * RPW + q, rows - 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * dim);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        v[q][i] = xr[idx];
        s += (v[q][i][0] + v[q][i][1]) + (v[q][i][2] + v[q][i][3]);
      }
    }
    const float mean = wave_sum(s) / (float)dim;
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < n4) {
        const float d0 = v[q][i][0] - mean, d1 = v[q][i][1] - mean, d2 = v[q][i][2] - mean, d3 = v[q][i][3] - mean;
        qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(qs) / (float)dim + eps);
    float amax = 0.f, ysq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        const f32x4 wv = reinterpret_cast<const f32x4*>(w)[idx], bv = reinterpret_cast<const f32x4*>(b)[idx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[q][i][j] = (v[q][i][j] - mean) * rstd * wv[j] + bv[j];
          amax = fmaxf(amax, fabsf(v[q][i][j]));
          ysq += v[q][i][j] * v[q][i][j];
        }
      }
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * RPW + q < rows) inv[row] = iv;
    if (bound_inv) {
      const float yn = sqrtf(wave_sum(ysq)) * 1.001f;                 // 0.1 % head room for the fp32 roundings
      const float bg = yn * bound4[0] + bound4[1];
      const float bd = (bound4[2] > 0.f || bound4[3] > 0.f) ? bg * (yn * bound4[2] + bound4[3]) : bg;
      float biv;
      h2_row_scale(fmaxf(bd * 1.001f, 1e-30f), biv);
      if (lane == 0 && row0 + wave * RPW + q < rows) bound_inv[row] = biv;
    }
  }
  h2_store_rows<NV, RPW, NW>(v, scale, tile, dim, row0, rows, out, R);
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
  const int lane = threadIdx.x;
  const int n4 = dim >> 2;
  const int64_t row = blockIdx.x;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * dim);
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (idx < n4) {
      v[i] = xr[idx];
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)dim;
  float qs = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < n4) {
      const float d0 = v[i][0] - mean, d1 = v[i][1] - mean, d2 = v[i][2] - mean, d3 = v[i][3] - mean;
      qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(qs) / (float)dim + eps);
  float amax = 0.f, ysq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (idx < n4) {
      const f32x4 wv = reinterpret_cast<const f32x4*>(w)[idx], bv = reinterpret_cast<const f32x4*>(b)[idx];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
        amax = fmaxf(amax, fabsf(v[i][j]));
        ysq += v[i][j] * v[i][j];
      }
    }
  }
  float iv;
  const float scale = h2_row_scale(wave_max(amax), iv);
  if (lane == 0) inv[row] = iv;
  if (bound_inv) {
    const float yn = sqrtf(wave_sum(ysq)) * 1.001f;
    const float bg = yn * bound4[0] + bound4[1];
    const float bd = (bound4[2] > 0.f || bound4[3] > 0.f) ? bg * (yn * bound4[2] + bound4[3]) : bg;
    float biv;
    h2_row_scale(fmaxf(bd * 1.001f, 1e-30f), biv);
    if (lane == 0) bound_inv[row] = biv;
  }
  const int swap = (int)((row >> 3) & 1);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;                        // columns 4 idx .. 4 idx + 3: k-block idx / 4, quarter idx % 4
    if (idx < n4) {
      unsigned h0, l0, h1, l1;
      h2_pack2(v[i][0] * scale, v[i][1] * scale, h0, l0);
      h2_pack2(v[i][2] * scale, v[i][3] * scale, h1, l1);
      const int q = idx & 3;
      unsigned char* dst = out + (((int64_t)(idx >> 2) * 2) * R + row) * 32 + (((q >> 1) ^ swap) << 4) + ((q & 1) << 3);
      *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(dst + R * 32) = uint2{l0, l1};
    }
  }
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
