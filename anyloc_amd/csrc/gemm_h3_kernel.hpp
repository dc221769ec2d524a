// gemm_h3_kernel<...>: the row-scaled two-term fp16 GEMM kernel template, shared by gemm_h3.hip (large GEMMs, quantisers,
// the C ABI) and gemm_h3s.hip (the small-M plans: other tile shapes, split-K).  Arithmetic and operand image: see the
// header of gemm_h3.hip.
#pragma once
#include "common.hpp"
#include "tile_order.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "gemm_h3_kernel.hpp: the split-K hand-off relies on the sc1 (write-through / coherent-read) cache policy of gfx942 / gfx950"
#endif

namespace anyloc {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));




// KB = k-blocks (of 16) per ring stage: one barrier and one counted wait per KB k-blocks instead of per k-block
template <int MI, int NI, int WM, int WN, int STAGES, int KB = 1>
struct H3Cfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  static constexpr int A_PLANE = BM * 32, W_PLANE = BN * 32;
  static constexpr int A_OP = 2 * A_PLANE, KSTAGE = A_OP + 2 * W_PLANE;   // one k-block: A planes, then W planes
  static constexpr int STAGE = KB * KSTAGE;
  static constexpr int LDS = STAGES * STAGE;
  // staging: a k-block of an operand is 2 planes x (rows / 32) pieces of 1 KiB (32 rows x 32 B, one DMA instruction of a
  // wave); piece f = plane * (rows / 32) + chunk goes to wave f % NW
  static constexpr int CA = BM / 32, CW = BN / 32;
  static constexpr int A_DMA = 2 * CA / NW, W_DMA = 2 * CW / NW;          // DMA instructions per wave, operand and k-block
  static constexpr int NDMA = KB * (A_DMA + W_DMA);
  static_assert((2 * CA) % NW == 0 && (2 * CW) % NW == 0, "every wave stages the same number of 1-KiB pieces");
  static_assert((STAGES - 2) * NDMA <= 63, "the counted vmcnt wait must be encodable");
};


// scale / inverse scale of a row (or tile) from its largest magnitude: amax * 2^e in [2^14, 2^15).  An all-zero (or
// denormal) row gets the LARGEST scale, 2^100: its planes are zero either way, and its 2^-100 never wins where the
// scales of several tiles are compared (attention_h3 takes the maximum over an image's V tiles as the output's bound)
__device__ __forceinline__ float h2_row_scale(float amax, float& inv) {
  const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
  const int e = ex == 0 ? 100 : max(-100, min(100, 14 - (ex - 127)));
  inv = __uint_as_float((unsigned)(127 - e) << 23);
  return __uint_as_float((unsigned)(127 + e) << 23);
}
// 2^e from a stored 2^-e (both normal powers of two: exact)
__device__ __forceinline__ float h2_scale_of_inv(float inv) { return __uint_as_float((254u << 23) - __float_as_uint(inv)); }
// two scaled values -> packed fp16 pair of the leading plane and of the residual plane
__device__ __forceinline__ void h2_pack2(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 pr;
  pr[0] = a; pr[1] = b;
  const f16x2 h = __builtin_convertvector(pr, f16x2);
  f32x2 res;
  res[0] = pr[0] - (float)h[0];
  res[1] = pr[1] - (float)h[1];
  const f16x2 l = __builtin_convertvector(res, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ float h3_gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float h3_silu(float v) { return v / (1.0f + expf(-v)); }
// the same on the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each -- ~5 instructions instead of ~25
// (expf's range reduction + an IEEE division).  |error| <= ~3 ulp of silu(v), i.e. below the 2^-22 quantisation the value
// gets on its way into the fc2 operand image.  v -> -inf: exp2 -> +inf, rcp -> 0, v * 0 = -0; v -> +inf: exp2 -> 0, v * 1.
__device__ __forceinline__ float h3_silu_fast(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// ---- quantisers: rows held in registers (NV float4 per lane and row, RPW rows per wave, NW * RPW rows per block) ----
// the scaled values of the block's rows go through an LDS tile of 256 columns, chunk by chunk, and are stored in IMAGE order
// (thread = (k-block, row, half): whole 512-byte runs per store instruction at 16 rows, 16 bytes per lane and plane).
// COH (the LayerNorm lead role of gemm_h3_kernel): the same stores WRITE-THROUGH (sc1) through the buffer descriptor `rs` of
// the image, so that a workgroup of the SAME launch on another XCD can read them once the rows' ticket says so.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t null_rsrc() { return __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(nullptr), 0, 0, 0); }
template <int RB = 16, int NT = 256, bool COH = false>
__device__ __forceinline__ void h2_store_chunk(float (*tile)[256 + 4], int i, int dim, int64_t row0, int64_t rows,
                                               unsigned char* out, int64_t R, __amdgpu_buffer_rsrc_t rs = null_rsrc()) {
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
      const int64_t off = (((int64_t)(k0 >> 4) * 2) * R + row) * 32 + ((half ^ (int)((row >> 3) & 1)) << 4);
      if constexpr (COH) {
        __builtin_amdgcn_raw_buffer_store_b128(ph, rs, (unsigned)off, 0, /*sc1*/ 16);
        __builtin_amdgcn_raw_buffer_store_b128(plo, rs, (unsigned)(off + R * 32), 0, /*sc1*/ 16);
      } else {
        *reinterpret_cast<hu32x4*>(out + off) = ph;
        *reinterpret_cast<hu32x4*>(out + off + R * 32) = plo;
      }
    }
  }
}

// rows held in registers: the scaled values of the block's rows go through the LDS tile chunk by chunk
template <int NV, int RPW = 4, int NW = 4, bool COH = false>
__device__ __forceinline__ void h2_store_rows(const f32x4 (&v)[RPW][NV], const float (&scale)[RPW], float (*tile)[256 + 4],
                                              int dim, int64_t row0, int64_t rows, unsigned char* out, int64_t R,
                                              __amdgpu_buffer_rsrc_t rs = null_rsrc()) {
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
    h2_store_chunk<NW * RPW, 64 * NW, COH>(tile, i, dim, row0, rows, out, R, rs);
  }
}

// LayerNorm (torch semantics, biased variance) of the NW * RPW rows from row0 on, quantised straight into the h2 image: the body of
// layernorm_h2_kernel (gemm_h3.hip) and of the LayerNorm LEAD role of gemm_h3_kernel<..., LNL = 2> below (COH: every store
// write-through).  bound4 / bound_inv: the FFN bound (see layernorm_h2_kernel).  Per-row arithmetic independent of RPW / NW / COH.
template <int NV, int RPW, int NW, bool COH>
__device__ __forceinline__ void ln_rows_tiled(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                              int dim, int64_t rows, float eps, unsigned char* __restrict__ out, float* __restrict__ inv,
                                              int64_t R, int64_t row0, const f32x4 bound4, float* __restrict__ bound_inv,
                                              float (*tile)[256 + 4], __amdgpu_buffer_rsrc_t rs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
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
    const bool mine = lane == 0 && row0 + wave * RPW + q < rows;
    if (mine) {
      if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(inv + row), __float_as_uint(iv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else inv[row] = iv;
    }
    if (bound_inv) {
      const float yn = sqrtf(wave_sum(ysq)) * 1.001f;                 // 0.1 % head room for the fp32 roundings
      const float bg = yn * bound4[0] + bound4[1];
      const float bd = (bound4[2] > 0.f || bound4[3] > 0.f) ? bg * (yn * bound4[2] + bound4[3]) : bg;
      float biv;
      h2_row_scale(fmaxf(bd * 1.001f, 1e-30f), biv);
      if (mine) {
        if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(bound_inv + row), __float_as_uint(biv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else bound_inv[row] = biv;
      }
    }
  }
  h2_store_rows<NV, RPW, NW, COH>(v, scale, tile, dim, row0, rows, out, R, rs);
}

// One LayerNorm row by ONE wave (torch semantics, biased variance), quantised straight into the h2 image: every lane stores its own
// four-column groups (8 bytes per plane and group) -- no LDS tile, no barriers.  The body of layernorm_h2_direct_kernel
// (gemm_h3.hip) and of the LayerNorm LEAD role of gemm_h3_kernel below (COH = true: every store write-through at agent scope, so
// that a workgroup of the SAME launch on another XCD can read the row once its tile's ticket says so).  bound4 / bound_inv: the
// FFN bound of layernorm_h2_kernel.  Per-row arithmetic identical to the tiled kernel: same bits.
template <int NV, bool COH>
__device__ __forceinline__ void ln_row_direct(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                              int dim, float eps, unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R,
                                              int64_t row, const f32x4 bound4, float* __restrict__ bound_inv) {
  const int lane = threadIdx.x & 63;
  const int n4 = dim >> 2;
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
  if (lane == 0) {
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(inv + row), __float_as_uint(iv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else inv[row] = iv;
  }
  if (bound_inv) {
    const float yn = sqrtf(wave_sum(ysq)) * 1.001f;
    const float bg = yn * bound4[0] + bound4[1];
    const float bd = (bound4[2] > 0.f || bound4[3] > 0.f) ? bg * (yn * bound4[2] + bound4[3]) : bg;
    float biv;
    h2_row_scale(fmaxf(bd * 1.001f, 1e-30f), biv);
    if (lane == 0) {
      if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(bound_inv + row), __float_as_uint(biv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else bound_inv[row] = biv;
    }
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
      if constexpr (COH) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), ((unsigned long long)h1 << 32) | h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + R * 32), ((unsigned long long)l1 << 32) | l0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(dst + R * 32) = uint2{l0, l1};
      }
    }
  }
}

// LNL (small-M plans of one image per call, gemm_h3s.hip): the launch's first p.ln_wgs workgroups are the LayerNorm LEAD role --
// one row per wave, ln_row_direct<.., true> into this GEMM's own operand image; every wave drains its write-through stores, the
// workgroup meets, one lane adds the workgroup's rows to the ticket of their BM-row tile (relaxed, agent scope: the hand-off of the
// split-K plans below).  A GEMM workgroup waits (one lane polls, bounded) until its row tile's ticket holds all the tile's rows,
// then stages A with sc1 loads.  Placement-independent: lead workgroups never wait, and there are fewer GEMM workgroups than CUs
// (h3_ln_lead_feasible), so whatever order the dispatcher picks, every lead workgroup finds a CU.
// LNL = 2 (batched GEMMs, gemm_h3.hip): lead workgroups of LEAD_ROWS = 16 rows (four per wave, the tiled arithmetic and store
// order of layernorm_h2_kernel) are INTERLEAVED with the GEMM tiles per XCD (tile_order.hpp, LeadPlan: the lead work of tile-row
// group j + 1 sits among the XCD's tiles of group j), so that LayerNorm's HBM traffic runs under the matrix work instead of in a
// launch of its own.  In-order dispatch + "every producer has a smaller workgroup id than its consumers" (checked on the host
// per shape) make it deadlock-free; the poll is bounded all the same.
template <int MI, int NI, int WM, int WN, int STAGES, int OCC, int EPI, int KB = 1, int LNL = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_h3_kernel(H3Problem p, int tiles_m, int tiles_n) {
  using Cfg = H3Cfg<MI, NI, WM, WN, STAGES, KB>;
  constexpr bool TR = EPI == EPI_SWIGLU_T || EPI == EPI_SWIGLU_T_H2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // split-K (p.ksplit > 1, small-M plans of gemm_h3s.hip): workgroup id = split * tiles + tile; split s contracts k-blocks
  // [s * kper, min(K16, (s + 1) * kper)) and the workgroup that draws the tile's last ticket sums the partial
  // accumulators IN SPLIT ORDER (deterministic whichever arrives last) and runs the epilogue
  int bid = blockIdx.x, split = 0;
  int lead_tm = -1, lead_tn = -1;
  if constexpr (LNL == 2) {
    const LeadPlan lp{tiles_m, tiles_n, p.group_m, Cfg::BM, (long long)p.M};
    const LeadRole role = lead_decode(lp, bid & 7, bid >> 3);
    if (role.kind == 0) return;
    if (role.kind == 2) {
      // ---- LayerNorm lead role: rows role.row0 .. + 15 (four per wave), through a 16 x 256 tile of the idle ring ----
      static_assert(Cfg::NW == 4 && Cfg::BM % LEAD_ROWS == 0 && Cfg::LDS >= LEAD_ROWS * 260 * 4, "lead role: four waves, whole lead blocks per tile row");
      unsigned char* o2 = const_cast<unsigned char*>(p.A2);
      const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(o2, 0, (int)((int64_t)p.K16 * 2 * p.RA * 32), 0x00020000);
      const f32x4 b4 = {p.ln_bound[0], p.ln_bound[1], p.ln_bound[2], p.ln_bound[3]};
      ln_rows_tiled<6, 4, 4, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.M, p.ln_eps, o2, const_cast<float*>(p.a_inv), p.RA, role.row0, b4,
                                   p.ln_has_bound ? const_cast<float*>(p.c_inv) : nullptr, reinterpret_cast<float(*)[256 + 4]>(smem), o_rsrc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave: its write-through stores have left
      __syncthreads();
      if (tid == 0) {
        const int cnt = (int)min((int64_t)LEAD_ROWS, p.M - role.row0);
        __hip_atomic_fetch_add(p.ln_tickets + role.row0 / Cfg::BM, (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    lead_tm = role.tm;
    lead_tn = role.tn;
  }
  if constexpr (LNL == 1) {
    if (bid < p.ln_wgs) {
      // ---- LayerNorm lead role: rows bid * NW + wave ----
      const int64_t row = (int64_t)bid * Cfg::NW + wave;
      if (row < p.M) {
        unsigned char* o2 = const_cast<unsigned char*>(p.A2);
        float* oinv = const_cast<float*>(p.a_inv);
        float* binv = p.ln_has_bound ? const_cast<float*>(p.c_inv) : nullptr;
        const f32x4 b4 = {p.ln_bound[0], p.ln_bound[1], p.ln_bound[2], p.ln_bound[3]};
        const int nv = (p.ln_dim / 4 + 63) / 64;
        if (nv <= 2) ln_row_direct<2, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.ln_eps, o2, oinv, p.RA, row, b4, binv);
        else if (nv <= 3) ln_row_direct<3, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.ln_eps, o2, oinv, p.RA, row, b4, binv);
        else if (nv <= 4) ln_row_direct<4, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.ln_eps, o2, oinv, p.RA, row, b4, binv);
        else if (nv <= 6) ln_row_direct<6, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.ln_eps, o2, oinv, p.RA, row, b4, binv);
        else ln_row_direct<8, true>(p.ln_x, p.ln_w, p.ln_b, p.ln_dim, p.ln_eps, o2, oinv, p.RA, row, b4, binv);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave: its write-through stores have left
      __syncthreads();
      if (tid == 0) {
        const int64_t r0 = (int64_t)bid * Cfg::NW;
        const int cnt = (int)max((int64_t)0, min((int64_t)Cfg::NW, p.M - r0));       // (BM % NW == 0: one tile per workgroup)
        if (cnt > 0) __hip_atomic_fetch_add(p.ln_tickets + r0 / Cfg::BM, (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    bid -= p.ln_wgs;                                       // (ln_wgs is a multiple of 8: the XCD of a GEMM workgroup is still bid % 8)
  }
  const int ntiles = tiles_m * tiles_n;
  if (p.ksplit > 1) {
    split = bid / ntiles;
    bid -= split * ntiles;
  }
  int tm, tn;
  if constexpr (LNL == 2) { tm = lead_tm; tn = lead_tn; }
  else xcd_grouped_tile(bid, tiles_m, tiles_n, p.group_m, tm, tn);
  const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;
  const int kb0 = p.ksplit > 1 ? split * p.kper : 0;
  const int kend = p.ksplit > 1 ? min(p.K16, kb0 + p.kper) : p.K16;      // the descriptors end here: later k-blocks zero-fill

  const unsigned a_slab = (unsigned)(2 * p.RA * 32), w_slab = (unsigned)(2 * p.RW * 32);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A2), 0, (int)((int64_t)kend * a_slab - p.a_off), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W2), 0, (int)((int64_t)kend * w_slab - p.w_off), 0x00020000);
  unsigned a_voff[Cfg::A_DMA], w_voff[Cfg::W_DMA], a_dst[Cfg::A_DMA], w_dst[Cfg::W_DMA];
#pragma unroll
  for (int i = 0; i < Cfg::A_DMA; ++i) {
    const int f = wave + Cfg::NW * i, pl = f / Cfg::CA, c = f % Cfg::CA;
    a_voff[i] = (unsigned)(((int64_t)pl * p.RA + m0 + 32 * c) * 32 + lane * 16);
    a_dst[i] = (unsigned)(pl * Cfg::A_PLANE + c * 1024);
  }
#pragma unroll
  for (int i = 0; i < Cfg::W_DMA; ++i) {
    const int f = wave + Cfg::NW * i, pl = f / Cfg::CW, c = f % Cfg::CW;
    w_voff[i] = (unsigned)(((int64_t)pl * p.RW + n0 + 32 * c) * 32 + lane * 16);
    w_dst[i] = (unsigned)(Cfg::A_OP + pl * Cfg::W_PLANE + c * 1024);
  }
  // stage step `ks` = k-blocks kb0 + ks * KB ... + KB - 1 (past kend: out of the descriptor's range, zero-fills)
  auto issue = [&](int ks, int stage) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      unsigned char* st = smem + stage * Cfg::STAGE + kb * Cfg::KSTAGE;
      const unsigned ao = (unsigned)(kb0 + ks * KB + kb) * a_slab, wo = (unsigned)(kb0 + ks * KB + kb) * w_slab;
#pragma unroll
      for (int i = 0; i < Cfg::A_DMA; ++i) {
        if constexpr (LNL) dma16_to_lds_sc1(a_rsrc, st + a_dst[i], a_voff[i], ao);
        else dma16_to_lds(a_rsrc, st + a_dst[i], a_voff[i], ao);
      }
#pragma unroll
      for (int i = 0; i < Cfg::W_DMA; ++i) dma16_to_lds(w_rsrc, st + w_dst[i], w_voff[i], wo);
    }
  };

  const int fr = lane & 31, fh = lane >> 5;
  const unsigned char* frag = smem + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int nk = (kend - kb0 + KB - 1) / KB;               // stage steps
  if constexpr (LNL) {
    // the rows of this tile come from the lead role of this launch: wait for the tile's ticket (one lane polls; bounded -- a
    // lost hand-off must not hang the device: it would show as a wrong result in every test instead)
    if (tid == 0) {
      const unsigned need = (unsigned)min((int64_t)Cfg::BM, p.M - m0);
      unsigned spins = 0;
      while (__hip_atomic_load(p.ln_tickets + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < (1u << 24))
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s);

  auto slab = [&](int kt, int stage) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned char* sa = frag + stage * Cfg::STAGE + kb * Cfg::KSTAGE + (wm * 32 * MI) * 32;
      const unsigned char* sw = frag + stage * Cfg::STAGE + kb * Cfg::KSTAGE + Cfg::A_OP + (wn * 32 * NI) * 32;
      f16x8 a[MI][2], b[NI][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi][pl] = *reinterpret_cast<const f16x8*>(sa + pl * Cfg::A_PLANE + mi * 1024);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni][pl] = *reinterpret_cast<const f16x8*>(sw + pl * Cfg::W_PLANE + ni * 1024);
      }
      if (kb == 0) issue(kt + STAGES - 1, (stage + STAGES - 1) % STAGES);
      // TR (EPI_SWIGLU_T*): the weight fragment is the MFMA's A operand, so acc[mi][ni] holds the TRANSPOSED 32 x 32 block --
      // lane = token, registers = 16 weight rows -- same products, same sums, other owner of each element.
      // The three products of a block are issued back to back on their accumulator (smallest first: lo x hi, hi x lo, hi x hi).
      // Round 3 issued each product for all blocks before the next one -- per accumulator the same additions in the same
      // order, so the same bits -- but the dependent chain measures 0.5 % faster end to end (w12 -0.5 %, qkv / fc2 -1 %,
      // interleaved A/B on one box, profiles/r04_ab_mfma_chain.log; column-block-outer traversal: the same, the bf16 and the
      // 16x16x32 kernels: neutral): a block's accumulator is forwarded from one MFMA to the next instead of making three
      // round trips through the register file, and at the power limit joules are time.
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {
        acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni][0], a[mi][1], acc[mi][ni], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][1], b[ni][0], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni][1], a[mi][0], acc[mi][ni], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ni][1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni][0], a[mi][0], acc[mi][ni], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ni][0], acc[mi][ni], 0, 0, 0);
      }
    }
    if constexpr (KB == 1) {
      constexpr int PIECES = Cfg::NDMA, G = (3 * MI * NI) / (PIECES + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };
  for (int kt = 0; kt < nk; kt += STAGES) {
    slab(kt, 0);
#pragma unroll
    for (int st = 1; st < STAGES; ++st)
      if (kt + st < nk) slab(kt + st, st);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (p.ksplit > 1) {
    // Partial accumulators -> workspace in register order, 16 bytes per lane (whole 1-KiB runs per instruction), as
    // WRITE-THROUGH (sc1) stores: they reach the device-coherent level themselves, so no release fence -- an agent-scope
    // release is an L2 write-back, and one per workgroup made a split GEMM 3-4 x SLOWER than the unsplit one
    // (profiles/r04_b1_plan_sweep_fences.log).  Every wave drains its stores, the workgroup meets, ONE lane takes the tile's
    // ticket (relaxed, agent scope), and the last arrival reads all slabs back with sc1 loads (the per-XCD L2s are not
    // coherent with each other) IN SPLIT ORDER -- deterministic whichever workgroup arrives last.  The hand-off recipe of
    // cdna_hip_programming.md section 6 (counter form, its sc1 variant: "sc1 slab stores -> every wave s_waitcnt vmcnt(0) ->
    // __syncthreads() -> lane 0 relaxed agent fetch_add; the reducer reads the slabs with sc1 loads").  This is a property of
    // the gfx942 / gfx950 cache policy bits (sc1 = write-through to / read from the device-coherent level), NOT of the HIP
    // memory model: the file refuses to compile for any other target (the #error below), and
    // tests/test_gpu_vit.py::test_split_k_hand_off_under_load hammers it (hundreds of concurrent tiles x up to 8 splits,
    // repeated; bit-identical to the first run and within the k-order bar of the unsplit plan).  The flag travels through
    // the (now idle) LDS ring: a second __shared__ object would cost every k-step of the pipeline a full vmcnt drain.
    constexpr int NT = 64 * Cfg::NW;
    constexpr int SC1 = 16;                                  // aux bit of the buffer instructions: sc1
    const int tile = tm * tiles_n + tn;
    const int64_t slab = (int64_t)(MI * NI * 4 * NT) * 16;   // bytes of one tile's partial
    const __amdgpu_buffer_rsrc_t sk_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(p.sk_part), 0, (int)((int64_t)p.ksplit * ntiles * slab), 0x00020000);
    const unsigned own = (unsigned)(((int64_t)split * ntiles + tile) * slab) + (unsigned)tid * 16u;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const hu32x4 v = {__float_as_uint(acc[mi][ni][4 * q]), __float_as_uint(acc[mi][ni][4 * q + 1]),
                            __float_as_uint(acc[mi][ni][4 * q + 2]), __float_as_uint(acc[mi][ni][4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, sk_rsrc, own + (unsigned)(((mi * NI + ni) * 4 + q) * NT * 16), 0, SC1);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // (also: nobody reads fragments from the ring any more)
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(p.sk_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(p.ksplit - 1);
      if (last) __hip_atomic_store(p.sk_tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();                                         // (the epilogues reuse the ring)
    if (!last) return;
    const unsigned base = (unsigned)((int64_t)tile * slab) + (unsigned)tid * 16u;
    const unsigned sstride = (unsigned)((int64_t)ntiles * slab);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned off = base + (unsigned)(((mi * NI + ni) * 4 + q) * NT * 16);
          hu32x4 u = __builtin_amdgcn_raw_buffer_load_b128(sk_rsrc, off, 0, SC1);
          f32x4 t = {__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
          for (int sp = 1; sp < p.ksplit; ++sp) {
            u = __builtin_amdgcn_raw_buffer_load_b128(sk_rsrc, off, (unsigned)sp * sstride, SC1);
            t[0] += __uint_as_float(u[0]); t[1] += __uint_as_float(u[1]); t[2] += __uint_as_float(u[2]); t[3] += __uint_as_float(u[3]);
          }
          acc[mi][ni][4 * q] = t[0]; acc[mi][ni][4 * q + 1] = t[1]; acc[mi][ni][4 * q + 2] = t[2]; acc[mi][ni][4 * q + 3] = t[3];
        }
  }

  if constexpr (TR) {
    // SwiGLU on transposed accumulators (weights in the 16-channel block layout, include/anyloc_hip.h): lane (token n =
    // lane & 31, half hl) holds, of the 32-row weight block ni, rows 4 hl + {0..3}, 8 + 4 hl + {0..3} = the GATES of hidden
    // channels c .. c+7 (c = 16 * block + 8 hl) in registers 0..7 and rows 16 + ..., 24 + ... = their VALUES in registers
    // 8..15.  silu(g) * v of one token and 8 consecutive channels is exactly one 16-byte chunk per plane of the fc2 operand
    // image: no LDS transposition, two 16-byte stores per block and lane, 32 rows x 32 B = whole 1-KiB runs per instruction.
    const int hl = lane >> 5;
    const int64_t wr0 = n0 + wn * 32 * NI;               // first weight row of this wave (N % (32 NI) == 0: all-in or all-out)
    if (wr0 >= p.N) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t row = m0 + wm * 32 * MI + mi * 32 + (lane & 31);
      const bool rok = row < p.M;
      const float ai = rok ? p.a_inv[row] : 0.0f;
      float cs = 1.0f;
      if constexpr (EPI == EPI_SWIGLU_T_H2) cs = rok ? h2_scale_of_inv(p.c_inv[row]) : 0.0f;
      float cmx = 0.0f;                                  // largest scaled magnitude of this lane's part of the row (telemetry)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int64_t wr = wr0 + ni * 32 + 4 * hl;
        f32x4 sc[4], bi[4];                              // [gate rows 0..3, gate rows 4..7, value rows 0..3, value rows 4..7]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sc[q] = *reinterpret_cast<const f32x4*>(p.w_inv + wr + 8 * q);
          bi[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wr + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = acc[mi][ni][j] * (ai * sc[j >> 2][j & 3]) + bi[j >> 2][j & 3];
          const float v = acc[mi][ni][8 + j] * (ai * sc[2 + (j >> 2)][j & 3]) + bi[2 + (j >> 2)][j & 3];
          o[j] = (p.fast_silu ? h3_silu_fast(g) : h3_silu(g)) * v;
        }
        const int64_t blk = (wr0 + ni * 32) >> 5;         // 16 hidden channels per weight block: channels 16 blk + 8 hl + j
        if (rok) {
          if constexpr (EPI == EPI_SWIGLU_T_H2) {
            unsigned qh[4], ql[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h2_pack2(o[2 * j] * cs, o[2 * j + 1] * cs, qh[j], ql[j]);
            if (p.c_max) {
#pragma unroll
              for (int j = 0; j < 8; ++j) cmx = fmaxf(cmx, fabsf(o[j] * cs));
            }
            hu32x4 ph, plo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
            unsigned char* dst = p.C2 + ((blk * 2) * p.RC + row) * 32 + ((hl ^ (int)((row >> 3) & 1)) << 4);
            *reinterpret_cast<hu32x4*>(dst) = ph;
            *reinterpret_cast<hu32x4*>(dst + p.RC * 32) = plo;
          } else {
            float* dst = p.C + row * p.ldc + 16 * blk + 8 * hl;
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
          }
        }
      }
      if constexpr (EPI == EPI_SWIGLU_T_H2) {
        // FFN-bound telemetry: c_max[row] = largest |value * 2^e_bound| the row holds in the fc2 operand image (positive floats
        // order as their bit patterns); one atomic per (row, wave) -- the two halves of a row meet by one lane exchange
        if (p.c_max) {
          cmx = fmaxf(cmx, __shfl_xor(cmx, 32, 64));
          if (rok && hl == 0) atomicMax(p.c_max + row, __float_as_uint(cmx));
        }
      }
    }
    return;
  }
  // ---- epilogue: acc * 2^-(e_row + e_col), then the same fused forms as gemm_x6.hip / gemm_f32.hip ----
  const int64_t wrow0 = m0 + wm * 32 * MI + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * 32 * NI + (lane & 31);
  if constexpr (EPI == EPI_LS_RESID) {
    if (p.epi_lds) {
      // x += gamma * (acc * 2^-(e_row+e_col) + bias) with 16-byte global accesses: the C/D layout gives a lane one
      // column of 16 rows (dword read-modify-write, 64 + 64 memory instructions per 32x64 block pair); instead each
      // wave transposes 32 x 64 sub-blocks through its own 8.5 KiB of the (now idle) LDS ring and every lane handles
      // four consecutive columns of a row: 8 + 8 memory instructions per sub-block.
      __builtin_amdgcn_s_barrier();                       // nobody reads fragments from the ring any more
      float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
      const int lr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float ai[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
          ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
        }
#pragma unroll
        for (int np = 0; np < NI / 2; ++np) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int ni = 2 * np + nb;
            const int64_t col = wcol0 + ni * 32;
            const bool cok = col < p.N;
            const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
            const float sw_ = cok ? p.w_inv[col] : 0.0f, gam = cok ? p.gamma[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 68 + nb * 32 + (lane & 31)] =
                  (acc[mi][ni][r] * (ai[r] * sw_) + bv) * gam;
          }
          const int64_t colb = n0 + wn * 32 * NI + np * 64 + c4;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + lr0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c4]);
            const int64_t row = m0 + wm * 32 * MI + mi * 32 + lr;
            if (row < p.M && colb < p.N) {
              const int64_t o = row * p.ldc + colb;
              f32x4 x4 = *reinterpret_cast<const f32x4*>(&p.resid[o]);
              x4[0] += t[0]; x4[1] += t[1]; x4[2] += t[2]; x4[3] += t[3];
              *reinterpret_cast<f32x4*>(&p.C[o]) = x4;
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_QKV_PLANES) {
    // q | k | v leave the kernel as the per-(head, 32-row group) two-plane fp16 tiles of attention_h3 (layout: common.hpp).
    // A wave owns 32 MI rows x 32 NI columns = MI row groups x NI/2 heads of ONE part (D % (32 NI) == 0).  v tiles are
    // written straight from the C/D layout (a lane holds one d and the 16 rows of its half in exactly the order the
    // consumer's B fragments want); q / k tiles are row-major, so they are transposed through the idle LDS ring.
    __builtin_amdgcn_s_barrier();
    float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const int Dm = p.heads * 64;
    const int64_t colw = n0 + wn * 32 * NI;
    const int part = (int)(colw / Dm);
    const int head0 = (int)((colw - (int64_t)part * Dm) >> 6);
    const int hl = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t rb = m0 + wm * 32 * MI + mi * 32;
      if (rb >= p.M || colw >= p.N) continue;
      float ai[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
      }
#pragma unroll
      for (int hh = 0; hh < NI / 2; ++hh) {
        float v[2][16];
        float amax = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int64_t col = wcol0 + (2 * hh + nb) * 32;
          const float bv = p.bias ? p.bias[col] : 0.0f, sw_ = p.w_inv[col];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            v[nb][r] = ai[r] != 0.0f ? acc[mi][2 * hh + nb][r] * (ai[r] * sw_) + bv : 0.0f;
            amax = fmaxf(amax, fabsf(v[nb][r]));
          }
        }
        amax = wave_max(amax);
        float inv;
        const float scale = h2_row_scale(amax, inv);
        const int64_t tile = (int64_t)(part * p.heads + head0 + hh) * p.groups + (rb >> 5);
        if (lane == 0) p.qkv_inv[tile] = inv;
        unsigned char* dst = p.qkv_planes + tile * 8192;
        if (part == 2) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int d = nb * 32 + (lane & 31);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              unsigned qh[4], ql[4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                h2_pack2(v[nb][8 * s2 + 2 * j] * scale, v[nb][8 * s2 + 2 * j + 1] * scale, qh[j], ql[j]);
              hu32x4 ph, plo;
#pragma unroll
              for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
              unsigned char* o = dst + d * 64 + (((hl * 2 + s2) ^ ((d >> 2) & 3)) << 4);
              *reinterpret_cast<hu32x4*>(o) = ph;
              *reinterpret_cast<hu32x4*>(o + 4096) = plo;
            }
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + nb * 32 + (lane & 31)] = v[nb][r] * scale;
          const int lr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + lr0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c4]);
            uint2 ph, plo;
            h2_pack2(t[0], t[1], ph.x, plo.x);
            h2_pack2(t[2], t[3], ph.y, plo.y);
            unsigned char* o = dst + lr * 128 + (((c4 >> 3) ^ ((lr >> 1) & 7)) << 4) + (c4 & 4) * 2;
            *reinterpret_cast<uint2*>(o) = ph;
            *reinterpret_cast<uint2*>(o + 4096) = plo;
          }
        }
      }
    }
    return;
  }
  if constexpr (EPI == EPI_GELU_H2 || EPI == EPI_SWIGLU_H2) {
    // the FFN hidden activation leaves the kernel as the h2 image of the fc2 GEMM, every row scaled by the caller's
    // power of two (1 / c_inv[row], an upper bound of the row: layernorm_h2).  32 x 64 sub-blocks are transposed through
    // the idle LDS ring so that a lane owns eight consecutive k of a row = one 16-byte half of an image row per plane.
    __builtin_amdgcn_s_barrier();
    float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    constexpr bool SW = EPI == EPI_SWIGLU_H2;
    constexpr int OC = SW ? 16 * NI : 64;                  // output columns per transposed sub-block
    constexpr int NSUB = SW ? 1 : NI / 2;
    constexpr int LPR = OC / 8;                            // lanes per row when reading back
    constexpr int RPI = 64 / LPR;                          // rows per read-back iteration
    const int hl = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t rb = m0 + wm * 32 * MI + mi * 32;
      if (rb >= p.M) continue;
      float ai[16], cs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
        cs[r] = row < p.M ? h2_scale_of_inv(p.c_inv[row]) : 0.0f;
      }
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        int64_t ocol0;                                     // first output column of this sub-block
        if constexpr (SW) {
          ocol0 = (n0 + wn * 32 * NI) / 2;
#pragma unroll
          for (int nj = 0; nj < NI; nj += 2) {
            const int64_t colg = wcol0 + nj * 32, colv = colg + 32;
            const bool cok = colv < p.N;
            const float bg = (cok && p.bias) ? p.bias[colg] : 0.0f, bv = (cok && p.bias) ? p.bias[colv] : 0.0f;
            const float sg = cok ? p.w_inv[colg] : 0.0f, sv = cok ? p.w_inv[colv] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float g = acc[mi][nj][r] * (ai[r] * sg) + bg;
              const float v = acc[mi][nj + 1][r] * (ai[r] * sv) + bv;
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + (nj / 2) * 32 + (lane & 31)] =
                  (p.fast_silu ? h3_silu_fast(g) : h3_silu(g)) * v * cs[r];
            }
          }
        } else {
          ocol0 = n0 + wn * 32 * NI + sub * 64;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int64_t col = wcol0 + (2 * sub + nb) * 32;
            const bool cok = col < p.N;
            const float bv = (cok && p.bias) ? p.bias[col] : 0.0f, sw_ = cok ? p.w_inv[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + nb * 32 + (lane & 31)] =
                  h3_gelu_erf(acc[mi][2 * sub + nb][r] * (ai[r] * sw_) + bv) * cs[r];
          }
        }
        const int lr0 = lane / LPR, c8 = (lane % LPR) * 8;
        const int64_t kcol = ocol0 + c8;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int lr = it * RPI + lr0;
          const int64_t row = rb + lr;
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c8]);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c8 + 4]);
          unsigned qh[4], ql[4];
          h2_pack2(t0[0], t0[1], qh[0], ql[0]);
          h2_pack2(t0[2], t0[3], qh[1], ql[1]);
          h2_pack2(t1[0], t1[1], qh[2], ql[2]);
          h2_pack2(t1[2], t1[3], qh[3], ql[3]);
          hu32x4 ph, plo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
          const bool live = row < p.M && kcol < (SW ? p.N / 2 : p.N);
          if (live) {
            unsigned char* o = p.C2 + (((kcol >> 4) * 2) * p.RC + row) * 32 + ((((kcol >> 3) & 1) ^ (int)((row >> 3) & 1)) << 4);
            *reinterpret_cast<hu32x4*>(o) = ph;
            *reinterpret_cast<hu32x4*>(o + p.RC * 32) = plo;
          }
          if (p.c_max) {
            // FFN-bound telemetry (see the transposed SwiGLU epilogue): the LPR lanes of a row meet by xor exchanges
            static_assert((LPR & (LPR - 1)) == 0, "lanes per row: a power of two");
            float cmx = live ? fmaxf(fmaxf(fmaxf(fabsf(t0[0]), fabsf(t0[1])), fmaxf(fabsf(t0[2]), fabsf(t0[3]))),
                                     fmaxf(fmaxf(fabsf(t1[0]), fabsf(t1[1])), fmaxf(fabsf(t1[2]), fabsf(t1[3])))) : 0.0f;
#pragma unroll
            for (int o2 = LPR / 2; o2 > 0; o2 >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o2, 64));
            if (row < p.M && lane % LPR == 0) atomicMax(p.c_max + row, __float_as_uint(cmx));
          }
        }
      }
    }
    return;
  }
  if constexpr (EPI == EPI_SWIGLU) {
    float bg[NI / 2], bv[NI / 2], sg[NI / 2], sv[NI / 2];
    bool cok[NI / 2];
#pragma unroll
    for (int nj = 0; nj < NI; nj += 2) {
      const int64_t colg = wcol0 + nj * 32, colv = colg + 32;
      cok[nj / 2] = colv < p.N;
      bg[nj / 2] = (cok[nj / 2] && p.bias) ? p.bias[colg] : 0.0f;
      bv[nj / 2] = (cok[nj / 2] && p.bias) ? p.bias[colv] : 0.0f;
      sg[nj / 2] = cok[nj / 2] ? p.w_inv[colg] : 0.0f;
      sv[nj / 2] = cok[nj / 2] ? p.w_inv[colv] : 0.0f;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          const float ai = p.a_inv[row];
#pragma unroll
          for (int nj = 0; nj < NI; nj += 2)
            if (cok[nj / 2]) {
              const int64_t ocol = (n0 + wn * 32 * NI + nj * 32) / 2 + (lane & 31);
              const float g = acc[mi][nj][r] * (ai * sg[nj / 2]) + bg[nj / 2];
              const float v = acc[mi][nj + 1][r] * (ai * sv[nj / 2]) + bv[nj / 2];
              p.C[row * p.ldc + ocol] = h3_silu(g) * v;
            }
        }
      }
  } else {
    float bv[NI], sw_[NI], gam[NI];
    bool cok[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t col = wcol0 + ni * 32;
      cok[ni] = col < p.N;
      bv[ni] = (cok[ni] && p.bias) ? p.bias[col] : 0.0f;
      sw_[ni] = cok[ni] ? p.w_inv[col] : 0.0f;
      gam[ni] = 0.0f;
      if constexpr (EPI == EPI_LS_RESID) gam[ni] = cok[ni] ? p.gamma[col] : 0.0f;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          const float ai = p.a_inv[row];
          int64_t orow = row, prow = 0;
          if constexpr (EPI == EPI_PATCH) {                // patch p of image b -> token row b * T + 1 + p, + pos[1 + p]
            const int64_t img = row / p.patches;
            prow = row - img * p.patches + 1;
            orow = img * (p.patches + 1) + prow;
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if (cok[ni]) {
              const float v = acc[mi][ni][r] * (ai * sw_[ni]) + bv[ni];
              const int64_t o = orow * p.ldc + wcol0 + ni * 32;
              if constexpr (EPI == EPI_PATCH) p.C[o] = v + p.pos[prow * p.N + wcol0 + ni * 32];
              else
              if constexpr (EPI == EPI_STORE) p.C[o] = p.accumulate ? p.C[o] + v : v;
              else if constexpr (EPI == EPI_GELU) p.C[o] = h3_gelu_erf(v);
              else p.C[o] = p.resid[o] + v * gam[ni];
            }
        }
      }
  }
}

}  // namespace
}  // namespace anyloc
