// Split-bf16 ("x6") GEMM: C[M,N] = A[M,K] * W[N,K]^T with fp32-level accuracy on the bf16 matrix cores.
//
// Every fp32 operand is the EXACT sum of three bf16 terms, x = x1 + x2 + x3 (24 mantissa bits = 3 x 8,
// round-to-nearest-even at every step).  The product a*b is then a sum of nine bf16 x bf16 products, each
// exact in fp32; the six leading ones (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2) carry everything above 2^-24
// relative, the other three are below one fp32 ulp of the product and are dropped.  Six
// v_mfma_f32_32x32x16_bf16 (16 k per instruction, 8 passes) replace eight v_mfma_f32_32x32x2_f32 (2 k per
// instruction, 16 passes): 2.67x the fp32-MFMA roofline, with fp32 accumulation throughout
// (tools/split_bf16_study.py: token / VLAD deviations of the whole ViT identical to the exact-fp32 path).
//
// Operand layout in HBM ("x3"): a matrix X[R, K] is stored as bf16 planes blocked by 16 k:
//     x3[kb][plane][row][16]      kb = k / 16,  plane = 0 (leading) .. 2,  32 bytes per (kb, plane, row)
// with the two 16-byte halves of a row swapped when (row >> 3) & 1.  This is exactly the image the kernel
// wants in LDS: a 128-row tile of one (kb, plane) is 4 KiB of contiguous memory, staged by four 1-KiB
// global->LDS DMA instructions (buffer_load ... lds, no VGPR round trip), and the half-swap makes the
// ds_read_b128 fragment loads (lane (i, h) reads k = 8h..8h+7 of row i) bank-conflict free.
//
// Kernel: 128x128 tile, 4 waves (2x2, 64x64 per wave = 2x2 MFMA blocks), BK = 16 (one MFMA k-step per slab),
// 3-stage LDS ring (3 x 24 KiB -> two workgroups per CU), one barrier per slab, DMA issued two slabs ahead.
#include <cstdlib>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));




// MI x NI 32x32 MFMA blocks per wave, WM x WN waves: block tile (32 MI WM) x (32 NI WN); STAGES-deep LDS ring
template <int MI, int NI, int WM, int WN, int STAGES>
struct X6Cfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  static constexpr int A_PLANE = BM * 32, W_PLANE = BN * 32;       // bytes of one (kb, plane) tile image
  static constexpr int A_OP = 3 * A_PLANE, STAGE = A_OP + 3 * W_PLANE;
  static constexpr int LDS = STAGES * STAGE;
  static constexpr int A_DMA = BM / (32 * NW), W_DMA = BN / (32 * NW);   // 1 KiB pieces per wave per plane
  static constexpr int NDMA = 3 * (A_DMA + W_DMA);                  // DMA instructions per wave per slab
  static_assert(BM % (32 * NW) == 0 && BN % (32 * NW) == 0, "each wave stages whole 32-row pieces");
};

// one fp32 value -> its three bf16 planes at (row, col) of an x3 image with R rows
__device__ __forceinline__ void store_x3(unsigned char* img, int64_t R, int64_t row, int64_t col, float v) {
  const int e = (int)(col & 15);
  unsigned char* dst = img + (((col >> 4) * 3) * R + row) * 32 + ((((e >> 3) ^ (int)((row >> 3) & 1))) << 4) + (e & 7) * 2;
  unsigned pk[3];
  split_pair_x3(v, 0.0f, pk);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned short*>(dst + pl * R * 32) = (unsigned short)pk[pl];
}

__device__ __forceinline__ float x6_gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float x6_silu(float v) { return v / (1.0f + expf(-v)); }

// OUT3: the activation (GELU / SwiGLU epilogues) is written as the plane image the NEXT GEMM reads (p.C3, p.RC rows)
// instead of fp32 -- the value never makes an fp32 round trip through HBM
template <int MI, int NI, int WM, int WN, int STAGES, int OCC, int EPI, int SCHED = 0, bool OUT3 = false>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_x6_kernel(X6Problem p, int tiles_m, int tiles_n) {
  using Cfg = X6Cfg<MI, NI, WM, WN, STAGES>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  xcd_grouped_tile(blockIdx.x, tiles_m, tiles_n, 8, tm, tn);
  const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;

  const unsigned a_slab = (unsigned)(3 * p.RA * 32), w_slab = (unsigned)(3 * p.RW * 32);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A3), 0, (int)((int64_t)p.K16 * a_slab - p.a_off), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W3), 0, (int)((int64_t)p.K16 * w_slab - p.w_off), 0x00020000);
  // wave w stages the 32-row pieces w, w+NW, ... (1 KiB each) of every plane tile of a slab
  unsigned a_voff[3], w_voff[3];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    a_voff[pl] = (unsigned)(((int64_t)pl * p.RA + m0 + 32 * wave) * 32 + lane * 16);
    w_voff[pl] = (unsigned)(((int64_t)pl * p.RW + n0 + 32 * wave) * 32 + lane * 16);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* st = smem + stage * Cfg::STAGE + wave * 1024;
    const unsigned ao = (unsigned)kt * a_slab, wo = (unsigned)kt * w_slab;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < Cfg::A_DMA; ++c)
        dma16_to_lds(a_rsrc, st + pl * Cfg::A_PLANE + c * (1024 * Cfg::NW), a_voff[pl] + c * (1024 * Cfg::NW), ao);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < Cfg::W_DMA; ++c)
        dma16_to_lds(w_rsrc, st + Cfg::A_OP + pl * Cfg::W_PLANE + c * (1024 * Cfg::NW), w_voff[pl] + c * (1024 * Cfg::NW), wo);
  };

  // fragment address: lane (i = lane & 31, h = lane >> 5) reads the 16 bytes holding k = 8h .. 8h+7 of row i
  const int fr = lane & 31, fh = lane >> 5;
  const unsigned char* frag = smem + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int nk = p.K16;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s);

  auto slab = [&](int kt, int stage) {
    // this wave's DMA pieces of slab kt have landed (those of the STAGES-2 later slabs may still be in flight)
    // (a refill is issued in EVERY slab -- past the last k-block it is out of the descriptor's range, moves no
    // data and zero-fills -- so the count of younger pieces in flight is a compile-time constant)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::NDMA) : "memory");
    __builtin_amdgcn_s_barrier();   // everyone's pieces landed, everyone finished slab kt-1 (no fence: waits are explicit)
    const unsigned char* sa = frag + stage * Cfg::STAGE + (wm * 32 * MI) * 32;
    const unsigned char* sw = frag + stage * Cfg::STAGE + Cfg::A_OP + (wn * 32 * NI) * 32;
    bf16x8 a[MI][3], b[NI][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * Cfg::A_PLANE + mi * 1024);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni][pl] = *reinterpret_cast<const bf16x8*>(sw + pl * Cfg::W_PLANE + ni * 1024);
    }
    // refill the buffer slab kt-1 used
    issue(kt + STAGES - 1, (stage + STAGES - 1) % STAGES);
    if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(1);
#define ANYLOC_X6_TERM(pa, pb)                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
    ANYLOC_X6_TERM(2, 0) ANYLOC_X6_TERM(0, 2) ANYLOC_X6_TERM(1, 1)
    ANYLOC_X6_TERM(1, 0) ANYLOC_X6_TERM(0, 1) ANYLOC_X6_TERM(0, 0)
#undef ANYLOC_X6_TERM
    if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(0);
    if constexpr (SCHED >= 1) {
      // spread the DMA pieces of the next slab between the MFMAs instead of issuing them in one burst:
      // all fragment reads first, then {MFMA x G, one VMEM piece} groups
      constexpr int PIECES = Cfg::NDMA, G = (6 * MI * NI) / (PIECES + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (MI + NI), 0);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };
  for (int kt = 0; kt < nk; kt += STAGES) {
    slab(kt, 0);
    if (kt + 1 < nk) slab(kt + 1, 1);
    if (STAGES > 2 && kt + 2 < nk) slab(kt + 2, 2);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail refills before LDS is released
  // ---- epilogue (same fused forms as gemm_f32.hip): C/D layout of the 32x32 MFMA:
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 ----
  const int64_t wrow0 = m0 + wm * 32 * MI + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * 32 * NI + (lane & 31);
  if constexpr (EPI == EPI_SWIGLU) {
    // W rows are interleaved in groups of 32: block 2j = gate[32j..], block 2j+1 = value[32j..]
#pragma unroll
    for (int nj = 0; nj < NI; nj += 2) {
      const int64_t colg = wcol0 + nj * 32, colv = colg + 32;
      const int64_t ocol = (n0 + wn * 32 * NI + nj * 32) / 2 + (lane & 31);
      const bool cok = colv < p.N;
      const float bg = (cok && p.bias) ? p.bias[colg] : 0.0f;
      const float bv = (cok && p.bias) ? p.bias[colv] : 0.0f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M && cok) {
            const float g = acc[mi][nj][r] + bg, v = acc[mi][nj + 1][r] + bv;
            if constexpr (OUT3) store_x3(p.C3, p.RC, row, ocol, x6_silu(g) * v);
            else p.C[row * p.ldc + ocol] = x6_silu(g) * v;
          }
        }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t col = wcol0 + ni * 32;
      const bool cok = col < p.N;
      const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
      float gam = 0.0f;
      if constexpr (EPI == EPI_LS_RESID) gam = cok ? p.gamma[col] : 0.0f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M && cok) {
            const float v = acc[mi][ni][r] + bv;
            const int64_t o = row * p.ldc + col;
            if constexpr (EPI == EPI_STORE) p.C[o] = v;
            else if constexpr (EPI == EPI_GELU) {
              if constexpr (OUT3) store_x3(p.C3, p.RC, row, col, x6_gelu_erf(v));
              else p.C[o] = x6_gelu_erf(v);
            }
            else p.C[o] = p.resid[o] + v * gam;
          }
        }
    }
  }
}

template <int MI, int NI, int WM, int WN, int STAGES, int OCC, int EPI, int SCHED = 0, bool OUT3 = false>
int launch_x6(const X6Problem& p, hipStream_t stream) {
  using Cfg = X6Cfg<MI, NI, WM, WN, STAGES>;
  const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_x6_kernel<MI, NI, WM, WN, STAGES, OCC, EPI, SCHED, OUT3>), (int)(Cfg::LDS)));
  hipLaunchKernelGGL((gemm_x6_kernel<MI, NI, WM, WN, STAGES, OCC, EPI, SCHED, OUT3>), dim3((unsigned)(tiles_m * tiles_n)), dim3(64 * WM * WN),
                     Cfg::LDS, stream, p, tiles_m, tiles_n);
  return launch_status("gemm_x6_kernel");
}

template <int EPI>
int dispatch_x6(const X6Problem& p, hipStream_t stream) {
  // few tiles (small batches): 128x128 tiles double the number of workgroups
  const bool small = ((p.M + 127) / 128) * ((p.N + 255) / 256) < 512;
  if constexpr (EPI == EPI_GELU || EPI == EPI_SWIGLU)
    if (p.C3) return small ? launch_x6<2, 2, 2, 2, 3, 2, EPI, 1, true>(p, stream)
                           : launch_x6<2, 4, 2, 2, 2, 2, EPI, 1, true>(p, stream);
  // option x6_cfg (micro-benchmarks): 0 = 128x256 tile, 4 waves, 2-deep ring, two blocks per CU (default);
  //   1 = 128x128 3-deep; 2 = 256x128 3-deep (1 block/CU); 3 = 256x128 2-deep; 7 / 8 = 256x256 with 8 waves
  switch ((int)option(OPT_X6_CFG)) {
    case 1: return launch_x6<2, 2, 2, 2, 3, 2, EPI>(p, stream);
    case 2: return launch_x6<4, 2, 2, 2, 3, 1, EPI>(p, stream);
    case 3: return launch_x6<4, 2, 2, 2, 2, 2, EPI>(p, stream);
    case 4: return launch_x6<2, 4, 2, 2, 2, 2, EPI, 0>(p, stream);   // default tile, refills issued in one burst
    case 5: return launch_x6<2, 2, 2, 2, 3, 2, EPI, 1>(p, stream);
    case 6: return launch_x6<4, 2, 2, 2, 2, 2, EPI, 1>(p, stream);
    case 7: return launch_x6<4, 2, 2, 4, 2, 1, EPI, 1>(p, stream);   // 256x256 tile, 8 waves, one block per CU
    case 8: return launch_x6<2, 4, 4, 2, 2, 1, EPI, 1>(p, stream);
    case 9: return launch_x6<2, 4, 2, 2, 2, 2, EPI, 2>(p, stream);
    default:                                                         // 128x256, refills interleaved with the MFMAs
      return small ? launch_x6<2, 2, 2, 2, 3, 2, EPI, 1>(p, stream) : launch_x6<2, 4, 2, 2, 2, 2, EPI, 1>(p, stream);
  }
}

// ---- fp32 row-major -> x3 planes ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int64_t K,
                                                       unsigned char* __restrict__ out, int64_t R, int K16) {
  const int64_t row = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
  const int half = threadIdx.x & 1;
  if (row >= rows) return;
  const int phys = half ^ (int)((row >> 3) & 1);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int kb = blockIdx.y * 2 + kk;
    if (kb >= K16) break;
    const int64_t k0 = (int64_t)kb * 16 + half * 8;
    float v[8];
    if (k0 + 8 <= K && (ldx & 3) == 0) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(x + row * ldx + k0);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(x + row * ldx + k0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K) ? x[row * ldx + k0 + j] : 0.0f;
    }
    unsigned pk[3][4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      unsigned t[3];
      split_pair_x3(v[j], v[j + 1], t);
      pk[0][j >> 1] = t[0]; pk[1][j >> 1] = t[1]; pk[2][j >> 1] = t[2];
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      u32x4 o;
      o[0] = pk[pl][0]; o[1] = pk[pl][1]; o[2] = pk[pl][2]; o[3] = pk[pl][3];
      *reinterpret_cast<u32x4*>(out + (((int64_t)kb * 3 + pl) * R + row) * 32 + phys * 16) = o;
    }
  }
}

// LayerNorm whose output is written directly as the plane image of the next GEMM's A operand (no fp32 y).
// 16 rows per 256-thread block.  Phase 1: wave w holds its rows 4w..4w+3 entirely in registers (NV float4 per lane
// and row) and reduces mean / variance with wave shuffles.  Phase 2, per chunk of 256 columns: the normalised
// values go through a [16][256] fp32 LDS tile and are read back in IMAGE order -- thread = (k-block, row, half) --
// so every store instruction writes whole 512-byte runs of the image, 16 bytes per lane.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, int dim, int64_t rows,
                                                           float eps, unsigned char* __restrict__ out, int64_t R) {
  constexpr int LDT = 256 + 4;                     // padded tile row (floats): rows land on different banks
  __shared__ __attribute__((aligned(16))) float tile[16][LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  f32x4 v[4][NV];
  float mean[4], rstd[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);       // tail rows recompute the last row, never stored
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
    mean[q] = wave_sum(s) / (float)dim;
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < n4) {
        const float d0 = v[q][i][0] - mean[q], d1 = v[q][i][1] - mean[q], d2 = v[q][i][2] - mean[q], d3 = v[q][i][3] - mean[q];
        qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    rstd[q] = 1.0f / sqrtf(wave_sum(qs) / (float)dim + eps);
  }
  // image-order role of this thread inside a 256-column chunk: two (k-block, row, half) items
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;                                 // float4 index inside the row = chunk i, lane
    if (i > 0) __syncthreads();
    if (idx < n4) {
      const f32x4 wv = reinterpret_cast<const f32x4*>(w)[idx], bv = reinterpret_cast<const f32x4*>(b)[idx];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[q][i][j] - mean[q]) * rstd[q] * wv[j] + bv[j];
        *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = tid + 256 * u;                              // 16 k-blocks x 16 rows x 2 halves
      const int kbl = item >> 5, r = (item >> 1) & 15, half = item & 1;
      const int k0 = 256 * i + 16 * kbl + 8 * half;
      const int64_t row = row0 + r;
      if (k0 < dim && row < rows) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half + 4]);
        unsigned pk[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned t[3];
          split_pair_x3(j < 2 ? lo[2 * j] : hi[2 * j - 4], j < 2 ? lo[2 * j + 1] : hi[2 * j - 3], t);
          pk[0][j] = t[0]; pk[1][j] = t[1]; pk[2][j] = t[2];
        }
        unsigned char* dst = out + (((int64_t)(k0 >> 4) * 3) * R + row) * 32 + ((half ^ (int)((row >> 3) & 1)) << 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          u32x4 o;
          o[0] = pk[pl][0]; o[1] = pk[pl][1]; o[2] = pk[pl][2]; o[3] = pk[pl][3];
          *reinterpret_cast<u32x4*>(dst + pl * R * 32) = o;
        }
      }
    }
  }
}

}  // namespace

int layernorm_x3(const float* x, const float* w, const float* b, int64_t rows, int dim, float eps, void* x3,
                 hipStream_t stream) {
  ANYLOC_CHECK_ARG(dim % 16 == 0 && dim <= 2048, "layernorm_x3: dim %d (needs a multiple of 16, at most 2048)", dim);
  ProfScope prof("layernorm_x3", stream, 8.0 * rows * dim, 10.0 * rows * dim);
  const dim3 grid((unsigned)((rows + 15) / 16));
  unsigned char* out = static_cast<unsigned char*>(x3);
  const int nv = (dim / 4 + 63) / 64;
  switch (nv) {
    case 1: hipLaunchKernelGGL(layernorm_x3_kernel<1>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    case 2: hipLaunchKernelGGL(layernorm_x3_kernel<2>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    case 3: hipLaunchKernelGGL(layernorm_x3_kernel<3>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    case 4: hipLaunchKernelGGL(layernorm_x3_kernel<4>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    case 5: hipLaunchKernelGGL(layernorm_x3_kernel<5>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    case 6: hipLaunchKernelGGL(layernorm_x3_kernel<6>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
    default: hipLaunchKernelGGL(layernorm_x3_kernel<8>, grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, rows); break;
  }
  return launch_status("layernorm_x3_kernel");
}

size_t x3_bytes(int64_t rows, int64_t K) { return (size_t)((K + 15) / 16) * 3 * (size_t)rows * 32; }

int split_x3(const float* x, int64_t ldx, int64_t rows, int64_t K, void* x3, hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && x3 && rows > 0 && K > 0 && ldx >= K, "split_x3: bad arguments");
  const int K16 = (int)((K + 15) / 16);
  ProfScope prof("split_x3", stream, 0.0, 10.0 * rows * K);
  hipLaunchKernelGGL(split_x3_kernel, dim3((unsigned)((rows + 127) / 128), (unsigned)((K16 + 1) / 2)), dim3(256), 0,
                     stream, x, ldx, rows, K, static_cast<unsigned char*>(x3), rows, K16);
  return launch_status("split_x3_kernel");
}

int gemm_x6(const X6Problem& p, int epilogue, hipStream_t stream) {
  ANYLOC_CHECK_ARG(p.A3 && p.W3 && (p.C || p.C3), "gemm_x6: null operand");
  if (p.C3) {
    const int64_t n_out = epilogue == EPI_SWIGLU ? p.N / 2 : p.N;
    ANYLOC_CHECK_ARG((epilogue == EPI_GELU || epilogue == EPI_SWIGLU) && n_out % 16 == 0 && p.RC >= p.M,
                     "gemm_x6: plane-image output needs a GELU / SwiGLU epilogue and a multiple of 16 output columns");
  }
  ANYLOC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K16 > 0 && p.RA >= p.M && p.RW >= p.N, "gemm_x6: bad shape");
  ANYLOC_CHECK_ARG((size_t)p.K16 * 3 * (size_t)p.RA * 32 < (1ull << 31) && (size_t)p.K16 * 3 * (size_t)p.RW * 32 < (1ull << 31),
                   "gemm_x6: operand image exceeds the 2 GiB buffer-addressing range");
  const int64_t K = 16ll * p.K16;
  ProfScope prof(p.tag ? p.tag : "gemm_x6", stream, 2.0 * p.M * p.N * K, 6.0 * (p.M + p.N) * K + 4.0 * p.M * p.N);
  switch (epilogue) {
    case EPI_STORE: return dispatch_x6<EPI_STORE>(p, stream);
    case EPI_GELU: return dispatch_x6<EPI_GELU>(p, stream);
    case EPI_LS_RESID:
      ANYLOC_CHECK_ARG(p.gamma && p.resid, "gemm_x6: LS_RESID needs gamma and resid");
      return dispatch_x6<EPI_LS_RESID>(p, stream);
    case EPI_SWIGLU:
      ANYLOC_CHECK_ARG(p.N % 64 == 0, "gemm_x6: SWIGLU needs N %% 64 == 0");
      return dispatch_x6<EPI_SWIGLU>(p, stream);
    default: set_error("gemm_x6: unsupported epilogue %d", epilogue); return ANYLOC_ERR_INVALID_ARG;
  }
}

}  // namespace anyloc

using namespace anyloc;

extern "C" size_t anyloc_x3_bytes(int64_t rows, int64_t K) { return x3_bytes(rows, K); }

extern "C" int anyloc_split_x3(const float* x, int64_t ldx, int64_t rows, int64_t K, void* x3, void* stream) {
  return split_x3(x, ldx, rows, K, x3, (hipStream_t)stream);
}

extern "C" int anyloc_gemm_nt_x6(const void* a3, const void* w3, const float* bias, float* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, void* stream) {
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_nt_x6: bad shape M=%lld N=%lld K=%lld", (long long)M,
                   (long long)N, (long long)K);
  X6Problem p{};
  p.A3 = static_cast<const unsigned char*>(a3); p.RA = M;
  p.W3 = static_cast<const unsigned char*>(w3); p.RW = N;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K16 = (int)((K + 15) / 16); p.bias = bias;
  return gemm_x6(p, EPI_STORE, (hipStream_t)stream);
}
