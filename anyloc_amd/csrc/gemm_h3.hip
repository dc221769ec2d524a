// EXPERIMENTAL -- row-scaled two-term fp16 split ("h3") GEMM: C = A W^T with fp32-level accuracy from THREE fp16
// matrix-core products per k-step (half the passes of gemm_x6.hip, which is limited by the chip's power budget).
//
// Every operand row is scaled by a power of two so that its largest magnitude lies in [2^14, 2^15):
//     x * 2^e = h + l,   h = fp16_rne(x * 2^e),   l = fp16_rne(x * 2^e - h)      (22 mantissa bits)
// l needs no extra scaling: the row scale keeps it a normal fp16 for every element within 2^16 of the row maximum,
// and below that its absolute error is 2^-39 of the row maximum.  a*b = (hh + hl + lh + ll) 2^-(ea+eb); ll is below
// 2^-24 and dropped, the other three accumulate in ONE fp32 accumulator (they have their natural magnitudes) and the
// epilogue multiplies by inv_a[row] * inv_w[col] (powers of two: exact).  CPU emulation of the whole ViT:
// tools/split_fp16_study.py.  Not used by the ViT forward yet: the attention output and the FFN hidden activation
// need their exact row maximum before they can be quantised (rows are produced by different workgroups).
//
// Operand image ("h2"): [k/16][plane 0..1][row][16] fp16, 32 bytes per (k-block, plane, row), 16-byte halves swapped
// when (row >> 3) & 1 -- the x3 image of gemm_x6.hip with two planes; same DMA staging, same fragment reads.
#include <cstdlib>

#include "common.hpp"

namespace anyloc {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void htile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nb = tiles_m * tiles_n;
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  constexpr int GM = 8;
  const int group_size = GM * tiles_n;
  const int g = logical / group_size;
  const int first_m = g * GM;
  const int gm = min(tiles_m - first_m, GM);
  const int within = logical - g * group_size;
  tm = first_m + within % gm;
  tn = within / gm;
}

__device__ __forceinline__ void hdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

struct H3Problem {
  const unsigned char* A2; int64_t RA; const float* a_inv;     // image of A, rows, 2^-e per row
  const unsigned char* W2; int64_t RW; const float* w_inv;
  float* C; int64_t ldc;
  int64_t M, N;
  int K16;
  const float* bias;
};

template <int MI, int NI, int WM, int WN, int STAGES>
struct H3Cfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  static constexpr int A_PLANE = BM * 32, W_PLANE = BN * 32;
  static constexpr int A_OP = 2 * A_PLANE, STAGE = A_OP + 2 * W_PLANE;
  static constexpr int LDS = STAGES * STAGE;
  static constexpr int A_DMA = BM / (32 * NW), W_DMA = BN / (32 * NW);
  static constexpr int NDMA = 2 * (A_DMA + W_DMA);
  static_assert(BM % (32 * NW) == 0 && BN % (32 * NW) == 0, "each wave stages whole 32-row pieces");
};

template <int MI, int NI, int WM, int WN, int STAGES, int OCC>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_h3_kernel(H3Problem p, int tiles_m, int tiles_n) {
  using Cfg = H3Cfg<MI, NI, WM, WN, STAGES>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  htile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;

  const unsigned a_slab = (unsigned)(2 * p.RA * 32), w_slab = (unsigned)(2 * p.RW * 32);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A2), 0, (int)((int64_t)p.K16 * a_slab), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W2), 0, (int)((int64_t)p.K16 * w_slab), 0x00020000);
  unsigned a_voff[2], w_voff[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    a_voff[pl] = (unsigned)(((int64_t)pl * p.RA + m0 + 32 * wave) * 32 + lane * 16);
    w_voff[pl] = (unsigned)(((int64_t)pl * p.RW + n0 + 32 * wave) * 32 + lane * 16);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* st = smem + stage * Cfg::STAGE + wave * 1024;
    const unsigned ao = (unsigned)kt * a_slab, wo = (unsigned)kt * w_slab;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < Cfg::A_DMA; ++c)
        hdma16(a_rsrc, st + pl * Cfg::A_PLANE + c * (1024 * Cfg::NW), a_voff[pl] + c * (1024 * Cfg::NW), ao);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < Cfg::W_DMA; ++c)
        hdma16(w_rsrc, st + Cfg::A_OP + pl * Cfg::W_PLANE + c * (1024 * Cfg::NW), w_voff[pl] + c * (1024 * Cfg::NW), wo);
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

  const int nk = p.K16;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s);

  auto slab = [&](int kt, int stage) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* sa = frag + stage * Cfg::STAGE + (wm * 32 * MI) * 32;
    const unsigned char* sw = frag + stage * Cfg::STAGE + Cfg::A_OP + (wn * 32 * NI) * 32;
    f16x8 a[MI][2], b[NI][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi][pl] = *reinterpret_cast<const f16x8*>(sa + pl * Cfg::A_PLANE + mi * 1024);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni][pl] = *reinterpret_cast<const f16x8*>(sw + pl * Cfg::W_PLANE + ni * 1024);
    }
    issue(kt + STAGES - 1, (stage + STAGES - 1) % STAGES);   // past the last k-block: out of range, zero-fills
#define ANYLOC_H3_TERM(pa, pb)                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
    ANYLOC_H3_TERM(1, 0) ANYLOC_H3_TERM(0, 1) ANYLOC_H3_TERM(0, 0)
#undef ANYLOC_H3_TERM
    constexpr int PIECES = Cfg::NDMA, G = (3 * MI * NI) / (PIECES + 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };
  for (int kt = 0; kt < nk; kt += STAGES) {
    slab(kt, 0);
    if (kt + 1 < nk) slab(kt + 1, 1);
    if (STAGES > 2 && kt + 2 < nk) slab(kt + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int64_t wrow0 = m0 + wm * 32 * MI + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * 32 * NI + (lane & 31);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int64_t col = wcol0 + ni * 32;
    const bool cok = col < p.N;
    const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
    const float sw_ = cok ? p.w_inv[col] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M && cok) p.C[row * p.ldc + col] = acc[mi][ni][r] * (p.a_inv[row] * sw_) + bv;
      }
  }
}

// fp32 row-major [rows, K] -> h2 image + inv[row] = 2^-e.  16 rows per block, wave w owns rows 4w..4w+3 with the row
// in registers (NV float4 per lane); the scaled values go through a 16 x 256 LDS tile and are stored in image order.
template <int NV>
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                       unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R) {
  constexpr int LDT = 256 + 4;
  __shared__ __attribute__((aligned(16))) float tile[16][LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    amax = wave_max(amax);
    // 2^e with amax * 2^e in [2^14, 2^15): e = 14 - floor(log2(amax)) from the exponent field (amax = 0 -> scale 1)
    const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
    const int e = ex == 0 ? 0 : max(-100, min(100, 14 - (ex - 127)));
    scale[q] = __uint_as_float((unsigned)(127 + e) << 23);
    if (lane == 0 && row0 + wave * 4 + q < rows) inv[row] = __uint_as_float((unsigned)(127 - e) << 23);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (i > 0) __syncthreads();
    if (idx < n4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[q][i][j] * scale[q];
        *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = tid + 256 * u;
      const int kbl = item >> 5, r = (item >> 1) & 15, half = item & 1;
      const int k0 = 256 * i + 16 * kbl + 8 * half;
      const int64_t row = row0 + r;
      if (k0 < dim && row < rows) {
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
}

}  // namespace

}  // namespace anyloc

using namespace anyloc;

extern "C" size_t anyloc_h2_bytes(int64_t rows, int64_t K) { return (size_t)((K + 15) / 16) * 2 * (size_t)rows * 32; }

extern "C" int anyloc_split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && rows > 0 && K > 0 && ldx >= K, "split_h2: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && K <= 4096 && ldx % 4 == 0, "split_h2: K must be a multiple of 16, at most 4096 (got %lld)",
                   (long long)K);
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
  else ANYLOC_SPLIT_H2(16);
#undef ANYLOC_SPLIT_H2
  return launch_status("split_h2_kernel");
}

extern "C" int anyloc_gemm_nt_h3(const void* a2, const float* a_inv, const void* w2, const float* w_inv, const float* bias,
                                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ANYLOC_CHECK_ARG(a2 && a_inv && w2 && w_inv && C, "gemm_nt_h3: null operand");
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 16 == 0 && ldc >= N, "gemm_nt_h3: bad shape");
  ANYLOC_CHECK_ARG(anyloc_h2_bytes(M, K) < (1ull << 31) && anyloc_h2_bytes(N, K) < (1ull << 31),
                   "gemm_nt_h3: operand image exceeds the 2 GiB buffer-addressing range");
  H3Problem p{};
  p.A2 = static_cast<const unsigned char*>(a2); p.RA = M; p.a_inv = a_inv;
  p.W2 = static_cast<const unsigned char*>(w2); p.RW = N; p.w_inv = w_inv;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K16 = (int)(K / 16); p.bias = bias;
  static int cfg = -1;
  if (cfg < 0) {
    const char* e = getenv("ANYLOC_H3_CFG");
    cfg = e ? atoi(e) : 0;
  }
  ProfScope prof("gemm_h3", stream, 2.0 * M * N * K, 4.0 * (M + N) * K + 4.0 * M * N);
#define ANYLOC_LAUNCH_H3(MI, NI, WM, WN, ST, OCC)                                                                     \
  do {                                                                                                                \
    using Cfg = H3Cfg<MI, NI, WM, WN, ST>;                                                                            \
    const int tiles_m = (int)((M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((N + Cfg::BN - 1) / Cfg::BN);             \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      ANYLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<MI, NI, WM, WN, ST, OCC>),          \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS));                          \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((gemm_h3_kernel<MI, NI, WM, WN, ST, OCC>), dim3((unsigned)(tiles_m * tiles_n)),                 \
                       dim3(64 * WM * WN), Cfg::LDS, stream, p, tiles_m, tiles_n);                                    \
  } while (0)
  switch (cfg) {
    case 1: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 2, 2); break;     // 128x256, 2-deep
    case 2: ANYLOC_LAUNCH_H3(4, 2, 2, 4, 2, 1); break;     // 256x256, 8 waves, 2-deep
    case 3: ANYLOC_LAUNCH_H3(4, 2, 2, 4, 3, 1); break;     // 256x256, 8 waves, 3-deep
    case 4: ANYLOC_LAUNCH_H3(4, 4, 2, 2, 2, 1); break;     // 256x256, 4 waves of 128x128 (256 accumulator registers)
    default: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 3, 2); break;    // 128x256, 3-deep ring
  }
#undef ANYLOC_LAUNCH_H3
  return launch_status("gemm_h3_kernel");
}
