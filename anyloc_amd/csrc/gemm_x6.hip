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

namespace anyloc {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XT = 128;                    // tile rows (both operands)
constexpr int X_PLANE = XT * 32;           // bytes of one (kb, plane) tile image
constexpr int X_OP = 3 * X_PLANE;          // one operand, one slab
constexpr int X_STAGE = 2 * X_OP;          // 24 KiB
constexpr int X_STAGES = 3;

__device__ __forceinline__ void xtile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
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

__device__ __forceinline__ void dma16b(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

struct X6Problem {
  const unsigned char* A3; int64_t RA;     // x3 image of A, rows allocated
  const unsigned char* W3; int64_t RW;
  float* C; int64_t ldc;
  int64_t M, N;
  int K16;                                  // k-blocks of 16
  const float* bias;
};

template <int DUMMY>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(X6Problem p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  xtile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * XT, n0 = (int64_t)tn * XT;

  const unsigned a_slab = (unsigned)(3 * p.RA * 32), w_slab = (unsigned)(3 * p.RW * 32);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A3), 0, (int)((int64_t)p.K16 * a_slab), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W3), 0, (int)((int64_t)p.K16 * w_slab), 0x00020000);
  // wave w stages row-chunk w (32 rows = 1 KiB) of each of the 3 + 3 plane tiles of a slab
  unsigned a_voff[3], w_voff[3];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    a_voff[pl] = (unsigned)(((int64_t)pl * p.RA + m0 + 32 * wave) * 32 + lane * 16);
    w_voff[pl] = (unsigned)(((int64_t)pl * p.RW + n0 + 32 * wave) * 32 + lane * 16);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* st = smem + stage * X_STAGE + wave * 1024;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dma16b(a_rsrc, st + pl * X_PLANE, a_voff[pl], (unsigned)kt * a_slab);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dma16b(w_rsrc, st + X_OP + pl * X_PLANE, w_voff[pl], (unsigned)kt * w_slab);
  };

  // fragment address: lane (i = lane & 31, h = lane >> 5) reads the 16 bytes holding k = 8h .. 8h+7 of row i
  const int fr = lane & 31, fh = lane >> 5;
  const unsigned char* frag = smem + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int nk = p.K16;
  issue(0, 0);
  if (nk > 1) issue(1, 1);

  auto slab = [&](int kt, int stage) {
    // this wave's DMA pieces of slab kt have landed (the 6 of slab kt+1 may still be in flight)
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone finished slab kt-1 (no fence: waits are explicit)
    const unsigned char* sa = frag + stage * X_STAGE + (wm * 64) * 32;
    const unsigned char* sw = frag + stage * X_STAGE + X_OP + (wn * 64) * 32;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[mi][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * X_PLANE + mi * 1024);
        b[mi][pl] = *reinterpret_cast<const bf16x8*>(sw + pl * X_PLANE + mi * 1024);
      }
    if (kt + 2 < nk) issue(kt + 2, stage == 0 ? 2 : stage - 1);   // (stage + 2) % 3: the buffer slab kt-1 used
#define ANYLOC_X6_TERM(pa, pb)                                                                     \
  _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
    ANYLOC_X6_TERM(2, 0) ANYLOC_X6_TERM(0, 2) ANYLOC_X6_TERM(1, 1)
    ANYLOC_X6_TERM(1, 0) ANYLOC_X6_TERM(0, 1) ANYLOC_X6_TERM(0, 0)
#undef ANYLOC_X6_TERM
  };
  for (int kt = 0; kt < nk; kt += 3) {
    slab(kt, 0);
    if (kt + 1 < nk) slab(kt + 1, 1);
    if (kt + 2 < nk) slab(kt + 2, 2);
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 ----
  const int64_t wrow0 = m0 + wm * 64 + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * 64 + (lane & 31);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t col = wcol0 + ni * 32;
    const bool cok = col < p.N;
    const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M && cok) p.C[row * p.ldc + col] = acc[mi][ni][r] + bv;
      }
  }
}

// ---- fp32 row-major -> x3 planes ------------------------------------------------------------------------
__device__ __forceinline__ unsigned bf16_rne(float x) {          // finite inputs
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

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
      unsigned t[3][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float r = v[j + e];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned b = bf16_rne(r);
          t[pl][e] = b;
          r -= __uint_as_float(b << 16);               // exact
        }
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pk[pl][j >> 1] = t[pl][0] | (t[pl][1] << 16);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      u32x4 o;
      o[0] = pk[pl][0]; o[1] = pk[pl][1]; o[2] = pk[pl][2]; o[3] = pk[pl][3];
      *reinterpret_cast<u32x4*>(out + (((int64_t)kb * 3 + pl) * R + row) * 32 + phys * 16) = o;
    }
  }
}

}  // namespace

}  // namespace anyloc

using namespace anyloc;

extern "C" size_t anyloc_x3_bytes(int64_t rows, int64_t K) {
  return (size_t)((K + 15) / 16) * 3 * (size_t)rows * 32;
}

extern "C" int anyloc_split_x3(const float* x, int64_t ldx, int64_t rows, int64_t K, void* x3, void* stream) {
  ANYLOC_CHECK_ARG(x && x3 && rows > 0 && K > 0 && ldx >= K, "split_x3: bad arguments");
  const int K16 = (int)((K + 15) / 16);
  ProfScope prof("split_x3", (hipStream_t)stream, 0.0, 10.0 * rows * K);
  hipLaunchKernelGGL(split_x3_kernel, dim3((unsigned)((rows + 127) / 128), (unsigned)((K16 + 1) / 2)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rows, K, static_cast<unsigned char*>(x3), rows, K16);
  return launch_status("split_x3_kernel");
}

extern "C" int anyloc_gemm_nt_x6(const void* a3, const void* w3, const float* bias, float* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, void* stream) {
  ANYLOC_CHECK_ARG(a3 && w3 && C, "gemm_nt_x6: null operand");
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_nt_x6: bad shape M=%lld N=%lld K=%lld", (long long)M,
                   (long long)N, (long long)K);
  X6Problem p;
  p.A3 = static_cast<const unsigned char*>(a3); p.RA = M;
  p.W3 = static_cast<const unsigned char*>(w3); p.RW = N;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K16 = (int)((K + 15) / 16); p.bias = bias;
  ANYLOC_CHECK_ARG(anyloc_x3_bytes(M, K) < (1ull << 31) && anyloc_x3_bytes(N, K) < (1ull << 31),
                   "gemm_nt_x6: operand image exceeds the 2 GiB buffer-addressing range");
  const int tiles_m = (int)((M + XT - 1) / XT), tiles_n = (int)((N + XT - 1) / XT);
  static bool attr_set = false;
  if (!attr_set) {
    ANYLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, X_STAGES * X_STAGE));
    attr_set = true;
  }
  ProfScope prof("gemm_x6", (hipStream_t)stream, 2.0 * M * N * K, 6.0 * (M + N) * K + 4.0 * M * N);
  hipLaunchKernelGGL(gemm_x6_kernel<0>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), X_STAGES * X_STAGE,
                     (hipStream_t)stream, p, tiles_m, tiles_n);
  return launch_status("gemm_x6_kernel");
}
