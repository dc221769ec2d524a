// PCA fit in float64 on the double-precision matrix cores (reference utilities.py:522-586: sklearn
// PCA(svd_solver='full') = a LAPACK SVD of the centred [n, f] matrix).
//
// anyloc_amd/pca.py reduces the fit to the smaller symmetric matrix of the centred data (Gram  Xc Xc^T  or scatter  Xc^T Xc),
// its eigendecomposition, and -- on the Gram side -- the back-projection  U^T Xc.  Both products square / carry the
// condition number, so they are formed in float64: C[i, j] = sum_c A(i, c) B(j, c) on v_mfma_f64_16x16x4_f64, with the
// operands read where they lie -- fp32 data converted and CENTRED on the way into LDS (no float64 copy of X: 3.9 GB at
// 10 000 x 49 152), eigenvectors as float64 columns in either storage order.  One 128 x 128 tile of C per 512-thread
// workgroup (a wave: 32 x 64 = 8 accumulator tiles, 8 MFMAs of 64 cycles per 6 LDS reads; two waves per SIMD), 32 contraction steps per LDS
// stage, the next stage's global loads (16-byte where alignment allows) in flight during the current stage's MFMAs.  A symmetric
// product computes the tiles on and above the diagonal and mirrors them.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {
namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// element (r, c) of an operand = value at p[r * rs + c * cs] (fp32 or float64) minus mean[mean_on_c ? c : r].
// How a thread fetches it is a template parameter of the kernel: 0 one element at a time (any strides, either type),
// 1 fp32 four at a time along c (cs == 1, the mean runs along c), 2 fp32 four at a time along r (rs == 1, the mean runs
// along r and is loaded once); the host picks 1 / 2 when pointer and strides are 16-byte aligned and 4 | extent.
struct F64Operand {
  const float* p32;
  const double* p64;
  int64_t rs, cs;
  const double* mean;
  int mean_on_c;
};

constexpr int TM = 128, TK = 32;
constexpr int LDT = TM + 16;                      // LDS tile [TK][LDT] doubles (row c, column = tile row); 144 doubles = 32 banks
                                                  // mod 64: the two k rows a half-wave's fragment read touches fill all banks
constexpr int NT = 512;                           // 8 waves: 4 x 2, a wave owns 32 x 64 of the 128 x 128 tile
constexpr int EPT = TM * TK / NT;                 // elements per thread and stage (8)

template <int V>
struct F64Regs {                                  // one operand's share of a stage in a thread's registers
  float4 q[V ? EPT / 4 : 1];
  double m[V ? EPT : 1];                          // the means that go with q (V = 2: loaded once)
  double d[V ? 1 : EPT];
};

template <int V>
__device__ __forceinline__ void f64_coords(int idx, int cs_is_1, int& rr, int& cc) {
  if (V == 1) { rr = idx / (TK / 4); cc = 4 * (idx % (TK / 4)); }          // 4 c's of one row
  else if (V == 2) { rr = 4 * (idx % (TM / 4)); cc = idx / (TM / 4); }     // 4 rows at one c
  else if (cs_is_1) { cc = idx % TK; rr = idx / TK; }
  else { rr = idx % TM; cc = idx / TM; }
}

template <int V>
__device__ __forceinline__ void f64_row_means(const F64Operand& o, F64Regs<V>& g, int tid, int64_t r0, int64_t R) {
  if constexpr (V == 2) {
#pragma unroll
    for (int e = 0; e < EPT / 4; ++e) {
      int rr, cc;
      f64_coords<V>(e * NT + tid, 0, rr, cc);
#pragma unroll
      for (int j = 0; j < 4; ++j) g.m[4 * e + j] = (o.mean && r0 + rr + j < R) ? o.mean[r0 + rr + j] : 0.0;
    }
  }
}

// (R, K are multiples of 4 in the vector modes -- the host checks -- so a fetched group is inside or outside as a whole;
// outside groups become zeros: padding contributes nothing and has no mean subtracted)
template <int V>
__device__ __forceinline__ void f64_fetch(const F64Operand& o, F64Regs<V>& g, int tid, int64_t r0, int64_t k0, int64_t R, int64_t K) {
  if constexpr (V != 0) {
#pragma unroll
    for (int e = 0; e < EPT / 4; ++e) {
      int rr, cc;
      f64_coords<V>(e * NT + tid, 0, rr, cc);
      const int64_t r = r0 + rr, c = k0 + cc;
      const bool in = r < R && c < K;
      const float* at = V == 1 ? o.p32 + r * o.rs + c : o.p32 + c * o.cs + r;
      g.q[e] = in ? *reinterpret_cast<const float4*>(at) : float4{0.f, 0.f, 0.f, 0.f};
      if constexpr (V == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g.m[4 * e + j] = (in && o.mean) ? o.mean[c + j] : 0.0;
      }
    }
  } else {
    const int cs1 = o.cs == 1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      int rr, cc;
      f64_coords<V>(e * NT + tid, cs1, rr, cc);
      const int64_t r = r0 + rr, c = k0 + cc;
      double v = 0.0;
      if (r < R && c < K) {
        const int64_t at = r * o.rs + c * o.cs;
        v = o.p32 ? (double)o.p32[at] : o.p64[at];
        if (o.mean) v -= o.mean[o.mean_on_c ? c : r];
      }
      g.d[e] = v;
    }
  }
}

template <int V>
__device__ __forceinline__ void f64_stash(const F64Operand& o, const F64Regs<V>& g, double (*T)[LDT], int tid, int64_t r0,
                                          int64_t k0, int64_t R, int64_t K) {
  if constexpr (V != 0) {
#pragma unroll
    for (int e = 0; e < EPT / 4; ++e) {
      int rr, cc;
      f64_coords<V>(e * NT + tid, 0, rr, cc);
      const bool in = r0 + rr < R && k0 + cc < K;
      const float v[4] = {g.q[e].x, g.q[e].y, g.q[e].z, g.q[e].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double x = in ? (double)v[j] - g.m[4 * e + j] : 0.0;
        if (V == 1) T[cc + j][rr] = x;
        else T[cc][rr + j] = x;
      }
    }
  } else {
    const int cs1 = o.cs == 1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      int rr, cc;
      f64_coords<V>(e * NT + tid, cs1, rr, cc);
      T[cc][rr] = g.d[e];
    }
  }
}

template <int VA, int VB>
__global__ __launch_bounds__(NT) void gemm_f64_kernel(F64Operand A, F64Operand B, int64_t M, int64_t N, int64_t K,
                                                      int symmetric, double* __restrict__ C) {
  // tile order: the workgroups resident on one XCD (one per CU: 144 KiB of LDS) cover 6 tile rows x ~5 tile columns, so
  // that its L2 serves each 128-row panel to 5 - 6 workgroups instead of every workgroup streaming its own from HBM
  // A symmetric product runs the tiles on and above the diagonal only: tile row v is paired with row T - 1 - v (T + 1 tiles
  // together), so that the grid is a rectangle of equal work per row and the XCDs finish together.
  int ti, tj;
  const int Tm = (int)((M + TM - 1) / TM), Tn = (int)((N + TM - 1) / TM);
  if (symmetric) {
    int v, u;
    xcd_grouped_tile((int)blockIdx.x, (Tm + 1) / 2, Tm + 1, 6, v, u);
    if (u < Tm - v) { ti = v; tj = v + u; }
    else if (Tm - 1 - v != v) { ti = Tm - 1 - v; tj = u - 1; }
    else return;                                   // the middle row of an odd T has no partner
  } else {
    xcd_grouped_tile((int)blockIdx.x, Tm, Tn, 6, ti, tj);
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char f64_smem[];
  double (*As)[TK][LDT] = reinterpret_cast<double (*)[TK][LDT]>(f64_smem);
  double (*Bs)[TK][LDT] = As + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;         // rows 32 wr .. + 31, columns 64 wc .. + 63: 2 x 4 MFMA tiles
  const int64_t i0 = (int64_t)ti * TM, j0 = (int64_t)tj * TM;

  f64x4 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int64_t steps = (K + TK - 1) / TK;
  F64Regs<VA> ga;
  F64Regs<VB> gb;
  f64_row_means<VA>(A, ga, tid, i0, M);
  f64_row_means<VB>(B, gb, tid, j0, N);
  f64_fetch<VA>(A, ga, tid, i0, 0, M, K);
  f64_fetch<VB>(B, gb, tid, j0, 0, N, K);
  f64_stash<VA>(A, ga, As[0], tid, i0, 0, M, K);
  f64_stash<VB>(B, gb, Bs[0], tid, j0, 0, N, K);
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;        // fragment: lane holds A[row fr][k fk] and B[k fk][col fr]
  for (int64_t s = 0; s < steps; ++s) {
    const int st = (int)(s & 1);
    const int64_t kn = (s + 1) * TK;
    if (s + 1 < steps) {
      f64_fetch<VA>(A, ga, tid, i0, kn, M, K);
      f64_fetch<VB>(B, gb, tid, j0, kn, N, K);
    }
#pragma unroll 2
    for (int ks = 0; ks < TK / 4; ++ks) {
      double a[2], b[4];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = As[st][4 * ks + fk][32 * wr + 16 * m + fr];
#pragma unroll
      for (int m = 0; m < 4; ++m) b[m] = Bs[st][4 * ks + fk][64 * wc + 16 * m + fr];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (s + 1 < steps) {                           // the other stage: its last readers passed the previous barrier
      f64_stash<VA>(A, ga, As[st ^ 1], tid, i0, kn, M, K);
      f64_stash<VB>(B, gb, Bs[st ^ 1], tid, j0, kn, N, K);
    }
    __syncthreads();
  }

  // C / D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + 32 * wr + 16 * mi + fk + 4 * r, j = j0 + 64 * wc + 16 * ni + fr;
        if (i < M && j < N) {
          C[i * N + j] = acc[mi][ni][r];
          if (symmetric && tj != ti) C[j * N + i] = acc[mi][ni][r];
        }
      }
}

// vector fetches need whole groups of four inside the matrix, 16-byte aligned addresses and the mean along the other index
int pick_vec(const F64Operand& o, int64_t R, int64_t K) {
  if (!o.p32 || (reinterpret_cast<uintptr_t>(o.p32) & 15)) return 0;
  if (o.cs == 1 && K % 4 == 0 && o.rs % 4 == 0 && (!o.mean || o.mean_on_c)) return 1;
  if (o.rs == 1 && R % 4 == 0 && o.cs % 4 == 0 && (!o.mean || !o.mean_on_c)) return 2;
  return 0;
}

template <int VA, int VB>
int launch_f64(const F64Operand& A, const F64Operand& B, int64_t M, int64_t N, int64_t K, bool symmetric, double* C,
               hipStream_t stream, const char* what) {
  const int64_t Tm = (M + TM - 1) / TM, Tn = (N + TM - 1) / TM;
  const dim3 grid((unsigned)(symmetric ? ((Tm + 1) / 2) * (Tm + 1) : Tm * Tn));
  constexpr size_t lds = sizeof(double) * 4 * TK * LDT;
  static DynLds dyn_lds_once;                            // (per device; a failure is reported, not remembered)
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(gemm_f64_kernel<VA, VB>), (int)lds));
  ProfScope prof(what, stream, 2.0 * (double)M * (double)N * (double)K * (symmetric ? 0.5 : 1.0), 0.0);
  hipLaunchKernelGGL((gemm_f64_kernel<VA, VB>), grid, dim3(NT), lds, stream, A, B, M, N, K, symmetric ? 1 : 0, C);
  return launch_status("gemm_f64_kernel");
}

int gemm_f64(const F64Operand& A, const F64Operand& B, int64_t M, int64_t N, int64_t K, bool symmetric, double* C,
             hipStream_t stream, const char* what) {
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && ((M + TM - 1) / TM) * ((N + TM - 1) / TM) < (1ll << 31),
                   "%s: bad shape %lld x %lld x %lld", what, (long long)M, (long long)N, (long long)K);
  const int va = pick_vec(A, M, K), vb = pick_vec(B, N, K);
  if (va == 1 && vb == 1) return launch_f64<1, 1>(A, B, M, N, K, symmetric, C, stream, what);   // Gram
  if (va == 2 && vb == 2) return launch_f64<2, 2>(A, B, M, N, K, symmetric, C, stream, what);   // scatter
  if (vb == 2) return launch_f64<0, 2>(A, B, M, N, K, symmetric, C, stream, what);              // eigenvectors x data
  return launch_f64<0, 0>(A, B, M, N, K, symmetric, C, stream, what);
}

}  // namespace
}  // namespace anyloc

extern "C" int anyloc_pca_gram_f64(const float* X, int64_t n, int64_t f, const double* mean, int side, double* out,
                                   void* stream) {
  using namespace anyloc;
  ANYLOC_CHECK_ARG(X && out, "pca_gram_f64: null pointer");
  ANYLOC_CHECK_ARG(n > 0 && f > 0, "pca_gram_f64: bad shape %lld x %lld", (long long)n, (long long)f);
  ANYLOC_CHECK_ARG(side == 0 || side == 1, "pca_gram_f64: side must be 0 (Gram, [n, n]) or 1 (scatter, [f, f])");
  if (side == 0) {                                 // rows of X against rows of X, contraction over the features
    const F64Operand a{X, nullptr, f, 1, mean, 1};
    return gemm_f64(a, a, n, n, f, true, out, (hipStream_t)stream, "pca_gram_f64");
  }
  const F64Operand a{X, nullptr, 1, f, mean, 0};   // columns of X against columns of X, contraction over the samples
  return gemm_f64(a, a, f, f, n, true, out, (hipStream_t)stream, "pca_gram_f64");
}

extern "C" int anyloc_pca_axes_f64(const double* vec, int64_t sample_stride, int64_t axis_stride, int64_t k, const float* X,
                                   int64_t n, int64_t f, const double* mean, double* out, void* stream) {
  using namespace anyloc;
  ANYLOC_CHECK_ARG(vec && X && out, "pca_axes_f64: null pointer");
  ANYLOC_CHECK_ARG(n > 0 && f > 0 && k > 0 && sample_stride > 0 && axis_stride > 0,
                   "pca_axes_f64: bad shape n=%lld f=%lld k=%lld strides %lld / %lld", (long long)n, (long long)f, (long long)k,
                   (long long)sample_stride, (long long)axis_stride);
  const F64Operand a{nullptr, vec, axis_stride, sample_stride, nullptr, 0};   // A(i, c) = component c of eigenvector i
  const F64Operand b{X, nullptr, 1, f, mean, 0};                              // B(j, c) = X[c, j] - mean[j]
  return gemm_f64(a, b, k, f, n, false, out, (hipStream_t)stream, "pca_axes_f64");
}
