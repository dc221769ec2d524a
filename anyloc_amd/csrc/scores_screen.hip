// Screened retrieval scores: ONE fp16 matrix-core product per k instead of three, with a proven error bound, and an exact
// re-scoring of the few database rows the bound cannot rule out.
//
// replaces: the score half of faiss.IndexFlatIP / IndexFlatL2 .search as called by get_top_k_recall (reference
// utilities.py:439-450) for MANY queries (csrc/topk.hip: the h3 score panels).
//
// Why: the score panels run three fp16 MFMA products per k (hi x hi + hi x lo + lo x hi of the row-scaled two-plane images:
// 22-bit operands, csrc/gemm_h3.hip) at the chip's power limit -- 0.47-0.49 of the 16-bit peak / 3.  A retrieval needs the
// fp32-accurate score only of the rows that can end up in the top-k list.  The LEADING planes alone give
//     s~ = sum_k a_hi w_hi,   |s - s~| <= (rho_a + rho_w + rho_a rho_w + accum) |a| |w|
// (rho = |x - x_hi| / |x|, the row's relative residual norm, MEASURED from its residual plane: ~1.9e-4 for ordinary rows against
// the 4.9e-4 worst case of an 11-bit rounding; Cauchy-Schwarz on the three neglected terms; accum = the fp32 accumulation)
// at a third of the matrix work and half the operand traffic.  With t~ the k-th largest screened score of a query and d the
// bound in the units of the compared value, every row of the true top-k list has s~ >= t~ - 2 d (k rows have s >= t~ - d, so
// the true k-th score is >= t~ - d; a row of the true list is at least that, and its screened score at most d below it).
// Those rows -- a few tens per query on the bench's database -- are re-scored from the fp32 rows in float64 (more accurate
// than the three-product panel: a correctly rounded dot product up to the final conversion) and ranked with the merge
// kernel's order (value, then lower index).  A query with more candidates than SCREEN_CMAX flags the call, which then runs
// the unscreened path: the result never depends on the bound being tight, only the time does.
//
// Kernels: gemm_screen_kernel (256 x 256 tiles, 8 waves, v_mfma_f32_16x16x32_f16, leading planes only, 4-deep DMA ring of
// 32-k stages), screen_compact_kernel (threshold filter of a query's screened row -> candidate columns),
// screen_rescore_kernel (one workgroup per query: the query in registers, four candidate rows per step, float64 sums),
// screen_select_kernel (candidates + running list -> new running list, rank by counting 64-bit keys).
#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

typedef _Float16 sc_f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------- screening GEMM
// C[M, N] (+)= (A_hi W_hi^T) * a_inv[row] * w_inv[col] on the leading planes of two h2 images (gemm_h3.hip layout:
// [k/16][plane][row][32 B], 16-byte halves swapped when (row >> 3) & 1).  Structure of gemm_h3m_kernel with the other three
// plane reads and two of three products removed: a stage is a PAIR of k-blocks [A hi 8 KiB | W hi 8 KiB] x 2 = 32 KiB, four
// stages = 128 KiB (one workgroup per CU, two waves per SIMD); wave w stages 1-KiB piece w of each operand and k-block (four
// DMA instructions per wave and stage), three stages in flight under the 32 MFMAs (512 matrix-core cycles) of the current one.
constexpr int SC_BM = 256, SC_BN = 256, SC_NW = 8, SC_WN = 2, SC_MB = 4, SC_NB = 8;
constexpr int SC_PLANE = 256 * 32;                         // one operand, one k-block, leading plane
constexpr int SC_KBLK = 2 * SC_PLANE;                      // A hi | W hi
constexpr int SC_STAGE = 2 * SC_KBLK;                      // a pair of k-blocks
constexpr int SC_STAGES = 4;
constexpr int SC_LDS = SC_STAGES * SC_STAGE;               // 128 KiB
constexpr int SC_NDMA = 4;                                 // DMA instructions per wave and stage

__global__ __launch_bounds__(64 * SC_NW, 2) void gemm_screen_kernel(H3Problem p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / SC_WN, wn = wave % SC_WN;
  int tm, tn;
  xcd_grouped_tile(blockIdx.x, tiles_m, tiles_n, p.group_m, tm, tn);
  const int64_t m0 = (int64_t)tm * SC_BM, n0 = (int64_t)tn * SC_BN;

  const unsigned a_slab = (unsigned)(2 * p.RA * 32), w_slab = (unsigned)(2 * p.RW * 32);
  // the descriptors end with the last k-block's LEADING plane (pairs past the end zero-fill)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A2), 0, (int)((int64_t)p.K16 * a_slab - p.a_off), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W2), 0, (int)((int64_t)p.K16 * w_slab - p.w_off), 0x00020000);
  // rows past M / N: inside the image when RA > M (other rows' data, discarded by the epilogue) or past its plane -- the
  // plane that follows is the residual plane of the same k-block (finite fp16), also discarded
  const unsigned a_voff = (unsigned)((m0 + 32 * wave) * 32 + lane * 16), w_voff = (unsigned)((n0 + 32 * wave) * 32 + lane * 16);
  auto issue = [&](int ks, int stage) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      unsigned char* st = sc_smem + stage * SC_STAGE + kb * SC_KBLK + wave * 1024;
      dma16_to_lds(a_rsrc, st, a_voff, (unsigned)(2 * ks + kb) * a_slab);
      dma16_to_lds(w_rsrc, st + SC_PLANE, w_voff, (unsigned)(2 * ks + kb) * w_slab);
    }
  };
  // fragment address of this lane inside a 16-row block: k-group kg reads (k-block kg & 1, half kg >> 1) (gemm_h3m.hip)
  const int fr = lane & 15, kg = lane >> 4;
  const unsigned char* frag = sc_smem + (kg & 1) * SC_KBLK + fr * 32 + (((kg >> 1) ^ ((fr >> 3) & 1)) << 4);

  f32x4 acc[SC_MB][SC_NB];
#pragma unroll
  for (int ma = 0; ma < SC_MB; ++ma)
#pragma unroll
    for (int nb = 0; nb < SC_NB; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K16 + 1) / 2;                          // pairs of k-blocks
#pragma unroll
  for (int s = 0; s < SC_STAGES - 1; ++s) issue(s, s);
  for (int ks = 0; ks < nk; ++ks) {
    const int stage = ks % SC_STAGES;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SC_STAGES - 2) * SC_NDMA) : "memory");   // this wave's pieces of pair ks have landed ...
    __builtin_amdgcn_s_barrier();                          // ... and everybody's; nobody reads the stage of pair ks - 1 any more
    const unsigned char* sa = frag + stage * SC_STAGE + (wm * 16 * SC_MB) * 32;
    const unsigned char* sw = frag + stage * SC_STAGE + SC_PLANE + (wn * 16 * SC_NB) * 32;
    sc_f16x8 a[SC_MB], b[SC_NB];
#pragma unroll
    for (int ma = 0; ma < SC_MB; ++ma) a[ma] = *reinterpret_cast<const sc_f16x8*>(sa + ma * 512);
#pragma unroll
    for (int nb = 0; nb < SC_NB; ++nb) b[nb] = *reinterpret_cast<const sc_f16x8*>(sw + nb * 512);
    issue(ks + SC_STAGES - 1, (stage + SC_STAGES - 1) % SC_STAGES);
#pragma unroll
    for (int ma = 0; ma < SC_MB; ++ma)
#pragma unroll
      for (int nb = 0; nb < SC_NB; ++nb) acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ma], b[nb], acc[ma][nb], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane = output column (lane & 15) of each 16-column block, rows 4 (lane >> 4) + r ----
  const int64_t col0 = n0 + wn * 16 * SC_NB + fr;
  float sw_[SC_NB];
  bool cok[SC_NB];
#pragma unroll
  for (int nb = 0; nb < SC_NB; ++nb) {
    const int64_t col = col0 + nb * 16;
    cok[nb] = col < p.N;
    sw_[nb] = cok[nb] ? p.w_inv[col] : 0.0f;
  }
#pragma unroll
  for (int ma = 0; ma < SC_MB; ++ma)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + wm * 16 * SC_MB + ma * 16 + 4 * kg + r;
      if (row < p.M) {
        const float ai = p.a_inv[row];
#pragma unroll
        for (int nb = 0; nb < SC_NB; ++nb)
          if (cok[nb]) {
            const float v = acc[ma][nb][r] * (ai * sw_[nb]);
            const int64_t o = row * p.ldc + col0 + nb * 16;
            p.C[o] = p.accumulate ? p.C[o] + v : v;
          }
      }
    }
}

// ------------------------------------------------------------------------------------------- candidates of a query
// the value the merge kernel compares (topk.hip: topk_merge_kernel), from a raw score
__device__ __forceinline__ float screen_value(float s, int64_t c, int metric, float qq, const float* dn, const float* dnorm) {
  float v = s;
  if (dnorm) v = v / dnorm[c];
  if (metric) v = -((qq + dn[c]) - 2.0f * v);
  return v;
}

// One workgroup per query: every column of the screened row whose value is within `margin[q]` of the k-th best screened value
// thr[q * k + k - 1] is a candidate -> cand[q * cmax + i] = column (order arbitrary: the selection ranks by value and index),
// count[q] = how many there are (may exceed cmax: the caller re-runs the query set unscreened), overflow |= count > cmax.
__global__ __launch_bounds__(256) void screen_compact_kernel(const float* __restrict__ scores, int64_t ld, int64_t ncols, int k,
                                                             int metric, const float* __restrict__ qn, const float* __restrict__ dn,
                                                             const float* __restrict__ dnorm, const float* __restrict__ thr,
                                                             const float* __restrict__ margin, int cmax, int* __restrict__ cand,
                                                             int* __restrict__ count, int* __restrict__ overflow) {
  __shared__ int n_s;
  const int tid = threadIdx.x;
  const int64_t q = blockIdx.x;
  if (tid == 0) n_s = 0;
  __syncthreads();
  const float* srow = scores + q * ld;
  const float qq = metric ? qn[q] : 0.f;
  const float t = thr[q * k + k - 1] - margin[q];
  for (int64_t c = tid; c < ncols; c += 256) {
    const float v = screen_value(srow[c], c, metric, qq, dn, dnorm);
    if (v >= t) {
      const int slot = atomicAdd(&n_s, 1);
      if (slot < cmax) cand[q * cmax + slot] = (int)c;
    }
  }
  __syncthreads();
  if (tid == 0) {
    count[q] = n_s;
    if (n_s > cmax) atomicOr(overflow, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------- re-score
// One workgroup of 512 threads per query: the query row sits in registers (thread t holds columns 4 t + 2048 j .. + 3,
// j < NJ: dim <= 2048 NJ), the candidate rows are read once each, FOUR per step (coalesced 16-byte loads, four rows' loads in
// flight per thread and column group), products and sums in float64, one fixed reduction tree per step: deterministic.
constexpr int RS_G = 4;
template <int NJ>
__global__ __launch_bounds__(512) void screen_rescore_kernel(const float* __restrict__ queries, const float* __restrict__ db,
                                                             int64_t dim, int cmax, const int* __restrict__ cand,
                                                             const int* __restrict__ count, int metric, const float* __restrict__ qn,
                                                             const float* __restrict__ dn, const float* __restrict__ dnorm,
                                                             float* __restrict__ cand_v) {
  __shared__ double red[8][RS_G];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q = blockIdx.x;
  const int n = min(count[q], cmax);
  const int n4 = (int)(dim >> 2);
  f32x4 qv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = tid + 512 * j;
    qv[j] = idx < n4 ? reinterpret_cast<const f32x4*>(queries + q * dim)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float qq = metric ? qn[q] : 0.f;
  for (int i0 = 0; i0 < n; i0 += RS_G) {
    int c[RS_G];
    const f32x4* dr[RS_G];
#pragma unroll
    for (int g = 0; g < RS_G; ++g) {
      c[g] = cand[q * cmax + min(i0 + g, n - 1)];        // (a short last group re-reads its last row; the copy is not stored)
      dr[g] = reinterpret_cast<const f32x4*>(db + (int64_t)c[g] * dim);
    }
    double s[RS_G];
#pragma unroll
    for (int g = 0; g < RS_G; ++g) s[g] = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 512 * j;
      if (idx < n4) {
        f32x4 d[RS_G];
#pragma unroll
        for (int g = 0; g < RS_G; ++g) d[g] = dr[g][idx];
#pragma unroll
        for (int g = 0; g < RS_G; ++g)
          s[g] += ((double)qv[j][0] * (double)d[g][0] + (double)qv[j][1] * (double)d[g][1]) +
                  ((double)qv[j][2] * (double)d[g][2] + (double)qv[j][3] * (double)d[g][3]);
      }
    }
#pragma unroll
    for (int g = 0; g < RS_G; ++g) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[g] += __shfl_xor(s[g], off, 64);
      if (lane == 0) red[wave][g] = s[g];
    }
    __syncthreads();
    if (tid < RS_G && i0 + tid < n) {
      const double tot = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
      cand_v[q * cmax + i0 + tid] = screen_value((float)tot, cand[q * cmax + i0 + tid], metric, qq, dn, dnorm);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------ select
// One workgroup per query: the re-scored candidates of this column range (global index col_base + column) and the running
// list (first == 0) -> the best k by (value, then lower index) -> the running list.  Rank by counting over 64-bit keys
// = order-preserving bits of the value | ~index (31 bits are enough: indices of one call's database range).
__device__ __forceinline__ unsigned sc_ord_bits(float v) {
  v += 0.0f;
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__global__ __launch_bounds__(256) void screen_select_kernel(const int* __restrict__ cand, const float* __restrict__ cand_v,
                                                            const int* __restrict__ count, int cmax, int64_t col_base, int k,
                                                            float* __restrict__ run_v, long long* __restrict__ run_i, int first) {
  extern __shared__ __attribute__((aligned(16))) char sel_raw[];
  unsigned long long* key = reinterpret_cast<unsigned long long*>(sel_raw);      // [cmax + k]
  float* ev = reinterpret_cast<float*>(key + cmax + k);                             // [cmax + k]
  long long* ei = reinterpret_cast<long long*>(ev + ((cmax + k + 1) & ~1));         // [cmax + k]
  const int tid = threadIdx.x;
  const int64_t q = blockIdx.x;
  const int n = min(count[q], cmax);
  const int nl = first ? 0 : k;
  const int tot = n + nl;
  for (int i = tid; i < tot; i += 256) {
    float v;
    long long gi;
    if (i < n) {
      v = cand_v[q * cmax + i];
      gi = col_base + cand[q * cmax + i];
    } else {
      v = run_v[q * k + (i - n)];
      gi = run_i[q * k + (i - n)];
    }
    ev[i] = v;
    ei[i] = gi;
    // padding entries of the running list (index -1, value -inf) rank below everything real: index field 0
    const unsigned long long low = gi < 0 ? 0ull : (0xffffffffull - (unsigned long long)(gi & 0xffffffffll));
    key[i] = ((unsigned long long)sc_ord_bits(v) << 32) | low;
  }
  __syncthreads();
  for (int e = tid; e < tot; e += 256) {
    const unsigned long long mine = key[e];
    int rank = 0;
    for (int j = 0; j < tot; ++j) rank += (key[j] > mine) || (key[j] == mine && j < e);
    if (rank < k) {
      run_v[q * k + rank] = ev[e];
      run_i[q * k + rank] = ei[e];
    }
  }
  // fewer than k entries in all: pad
  for (int i = tot + tid; i < k; i += 256) {
    run_v[q * k + i] = -INFINITY;
    run_i[q * k + i] = -1;
  }
}

// rho[row] = |x - x_hi| / |x| of every row of an h2 image, from its RESIDUAL plane (lo = fp16(x 2^e - hi): |x 2^e - hi| <=
// |lo| (1 + 2^-11), and what lies below fp16's normal range adds < 1e-9 |x|), with the row's raw sum of squares `ss` and
// 2^-e `inv`; rho_max (bits of a positive float) = running maximum over the rows seen (atomicMax: order-independent).
// One workgroup per 32 rows; wave w reads the 1-KiB pieces (32 rows x 32 B) of k-blocks w, w + 4, ...
__global__ __launch_bounds__(256) void plane_resid_kernel(const unsigned char* __restrict__ img, int64_t R, int K16, int64_t rows,
                                                          const float* __restrict__ inv, const float* __restrict__ ss,
                                                          float* __restrict__ rho, unsigned* __restrict__ rho_max) {
  __shared__ float part[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 32, row = row0 + (lane >> 1);
  float acc = 0.f;
  if (row < R)
    for (int kb = wave; kb < K16; kb += 4) {
      const sc_f16x8 v = *reinterpret_cast<const sc_f16x8*>(img + (((int64_t)kb * 2 + 1) * R + row0) * 32 + lane * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)v[j] * (float)v[j];
    }
  acc += __shfl_xor(acc, 1, 64);
  if ((lane & 1) == 0) part[wave][lane >> 1] = acc;
  __syncthreads();
  if (tid < 32 && row0 + tid < rows) {
    const float sq = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    const float nrm2 = ss[row0 + tid];
    float r = 0.f;
    if (nrm2 > 0.f) r = sqrtf(sq) * inv[row0 + tid] / sqrtf(nrm2) * 1.01f + 1e-9f;
    r = fminf(r, 1.0f / 2048.0f * 1.01f);                // never above the worst case of the rounding itself
    rho[row0 + tid] = r;
    if (rho_max) atomicMax(rho_max, __float_as_uint(r));
  }
}

// the same rho from the quantiser's own residual sums (split_h1_wide: sum (x 2^e - hi)^2 per row, the exact residual)
__global__ void rho_from_resid_kernel(const float* __restrict__ resid_sq, const float* __restrict__ inv, const float* __restrict__ ss,
                                      int64_t rows, float* __restrict__ rho, unsigned* __restrict__ rho_max) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float r = 0.f;
  if (i < rows) {
    if (ss[i] > 0.f) r = sqrtf(resid_sq[i]) * inv[i] / sqrtf(ss[i]) * 1.01f + 1e-9f;
    r = fminf(r, 1.0f / 2048.0f * 1.01f);
    rho[i] = r;
  }
  const float m = wave_max(r);
  if (rho_max && (threadIdx.x & 63) == 0) atomicMax(rho_max, __float_as_uint(m));
}

__global__ void rho_max_kernel(const float* __restrict__ rho, int64_t n, unsigned* __restrict__ rho_max) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, rho[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(rho_max, __float_as_uint(m));
}

// margin[q] = 2 * bound of query q in the units of the compared value + slack for the fp32 roundings of the transform
// (norm_db only: every database row counts with norm 1).  bound = ((rho_q + rho_db + rho_q rho_db) (1 + 2^-10) + accum) |q|,
// rho_db = the largest relative residual norm among the database rows scored so far (Cauchy-Schwarz on the three neglected
// terms a_hi r_w + r_a w_hi + r_a r_w), accum = the fp32 accumulation term of screen_accum().
// ss_max (searches WITHOUT ANYLOC_TOPK_NORMALIZE_DB): bits of the largest raw sum of squares among the database rows scored so
// far -- the rows then count with their raw norms, and the bound of every column is at most the one of the longest row.
__global__ void screen_margin_kernel(const float* __restrict__ qn, const float* __restrict__ rho_q, const unsigned* __restrict__ rho_max,
                                     const unsigned* __restrict__ ss_max, int64_t nq, int metric, float accum,
                                     float* __restrict__ margin) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const float nrm = sqrtf(qn[q]);
  const float dmax = ss_max ? sqrtf(__uint_as_float(*ss_max)) * 1.0001f : 1.0f;
  const float rd = __uint_as_float(*rho_max), rq = rho_q[q];
  const float eps = (rq + rd + rq * rd) * (1.0f + 1.0f / 1024.0f) + accum;
  const float d = eps * nrm * dmax * (metric ? 2.0f : 1.0f);
  margin[q] = 2.0f * d * 1.0001f + 1e-6f * (1.0f + (metric ? qn[q] + dmax * dmax + 2.0f * nrm * dmax : nrm * dmax));
}

}  // namespace

// fp32 accumulation term of the bound, relative to |a| |w|: one accumulator over K16 / 2 matrix-core instructions per chunk
// (each adds a 32-term partial sum: K16 / 2 + 32 roundings relative to sum |a_i w_i| <= |a| |w|), one more rounding per chunk
// added into the panel and two for the descaling -- each priced at 2^-23, not 2^-24: the matrix core's internal rounding is not
// documented as round-to-nearest, and a truncating adder would err by a whole ulp
float screen_accum(int k16_chunk, int chunks) {
  return (float)((((double)k16_chunk / 2.0 + 32.0 + 3.0 * chunks) / 8388608.0) * 1.01);
}

int screen_resid(const unsigned char* img, int64_t R, int K16, int64_t rows, const float* inv, const float* ss, float* rho,
                 unsigned* rho_max, hipStream_t stream) {
  ProfScope prof("topk_screen_resid", stream, 0.0, 32.0 * rows * K16);
  hipLaunchKernelGGL(plane_resid_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, stream, img, R, K16, rows, inv, ss, rho, rho_max);
  return launch_status("plane_resid_kernel");
}

int gemm_screen(const H3Problem& p_in, hipStream_t stream) {
  H3Problem p = p_in;
  ANYLOC_CHECK_ARG(p.A2 && p.a_inv && p.W2 && p.w_inv && p.C && p.M > 0 && p.N > 0 && p.K16 > 0 && p.RA >= p.M && p.RW >= p.N,
                   "gemm_screen: bad operands");
  ANYLOC_CHECK_ARG((size_t)p.K16 * 2 * (size_t)p.RA * 32 < (1ull << 31) && (size_t)p.K16 * 2 * (size_t)p.RW * 32 < (1ull << 31),
                   "gemm_screen: operand image exceeds the 2 GiB buffer-addressing range");
  ProfScope prof(p.tag ? p.tag : "gemm_screen", stream, 2.0 * p.M * p.N * 16.0 * p.K16, 2.0 * (p.M + p.N) * 16.0 * p.K16 + 4.0 * p.M * p.N);
  p.group_m = (int)std::max<int64_t>(1, option(OPT_H3_GROUP_M));
  const int tiles_m = (int)((p.M + SC_BM - 1) / SC_BM), tiles_n = (int)((p.N + SC_BN - 1) / SC_BN);
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_screen_kernel), SC_LDS));
  hipLaunchKernelGGL(gemm_screen_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(64 * SC_NW), SC_LDS, stream, p, tiles_m, tiles_n);
  return launch_status("gemm_screen_kernel");
}

int screen_rho_from_resid(const float* resid_sq, const float* inv, const float* ss, int64_t rows, float* rho, unsigned* rho_max,
                          hipStream_t stream) {
  hipLaunchKernelGGL(rho_from_resid_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, resid_sq, inv, ss, rows, rho, rho_max);
  return launch_status("rho_from_resid_kernel");
}

int screen_rho_max(const float* rho, int64_t n, unsigned* rho_max, hipStream_t stream) {
  hipLaunchKernelGGL(rho_max_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, stream, rho, n, rho_max);
  return launch_status("rho_max_kernel");
}

int screen_margins(const float* qn, const float* rho_q, const unsigned* rho_max, const unsigned* ss_max, int64_t nq, int metric,
                   float accum, float* margin, hipStream_t stream) {
  hipLaunchKernelGGL(screen_margin_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, qn, rho_q, rho_max, ss_max, nq, metric,
                     accum, margin);
  return launch_status("screen_margin_kernel");
}

int screen_compact(const float* scores, int64_t ld, int64_t ncols, int64_t nq, int k, int metric, const float* qn, const float* dn,
                   const float* dnorm, const float* thr, const float* margin, int cmax, int* cand, int* count, int* overflow,
                   hipStream_t stream) {
  ProfScope prof("topk_screen_compact", stream, 0.0, 4.0 * nq * ncols);
  hipLaunchKernelGGL(screen_compact_kernel, dim3((unsigned)nq), dim3(256), 0, stream, scores, ld, ncols, k, metric, qn, dn, dnorm, thr,
                     margin, cmax, cand, count, overflow);
  return launch_status("screen_compact_kernel");
}

bool screen_rescore_supported(int64_t dim) { return dim % 4 == 0 && dim <= 2048 * 24; }

int screen_rescore(const float* queries, const float* db, int64_t dim, int64_t nq, int cmax, const int* cand, const int* count,
                   int metric, const float* qn, const float* dn, const float* dnorm, float* cand_v, hipStream_t stream) {
  ANYLOC_CHECK_ARG(screen_rescore_supported(dim), "screen_rescore: dim %lld not served", (long long)dim);
  ProfScope prof("topk_screen_rescore", stream, 0.0, 0.0);
  const int nj = (int)((dim / 4 + 511) / 512);
#define ANYLOC_RESCORE(NJ)                                                                                                      \
  hipLaunchKernelGGL((screen_rescore_kernel<NJ>), dim3((unsigned)nq), dim3(512), 0, stream, queries, db, dim, cmax, cand, count, \
                     metric, qn, dn, dnorm, cand_v)
  if (nj <= 1) ANYLOC_RESCORE(1);
  else if (nj <= 2) ANYLOC_RESCORE(2);
  else if (nj <= 4) ANYLOC_RESCORE(4);
  else if (nj <= 8) ANYLOC_RESCORE(8);
  else if (nj <= 16) ANYLOC_RESCORE(16);
  else ANYLOC_RESCORE(24);
#undef ANYLOC_RESCORE
  return launch_status("screen_rescore_kernel");
}

size_t screen_select_lds(int cmax, int k) { return (size_t)(cmax + k) * 8 + (size_t)((cmax + k + 1) & ~1) * 4 + (size_t)(cmax + k) * 8 + 16; }

int screen_select(const int* cand, const float* cand_v, const int* count, int cmax, int64_t col_base, int64_t nq, int k, float* run_v,
                  int64_t* run_i, int first, hipStream_t stream) {
  ProfScope prof("topk_screen_select", stream, 0.0, 0.0);
  hipLaunchKernelGGL(screen_select_kernel, dim3((unsigned)nq), dim3(256), screen_select_lds(cmax, k), stream, cand, cand_v, count, cmax,
                     col_base, k, run_v, reinterpret_cast<long long*>(run_i), first);
  return launch_status("screen_select_kernel");
}

}  // namespace anyloc
