// Exact brute-force top-k retrieval (inner product / squared L2).
//
// replaces: faiss.IndexFlatIP / IndexFlatL2 .add + .search as called by
// get_top_k_recall (reference utilities.py:439-450).
//
// The database is processed in column panels: an fp32 MFMA GEMM (gemm_f32.hip)
// writes the [nq, panel] score block, then one block per query merges the
// panel into that query's running top-k list (threshold filter against the
// list's k-th entry + one-wave selection; ties -> lower database index,
// deterministic).
// Governing roofline: fp32 MFMA (2*nq flop per database float).
#include <algorithm>

#include "common.hpp"

namespace anyloc {

namespace {

constexpr int TILE = 2048;        // score columns per filtering step (at most TILE new candidates)
constexpr int BOOT = 256;         // columns of the very first step (no threshold yet: every column is a candidate)
constexpr int CAP = TILE + BOOT;  // candidates that beat the running k-th entry, buffered in LDS between selections
constexpr int64_t PANEL = 32768;  // database rows per GEMM panel
constexpr int KMAX = 1024;

__device__ __forceinline__ bool better(float v, long long i, float bv, long long bi) {
  return v > bv || (v == bv && i < bi);
}
__global__ __launch_bounds__(256) void rownorm_sq_kernel(const float* __restrict__ x, int64_t dim,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  const float* r = x + (int64_t)blockIdx.x * dim;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < dim; i += 256) ss += r[i] * r[i];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One block per query.  Running list (best first) lives in run_v/run_i [nq,k]; `first` != 0
// initialises it to (-inf, -1).  metric 1: candidate value = -(qn + dn - 2 ip).
// dnorm != nullptr: the database rows were scored RAW and are normalised here, score / dnorm[col]
// (dnorm = max(||row||, 1e-12): F.normalize of the row, reference utilities.py:436, without a normalised copy).
//
// Threshold filter + rank selection.  The k-th entry of the running list bounds everything that can still enter it, so
// the block streams the score row in steps of TILE columns and keeps only the candidates that beat that entry (value,
// then lower index) in an LDS buffer -- for a list that has seen n columns about k / n of a step.  When the buffer could
// overflow on the next step, and at the end, the best k of list + buffer are found WITHOUT selection rounds: every entry
// carries a 64-bit key whose unsigned order is the retrieval order,
//     key = order-preserving bits of the value | tie field (list entries 0xFFFF, candidates 0x7FFF - column) | ~slot,
// (list entries come from earlier columns than any buffered candidate, and the list is sorted: on equal values a list
// entry precedes a candidate and a lower slot precedes a higher one -- exactly "ties -> lower database index"), each
// thread counts the keys above those of its own entries while the whole block reads the key array as LDS broadcasts, and an
// entry of rank r < k goes to position r of the new list.  Keys are unique (the slot), so ranks are; the buffer is filled
// through an LDS counter in a varying order, which the ranks do not depend on: the result is deterministic.
__device__ __forceinline__ unsigned ord_bits(float v) {
  v += 0.0f;                                               // -0 -> +0: equal under the float comparison of the filter
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float ord_value(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}
__device__ __forceinline__ unsigned long long merge_key(float v, unsigned tie, int slot) {
  return ((unsigned long long)ord_bits(v) << 32) | ((unsigned long long)(tie & 0xffffu) << 16) | (unsigned long long)(0xffff - slot);
}

// rank of every entry among the `tot` keys (number of keys above it); entries of rank < k go to position rank of the new
// list.  U of a thread's entries share one pass over the key array, which every lane reads at the same address (broadcast).
template <int U>
__device__ __forceinline__ void rank_entries(const unsigned long long* key, const long long* ei, float* nv, long long* ni,
                                             int tot, int tot2, int k) {
  const int tid = threadIdx.x;
  for (int e0 = tid; e0 < tot; e0 += 256 * U) {
    unsigned long long mine[U];
    int rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      mine[u] = e < tot ? key[e] : ~0ull;                  // no entry: nothing ranks above it, and it is never written
      rank[u] = 0;
    }
    for (int j = 0; j < tot2; j += 2) {
      const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&key[j]);
#pragma unroll
      for (int u = 0; u < U; ++u) rank[u] += (kk.x > mine[u]) + (kk.y > mine[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      if (e < tot && rank[u] < k) {
        nv[rank[u]] = ord_value((unsigned)(mine[u] >> 32));
        ni[rank[u]] = ei[e];
      }
    }
  }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ scores, int64_t ld, int64_t ncols,
                                                         int64_t col_base, int k, int metric,
                                                         const float* __restrict__ qn, const float* __restrict__ dn,
                                                         const float* __restrict__ dnorm,
                                                         float* __restrict__ run_v, long long* __restrict__ run_i,
                                                         int first) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // slots [0, k): running list, best first; slots [k, k + CAP): buffered candidates
  unsigned long long* key = reinterpret_cast<unsigned long long*>(smem_raw);          // [k2 + CAP], k2 = k rounded up to even
  const int k2 = (k + 1) & ~1;
  long long* ei = reinterpret_cast<long long*>(key + k2 + CAP);                        // [k2 + CAP] global indices
  float* nv = reinterpret_cast<float*>(ei + k2 + CAP);                                 // [k] list under construction
  long long* ni = reinterpret_cast<long long*>(nv + k2);
  __shared__ int n_s;
  __shared__ float thr_v;
  __shared__ long long thr_i;

  const int tid = threadIdx.x;
  const int64_t q = blockIdx.x;
  const float* srow = scores + q * ld;
  const float qq = metric ? qn[q] : 0.f;

  for (int i = tid; i < k; i += 256) {
    const float v = first ? -INFINITY : run_v[q * k + i];
    key[i] = merge_key(v, 0xffffu, i);
    ei[i] = first ? -1 : run_i[q * k + i];
  }
  if (tid == 0) {
    n_s = 0;
    thr_v = first ? -INFINITY : run_v[q * k + k - 1];
    thr_i = first ? -1 : run_i[q * k + k - 1];
  }
  __syncthreads();

  // best k of slots [0, k + n_s) -> list; called by the whole block after a barrier
  auto select = [&]() {
    const int tot = k + n_s;
    const int tot2 = (tot + 1) & ~1;
    if (tid == 0 && (tot & 1)) key[tot] = 0;               // pad to a whole 16-byte read: below every real key
    __syncthreads();
    if (tot <= 256) rank_entries<1>(key, ei, nv, ni, tot, tot2, k);
    else if (tot <= 512) rank_entries<2>(key, ei, nv, ni, tot, tot2, k);
    else rank_entries<4>(key, ei, nv, ni, tot, tot2, k);
    __syncthreads();
    for (int i = tid; i < k; i += 256) {
      key[i] = merge_key(nv[i], 0xffffu, i);
      ei[i] = ni[i];
    }
    if (tid == 0) {
      thr_v = nv[k - 1];
      thr_i = ni[k - 1];
      n_s = 0;
    }
    __syncthreads();
  };

  int64_t c0 = 0;
  // block-uniform copy of n_s: read after the post-scatter barrier of a step, and nobody appends again before the
  // pre-scatter barrier of the next step, which every thread reaches only after its own read -- so all threads take the
  // same branch below (a decision made from n_s itself could see a faster wave's atomicAdd of the next step)
  int n_now = 0;
  while (c0 < ncols) {
    const int step = (first && c0 == 0) ? BOOT : TILE;    // the first step only seeds the threshold
    if (n_now + step > CAP) {
      select();
      n_now = 0;
    }
    const float tv = thr_v;
    const long long ti = thr_i;
    float v[TILE / 256];
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j) {                // all loads of the step first
      const int64_t c = c0 + tid + 256 * j;
      v[j] = -INFINITY;
      if (256 * j < step && c < ncols) {
        v[j] = srow[c];
        if (dnorm) v[j] = v[j] / dnorm[c];
        if (metric) v[j] = -((qq + dn[c]) - 2.0f * v[j]);
      }
    }
    __syncthreads();                                      // every thread has read n_s / the threshold of this step
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j) {
      const int64_t c = c0 + tid + 256 * j;
      const long long gi = col_base + c;
      if (256 * j < step && c < ncols && better(v[j], gi, tv, ti)) {
        const int slot = k + atomicAdd(&n_s, 1);
        key[slot] = merge_key(v[j], 0x7fffu - (unsigned)c, slot);
        ei[slot] = gi;
      }
    }
    __syncthreads();
    if (first && c0 == 0) {
      select();                                           // the seed columns become the first list: a real threshold from here on
      n_now = 0;
    } else {
      n_now = n_s;
    }
    c0 += step;
  }
  select();
  for (int i = tid; i < k; i += 256) {
    run_v[q * k + i] = ord_value((unsigned)(key[i] >> 32));
    run_i[q * k + i] = ei[i];
  }
}

// raw sums of squares -> dnorm = max(sqrt(ss), 1e-12) (F.normalize's denominator) and, for the L2 metric, the squared
// norm of the normalised row dn = ss / dnorm^2
__global__ void dbnorm_kernel(const float* __restrict__ ss, int64_t n, float* __restrict__ dnorm, float* __restrict__ dn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float d = fmaxf(sqrtf(ss[i]), 1e-12f);
  dnorm[i] = d;
  dn[i] = (ss[i] / d) / d;
}

// few-query path: sum the split-K slices of C_s[row, 0..63] (and of the row sums of squares) in slice order and write
// the [nq, ncols] score panel the merge kernel reads, plus the raw sum of squares of every database row
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ part, const float* __restrict__ rsq_part,
                                                             int S, int64_t rows, int nq, float* __restrict__ scores,
                                                             float* __restrict__ ss) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= rows) return;
  float acc[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) acc[q] = 0.f;
  float r = 0.f;
  for (int s = 0; s < S; ++s) {
    const f32x4* pr = reinterpret_cast<const f32x4*>(part + ((int64_t)s * rows + j) * 64);
#pragma unroll
    for (int q4 = 0; q4 < 16; ++q4) {
      const f32x4 v = pr[q4];
      acc[4 * q4] += v[0]; acc[4 * q4 + 1] += v[1]; acc[4 * q4 + 2] += v[2]; acc[4 * q4 + 3] += v[3];
    }
    r += rsq_part[(int64_t)s * rows + j];
  }
  ss[j] = r;
#pragma unroll
  for (int q = 0; q < 64; ++q)
    if (q < nq) scores[(int64_t)q * rows + j] = acc[q];
}

// metric 1: stored values are negated squared distances -> flip sign; padding -> +inf
__global__ void topk_finish_kernel(float* __restrict__ v, const long long* __restrict__ idx, int64_t n, int metric) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (metric) v[i] = idx[i] < 0 ? INFINITY : -v[i];
}

// few queries (<= 64) against long rows: a [nq, panel] GEMM has too few tiles to stream the database, so the panel is
// scored by a split-K launch with the database as the M operand (gemm_nt_splitk)
constexpr int SPLITK_MAX = 16;
bool few_queries(int64_t nq, int64_t dim) { return nq <= 64 && dim % 32 == 0 && dim >= 4096; }
// number of K slices: a divisor of dim/32 that fills the 512 resident workgroups (2 per CU) best, slices >= 1024 long
int choose_ksplit(int64_t rows, int64_t dim) {
  const int64_t tiles = (rows + 127) / 128, kb = dim / 32;
  int best = 1;
  double best_u = 0.0;
  for (int s = 1; s <= SPLITK_MAX; ++s) {
    if (kb % s != 0 || dim / s < 1024) continue;
    const int64_t blocks = tiles * s;
    const double u = (double)blocks / (double)((blocks + 511) / 512 * 512) * (blocks >= 256 ? 1.0 : (double)blocks / 256.0);
    if (u > best_u + 1e-9) { best_u = u; best = s; }
  }
  return best;
}

// Many queries against long rows: the score panels run on the two-term fp16 GEMM (gemm_h3.hip: three fp16 matrix-core products
// per k-step of row-scaled 22-bit operands, fp32 accumulate -- the arithmetic of the ViT block GEMMs, 2.5-3x the fp32-MFMA
// rate, as accurate as an fp32 GEMM).  Queries are quantised once per call, every database panel once per panel
// (split_h2_wide: two reads + one write of the panel, a few % of its GEMM; the rows' sums of squares for F.normalize / L2
// come out of the same pass).  An operand image must stay inside 2 GiB of buffer addressing: rows per image <=
// (2^31 - 1) / (64 * dim / 16).  Option topk_h3: -1 (default) = where it pays (>= 256 queries, dim >= 1024, >= 2048 rows),
// 0 = never, 1 = wherever the shape allows (tests).
constexpr int64_t H3_PANEL = 8192;
int64_t h3_rows_limit(int64_t dim) { return ((1ll << 31) - 1) / (4 * dim) / 256 * 256; }
bool h3_scores(int64_t nq, int64_t ndb, int64_t dim) {
  const int64_t mode = option(OPT_TOPK_H3);
  if (mode == 0 || nq <= 64 || dim % 16 != 0 || h3_rows_limit(dim) < 256) return false;
  return mode > 0 || (nq >= 256 && dim >= 1024 && ndb >= 2048);
}

// Screened search (scores_screen.hip): the leading-plane scores of a whole range of database columns [nq, sc_cols], the
// screened running lists, the per-query margin, the candidates and their re-scored values
constexpr int SCREEN_CMAX = 512;          // candidates per query and column range; more: the call re-runs unscreened
constexpr int SCREEN_KMAX = 128;
constexpr int64_t SCREEN_COLS = 131072;   // database columns per range (a multiple of the panel)
constexpr size_t SCREEN_SBUF_MAX = 12ull << 30;
struct TopkWs {
  float *sbuf, *scr_v, *margin, *cand_v, *rho_q, *rho_d, *resid;
  long long* scr_i;
  int *cand, *count, *overflow;
  int64_t sc_cols;                 // 0: no screened search for this shape
  float *scores, *qn, *dn, *dss, *dnorm, *part, *rsq_part;
  unsigned char *qimg, *dimg;      // h3 path: operand images of the queries (per chunk of q_chunk rows) and of one panel
  float *qinv, *dinv;
  int64_t panel, q_chunk;
  size_t bytes;
};
// A prepared database (anyloc_topk_index_build: faiss' index.add): per panel of index_panel(dim) rows the two-plane fp16 image
// the score GEMM reads, then the rows' 2^-e and their raw sums of squares.  Layout inside the caller's buffer:
//   [n_panels][align256(h2_bytes(panel, dim))] images (a shorter last panel: an image of its own row count at its slot)
//   [ndb] float 2^-e      [ndb] float sum of squares      [ndb] float relative residual norm (ABI 9)
int64_t index_panel(int64_t dim) { return std::min(H3_PANEL, h3_rows_limit(dim)); }
bool index_supported(int64_t ndb, int64_t dim) { return ndb > 0 && dim % 16 == 0 && dim >= 16 && h3_rows_limit(dim) >= 256; }
struct IndexView {
  unsigned char* img;
  float *dinv, *dss, *drho;        // drho (ABI 9): |row - leading plane| / |row|, the screened search's bound (scores_screen.hip)
  int64_t panel;
  size_t slot, bytes;
};
IndexView index_view(void* p, int64_t ndb, int64_t dim) {
  IndexView v;
  v.panel = index_panel(dim);
  const int64_t np = (ndb + v.panel - 1) / v.panel;
  v.slot = align_up(h2_bytes(v.panel, dim), 256);
  unsigned char* b = static_cast<unsigned char*>(p);
  v.img = b;
  v.dinv = reinterpret_cast<float*>(b + (size_t)np * v.slot);
  v.dss = v.dinv + align_up((size_t)ndb, 64);
  v.drho = v.dss + align_up((size_t)ndb, 64);
  v.bytes = (size_t)np * v.slot + 3 * align_up((size_t)ndb, 64) * sizeof(float);
  return v;
}

// Shapes the screened search serves (option topk_screen: 0 = never, 1 = wherever the shape allows, -1 (default) = where it
// pays: >= 256 queries against >= 16 384 rows of >= 4096 columns): the h3 score panels' shapes with k <= 128, rows the re-scoring kernel holds
// in registers, and a score buffer of at most 12 GiB (fewer columns per range for more queries)
int64_t screen_cols(int64_t nq, int64_t ndb, int64_t dim, int64_t k, bool h3) {
  const int64_t mode = option(OPT_TOPK_SCREEN);
  if (!h3 || mode == 0 || k > SCREEN_KMAX || k <= 0 || !screen_rescore_supported(dim) || nq <= 0 || ndb <= 0) return 0;
  if (mode < 0 && !(nq >= 256 && ndb >= 16384 && dim >= 4096)) return 0;
  const int64_t panel = std::min(H3_PANEL, h3_rows_limit(dim));
  int64_t cols = std::min<int64_t>(SCREEN_COLS, (ndb + panel - 1) / panel * panel);
  const int64_t fit = (int64_t)(SCREEN_SBUF_MAX / 4) / nq / panel * panel;
  cols = std::min(cols, fit);
  return cols >= panel ? cols : 0;
}

TopkWs carve(void* ws, size_t cap, int64_t nq, int64_t ndb, int64_t dim, bool indexed = false, int64_t k = 0) {
  Arena a(ws, cap);
  TopkWs w;
  const bool h3 = indexed || h3_scores(nq, ndb, dim);
  w.panel = h3 ? std::min(H3_PANEL, h3_rows_limit(dim)) : PANEL;
  w.q_chunk = h3 ? std::min<int64_t>(nq, h3_rows_limit(dim)) : nq;
  const int64_t panel = std::min<int64_t>(w.panel, std::max<int64_t>(ndb, 1));
  w.scores = a.take<float>(std::max<int64_t>(nq, 1) * panel);
  const int64_t n_qchunks = h3 ? (nq + w.q_chunk - 1) / w.q_chunk : 0;
  // (few-query fp16 path: the queries' pre-split planes, 10 KiB per 32-k slab)
  w.qimg = a.take<unsigned char>(h3 ? (size_t)n_qchunks * h2_bytes(w.q_chunk, dim) : few_queries(nq, dim) ? fewq_query_image_bytes(dim) : 1);
  w.dimg = a.take<unsigned char>(h3 && !indexed ? h2_bytes(panel, dim) : 1);
  w.qinv = a.take<float>(h3 ? nq : 64);                  // (few-query fp16 path: the <= 64 queries' row scales)
  w.dinv = a.take<float>(h3 && !indexed ? panel : 1);
  w.qn = a.take<float>(std::max<int64_t>(nq, 1));
  w.dn = a.take<float>(std::max<int64_t>(ndb, 1));
  w.dss = a.take<float>(std::max<int64_t>(ndb, 1));
  w.dnorm = a.take<float>(std::max<int64_t>(ndb, 1));
  const bool few = !indexed && few_queries(nq, dim);
  w.part = a.take<float>(few ? (size_t)SPLITK_MAX * panel * 64 : 1);
  w.rsq_part = a.take<float>(few ? (size_t)SPLITK_MAX * panel : 1);
  w.sc_cols = screen_cols(nq, ndb, dim, k, h3);
  const bool sc = w.sc_cols > 0;
  w.sbuf = a.take<float>(sc ? (size_t)nq * w.sc_cols : 1);
  w.scr_v = a.take<float>(sc ? (size_t)nq * k : 1);
  w.scr_i = a.take<long long>(sc ? (size_t)nq * k : 1);
  w.margin = a.take<float>(sc ? (size_t)nq : 1);
  w.cand = a.take<int>(sc ? (size_t)nq * SCREEN_CMAX : 1);
  w.cand_v = a.take<float>(sc ? (size_t)nq * SCREEN_CMAX : 1);
  w.count = a.take<int>(sc ? (size_t)nq : 1);
  w.rho_q = a.take<float>(sc ? (size_t)nq : 1);
  w.rho_d = a.take<float>(sc && !indexed ? (size_t)std::max<int64_t>(ndb, 1) : 1);
  w.resid = a.take<float>(sc && !indexed ? (size_t)panel : 1);
  w.overflow = a.take<int>(4);                            // [0] overflow flag, [1] bits of the largest database rho, [2] of the largest raw sum of squares
  w.bytes = a.off;
  return w;
}

}  // namespace
}  // namespace anyloc

using namespace anyloc;

extern "C" {

size_t anyloc_topk_workspace_bytes(int64_t nq, int64_t ndb, int64_t dim, int64_t k) {
  return carve(nullptr, 0, nq, ndb, dim, false, k).bytes + 256;
}

}  // extern "C"

// the search; `index` != nullptr: the database side comes from a prepared index (db may be null), every query count runs
// on the fp16 score panels
static int topk_impl(const float* queries, int64_t nq, const float* db, int64_t ndb, int64_t dim, int64_t k, int metric,
                     unsigned flags, int64_t index_base, float* dist, int64_t* idx, void* workspace, size_t workspace_bytes,
                     const void* index, hipStream_t stream, bool allow_screen = true) {
  ANYLOC_CHECK_ARG(nq >= 0 && ndb >= 0, "topk: negative size");
  if (nq == 0 || k == 0) return ANYLOC_OK;
  ANYLOC_CHECK_ARG(queries && dist && idx, "topk: null pointer");
  ANYLOC_CHECK_ARG(db || ndb == 0 || index, "topk: null database");
  const bool indexed = index != nullptr;
  const IndexView iv = indexed ? index_view(const_cast<void*>(index), ndb, dim) : IndexView{};
  ANYLOC_CHECK_ARG(k >= 1 && k <= KMAX, "topk: k=%lld outside [1,%d]", (long long)k, KMAX);
  ANYLOC_CHECK_ARG(metric == 0 || metric == 1, "topk: metric %d", metric);
  ANYLOC_CHECK_ARG(dim >= 4 && dim % 4 == 0, "topk: dim %lld must be a positive multiple of 4", (long long)dim);
  ANYLOC_CHECK_ARG(nq < (1ll << 31), "topk: too many queries");
  ANYLOC_CHECK_ARG((flags & ~ANYLOC_TOPK_NORMALIZE_DB) == 0, "topk: unknown flags %u", flags);
  TopkWs w = carve(workspace, workspace_bytes, nq, ndb, dim, indexed, k);
  const bool norm_db = (flags & ANYLOC_TOPK_NORMALIZE_DB) != 0;
  const bool few = !indexed && few_queries(nq, dim);
  const bool h3 = indexed || h3_scores(nq, ndb, dim);
  const int64_t PANEL_ROWS = w.panel;
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("topk: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  static_assert(PANEL <= 0x8000 && H3_PANEL <= PANEL && CAP + KMAX + 2 <= 0xffff,
                "merge keys hold the column in 15 bits and the slot in 16");
  const size_t k2 = (size_t)((k + 1) & ~1ll);
  const size_t lds = 16 * (k2 + CAP) + 12 * k2 + 16;      // 37 KiB at k = 20: four blocks per CU
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(topk_merge_kernel), (int)(16 * (KMAX + CAP) + 12 * KMAX + 16)));
  // screened search (scores_screen.hip): the panels are scored on the leading fp16 planes alone, the rows their error bound
  // cannot rule out are re-scored exactly from the fp32 rows -- needs those rows (a prepared index WITH its rows:
  // anyloc_topk_search_index_rows).  The bound of a query is one number: its own norm and residual x the LARGEST row norm the
  // compared value sees -- 1 with ANYLOC_TOPK_NORMALIZE_DB, the largest raw row norm of the database without it
  const bool screen = allow_screen && h3 && w.sc_cols > 0 && db != nullptr && ndb > 0;
  if (metric == 1 || screen) {
    hipLaunchKernelGGL(rownorm_sq_kernel, dim3((unsigned)nq), dim3(256), 0, stream, queries, dim, w.qn);
    ANYLOC_TRY(launch_status("rownorm_sq_kernel(q)"));
  }
  if ((metric == 1 || norm_db) && !few && !h3) {   // (the few-query and fp16 paths get the rows' sums of squares from their own pass)
    ProfScope prof("topk_db_norms", stream, 2.0 * ndb * dim, 4.0 * ndb * dim);
    float* ss = norm_db ? w.dss : w.dn;
    for (int64_t r0 = 0; r0 < ndb; r0 += (1ll << 30)) {
      const int64_t cnt = std::min<int64_t>(1ll << 30, ndb - r0);
      hipLaunchKernelGGL(rownorm_sq_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, db + r0 * dim, dim, ss + r0);
      ANYLOC_TRY(launch_status("rownorm_sq_kernel(db)"));
    }
    if (norm_db && ndb > 0) {
      hipLaunchKernelGGL(dbnorm_kernel, dim3((unsigned)((ndb + 255) / 256)), dim3(256), 0, stream, w.dss, ndb, w.dnorm, w.dn);
      ANYLOC_TRY(launch_status("dbnorm_kernel"));
    }
  }
  long long* idx_ll = reinterpret_cast<long long*>(idx);
  int first = 1;
  if (ndb == 0) {
    // nothing to search: emit the padding list
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)nq), dim3(256), lds, stream, w.scores, (int64_t)0,
                       (int64_t)0, index_base, (int)k, metric, w.qn, w.dn, (const float*)nullptr, dist, idx_ll, 1);
    ANYLOC_TRY(launch_status("topk_merge_kernel"));
  }
  if (h3 && ndb > 0)
    for (int64_t q0 = 0, c = 0; q0 < nq; q0 += w.q_chunk, ++c) {
      const int64_t qc = std::min<int64_t>(w.q_chunk, nq - q0);
      ANYLOC_TRY(split_h2_wide(queries + q0 * dim, dim, qc, dim, w.qimg + c * h2_bytes(w.q_chunk, dim), w.qinv + q0, nullptr, stream));
    }
  if (screen) {
    const int64_t K16 = dim / 16;
    const int64_t KC16 = 1536;                               // k-blocks per accumulated chunk (the bound's accumulation term)
    const int nchunks = (int)((K16 + KC16 - 1) / KC16);
    ANYLOC_HIP(hipMemsetAsync(w.overflow, 0, 3 * sizeof(int), stream));
    unsigned* rho_max = reinterpret_cast<unsigned*>(w.overflow + 1);
    unsigned* ss_max = reinterpret_cast<unsigned*>(w.overflow + 2);   // bits of the largest raw sum of squares (searches without NORMALIZE_DB)
    const float* dnorm_all = norm_db ? w.dnorm : nullptr;             // the divisor array of the compared value, or none
    // the queries' relative residual norms, from the residual planes of their images
    for (int64_t q0 = 0, c = 0; q0 < nq; q0 += w.q_chunk, ++c) {
      const int64_t qc = std::min<int64_t>(w.q_chunk, nq - q0);
      ANYLOC_TRY(screen_resid(w.qimg + c * h2_bytes(w.q_chunk, dim), qc, (int)K16, qc, w.qinv + q0, w.qn + q0, w.rho_q + q0, nullptr, stream));
    }
    int first_exact = 1;
    for (int64_t s0 = 0; s0 < ndb; s0 += w.sc_cols) {
      const int64_t sn = std::min<int64_t>(w.sc_cols, ndb - s0);
      int first_scr = 1;
      for (int64_t c0 = s0; c0 < s0 + sn; c0 += PANEL_ROWS) {
        const int64_t pc = std::min<int64_t>(PANEL_ROWS, s0 + sn - c0);
        const unsigned char* dimg = w.dimg;
        const float* dinv = w.dinv;
        const float* dss = w.dss + c0;
        if (indexed) {
          dimg = iv.img + (size_t)(c0 / iv.panel) * iv.slot;
          dinv = iv.dinv + c0;
          dss = iv.dss + c0;
        } else {
          // leading plane only; the residual norms come out of the quantiser itself
          ANYLOC_HIP(hipMemsetAsync(w.resid, 0, (size_t)pc * sizeof(float), stream));
          ANYLOC_TRY(split_h1_wide(db + c0 * dim, dim, pc, dim, w.dimg, w.dinv, w.dss + c0, w.resid, stream));
          ANYLOC_TRY(screen_rho_from_resid(w.resid, w.dinv, w.dss + c0, pc, w.rho_d + c0, rho_max, stream));
        }
        if (norm_db) {
          hipLaunchKernelGGL(dbnorm_kernel, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, stream, dss, pc, w.dnorm + c0, w.dn + c0);
          ANYLOC_TRY(launch_status("dbnorm_kernel"));
        } else {
          // raw rows: the L2 term is the raw sum of squares; the bound scales with the largest raw norm seen so far
          if (dss != w.dn + c0) ANYLOC_HIP(hipMemcpyAsync(w.dn + c0, dss, (size_t)pc * sizeof(float), hipMemcpyDeviceToDevice, stream));
          ANYLOC_TRY(screen_rho_max(dss, pc, ss_max, stream));
        }
        for (int64_t q0 = 0, c = 0; q0 < nq; q0 += w.q_chunk, ++c) {
          const int64_t qc = std::min<int64_t>(w.q_chunk, nq - q0);
          for (int64_t kb0 = 0; kb0 < K16; kb0 += KC16) {
            H3Problem h{};
            h.A2 = w.qimg + c * h2_bytes(w.q_chunk, dim) + kb0 * (2 * qc * 32); h.RA = qc; h.a_inv = w.qinv + q0;
            h.W2 = dimg + kb0 * (2 * pc * 32); h.RW = pc; h.w_inv = dinv;
            h.C = w.sbuf + q0 * sn + (c0 - s0); h.ldc = sn;
            h.M = qc; h.N = pc; h.K16 = (int)std::min<int64_t>(KC16, K16 - kb0);
            h.accumulate = kb0 > 0;
            h.tag = "topk_screen_gemm";
            ANYLOC_TRY(gemm_screen(h, stream));
          }
        }
        {
          ProfScope prof("topk_merge", stream, 0.0, 4.0 * nq * pc);
          hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)nq), dim3(256), lds, stream, w.sbuf + (c0 - s0), sn, pc, index_base + c0,
                             (int)k, metric, w.qn, w.dn + c0, dnorm_all ? dnorm_all + c0 : (const float*)nullptr, w.scr_v, w.scr_i, first_scr);
          ANYLOC_TRY(launch_status("topk_merge_kernel"));
        }
        first_scr = 0;
      }
      if (indexed) ANYLOC_TRY(screen_rho_max(iv.drho + s0, sn, rho_max, stream));
      ANYLOC_TRY(screen_margins(w.qn, w.rho_q, rho_max, norm_db ? nullptr : ss_max, nq, metric,
                                screen_accum((int)std::min(KC16, K16), nchunks), w.margin, stream));
      const float* dnorm_s = dnorm_all ? dnorm_all + s0 : nullptr;
      ANYLOC_TRY(screen_compact(w.sbuf, sn, sn, nq, (int)k, metric, w.qn, w.dn + s0, dnorm_s, w.scr_v, w.margin, SCREEN_CMAX, w.cand,
                                w.count, w.overflow, stream));
      ANYLOC_TRY(screen_rescore(queries, db + s0 * dim, dim, nq, SCREEN_CMAX, w.cand, w.count, metric, w.qn, w.dn + s0, dnorm_s,
                                w.cand_v, stream));
      ANYLOC_TRY(screen_select(w.cand, w.cand_v, w.count, SCREEN_CMAX, index_base + s0, nq, (int)k, dist, idx, first_exact, stream));
      first_exact = 0;
    }
    int over = 0;
    ANYLOC_HIP(hipMemcpyAsync(&over, w.overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
    ANYLOC_HIP(hipStreamSynchronize(stream));
    if (over)   // some query has more candidates than SCREEN_CMAX inside its bound (near-duplicate rows): the unscreened search
      return topk_impl(queries, nq, db, ndb, dim, k, metric, flags, index_base, dist, idx, workspace, workspace_bytes, index, stream, false);
    hipLaunchKernelGGL(topk_finish_kernel, dim3((unsigned)((nq * k + 255) / 256)), dim3(256), 0, stream, dist, idx_ll, nq * k, metric);
    return launch_status("topk_finish_kernel");
  }
  for (int64_t c0 = 0; c0 < ndb; c0 += PANEL_ROWS) {
    const int64_t pc = std::min<int64_t>(PANEL_ROWS, ndb - c0);
    GemmProblem g{};
    g.tag = "topk_scores_gemm";
    const float* dn_panel = w.dn + c0;                       // the merge kernel's squared-norm term of the panel's rows (L2)
    if (h3) {
      // the panel's operand image + its rows' sums of squares (raw: the L2 term; or, normalising, the F.normalize divisor):
      // quantised here, or -- prepared index -- read where anyloc_topk_index_build left them
      const bool want_ss = metric == 1 || norm_db;
      const unsigned char* dimg = w.dimg;
      const float* dinv = w.dinv;
      const float* dss = w.dss + c0;
      if (indexed) {
        dimg = iv.img + (size_t)(c0 / iv.panel) * iv.slot;
        dinv = iv.dinv + c0;
        dss = iv.dss + c0;
        if (metric == 1 && !norm_db) dn_panel = dss;
      } else {
        ANYLOC_TRY(split_h2_wide(db + c0 * dim, dim, pc, dim, w.dimg, w.dinv, want_ss ? (norm_db ? w.dss : w.dn) + c0 : nullptr, stream));
      }
      if (norm_db) {
        hipLaunchKernelGGL(dbnorm_kernel, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, stream, dss, pc, w.dnorm + c0,
                           w.dn + c0);
        ANYLOC_TRY(launch_status("dbnorm_kernel"));
      }
      // A 49 152-long contraction in ONE fp32 accumulator takes ~9 000 rounded additions: 4e-6 on a score of 1.  Cut into
      // chunks of 8192 k (one launch each, the later ones adding into the panel) the error is that of ~1 500 additions plus
      // six: 5e-7 -- the chunked summation of the fp32-MFMA path (gemm_f32.hip, ABL bit 5) at the price of re-reading and
      // re-writing the score panel per chunk (3 % of the GEMM time).
      const int64_t K16 = dim / 16, KC16 = 512;
      for (int64_t q0 = 0, c = 0; q0 < nq; q0 += w.q_chunk, ++c) {
        const int64_t qc = std::min<int64_t>(w.q_chunk, nq - q0);
        for (int64_t kb0 = 0; kb0 < K16; kb0 += KC16) {
          H3Problem h{};
          h.A2 = w.qimg + c * h2_bytes(w.q_chunk, dim) + kb0 * (2 * qc * 32); h.RA = qc; h.a_inv = w.qinv + q0;
          h.W2 = dimg + kb0 * (2 * pc * 32); h.RW = pc; h.w_inv = dinv;
          h.C = w.scores + q0 * pc; h.ldc = pc;
          h.M = qc; h.N = pc; h.K16 = (int)std::min<int64_t>(KC16, K16 - kb0);
          h.accumulate = kb0 > 0;
          h.tag = "topk_scores_gemm";
          ANYLOC_TRY(gemm_h3(h, EPI_STORE, stream));
        }
      }
    } else if (few) {
      // database rows as the M operand, the (<= 64) queries as N, K cut into slices: enough workgroups to stream the
      // panel at HBM rate; the row sums of squares of the database come out of the same pass
      const int S = choose_ksplit(pc, dim);
      g.A = db + c0 * dim; g.lda = dim;
      g.W = queries; g.ldw = dim;
      g.C = w.part; g.ldc = 64;
      g.M = pc; g.N = nq; g.K = dim / S;
      g.ksplit = S; g.c_split_stride = pc * 64;
      g.rowsq = w.rsq_part;
      const int64_t fewq = option(OPT_TOPK_FEWQ_X6);
      if (fewq == 2) {
        const bool qdma = option(OPT_TOPK_FEWQ_QDMA) != 0;
        if (c0 == 0) {                                                                 // once per call: the queries' row scales and planes
          ANYLOC_TRY(row_scales_h2(queries, dim, nq, dim, w.qinv, nullptr, stream));
          if (qdma) ANYLOC_TRY(fewq_query_image(queries, dim, nq, w.qinv, dim, w.qimg, stream));
        }
        ANYLOC_TRY(scores_fewq_h3(g.A, g.lda, pc, queries, dim, nq, w.qinv, qdma ? w.qimg : nullptr, g.K, S, w.part, w.rsq_part, stream));
      } else if (fewq != 0)
        ANYLOC_TRY(scores_fewq_x6(g.A, g.lda, pc, queries, dim, nq, g.K, S, w.part, w.rsq_part, stream));
      else
        ANYLOC_TRY(gemm_nt_splitk(g, stream));
      ProfScope prof("topk_combine", stream, (double)S * pc * 64, 4.0 * ((double)S * pc * 65 + (double)nq * pc));
      hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, stream, w.part, w.rsq_part, S,
                         pc, (int)nq, w.scores, (norm_db ? w.dss : w.dn) + c0);
      ANYLOC_TRY(launch_status("splitk_combine_kernel"));
      if (norm_db) {
        hipLaunchKernelGGL(dbnorm_kernel, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, stream, w.dss + c0, pc, w.dnorm + c0,
                           w.dn + c0);
        ANYLOC_TRY(launch_status("dbnorm_kernel"));
      }
    } else {
      g.A = queries; g.lda = dim;
      g.W = db + c0 * dim; g.ldw = dim;
      g.C = w.scores; g.ldc = pc;
      g.M = nq; g.N = pc; g.K = dim;
      ANYLOC_TRY(gemm_nt(g, EPI_STORE, stream));
    }
    {
      ProfScope prof("topk_merge", stream, 0.0, 4.0 * nq * pc);
      hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)nq), dim3(256), lds, stream, w.scores, pc, pc,
                         index_base + c0, (int)k, metric, w.qn, dn_panel, norm_db ? w.dnorm + c0 : (const float*)nullptr, dist,
                         idx_ll, first);
      ANYLOC_TRY(launch_status("topk_merge_kernel"));
    }
    first = 0;
  }
  // padding entries carry index -1 regardless of index_base (faiss); L2 distances are sign-flipped back
  hipLaunchKernelGGL(topk_finish_kernel, dim3((unsigned)((nq * k + 255) / 256)), dim3(256), 0, stream, dist, idx_ll,
                     nq * k, metric);
  return launch_status("topk_finish_kernel");
}

extern "C" {

int anyloc_topk(const float* queries, int64_t nq, const float* db, int64_t ndb, int64_t dim, int64_t k, int metric,
                unsigned flags, int64_t index_base, float* dist, int64_t* idx, void* workspace, size_t workspace_bytes,
                void* stream) {
  return topk_impl(queries, nq, db, ndb, dim, k, metric, flags, index_base, dist, idx, workspace, workspace_bytes, nullptr,
                   static_cast<hipStream_t>(stream));
}

int anyloc_topk_path(int64_t nq, int64_t ndb, int64_t dim) {
  if (nq <= 0 || ndb < 0 || dim < 4 || dim % 4) return -1;
  return h3_scores(nq, ndb, dim) ? 2 : few_queries(nq, dim) ? 1 : 0;
}

size_t anyloc_topk_index_bytes(int64_t ndb, int64_t dim) {
  return index_supported(ndb, dim) ? index_view(nullptr, ndb, dim).bytes + 256 : 0;
}

int anyloc_topk_index_build(const float* db, int64_t ndb, int64_t dim, void* index, size_t index_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(db && index && ndb > 0, "topk_index_build: null pointer / empty database");
  if (!index_supported(ndb, dim)) {
    set_error("topk_index_build: dim %lld is not served by the fp16 score panels", (long long)dim);
    return ANYLOC_ERR_UNSUPPORTED;
  }
  const IndexView iv = index_view(index, ndb, dim);
  if (iv.bytes > index_bytes) {
    set_error("topk_index_build: index buffer %zu < %zu", index_bytes, iv.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  for (int64_t c0 = 0; c0 < ndb; c0 += iv.panel) {
    const int64_t pc = std::min<int64_t>(iv.panel, ndb - c0);
    ANYLOC_TRY(split_h2_wide(db + c0 * dim, dim, pc, dim, iv.img + (size_t)(c0 / iv.panel) * iv.slot, iv.dinv + c0, iv.dss + c0, stream));
    ANYLOC_TRY(screen_resid(iv.img + (size_t)(c0 / iv.panel) * iv.slot, pc, (int)(dim / 16), pc, iv.dinv + c0, iv.dss + c0, iv.drho + c0, nullptr, stream));
  }
  return ANYLOC_OK;
}

size_t anyloc_topk_index_workspace_bytes(int64_t nq, int64_t ndb, int64_t dim, int64_t k) {
  return index_supported(ndb, dim) ? carve(nullptr, 0, nq, ndb, dim, true, k).bytes + 256 : 0;
}

int anyloc_topk_search_index(const float* queries, int64_t nq, const void* index, int64_t ndb, int64_t dim, int64_t k, int metric,
                             unsigned flags, int64_t index_base, float* dist, int64_t* idx, void* workspace,
                             size_t workspace_bytes, void* stream) {
  ANYLOC_CHECK_ARG(index && index_supported(ndb, dim), "topk_search_index: no index / shape not served by the fp16 score panels");
  return topk_impl(queries, nq, nullptr, ndb, dim, k, metric, flags, index_base, dist, idx, workspace, workspace_bytes, index,
                   static_cast<hipStream_t>(stream));
}

int anyloc_topk_search_index_rows(const float* queries, int64_t nq, const float* db, const void* index, int64_t ndb, int64_t dim,
                                  int64_t k, int metric, unsigned flags, int64_t index_base, float* dist, int64_t* idx, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  ANYLOC_CHECK_ARG(index && index_supported(ndb, dim), "topk_search_index_rows: no index / shape not served by the fp16 score panels");
  return topk_impl(queries, nq, db, ndb, dim, k, metric, flags, index_base, dist, idx, workspace, workspace_bytes, index,
                   static_cast<hipStream_t>(stream));
}

}  // extern "C"
