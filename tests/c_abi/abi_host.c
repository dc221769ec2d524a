/*
 * abi_host.c -- a host program in plain C (no Python, no torch) that drives libanyloc_hip.so through
 * include/anyloc_hip.h exactly as a non-Python caller would: hipMalloc'd buffers, the *_workspace_bytes queries,
 * a caller-owned stream, status codes + anyloc_last_error().  It is the proof that the drop-in boundary is the C ABI
 * and not the ctypes layer above it.  Every result is compared with the C restatement of the reference's arithmetic
 * (oracle/c/anyloc_oracle.c -- the checker, linked only into this test binary).
 *
 * Cases (reference lines: see the entry points in include/anyloc_hip.h):
 *   vlad     hard-assignment VLAD of packed ragged images incl. an empty one, at the headline width (D = 1536,
 *            K = 32: the fused one-pass kernel) and at an odd width (D = 100, K = 5: the general kernels), with and
 *            without re-normalisation / intra-normalisation, cosine and euclidean labels
 *   kmeans   one anyloc_kmeans_step + anyloc_kmeans_update against one fpk iteration
 *   topk     anyloc_l2norm_rows + anyloc_topk(NORMALIZE_DB): a few queries of long rows (the split-K stream over the
 *            database) and many queries of short rows (score panels), IP and L2, k > ndb padding, index_base; two of them a
 *            second time through anyloc_topk_index_build + anyloc_topk_search_index (ABI 8: the database side prepared once)
 *   vit      anyloc_vit_create / anyloc_vit_forward (two taps in one call) on GELU-mlp and SwiGLU models, on the fp32 matrix-core
 *            kernels and with anyloc_split_h2 images attached (the default two-term fp16 arithmetic), against the C restatement
 *            of the hub model
 *   errors   a too-small workspace and a null pointer come back as status codes with a message, nothing crashes
 *
 * Exit code 0 = all cases agree, 1 = a mismatch or an unexpected status, 77 = no HIP device (the product has no CPU
 * path: the program says so and stops).  One line per case on stdout, a final JSON line.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "anyloc_hip.h"
#include "anyloc_oracle.h"

static int failures = 0;

#define HIP_OK(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

#define ANYLOC_OK_OR_FAIL(call)                                                                      \
  do {                                                                                               \
    int s_ = (call);                                                                                 \
    if (s_ != ANYLOC_OK) {                                                                           \
      fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, s_, anyloc_last_error()); \
      exit(1);                                                                                       \
    }                                                                                                \
  } while (0)

/* ---- seeded data (SplitMix64 + Box-Muller): the same bytes on every run and every machine ---- */
static uint64_t rng_state;
static uint64_t rng_u64(void) {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double rng_uniform(void) { return ((double)(rng_u64() >> 11) + 0.5) / 9007199254740992.0; }
static float rng_normal(void) {
  return (float)(sqrt(-2.0 * log(rng_uniform())) * cos(6.283185307179586 * rng_uniform()));
}

static void* dev_alloc(size_t bytes) {
  void* p = NULL;
  HIP_OK(hipMalloc(&p, bytes ? bytes : 16));
  return p;
}
static void* dev_from(const void* host, size_t bytes) {
  void* p = dev_alloc(bytes);
  if (bytes) HIP_OK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
  return p;
}

static void report(const char* name, int ok, const char* detail) {
  printf("%-44s %s  %s\n", name, ok ? "ok  " : "FAIL", detail);
  if (!ok) ++failures;
}

/* tokens scattered around `modes` random directions (what patch descriptors look like to k-means), rows scaled over
   three decades so that "tokens as passed" and "re-normalised tokens" differ */
static void clustered_rows(float* x, int64_t n, int64_t D, int modes, float noise, int scale_rows) {
  float* m = (float*)malloc(sizeof(float) * (size_t)(modes * D));
  for (int64_t i = 0; i < modes * D; ++i) m[i] = rng_normal();
  for (int64_t r = 0; r < n; ++r) {
    const float* c = m + (int64_t)(rng_u64() % (uint64_t)modes) * D;
    float s = scale_rows ? (float)pow(10.0, 3.0 * rng_uniform() - 1.5) : 1.0f;
    for (int64_t j = 0; j < D; ++j) x[r * D + j] = s * (c[j] + noise * rng_normal());
  }
  free(m);
}

/* ------------------------------------------------------------------ VLAD */
static void case_vlad(hipStream_t stream, int64_t D, int64_t K, unsigned flags, const int64_t* offsets, int64_t n_img,
                      const char* name) {
  const int64_t total = offsets[n_img];
  float* tok = (float*)malloc(sizeof(float) * (size_t)(total * D));
  float* cen = (float*)malloc(sizeof(float) * (size_t)(K * D));
  clustered_rows(tok, total, D, (int)K + 3, 0.7f, 1);
  for (int64_t k = 0; k < K; ++k) {            /* centres = scaled tokens: raw (un-normalised) centroids as k-means leaves them */
    int64_t r = (int64_t)(rng_u64() % (uint64_t)total);
    for (int64_t j = 0; j < D; ++j) cen[k * D + j] = 0.6f * tok[r * D + j] + 0.05f * rng_normal();
  }
  float* want = (float*)malloc(sizeof(float) * (size_t)(n_img * K * D));
  int64_t* want_lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)total);
  double* gap = (double*)malloc(sizeof(double) * (size_t)total);
  oracle_vlad_hard(tok, offsets, n_img, D, cen, K, flags, want, want_lab, gap);

  float* d_tok = (float*)dev_from(tok, sizeof(float) * (size_t)(total * D));
  float* d_cen = (float*)dev_from(cen, sizeof(float) * (size_t)(K * D));
  int64_t* d_off = (int64_t*)dev_from(offsets, sizeof(int64_t) * (size_t)(n_img + 1));
  float* d_out = (float*)dev_alloc(sizeof(float) * (size_t)(n_img * K * D));
  int64_t* d_lab = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)total);
  size_t ws_bytes = anyloc_vlad_workspace_bytes(total, n_img, D, K);
  void* d_ws = dev_alloc(ws_bytes);
  HIP_OK(hipMemsetAsync(d_out, 0xff, sizeof(float) * (size_t)(n_img * K * D), stream));   /* NaN pattern: every element must be written */
  ANYLOC_OK_OR_FAIL(anyloc_vlad_hard(d_tok, d_off, n_img, total, D, d_cen, K, flags, d_out, d_lab, d_ws, ws_bytes, stream));
  float* got = (float*)malloc(sizeof(float) * (size_t)(n_img * K * D));
  int64_t* got_lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)total);
  HIP_OK(hipMemcpyAsync(got, d_out, sizeof(float) * (size_t)(n_img * K * D), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(got_lab, d_lab, sizeof(int64_t) * (size_t)total, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  /* labels: identical, except where the exact top-2 similarity gap is below what fp32 resolves (cosine scores are
     O(1): 1e-6; euclidean similarities scale with the squared row norms) */
  int64_t flips = 0, bad_flips = 0;
  double worst = 0.0;
  int ok = 1;
  for (int64_t t = 0; t < total; ++t) {
    if (got_lab[t] < 0 || got_lab[t] >= K) { ok = 0; got_lab[t] = 0; continue; }
    if (got_lab[t] == want_lab[t]) continue;
    ++flips;
    double bar = 1e-6;
    if (flags & ANYLOC_VLAD_EUCLIDEAN) {
      double ss = 0.0;
      for (int64_t j = 0; j < D; ++j) ss += (double)tok[t * D + j] * tok[t * D + j];
      bar = 1e-5 * (ss > 1.0 ? ss : 1.0);
    }
    if (gap[t] > bar) ++bad_flips;
  }
  /* descriptors: against the exact VLAD of the assignment the library reported (equal to `want` when nothing flipped),
     so a legitimate near-tie does not hide the sums from the check */
  if (flips) oracle_vlad_assigned(tok, offsets, n_img, D, cen, K, flags, got_lab, want);
  for (int64_t im = 0; im < n_img; ++im) {
    double num = 0.0, den = 0.0;
    for (int64_t j = 0; j < K * D; ++j) {
      double g = got[im * K * D + j], w = want[im * K * D + j];
      if (!(g == g)) ok = 0;                   /* NaN: an element the launch never wrote, or a bad norm */
      num += (g - w) * (g - w);
      den += w * w;
    }
    double rel = den > 0.0 ? sqrt(num / den) : sqrt(num);   /* an empty image: the zero vector, exactly */
    if (!(rel <= worst)) worst = rel;
  }
  ok = ok && bad_flips == 0 && worst <= 1e-5;
  char detail[160];
  snprintf(detail, sizeof detail, "%lld images, %lld tokens: %lld label flips (%lld outside the gap bar), worst rel err %.2e",
           (long long)n_img, (long long)total, (long long)flips, (long long)bad_flips, worst);
  report(name, ok, detail);
  hipFree(d_tok); hipFree(d_cen); hipFree(d_off); hipFree(d_out); hipFree(d_lab); hipFree(d_ws);
  free(tok); free(cen); free(want); free(want_lab); free(gap); free(got); free(got_lab);
}

/* --------------------------------------------------------------- k-means */
static void case_kmeans(hipStream_t stream, int64_t n, int64_t D, int64_t K, int mode, const char* name) {
  float* x = (float*)malloc(sizeof(float) * (size_t)(n * D));
  clustered_rows(x, n, D, (int)K, 0.5f, 0);
  if (mode == 0) oracle_l2norm_rows(x, x, n, D);             /* VLAD.fit normalises the rows first (utilities.py:782) */
  float* cen = (float*)malloc(sizeof(float) * (size_t)(K * D));
  for (int64_t k = 0; k < K; ++k) memcpy(cen + k * D, x + (int64_t)(rng_u64() % (uint64_t)n) * D, sizeof(float) * (size_t)D);
  memcpy(cen + (K - 1) * D, cen, sizeof(float) * (size_t)D);  /* a duplicated initial row: the later copy ends up EMPTY -> centre 0 */
  float* want = (float*)malloc(sizeof(float) * (size_t)(K * D));
  int64_t* want_lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  double* want_cnt = (double*)malloc(sizeof(double) * (size_t)K);
  double want_err = oracle_kmeans_iteration(x, n, D, cen, K, mode, want, want_lab, want_cnt);
  double* gap = (double*)malloc(sizeof(double) * (size_t)n);
  oracle_fpk_labels(x, n, D, cen, K, mode, want_lab, gap);

  float* d_x = (float*)dev_from(x, sizeof(float) * (size_t)(n * D));
  float* d_cen = (float*)dev_from(cen, sizeof(float) * (size_t)(K * D));
  float* d_sums = (float*)dev_alloc(sizeof(float) * (size_t)(K * D));
  float* d_cnt = (float*)dev_alloc(sizeof(float) * (size_t)K);
  float* d_new = (float*)dev_alloc(sizeof(float) * (size_t)(K * D));
  double* d_err = (double*)dev_alloc(sizeof(double));
  int64_t* d_lab = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)n);
  size_t ws_bytes = anyloc_kmeans_workspace_bytes(n, D, K);
  void* d_ws = dev_alloc(ws_bytes);
  ANYLOC_OK_OR_FAIL(anyloc_kmeans_step(d_x, n, D, d_cen, K, mode, d_sums, d_cnt, d_lab, d_ws, ws_bytes, stream));
  ANYLOC_OK_OR_FAIL(anyloc_kmeans_update(d_sums, d_cnt, d_cen, K, D, d_new, d_err, stream));
  float* got = (float*)malloc(sizeof(float) * (size_t)(K * D));
  float* got_cnt = (float*)malloc(sizeof(float) * (size_t)K);
  int64_t* got_lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  double got_err = 0.0;
  HIP_OK(hipMemcpyAsync(got, d_new, sizeof(float) * (size_t)(K * D), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(got_cnt, d_cnt, sizeof(float) * (size_t)K, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(got_lab, d_lab, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(&got_err, d_err, sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int64_t flips = 0, bad = 0;
  int lab_ok = 1;
  for (int64_t i = 0; i < n; ++i) {
    if (got_lab[i] < 0 || got_lab[i] >= K) { lab_ok = 0; got_lab[i] = 0; continue; }
    if (got_lab[i] != want_lab[i]) { ++flips; if (gap[i] > (mode == 0 ? 1e-6 : 1e-4)) ++bad; }
  }
  /* sums, counts, centres and err against the exact update of the assignment the library reported */
  if (flips) want_err = oracle_kmeans_update_from_labels(x, n, D, cen, K, got_lab, want, want_cnt);
  double worst = 0.0;
  int counts_ok = lab_ok;
  for (int64_t k = 0; k < K; ++k) {
    if ((double)got_cnt[k] != want_cnt[k]) counts_ok = 0;
    for (int64_t j = 0; j < D; ++j) {
      double d = fabs((double)got[k * D + j] - (double)want[k * D + j]);
      if (!(d <= worst)) worst = d;
    }
  }
  int empty_ok = want_cnt[K - 1] == 0.0 && got_cnt[K - 1] == 0.0f;
  for (int64_t j = 0; j < D; ++j) empty_ok = empty_ok && got[(K - 1) * D + j] == 0.0f;
  int err_ok = fabs(got_err - want_err) <= 1e-5 * (want_err > 1.0 ? want_err : 1.0);
  int ok = bad == 0 && counts_ok && empty_ok && err_ok && worst <= 1e-5;
  char detail[200];
  snprintf(detail, sizeof detail, "%lld rows: %lld flips (%lld outside the bar), centres max err %.2e, empty cluster -> 0: %s, err %.6g vs %.6g",
           (long long)n, (long long)flips, (long long)bad, worst, empty_ok ? "yes" : "NO", got_err, want_err);
  report(name, ok, detail);
  hipFree(d_x); hipFree(d_cen); hipFree(d_sums); hipFree(d_cnt); hipFree(d_new); hipFree(d_err); hipFree(d_lab); hipFree(d_ws);
  free(x); free(cen); free(want); free(want_lab); free(want_cnt); free(gap); free(got); free(got_cnt); free(got_lab);
}

/* ----------------------------------------------------------------- top-k */
/* indexed != 0: the same search a second time through the database side prepared once (ABI 8: anyloc_topk_index_build +
 * anyloc_topk_search_index -- faiss' index.add apart from index.search), and a third time SCREENED (ABI 9:
 * anyloc_topk_search_index_rows with the fp32 rows next to the index, option topk_screen = 1: leading-plane scores under a
 * proven bound + float64 re-scoring of the rows inside it), all checked against the same oracle lists */
static void case_topk(hipStream_t stream, int64_t nq, int64_t ndb, int64_t dim, int64_t k, int metric, int64_t base,
                      int indexed, const char* name) {
  float* db = (float*)malloc(sizeof(float) * (size_t)(ndb * dim));
  float* qu = (float*)malloc(sizeof(float) * (size_t)(nq * dim));
  clustered_rows(db, ndb, dim, 12, 1.0f, 1);                 /* raw rows of very different norms: NORMALIZE_DB has work to do */
  clustered_rows(qu, nq, dim, 12, 1.0f, 1);
  for (int64_t q = 0; q < nq && q < ndb; q += 3)             /* every third query depicts a database row */
    for (int64_t j = 0; j < dim; ++j) qu[q * dim + j] = 0.8f * db[((q * 7) % ndb) * dim + j] + 0.2f * qu[q * dim + j];
  if (ndb > 10) memcpy(db + 9 * dim, db + 4 * dim, sizeof(float) * (size_t)dim);   /* an exact duplicate: the tie goes to row 4 */
  float* qn = (float*)malloc(sizeof(float) * (size_t)(nq * dim));
  oracle_l2norm_rows(qu, qn, nq, dim);
  float* want_d = (float*)malloc(sizeof(float) * (size_t)(nq * k));
  int64_t* want_i = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq * k));
  oracle_flat_topk(qn, nq, db, ndb, dim, k, metric, 1, want_d, want_i);

  float* d_db = (float*)dev_from(db, sizeof(float) * (size_t)(ndb * dim));
  float* d_qu = (float*)dev_from(qu, sizeof(float) * (size_t)(nq * dim));
  float* d_qn = (float*)dev_alloc(sizeof(float) * (size_t)(nq * dim));
  float* d_dist = (float*)dev_alloc(sizeof(float) * (size_t)(nq * k));
  int64_t* d_idx = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)(nq * k));
  size_t ws_bytes = anyloc_topk_workspace_bytes(nq, ndb, dim, k);
  void* d_ws = dev_alloc(ws_bytes);
  ANYLOC_OK_OR_FAIL(anyloc_l2norm_rows(d_qu, d_qn, nq, dim, 1e-12f, stream));
  float* got_d = (float*)malloc(sizeof(float) * (size_t)(nq * k));
  int64_t* got_i = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq * k));
  void *d_index = NULL, *d_iws = NULL;
  int ok = 1;
  int64_t swaps = 0, bad = 0, pad_bad = 0;
  double worst = 0.0;
  for (int pass = 0; pass < (indexed ? 3 : 1); ++pass) {
  if (pass == 0) {
    ANYLOC_OK_OR_FAIL(anyloc_topk(d_qn, nq, d_db, ndb, dim, k, metric, ANYLOC_TOPK_NORMALIZE_DB, base, d_dist, d_idx, d_ws,
                                  ws_bytes, stream));
  } else if (pass == 2) {
    ANYLOC_OK_OR_FAIL(anyloc_set_option("topk_screen", 1));
    const size_t iw2 = anyloc_topk_index_workspace_bytes(nq, ndb, dim, k);    /* (sized with the option set: the screened buffers) */
    void* d_iws2 = dev_alloc(iw2);
    HIP_OK(hipMemsetAsync(d_dist, 0xff, sizeof(float) * (size_t)(nq * k), stream));
    HIP_OK(hipMemsetAsync(d_idx, 0xff, sizeof(int64_t) * (size_t)(nq * k), stream));
    ANYLOC_OK_OR_FAIL(anyloc_topk_search_index_rows(d_qn, nq, d_db, d_index, ndb, dim, k, metric, ANYLOC_TOPK_NORMALIZE_DB, base, d_dist,
                                                    d_idx, d_iws2, iw2, stream));
    HIP_OK(hipStreamSynchronize(stream));
    hipFree(d_iws2);
    ANYLOC_OK_OR_FAIL(anyloc_set_option("topk_screen", -1));
  } else {
    const size_t ib = anyloc_topk_index_bytes(ndb, dim), iw = anyloc_topk_index_workspace_bytes(nq, ndb, dim, k);
    if (ib == 0 || iw == 0) { report(name, 0, "anyloc_topk_index_bytes says the shape is not served"); ok = -1; break; }
    d_index = dev_alloc(ib);
    d_iws = dev_alloc(iw);
    HIP_OK(hipMemsetAsync(d_dist, 0xff, sizeof(float) * (size_t)(nq * k), stream));
    HIP_OK(hipMemsetAsync(d_idx, 0xff, sizeof(int64_t) * (size_t)(nq * k), stream));
    ANYLOC_OK_OR_FAIL(anyloc_topk_index_build(d_db, ndb, dim, d_index, ib, stream));
    ANYLOC_OK_OR_FAIL(anyloc_topk_search_index(d_qn, nq, d_index, ndb, dim, k, metric, ANYLOC_TOPK_NORMALIZE_DB, base, d_dist, d_idx,
                                               d_iws, iw, stream));
  }
  HIP_OK(hipMemcpyAsync(got_d, d_dist, sizeof(float) * (size_t)(nq * k), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(got_i, d_idx, sizeof(int64_t) * (size_t)(nq * k), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  for (int64_t q = 0; q < nq; ++q)
    for (int64_t j = 0; j < k; ++j) {
      int64_t w = want_i[q * k + j], g = got_i[q * k + j];
      if (w < 0) { if (g != -1) ++pad_bad; continue; }        /* k > ndb: idx -1 (faiss) */
      double dd = fabs((double)got_d[q * k + j] - (double)want_d[q * k + j]);
      if (dd > worst) worst = dd;
      if (g != w + base) {
        ++swaps;                                              /* acceptable only between candidates whose exact scores tie to 3e-6 */
        if (dd > 3e-6) ++bad;
      }
    }
  if (ok > 0) ok = bad == 0 && pad_bad == 0 && worst <= 3e-6;
  if (ok > 0 && ndb > 10) {                                            /* the duplicated row: wherever both appear, 4 comes before 9 */
    for (int64_t q = 0; q < nq; ++q) {
      int64_t p4 = -1, p9 = -1;
      for (int64_t j = 0; j < k; ++j) { if (got_i[q * k + j] == 4 + base) p4 = j; if (got_i[q * k + j] == 9 + base) p9 = j; }
      if (p9 >= 0 && (p4 < 0 || p4 > p9)) ok = 0;
    }
  }
  }                                                          /* (both passes accumulate into the same counters) */
  char detail[240];
  snprintf(detail, sizeof detail, "%lld x %lld x %lld, k=%lld%s: %lld near-tie swaps (%lld outside 3e-6), max |dist err| %.2e, padding %s",
           (long long)nq, (long long)ndb, (long long)dim, (long long)k, indexed ? " (one-shot + prepared index + screened with rows)" : "", (long long)swaps,
           (long long)bad, worst, pad_bad ? "WRONG" : "ok");
  if (ok >= 0) report(name, ok, detail);                     /* (-1: already reported) */
  hipFree(d_db); hipFree(d_qu); hipFree(d_qn); hipFree(d_dist); hipFree(d_idx); hipFree(d_ws);
  if (d_index) hipFree(d_index);
  if (d_iws) hipFree(d_iws);
  free(db); free(qu); free(qn); free(want_d); free(want_i); free(got_d); free(got_i);
}


/* ------------------------------------------------------------------- ViT */
typedef struct { float *h, *d; size_t n; } pair_t;            /* a host array and its device copy */
static pair_t mk(size_t n, float mean, float std) {
  pair_t p;
  p.n = n;
  p.h = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) p.h[i] = mean + std * rng_normal();
  p.d = (float*)dev_from(p.h, sizeof(float) * n);
  return p;
}
static void rm(pair_t* p) { free(p->h); hipFree(p->d); }

/* DinoV2ExtractFeatures.__call__ through anyloc_vit_create / anyloc_vit_forward on a 3-block model of the ViT-S width
   (ffn_kind 0: GELU mlp; 1: SwiGLU as in ViT-g), two taps in one call (token of block 1 | value of block 2), first on the
   exact-fp32 matrix-core kernels, then with the two-term fp16 images attached (anyloc_split_h2 of every weight matrix +
   the Cauchy-Schwarz constants of the FFN input projection) -- against the C restatement of the hub model. */
static void case_vit(hipStream_t stream, int ffn_kind, const char* name) {
  enum { D = 384, HEADS = 6, DEPTH = 3, P = 14, PK = 3 * 14 * 14 };
  const int64_t HID = ffn_kind ? 1024 : 1536, F1 = ffn_kind ? 2 * HID : HID;
  const int64_t B = 3, H = 112, W = 168, N = (H / P) * (W / P), T = N + 1;
  pair_t img = mk((size_t)(B * 3 * H * W), 0.0f, 1.0f), pos = mk((size_t)(T * D), 0.0f, 0.3f);
  pair_t pw = mk((size_t)D * PK, 0.0f, 0.04f), pb = mk(D, 0.0f, 0.1f), cls = mk(D, 0.0f, 0.5f);
  pair_t w[DEPTH][14];
  oracle_vit_block ob[DEPTH];
  anyloc_vit_block_weights ab[DEPTH];
  float* fc1_il[DEPTH] = {0};                                 /* SwiGLU: the 32 gate / 32 value interleave the ABI takes */
  float* fc1b_il[DEPTH] = {0};
  for (int l = 0; l < DEPTH; ++l) {
    const size_t sz[14] = {D, D, (size_t)3 * D * D, 3 * D, (size_t)D * D, D, D, D, D, (size_t)F1 * D, (size_t)F1, (size_t)D * HID, D, D};
    const float mean[14] = {1, 0, 0, 0, 0, 0, 0.5f, 1, 0, 0, 0, 0, 0, 0.5f};
    const float sd[14] = {0.1f, 0.1f, 0.05f, 0.1f, 0.05f, 0.1f, 0.2f, 0.1f, 0.1f, 0.05f, 0.1f, 0.03f, 0.1f, 0.2f};
    for (int f = 0; f < 14; ++f) w[l][f] = mk(sz[f], mean[f], sd[f]);
    ob[l].norm1_w = w[l][0].h; ob[l].norm1_b = w[l][1].h; ob[l].qkv_w = w[l][2].h; ob[l].qkv_b = w[l][3].h;
    ob[l].proj_w = w[l][4].h; ob[l].proj_b = w[l][5].h; ob[l].ls1 = w[l][6].h; ob[l].norm2_w = w[l][7].h; ob[l].norm2_b = w[l][8].h;
    ob[l].fc1_w = w[l][9].h; ob[l].fc1_b = w[l][10].h; ob[l].fc2_w = w[l][11].h; ob[l].fc2_b = w[l][12].h; ob[l].ls2 = w[l][13].h;
    ab[l].norm1_w = w[l][0].d; ab[l].norm1_b = w[l][1].d; ab[l].qkv_w = w[l][2].d; ab[l].qkv_b = w[l][3].d;
    ab[l].proj_w = w[l][4].d; ab[l].proj_b = w[l][5].d; ab[l].ls1 = w[l][6].d; ab[l].norm2_w = w[l][7].d; ab[l].norm2_b = w[l][8].d;
    ab[l].fc1_w = w[l][9].d; ab[l].fc1_b = w[l][10].d; ab[l].fc2_w = w[l][11].d; ab[l].fc2_b = w[l][12].d; ab[l].ls2 = w[l][13].d;
    if (ffn_kind) {                                           /* hub w12 = [HID gate rows; HID value rows] -> per 32 channels: 32 gate rows, 32 value rows */
      float* il = (float*)malloc(sizeof(float) * (size_t)(F1 * D));
      float* bl = (float*)malloc(sizeof(float) * (size_t)F1);
      for (int64_t c = 0; c < HID; ++c)
        for (int v = 0; v < 2; ++v) {
          const int64_t dst = (c / 32) * 64 + v * 32 + c % 32, src = v * HID + c;
          memcpy(il + dst * D, w[l][9].h + src * D, sizeof(float) * D);
          bl[dst] = w[l][10].h[src];
        }
      fc1_il[l] = (float*)dev_from(il, sizeof(float) * (size_t)(F1 * D));
      fc1b_il[l] = (float*)dev_from(bl, sizeof(float) * (size_t)F1);
      ab[l].fc1_w = fc1_il[l];
      ab[l].fc1_b = fc1b_il[l];
      free(il); free(bl);
    }
  }
  oracle_vit_config oc = {D, DEPTH, HEADS, ffn_kind, (int32_t)HID, P};
  float* want_tok = (float*)malloc(sizeof(float) * (size_t)(B * N * D));
  float* want_val = (float*)malloc(sizeof(float) * (size_t)(B * N * D));
  oracle_vit_facet(&oc, pw.h, pb.h, cls.h, pos.h, ob, img.h, B, H, W, 1, 3, 0, 1, want_tok);
  oracle_vit_facet(&oc, pw.h, pb.h, cls.h, pos.h, ob, img.h, B, H, W, 2, 2, 0, 1, want_val);

  anyloc_vit_config cfg = {D, DEPTH, HEADS, ffn_kind, (int32_t)HID, P, PK};
  anyloc_vit_t* vit = NULL;
  ANYLOC_OK_OR_FAIL(anyloc_vit_create(&vit, &cfg, pw.d, pb.d, cls.d, ab));
  const size_t ws_bytes = anyloc_vit_workspace_bytes(vit, B, H, W);
  void* d_ws = dev_alloc(ws_bytes);
  float* d_out = (float*)dev_alloc(sizeof(float) * (size_t)(B * N * 2 * D));
  float* got = (float*)malloc(sizeof(float) * (size_t)(B * N * 2 * D));
  const int32_t layers[2] = {1, 2}, facets[2] = {ANYLOC_FACET_TOKEN, ANYLOC_FACET_VALUE};
  void* img2[DEPTH][4] = {{0}};
  float* inv2[DEPTH][4] = {{0}};
  anyloc_vit_block_h2 hb[DEPTH];
  memset(hb, 0, sizeof hb);
  for (int mode = 0; mode < 2; ++mode) {
    unsigned flags = ANYLOC_VIT_NORM_TAPS;
    if (mode == 1) {
      for (int l = 0; l < DEPTH; ++l) {
        const float* mat[4] = {ab[l].qkv_w, ab[l].proj_w, ab[l].fc1_w, ab[l].fc2_w};
        const int64_t rows[4] = {3 * D, D, F1, D}, kk[4] = {D, D, D, HID};
        for (int f = 0; f < 4; ++f) {
          img2[l][f] = dev_alloc(anyloc_h2_bytes(rows[f], kk[f]));
          inv2[l][f] = (float*)dev_alloc(sizeof(float) * (size_t)rows[f]);
          ANYLOC_OK_OR_FAIL(anyloc_split_h2(mat[f], kk[f], rows[f], kk[f], img2[l][f], inv2[l][f], stream));
        }
        hb[l].qkv_w2 = img2[l][0]; hb[l].qkv_inv = inv2[l][0]; hb[l].proj_w2 = img2[l][1]; hb[l].proj_inv = inv2[l][1];
        hb[l].fc1_w2 = img2[l][2]; hb[l].fc1_inv = inv2[l][2]; hb[l].fc2_w2 = img2[l][3]; hb[l].fc2_inv = inv2[l][3];
        /* |fc1_j(y)| <= ||y|| max_j ||W_j|| + max_j |b_j|: largest row norm and bias of the gate (mlp: fc1) rows, then of the value rows */
        double bound[4] = {0, 0, 0, 0};
        for (int64_t r = 0; r < F1; ++r) {
          double ss = 0.0;
          for (int64_t c = 0; c < D; ++c) ss += (double)w[l][9].h[r * D + c] * w[l][9].h[r * D + c];
          const int v = (ffn_kind && r >= HID) ? 2 : 0;
          if (sqrt(ss) > bound[v]) bound[v] = sqrt(ss);
          if (fabs((double)w[l][10].h[r]) > bound[v + 1]) bound[v + 1] = fabs((double)w[l][10].h[r]);
        }
        for (int j = 0; j < 4; ++j) hb[l].fc1_bound[j] = (float)(bound[j] * (1.0 + 1e-6));
        hb[l].fc1_b2 = NULL;
        hb[l].fc1_layout = 0;
      }
      ANYLOC_OK_OR_FAIL(anyloc_vit_attach_h2(vit, hb));
      flags |= ANYLOC_VIT_SPLIT_FP16;
    }
    HIP_OK(hipMemsetAsync(d_out, 0xff, sizeof(float) * (size_t)(B * N * 2 * D), stream));
    ANYLOC_OK_OR_FAIL(anyloc_vit_forward(vit, img.d, B, H, W, pos.d, 2, layers, facets, flags, d_out, d_ws, ws_bytes, stream));
    HIP_OK(hipMemcpyAsync(got, d_out, sizeof(float) * (size_t)(B * N * 2 * D), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    double worst_t = 0.0, worst_v = 0.0;
    for (int64_t r = 0; r < B * N; ++r)
      for (int64_t c = 0; c < D; ++c) {
        const double dt = fabs((double)got[r * 2 * D + c] - (double)want_tok[r * D + c]);
        const double dv = fabs((double)got[r * 2 * D + D + c] - (double)want_val[r * D + c]);
        if (!(dt <= worst_t)) worst_t = dt;
        if (!(dv <= worst_v)) worst_v = dv;
      }
    char label[96], detail[200];
    snprintf(label, sizeof label, "%s, %s", name, mode ? "two-term fp16 GEMMs" : "fp32 matrix-core GEMMs");
    snprintf(detail, sizeof detail, "%lld images of %lldx%lld, %lld tokens x %d: max |err| token tap %.2e, value tap %.2e (unit rows; bar 2e-5)",
             (long long)B, (long long)H, (long long)W, (long long)N, D, worst_t, worst_v);
    report(label, worst_t <= 2e-5 && worst_v <= 2e-5, detail);
  }
  anyloc_vit_destroy(vit);
  for (int l = 0; l < DEPTH; ++l) {
    for (int f = 0; f < 14; ++f) rm(&w[l][f]);
    for (int f = 0; f < 4; ++f) { hipFree(img2[l][f]); hipFree(inv2[l][f]); }
    hipFree(fc1_il[l]); hipFree(fc1b_il[l]);
  }
  rm(&img); rm(&pos); rm(&pw); rm(&pb); rm(&cls);
  hipFree(d_ws); hipFree(d_out);
  free(got); free(want_tok); free(want_val);
}

/* ---------------------------------------------------------------- errors */
/* ------------------------------------------------------------------- PCA fit products (float64 matrix cores) */
static void case_pca(hipStream_t stream, int64_t n, int64_t f, int64_t k, const char* name) {
  float* x = (float*)malloc(sizeof(float) * (size_t)(n * f));
  double* mean = (double*)calloc((size_t)f, sizeof(double));
  double* vec = (double*)malloc(sizeof(double) * (size_t)(n * n));
  for (int64_t i = 0; i < n * f; ++i) x[i] = 25.0f + 2.0f * rng_normal();      /* a common offset the centring has to remove */
  for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < f; ++j) mean[j] += (double)x[i * f + j] / (double)n;
  for (int64_t i = 0; i < n * n; ++i) vec[i] = rng_normal();
  float* d_x = (float*)dev_from(x, sizeof(float) * (size_t)(n * f));
  double* d_mean = (double*)dev_from(mean, sizeof(double) * (size_t)f);
  double* d_vec = (double*)dev_from(vec, sizeof(double) * (size_t)(n * n));
  double* d_gram = (double*)dev_alloc(sizeof(double) * (size_t)(n * n));
  double* d_scat = (double*)dev_alloc(sizeof(double) * (size_t)(f * f));
  double* d_axes = (double*)dev_alloc(sizeof(double) * (size_t)(k * f));
  ANYLOC_OK_OR_FAIL(anyloc_pca_gram_f64(d_x, n, f, d_mean, 0, d_gram, stream));
  ANYLOC_OK_OR_FAIL(anyloc_pca_gram_f64(d_x, n, f, d_mean, 1, d_scat, stream));
  ANYLOC_OK_OR_FAIL(anyloc_pca_axes_f64(d_vec, n, 1, k, d_x, n, f, d_mean, d_axes, stream));
  double* gram = (double*)malloc(sizeof(double) * (size_t)(n * n));
  double* scat = (double*)malloc(sizeof(double) * (size_t)(f * f));
  double* axes = (double*)malloc(sizeof(double) * (size_t)(k * f));
  HIP_OK(hipMemcpyAsync(gram, d_gram, sizeof(double) * (size_t)(n * n), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(scat, d_scat, sizeof(double) * (size_t)(f * f), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(axes, d_axes, sizeof(double) * (size_t)(k * f), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  double worst = 0.0, top = 0.0;                               /* the same sums in long double on the host */
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) {
      long double acc = 0.0L;
      for (int64_t c = 0; c < f; ++c) acc += ((long double)x[i * f + c] - mean[c]) * ((long double)x[j * f + c] - mean[c]);
      double e = fabs(gram[i * n + j] - (double)acc);
      if (e > worst) worst = e;
      if (fabs((double)acc) > top) top = fabs((double)acc);
    }
  for (int64_t i = 0; i < f; ++i)
    for (int64_t j = 0; j < f; ++j) {
      long double acc = 0.0L;
      for (int64_t c = 0; c < n; ++c) acc += ((long double)x[c * f + i] - mean[i]) * ((long double)x[c * f + j] - mean[j]);
      double e = fabs(scat[i * f + j] - (double)acc);
      if (e > worst) worst = e;
    }
  for (int64_t i = 0; i < k; ++i)
    for (int64_t j = 0; j < f; ++j) {
      long double acc = 0.0L;
      for (int64_t c = 0; c < n; ++c) acc += (long double)vec[c * n + i] * ((long double)x[c * f + j] - mean[j]);
      double e = fabs(axes[i * f + j] - (double)acc);
      if (e > worst) worst = e;
    }
  int ok = worst <= 1e-12 * top;
  char detail[200];
  snprintf(detail, sizeof detail, "%lld x %lld, k=%lld: Gram, scatter, U^T Xc vs long double: max |err| %.2e (largest entry %.2e)",
           (long long)n, (long long)f, (long long)k, worst, top);
  report(name, ok, detail);
  hipFree(d_x); hipFree(d_mean); hipFree(d_vec); hipFree(d_gram); hipFree(d_scat); hipFree(d_axes);
  free(x); free(mean); free(vec); free(gram); free(scat); free(axes);
}

static void case_errors(hipStream_t stream) {
  float* d = (float*)dev_alloc(sizeof(float) * 64 * 8);
  int64_t* di = (int64_t*)dev_alloc(sizeof(int64_t) * 64);
  int s1 = anyloc_topk(d, 4, d, 8, 8, 2, 0, 0, 0, d, di, d, 0 /* workspace_bytes */, stream);
  const char* m1 = anyloc_last_error();
  int ok1 = s1 == ANYLOC_ERR_WORKSPACE && m1 && m1[0];
  int s2 = anyloc_l2norm_rows(NULL, d, 4, 8, 1e-12f, stream);
  const char* m2 = anyloc_last_error();
  int ok2 = s2 == ANYLOC_ERR_INVALID_ARG && m2 && m2[0];
  int64_t v = -1;
  int s3 = anyloc_get_option("no_such_option", &v);
  int ok3 = s3 != ANYLOC_OK;
  HIP_OK(hipStreamSynchronize(stream));
  char detail[200];
  snprintf(detail, sizeof detail, "small workspace -> %d, null pointer -> %d, unknown option -> %d", s1, s2, s3);
  report("errors are status codes", ok1 && ok2 && ok3, detail);
  hipFree(d); hipFree(di);
}

int main(void) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    fprintf(stderr, "abi_host: no HIP device -- libanyloc_hip.so has no CPU path, nothing to run\n");
    return 77;
  }
  if (anyloc_version() != ANYLOC_ABI_VERSION) {
    fprintf(stderr, "abi_host: library ABI %d, header ABI %d -- rebuild\n", anyloc_version(), ANYLOC_ABI_VERSION);
    return 1;
  }
  HIP_OK(hipSetDevice(0));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));                          /* a caller-owned, non-default stream */
  rng_state = 20240925u;

  {  /* headline width: three full images + an empty one + two ragged ones */
    const int64_t off[] = {0, 529, 529, 1058, 1075, 1076, 1605};
    case_vlad(stream, 1536, 32, ANYLOC_VLAD_NORM_DESCS | ANYLOC_VLAD_INTRA_NORM, off, 6, "vlad D=1536 K=32 norm+intra");
    case_vlad(stream, 1536, 32, ANYLOC_VLAD_NORM_DESCS, off, 6, "vlad D=1536 K=32 norm only");
  }
  {
    const int64_t off[] = {0, 300, 300, 301, 700};
    case_vlad(stream, 100, 5, ANYLOC_VLAD_NORM_DESCS | ANYLOC_VLAD_INTRA_NORM, off, 4, "vlad D=100 K=5 norm+intra");
    case_vlad(stream, 100, 5, ANYLOC_VLAD_INTRA_NORM, off, 4, "vlad D=100 K=5 tokens as passed");
    case_vlad(stream, 100, 5, ANYLOC_VLAD_NORM_DESCS | ANYLOC_VLAD_INTRA_NORM | ANYLOC_VLAD_EUCLIDEAN, off, 4,
              "vlad D=100 K=5 euclidean labels");
  }
  case_kmeans(stream, 40000, 1536, 32, 0, "kmeans step+update 40000x1536 K=32 cosine");
  case_kmeans(stream, 5000, 64, 16, 1, "kmeans step+update 5000x64 K=16 euclidean");
  case_topk(stream, 5, 2000, 49152, 20, 0, 0, 0, "topk 5 queries x 49152 dims, IP");
  case_topk(stream, 5, 2000, 49152, 20, 1, 1000000, 0, "topk 5 queries x 49152 dims, L2, base 1e6");
  case_topk(stream, 300, 3000, 256, 10, 0, 0, 1, "topk 300 queries x 256 dims, IP");
  case_topk(stream, 280, 9000, 1024, 20, 1, 77, 1, "topk 280 queries x 9000 rows x 1024 dims, L2, base 77 (ragged last panel)");
  case_topk(stream, 3, 12, 64, 20, 0, 0, 0, "topk k > ndb padding");
  case_vit(stream, 0, "vit 3 blocks D=384 mlp");
  case_vit(stream, 1, "vit 3 blocks D=384 swiglu");
  case_pca(stream, 70, 131, 9, "pca fit products 70 x 131 (float64 matrix cores)");
  case_errors(stream);

  HIP_OK(hipStreamDestroy(stream));
  printf("{\"abi_host\": \"%s\", \"failures\": %d, \"abi\": %d}\n", failures ? "FAIL" : "ok", failures, anyloc_version());
  return failures ? 1 : 0;
}
