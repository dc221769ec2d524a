/*
 * anyloc_oracle.c -- TEST INFRASTRUCTURE ONLY: see anyloc_oracle.h.  Plain C11, no dependencies beyond libm; the
 * OpenMP pragmas only spread independent rows over the host cores (the results do not depend on the thread count:
 * every output element is produced by one thread in a fixed order).
 */
#include "anyloc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static double dot_d(const float* a, const float* b, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
  return s;
}

/* F.normalize: v / max(||v||_2, eps), eps = 1e-12 (torch default; utilities.py:436-437, :782, :859-860, :889, :960) */
static double fnorm_denominator(double sumsq) {
  double n = sqrt(sumsq);
  return n > 1e-12 ? n : 1e-12;
}

void oracle_l2norm_rows(const float* x, float* out, int64_t rows, int64_t dim) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    const float* p = x + r * dim;
    double den = fnorm_denominator(dot_d(p, p, dim));
    for (int64_t c = 0; c < dim; ++c) out[r * dim + c] = (float)((double)p[c] / den);
  }
}

/* fpk cos_sim: a / (||a|| + 1e-8) . b / (||b|| + 1e-8);  euc_sim: 2 a.b - a.a - b.b */
void oracle_fpk_labels(const float* x, int64_t n, int64_t D, const float* centers, int64_t K, int mode,
                       int64_t* labels, double* gap) {
  double* cden = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
  for (int64_t k = 0; k < K; ++k) {
    double ss = dot_d(centers + k * D, centers + k * D, D);
    cden[k] = mode == 0 ? sqrt(ss) + 1e-8 : ss;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* p = x + i * D;
    double ss = dot_d(p, p, D);
    double xden = mode == 0 ? sqrt(ss) + 1e-8 : ss;
    double best = -INFINITY, second = -INFINITY;
    int64_t arg = 0;
    for (int64_t k = 0; k < K; ++k) {
      double ab = dot_d(p, centers + k * D, D);
      double s = mode == 0 ? ab / (xden * cden[k]) : 2.0 * ab - xden - cden[k];
      if (s > best) { second = best; best = s; arg = k; }   /* strict: the first maximum wins, as torch.max on CPU */
      else if (s > second) second = s;
    }
    labels[i] = arg;
    if (gap) gap[i] = K > 1 ? best - second : INFINITY;
  }
  free(cden);
}

void oracle_vlad_assigned(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t D, const float* centers,
                          int64_t K, unsigned flags, const int64_t* lab, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t im = 0; im < n_img; ++im) {
    double* acc = (double*)calloc((size_t)(K * D), sizeof(double));
    char* used = (char*)calloc((size_t)K, 1);
    for (int64_t t = offsets[im]; t < offsets[im + 1]; ++t) {
      const float* p = tokens + t * D;
      const int64_t k = lab[t];
      /* residual of the RE-NORMALISED token w.r.t. the RAW centre (utilities.py:959-962), fp32 values as the
         reference forms them, summed exactly */
      double den = (flags & ORACLE_VLAD_NORM_DESCS) ? fnorm_denominator(dot_d(p, p, D)) : 1.0;
      for (int64_t c = 0; c < D; ++c) {
        float xh = (flags & ORACLE_VLAD_NORM_DESCS) ? (float)((double)p[c] / den) : p[c];
        acc[k * D + c] += (double)xh - (double)centers[k * D + c];
      }
      used[k] = 1;
    }
    double gss = 0.0;
    for (int64_t k = 0; k < K; ++k) {
      if (!used[k]) continue;                                  /* unused clusters stay zero (:854) */
      double ss = 0.0;
      for (int64_t c = 0; c < D; ++c) ss += acc[k * D + c] * acc[k * D + c];
      if (flags & ORACLE_VLAD_INTRA_NORM) {
        double den = fnorm_denominator(ss);
        for (int64_t c = 0; c < D; ++c) acc[k * D + c] /= den;
        ss = ss / (den * den);
      }
      gss += ss;
    }
    double gden = fnorm_denominator(gss);                      /* :889 */
    for (int64_t j = 0; j < K * D; ++j) out[im * K * D + j] = (float)(acc[j] / gden);
    free(acc);
    free(used);
  }
}

void oracle_vlad_hard(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t D, const float* centers,
                      int64_t K, unsigned flags, float* out, int64_t* labels, double* gap) {
  const int64_t total = offsets[n_img];
  int64_t* lab = labels ? labels : (int64_t*)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
  oracle_fpk_labels(tokens, total, D, centers, K, (flags & ORACLE_VLAD_EUCLIDEAN) ? 1 : 0, lab, gap);
  oracle_vlad_assigned(tokens, offsets, n_img, D, centers, K, flags, lab, out);
  if (!labels) free(lab);
}

double oracle_kmeans_update_from_labels(const float* x, int64_t n, int64_t D, const float* centers, int64_t K,
                                        const int64_t* lab, float* centers_new, double* counts) {
  double* sums = (double*)calloc((size_t)(K * D), sizeof(double));
  double* cnt = (double*)calloc((size_t)K, sizeof(double));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t k = lab[i];
    cnt[k] += 1.0;
    for (int64_t c = 0; c < D; ++c) sums[k * D + c] += (double)x[i * D + c];
  }
  double err = 0.0;
  for (int64_t k = 0; k < K; ++k)
    for (int64_t c = 0; c < D; ++c) {
      /* (onehot @ X) / onehot.sum(-1): an empty cluster is 0 / 0 = NaN, which fpk replaces by 0 */
      float v = cnt[k] > 0.0 ? (float)(sums[k * D + c] / cnt[k]) : 0.0f;
      double d = (double)v - (double)centers[k * D + c];
      err += d * d;
      centers_new[k * D + c] = v;
    }
  if (counts) memcpy(counts, cnt, sizeof(double) * (size_t)K);
  free(sums);
  free(cnt);
  return err;
}

double oracle_kmeans_iteration(const float* x, int64_t n, int64_t D, const float* centers, int64_t K, int mode,
                               float* centers_new, int64_t* labels, double* counts) {
  int64_t* lab = labels ? labels : (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  oracle_fpk_labels(x, n, D, centers, K, mode, lab, NULL);
  double err = oracle_kmeans_update_from_labels(x, n, D, centers, K, lab, centers_new, counts);
  if (!labels) free(lab);
  return err;
}

int oracle_kmeans_fit(const float* x, int64_t n, int64_t D, int64_t K, int mode, const int64_t* init_rows,
                      int max_iter, double tol, float* centers) {
  float* next = (float*)malloc(sizeof(float) * (size_t)(K * D));
  for (int64_t k = 0; k < K; ++k) memcpy(centers + k * D, x + init_rows[k] * D, sizeof(float) * (size_t)D);
  int it = 0;
  while (it < max_iter) {
    double err = oracle_kmeans_iteration(x, n, D, centers, K, mode, next, NULL, NULL);
    memcpy(centers, next, sizeof(float) * (size_t)(K * D));   /* lr = 1 for full-batch fits */
    ++it;
    if (err <= tol) break;
  }
  free(next);
  return it;
}

typedef struct { double key; int64_t idx; } scored;

static int scored_cmp(const void* a, const void* b) {
  const scored* x = (const scored*)a;
  const scored* y = (const scored*)b;
  if (x->key < y->key) return -1;                              /* smaller key = better */
  if (x->key > y->key) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);     /* ties -> lower database index */
}

void oracle_flat_topk(const float* qu, int64_t nq, const float* db, int64_t ndb, int64_t dim, int64_t k, int metric,
                      int normalize_db, float* dist, int64_t* idx) {
  double* dden = (double*)malloc(sizeof(double) * (size_t)(ndb > 0 ? ndb : 1));
  for (int64_t r = 0; r < ndb; ++r)
    dden[r] = normalize_db ? fnorm_denominator(dot_d(db + r * dim, db + r * dim, dim)) : 1.0;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < nq; ++q) {
    scored* s = (scored*)malloc(sizeof(scored) * (size_t)(ndb > 0 ? ndb : 1));
    const float* p = qu + q * dim;
    const double qq = dot_d(p, p, dim);
    for (int64_t r = 0; r < ndb; ++r) {
      const float* d = db + r * dim;
      double ab = dot_d(p, d, dim) / dden[r];
      double score;
      if (metric == 0) score = ab;                             /* IndexFlatIP */
      else {                                                   /* IndexFlatL2: squared distance to the (normalised) row */
        double dd = dot_d(d, d, dim) / (dden[r] * dden[r]);
        score = qq + dd - 2.0 * ab;
      }
      s[r].key = metric == 0 ? -score : score;
      s[r].idx = r;
    }
    qsort(s, (size_t)ndb, sizeof(scored), scored_cmp);
    for (int64_t j = 0; j < k; ++j) {
      if (j < ndb) {
        dist[q * k + j] = (float)(metric == 0 ? -s[j].key : s[j].key);
        idx[q * k + j] = s[j].idx;
      } else {
        dist[q * k + j] = metric == 0 ? -INFINITY : INFINITY;
        idx[q * k + j] = -1;
      }
    }
    free(s);
  }
  free(dden);
}

void oracle_recalls(const int64_t* idx, int64_t nq, int64_t kmax, const int64_t* top_k, int64_t n_k,
                    const int64_t* gt, const int64_t* gt_off, double* recalls) {
  for (int64_t j = 0; j < n_k; ++j) recalls[j] = 0.0;
  for (int64_t q = 0; q < nq; ++q)
    for (int64_t j = 0; j < n_k; ++j) {
      int hit = 0;
      for (int64_t r = 0; r < top_k[j] && r < kmax && !hit; ++r)
        for (int64_t g = gt_off[q]; g < gt_off[q + 1]; ++g)
          if (idx[q * kmax + r] == gt[g]) { hit = 1; break; }
      recalls[j] += hit;
    }
  for (int64_t j = 0; j < n_k; ++j) recalls[j] = nq > 0 ? recalls[j] / (double)nq : 0.0;
}

/* ------------------------------------------------------------------ DINOv2 ViT (see anyloc_oracle.h) */
static void linear_rows(const float* x, int64_t rows, int64_t in, const float* w, const float* b, int64_t out_dim, float* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t o = 0; o < out_dim; ++o)
      y[r * out_dim + o] = (float)(dot_d(x + r * in, w + o * in, in) + (b ? (double)b[o] : 0.0));
}

static void layernorm_rows(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, float* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {                     /* nn.LayerNorm(dim, eps=1e-6): biased variance */
    const float* p = x + r * dim;
    double mean = 0.0, var = 0.0;
    for (int64_t c = 0; c < dim; ++c) mean += p[c];
    mean /= (double)dim;
    for (int64_t c = 0; c < dim; ++c) var += ((double)p[c] - mean) * ((double)p[c] - mean);
    const double rstd = 1.0 / sqrt(var / (double)dim + 1e-6);
    for (int64_t c = 0; c < dim; ++c) y[r * dim + c] = (float)(((double)p[c] - mean) * rstd * (double)w[c] + (double)b[c]);
  }
}

void oracle_vit_facet(const oracle_vit_config* cfg, const float* patch_w, const float* patch_b, const float* cls_token,
                      const float* pos, const oracle_vit_block* blocks, const float* img, int64_t B, int64_t H, int64_t W,
                      int32_t layer, int32_t facet, int use_cls, int norm, float* out) {
  const int64_t D = cfg->dim, P = cfg->patch, gh = H / P, gw = W / P, N = gh * gw, T = N + 1, R = B * T;
  const int64_t heads = cfg->heads, hd = D / heads, Hh = cfg->ffn_hidden, PK = 3 * P * P;
  float* x = (float*)malloc(sizeof(float) * (size_t)(R * D));
  float* y = (float*)malloc(sizeof(float) * (size_t)(R * D));
  float* qkv = (float*)malloc(sizeof(float) * (size_t)(R * 3 * D));
  float* att = (float*)malloc(sizeof(float) * (size_t)(R * D));
  float* hid = (float*)malloc(sizeof(float) * (size_t)(R * 2 * Hh));
  /* prepare_tokens: conv 14x14 stride 14 as a contraction over (channel, py, px), CLS in front, + positional table */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const int64_t b = r / T, t = r % T;
    if (t == 0) {
      for (int64_t c = 0; c < D; ++c) x[r * D + c] = (float)((double)cls_token[c] + (double)pos[c]);
      continue;
    }
    const int64_t py0 = ((t - 1) / gw) * P, px0 = ((t - 1) % gw) * P;
    for (int64_t c = 0; c < D; ++c) {
      double s = patch_b[c];
      for (int64_t ch = 0; ch < 3; ++ch)
        for (int64_t py = 0; py < P; ++py) {
          const float* ip = img + ((b * 3 + ch) * H + py0 + py) * W + px0;
          const float* wp = patch_w + c * PK + (ch * P + py) * P;
          for (int64_t px = 0; px < P; ++px) s += (double)ip[px] * (double)wp[px];
        }
      x[r * D + c] = (float)((double)(float)s + (double)pos[t * D + c]);      /* conv output is an fp32 tensor, then + pos */
    }
  }
  const float* tap = NULL;
  int64_t tap_ld = D, tap_off = 0;
  for (int32_t l = 0; l <= layer; ++l) {
    const oracle_vit_block* bw = blocks + l;
    layernorm_rows(x, R, D, bw->norm1_w, bw->norm1_b, y);
    linear_rows(y, R, D, bw->qkv_w, bw->qkv_b, 3 * D, qkv);
    if (l == layer && facet != 3) { tap = qkv; tap_ld = 3 * D; tap_off = (int64_t)facet * D; break; }
    /* softmax((q * hd^-0.5) k^T) v per (image, head); qkv row = [3][heads][hd] */
    const double scale = 1.0 / sqrt((double)hd);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t bh = 0; bh < B * heads * T; ++bh) {
      const int64_t b = bh / (heads * T), h = (bh / T) % heads, i = bh % T;
      const float* q = qkv + (b * T + i) * 3 * D + h * hd;
      double* sc = (double*)malloc(sizeof(double) * (size_t)T);
      double m = -INFINITY, den = 0.0;
      for (int64_t j = 0; j < T; ++j) {
        const float* k = qkv + (b * T + j) * 3 * D + D + h * hd;
        double s = 0.0;
        for (int64_t d = 0; d < hd; ++d) s += (double)(float)((double)q[d] * scale) * (double)k[d];
        sc[j] = s;
        if (s > m) m = s;
      }
      for (int64_t j = 0; j < T; ++j) { sc[j] = exp(sc[j] - m); den += sc[j]; }
      for (int64_t d = 0; d < hd; ++d) {
        double o = 0.0;
        for (int64_t j = 0; j < T; ++j) o += sc[j] * (double)qkv[(b * T + j) * 3 * D + 2 * D + h * hd + d];
        att[(b * T + i) * D + h * hd + d] = (float)(o / den);
      }
      free(sc);
    }
    linear_rows(att, R, D, bw->proj_w, bw->proj_b, D, y);
    for (int64_t i = 0; i < R * D; ++i) x[i] = (float)((double)x[i] + (double)(float)((double)y[i] * (double)bw->ls1[i % D]));
    layernorm_rows(x, R, D, bw->norm2_w, bw->norm2_b, y);
    if (cfg->ffn_kind == 0) {
      linear_rows(y, R, D, bw->fc1_w, bw->fc1_b, Hh, hid);
      for (int64_t i = 0; i < R * Hh; ++i) {
        const double v = hid[i];
        hid[i] = (float)(0.5 * v * (1.0 + erf(v * 0.70710678118654752440)));
      }
    } else {
      linear_rows(y, R, D, bw->fc1_w, bw->fc1_b, 2 * Hh, hid);           /* w12: [gate | value] */
      for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < Hh; ++c) {
          const double g = hid[r * 2 * Hh + c], v = hid[r * 2 * Hh + Hh + c];
          hid[r * Hh + c] = (float)((double)(float)(g / (1.0 + exp(-g))) * v);    /* F.silu(x1) * x2; packed in place (c < Hh <= 2 Hh r) */
        }
    }
    linear_rows(hid, R, Hh, bw->fc2_w, bw->fc2_b, D, y);
    for (int64_t i = 0; i < R * D; ++i) x[i] = (float)((double)x[i] + (double)(float)((double)y[i] * (double)bw->ls2[i % D]));
    if (l == layer) { tap = x; tap_ld = D; tap_off = 0; }
  }
  const int64_t rows_out = use_cls ? T : N, skip = use_cls ? 0 : 1;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t t = 0; t < rows_out; ++t) {
      const float* src = tap + (b * T + t + skip) * tap_ld + tap_off;
      float* dst = out + (b * rows_out + t) * D;
      const double den = norm ? fnorm_denominator(dot_d(src, src, D)) : 1.0;
      for (int64_t c = 0; c < D; ++c) dst[c] = (float)((double)src[c] / den);
    }
  free(x); free(y); free(qkv); free(att); free(hid);
}
