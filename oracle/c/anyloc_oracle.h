/*
 * anyloc_oracle.h -- TEST INFRASTRUCTURE ONLY (oracle/): a scalar C restatement, in double precision, of the
 * arithmetic the reference runs between the DINOv2 tokens and the recall numbers.  Nothing under anyloc_amd/
 * (the product) links or loads this; tests/ use it as the checker of the C ABI (tests/c_abi/abi_host.c, which
 * contains no Python and no torch) and tests/test_oracle_c.py pins it against the golden vectors recorded from the
 * reference's own code (tests/golden/, oracle/make_golden.py).
 *
 * Every function takes HOST pointers to dense row-major arrays and cites the reference lines
 * (/root/reference = AnyLoc/AnyLoc) or the pinned third-party algorithm it follows.  Sums are accumulated in double
 * from the fp32 inputs, so the results are the exact-arithmetic answer to ~1e-15 relative: the reference's own fp32
 * results (torch CPU kernels) and the HIP kernels' are both compared against it with the tolerances written in the
 * tests (labels identical outside top-2 gaps < 1e-6, descriptors <= 1e-5 relative, top-k indices identical outside
 * ties).
 */
#ifndef ANYLOC_ORACLE_H
#define ANYLOC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_VLAD_NORM_DESCS 1u /* same bit values as include/anyloc_hip.h */
#define ORACLE_VLAD_INTRA_NORM 2u
#define ORACLE_VLAD_EUCLIDEAN 4u

/* F.normalize(x, dim=1): row / max(||row||_2, 1e-12)  (utilities.py:436-437, :782, :960). */
void oracle_l2norm_rows(const float* x, float* out, int64_t rows, int64_t dim);

/* fast-pytorch-kmeans 0.1.6 max_sim (reached from utilities.py:849 predict and :786 fit):
 * mode 0 cosine = rows / (norm + 1e-8) on both sides, mode 1 euclidean = 2ab - a^2 - b^2; labels = first arg-max.
 * gap (nullable) [n]: best similarity minus the runner-up (+inf when K == 1) -- what a fp32 implementation may
 * legitimately resolve the other way when it is below its rounding error. */
void oracle_fpk_labels(const float* x, int64_t n, int64_t D, const float* centers, int64_t K, int mode,
                       int64_t* labels, double* gap);

/* VLAD.generate, hard assignment, for n_img packed images (utilities.py:838-861, :888-890 on the residuals of
 * :959-962): labels from the tokens as passed (:849); residual = F.normalize(token) (flag NORM_DESCS) minus the RAW
 * centre; per used cluster the sum of its members' residuals, optional intra-normalisation (:859-860), unused
 * clusters zero, global L2 normalisation (:889).  out [n_img, K*D]; labels / gap [total tokens], nullable. */
void oracle_vlad_hard(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t D, const float* centers,
                      int64_t K, unsigned flags, float* out, int64_t* labels, double* gap);

/* The same from a GIVEN assignment (the cache-hit branch utilities.py:843-847; also what a checker needs to judge the
 * sums of an implementation whose labels legitimately differ at exact near-ties). */
void oracle_vlad_assigned(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t D, const float* centers,
                          int64_t K, unsigned flags, const int64_t* labels, float* out);

/* One fpk iteration (fit_predict loop body): labels, per-cluster sums and counts, new centres = sums / counts with an
 * empty cluster's 0/0 -> 0, err = sum((new - old)^2).  Returns err. */
double oracle_kmeans_iteration(const float* x, int64_t n, int64_t D, const float* centers, int64_t K, int mode,
                               float* centers_new, int64_t* labels, double* counts);

/* The update half alone, from given labels: per-cluster means, empty -> 0, err; counts [K] nullable. */
double oracle_kmeans_update_from_labels(const float* x, int64_t n, int64_t D, const float* centers, int64_t K,
                                        const int64_t* labels, float* centers_new, double* counts);

/* fpk KMeans.fit from given initial rows (the reference draws them with np.random.choice under seed 42;
 * tests pass the recorded draw): iterate until err <= tol or max_iter (fpk: tol 1e-4, max_iter 100).
 * centers [K, D] out; returns the number of iterations run.  (VLAD.fit, utilities.py:766, :786-787.) */
int oracle_kmeans_fit(const float* x, int64_t n, int64_t D, int64_t K, int mode, const int64_t* init_rows,
                      int max_iter, double tol, float* centers);

/* faiss 1.7.2 IndexFlatIP / IndexFlatL2 add + search (utilities.py:439-450): exact brute force, best first
 * (metric 0: descending inner product; 1: ascending squared L2), ties -> lower database index, k > ndb pads idx
 * with -1 and dist with -inf / +inf.  normalize_db != 0 uses every database row as row / max(||row||, 1e-12)
 * (the F.normalize(db) of :436) without the caller materialising it. */
void oracle_flat_topk(const float* qu, int64_t nq, const float* db, int64_t ndb, int64_t dim, int64_t k, int metric,
                      int normalize_db, float* dist, int64_t* idx);

/* Recall@k loop (utilities.py:451-468) for one positive per query list: gt_off [nq+1] indexes gt [*]; a query is a hit
 * at k if any of idx[q, :k] is among its positives.  recalls [n_k] = hits / nq. */
void oracle_recalls(const int64_t* idx, int64_t nq, int64_t kmax, const int64_t* top_k, int64_t n_k,
                    const int64_t* gt, const int64_t* gt_off, double* recalls);

/* ---- DINOv2 ViT facet extraction: DinoV2ExtractFeatures.__call__ (utilities.py:263-285) on the hub model
 * (facebookresearch/dinov2 @ main, not vendored in the reference: dinov2/models/vision_transformer.py and dinov2/layers,
 * restated as in oracle/dinov2_ref.py).  Weights in the HUB layout (torch Linear [out, in]; SwiGLU w12 = H gate rows then
 * H value rows); activations are kept in fp32 between layers as the fp32 model keeps them, every contraction, LayerNorm
 * statistic and softmax is accumulated in double.  `pos` is the positional table ALREADY interpolated for (H, W)
 * ([1 + N, D], row 0 = CLS), as the C ABI takes it.  facet: 0 query, 1 key, 2 value (slices of blocks[layer].attn.qkv's
 * output, :270-281), 3 token (output of blocks[layer]); CLS row dropped unless use_cls (:270-273); rows L2-normalised when
 * norm != 0 (:282-283).  out [B, N (+1), D]. */
typedef struct oracle_vit_block {
  const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ls1;
  const float *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ls2;   /* fc1 = mlp.fc1 / mlp.w12, fc2 = mlp.fc2 / mlp.w3 */
} oracle_vit_block;
typedef struct oracle_vit_config {
  int32_t dim, depth, heads, ffn_kind /* 0 mlp (erf GELU), 1 swiglu */, ffn_hidden, patch;
} oracle_vit_config;
void oracle_vit_facet(const oracle_vit_config* cfg, const float* patch_w /* [D, 3*P*P] */, const float* patch_b,
                      const float* cls_token, const float* pos, const oracle_vit_block* blocks, const float* img,
                      int64_t B, int64_t H, int64_t W, int32_t layer, int32_t facet, int use_cls, int norm, float* out);

#ifdef __cplusplus
}
#endif
#endif
