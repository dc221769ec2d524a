/*
 * anyloc_hip.h -- C ABI of libanyloc_hip.so: hand-written HIP (gfx950 / CDNA4)
 * kernels for the AnyLoc-VLAD-DINOv2 hot path.
 *
 * The reference (AnyLoc/AnyLoc) is 100 % Python; its "operator boundary" for
 * this path is the class surface of utilities.py.  Each entry point below
 * names the reference interface (file:line under the reference tree) whose
 * arithmetic it replaces.  The Python host in anyloc_amd/ binds these with
 * ctypes (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned memory unless the
 *     parameter is documented "host"; the library never returns memory it
 *     allocated; sizes are explicit; tensors are dense row-major fp32 unless
 *     stated; indices / labels are int64 (the reference's torch.long).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     work is enqueued on it; no call synchronises the device.
 *   - scratch space is caller-provided: query the size with the matching
 *     *_workspace_bytes() and pass a buffer at least that large.
 *   - return value: 0 on success, a negative anyloc_status otherwise;
 *     anyloc_last_error() returns a thread-local description of the last
 *     failure.  No C++ exception crosses the ABI.
 */
#ifndef ANYLOC_HIP_H
#define ANYLOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANYLOC_ABI_VERSION 9

typedef enum anyloc_status {
  ANYLOC_OK = 0,
  ANYLOC_ERR_INVALID_ARG = -1,   /* bad shape / null pointer / unsupported value */
  ANYLOC_ERR_WORKSPACE = -2,     /* workspace too small */
  ANYLOC_ERR_HIP = -3,           /* a HIP runtime call or kernel launch failed */
  ANYLOC_ERR_UNSUPPORTED = -4    /* valid request this build cannot serve */
} anyloc_status;

int anyloc_version(void);
const char* anyloc_last_error(void);

/* ------------------------------------------------------------ options ----
 * Process-wide integer tuning options of the host-side dispatch (tile shapes,
 * thresholds between kernel variants, the unfused data flows kept for A/B
 * measurements and tests).  Defaults are the measured winners; nothing on a
 * call path reads the environment.  The one environment variable,
 * ANYLOC_OPTIONS="name=value,name=value", is parsed once, when the first
 * option is looked up (and again by anyloc_reset_options).  Names:
 *   gemm_f32_cfg x6_cfg h3_cfg        tile configuration of the three GEMM kernels (0 = default)
 *   h3_group_m (8)                    gemm_h3 tile rows per XCD scheduling group
 *   h3_tiny_max (256) h3_deep_max (320) h3_deep2_max (500)
 *                                     gemm_h3 small-problem tile / ring-depth thresholds (tile counts)
 *   h3_epi_lds (1)                    LayerScale-residual epilogue with 16-byte accesses through LDS
 *   ln_rows_per_wave (0)              layernorm_h2: 1 / 2 / 4 rows per wave at every size (0 = by ln_small_rows)
 *   ln_small_rows (4096)              layernorm_h2: one row per wave below this many rows
 *   ln_waves (8)                      layernorm_h2, two rows per wave: waves per block (8 = 16 rows per block, 512-byte store runs; 4)
 *   ln_direct_rows (1200)             layernorm_h2: below this many rows one single-wave workgroup per row writes the image
 *                                     straight from registers (no LDS tile, no barriers); 0 = never
 *   h3_fuse (1) x6_fuse (1)           activations stay in fp16 / bf16 planes between kernels
 *   h3_min_rows (0) x6_min_rows (1600) token rows below which a split-mode forward uses the fp32-MFMA kernels
 *   attn_cfg (0) attn_x6 (-1)         anyloc_attention: kernel variant; split-bf16 products (1 always, 0 never, -1 caller)
 *   attn_h3_qg (1)                    attention of the two-term fp16 forward: 1 = four waves of 32 queries per workgroup, 2 = two waves of
 *                                     64 queries (A/B: measured slower, profiles/r05_attention_qg2.log)
 *   attn_h3_ks (0)                    the same kernel's key splits: 2 = two query waves x two key waves per workgroup (half the serial
 *                                     chain of key tiles; partial sums meet in LDS), 1 = none, 0 = 2 when all workgroups are resident
 *   vlad_parts (0 = auto) vlad_two_pass (0) vlad_fused_v (0) kmeans_fused_v (0)
 *                                     which VLAD / k-means kernel serves a call
 *   topk_screen (-1)                  many queries against long rows: the screened search described at anyloc_topk_search_index_rows,
 *                                     -1 = where it pays (>= 256 queries, >= 16 384 rows, dim >= 4096), 1 = wherever the shape allows,
 *                                     0 = never (every panel on the three-product scores)
 *   h3_ln_lead (0)                    batched calls (>= 64 tile rows of 128 tokens): 1 = LayerNorm 1 / 2 run as LEAD workgroups of 16 rows
 *                                     interleaved with the tiles of the qkv / fc1 (w12) GEMM's own launch (per XCD: the lead work of the next
 *                                     tile-row group sits among the tiles of the current one; write-through stores, one ticket per tile row);
 *                                     same bits as the separate launch -- and 5 % SLOWER end to end (profiles/r06_batched_ln_lead.log: a lead
 *                                     workgroup holds a GEMM slot for ~50 us and its vector work takes issue slots from the matrix work)
 *   h3s_ln_lead (0)                   one image per call: 1 = LayerNorm runs as the LEAD role of its consumer GEMM's launch (LN1 + qkv, LN2 + w12:
 *                                     the first workgroups normalise the rows write-through and count them per row tile, a GEMM workgroup waits
 *                                     for its tile's count) instead of as a launch of its own; same bits, 163 launches per ViT-g forward instead
 *                                     of 225 -- and the same 5.55 ms (profiles/r06_b1_ln_lead.log: the hand-off costs what the launch cost)
 *   vlad_gather_v (0)                 one-pass VLAD kernel at D = 1536: variants of the register-indexed gather kept for the round-6
 *                                     hazard study (tools/stress_vlad.py); 0 = the shipped arithmetic
 *   kmeans_max_chunks (0 = two per CU)
 *   h3_mfma16 (-1)                    plain-store two-term fp16 GEMMs of >= 256 tiles of 256 x 256 on the 16 x 16 x 32 MFMA kernel
 *                                     (csrc/gemm_h3m.hip): -1 when the contraction is >= 4096 long (retrieval panels), 0 never, 1 always
 *   h3_swiglu_t (1)                   read by the Python host when a model is built: SwiGLU fc1 image in the 16-channel block
 *                                     layout (anyloc_vit_block_h2.fc1_layout = 1: epilogue straight from transposed accumulators)
 *   h3_fast_silu (1)                  fused SwiGLU epilogue: SiLU on the hardware exp2 / rcp (1 ulp each)
 *   topk_fewq_x6 (2)                  anyloc_topk with <= 64 queries, the database read once and split on the fly: 2 = two fp16 planes
 *                                     under a running power-of-two row scale (three fp16 MFMA products, csrc/scores_h3.hip),
 *                                     1 = three bf16 planes (six bf16 products, csrc/scores_x6.hip), 0 = fp32 MFMA
 *   topk_fewq_qdma (1)                few-query fp16 scores: the queries are split into planes once per call and DMA'd into LDS per slab;
 *                                     0 = split by the staging lanes at every slab (same bits)
 *   topk_h3 (-1)                      anyloc_topk score panels on the two-term fp16 GEMM: -1 where it pays, 0 never, 1 wherever possible
 *   h3_patch (1)                      ANYLOC_VIT_SPLIT_FP16: the patch embedding runs on the two-term fp16 GEMM too (weights quantised by
 *                                     anyloc_vit_attach_h2); 0 = on the fp32 matrix-core GEMM
 *   h3s_enable (1)                    small-M plans of the two-term fp16 GEMM (csrc/gemm_h3s.hip: tile shape, ring depth and split-K
 *                                     factor per GEMM shape when a call has one or a few images); 0 = the round-3 small-batch kernels
 *   h3s_cfg (-1) h3s_ksplit (0) h3s_kb (0) h3s_stages (0) h3s_mask (31)
 *                                     overrides of that plan table for sweeps: tile configuration id, split-K factor, k-blocks per
 *                                     ring stage, ring depth (3 or 6); mask bit 0 qkv, 1 proj, 2 fc1 / w12, 3 fc2, 4 other GEMMs
 *   h3s_w12_tall (1)                  the FFN input GEMM (N >= 8192) of ONE image of 385 ... 576 token rows on 192 x 128 tiles: three
 *                                     row tiles, one workgroup per CU; 0 = 128 x 128 tiles (320 workgroups: two on 64 of the CUs)
 * Unknown names are rejected (ANYLOC_ERR_INVALID_ARG).  Not thread-safe against
 * concurrent launches that read the option being changed. */
int anyloc_set_option(const char* name, int64_t value);
int anyloc_get_option(const char* name, int64_t* value /*host*/);
int anyloc_reset_options(void);      /* defaults, then ANYLOC_OPTIONS */

/* ---------------------------------------------------------------- rows ---
 * out[r,:] = x[r,:] / max(||x[r,:]||_2, eps)      (torch F.normalize, eps 1e-12)
 * replaces: F.normalize calls at utilities.py:283, :436-437, :785, :960.
 * x and out may alias. */
int anyloc_l2norm_rows(const float* x, float* out, int64_t rows, int64_t dim,
                       float eps, void* stream);

/* ------------------------------------------------------------ ingest ----
 * uint8 HWC images -> float32 CHW model input in one pass: centre crop to
 * crop_h x crop_w (torchvision rule: top = round_half_even((H - crop_h) / 2)),
 * x / 255, (x - mean[c]) / std[c].
 * replaces: ToTensor + Normalize (dvgl_benchmark/datasets_ws.py:20-23) and the
 *   CenterCrop to multiples of 14 (scripts/dino_v2_vlad.py:173-176,
 *   demo/anyloc_vlad_generate.py:179-181) that precede the extractor.
 *   img_hwc [B,H,W,3] uint8 (device), mean3/std3: HOST arrays of 3 floats,
 *   out [B,3,crop_h,crop_w] float32 (device). */
int anyloc_preprocess_u8(const unsigned char* img_hwc, int64_t batch, int64_t height,
                         int64_t width, int64_t crop_h, int64_t crop_w,
                         const float* mean3, const float* std3, float* out, void* stream);

/* Optional downscale of the ingest (demo/anyloc_vlad_generate.py:163-181: images whose longer side exceeds
 * max_img_size are resized with torchvision's BICUBIC on the normalised float tensor, then centre-cropped to
 * multiples of 14).  in [planes, H, W] fp32 (planes = B*3) is resized to (out_h, out_w) with torch's bicubic
 * convolution (A = -0.75, align_corners = False, no antialias) and only the window
 * [crop_top, crop_top + crop_h) x [crop_left, crop_left + crop_w) of the resized image is written:
 * out [planes, crop_h, crop_w]. */
int anyloc_resize_bicubic(const float* in, int64_t planes, int64_t height, int64_t width,
                          int64_t out_h, int64_t out_w, int64_t crop_top, int64_t crop_left,
                          int64_t crop_h, int64_t crop_w, float* out, void* stream);

/* ------------------------------------------------- split-bf16 matmul ----
 * The same contraction as anyloc_gemm_nt (torch Linear layout, reference
 * utilities.py:269 model forward) on the bf16 matrix cores with fp32-level
 * accuracy: every fp32 operand is the exact sum of three bf16 planes and the six
 * leading plane products are accumulated in fp32 (csrc/gemm_x6.hip).
 *   anyloc_x3_bytes   size of the plane image of an fp32 matrix [rows, K]
 *   anyloc_split_x3   fp32 row-major [rows, K] (leading dim ldx) -> plane image
 *   anyloc_gemm_nt_x6 C[M,N] = A * W^T (+ bias[N]) from the plane images of A [M,K]
 *                     and W [N,K]; C fp32, leading dim ldc. */
size_t anyloc_x3_bytes(int64_t rows, int64_t K);
int anyloc_split_x3(const float* x, int64_t ldx, int64_t rows, int64_t K, void* x3,
                    void* stream);
int anyloc_gemm_nt_x6(const void* a3, const void* w3, const float* bias, float* C,
                      int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);

/* ------------------------------------------------ fp16 two-term matmul ----
 * Same contraction with THREE fp16 matrix-core products per k-step: every operand
 * row is scaled by a power of two into [2^14, 2^15) and split into two fp16 planes
 * (22 mantissa bits); the epilogue descales with inv_a[row] * inv_w[col]
 * (csrc/gemm_h3.hip, tools/split_fp16_study.py).
 *   anyloc_split_h2    fp32 [rows, K] (K % 16 == 0, K <= 4096, ldx % 4 == 0) -> plane image
 *                      (anyloc_h2_bytes) + inv_scale[rows] (2^-e of each row)
 *   anyloc_gemm_nt_h3  C[M,N] = A W^T (+ bias[N]) from the two images */
size_t anyloc_h2_bytes(int64_t rows, int64_t K);
int anyloc_split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2,
                    float* inv_scale, void* stream);
int anyloc_gemm_nt_h3(const void* a2, const float* a_inv, const void* w2,
                      const float* w_inv, const float* bias, float* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, void* stream);
/* ABI 9 (diagnostic, host only -- no device is touched): the batched ViT forward runs LayerNorm as LEAD workgroups inside its
 * consumer GEMM's launch (option h3_ln_lead; csrc/tile_order.hpp: LeadPlan).  The plan of a shape -- tiles_m x tiles_n tiles of
 * 128 rows, scheduling groups of group_m tile rows, M rows -- is simulated on the host before it is used: every GEMM tile exactly
 * once, every row normalised exactly once, every producer ahead of its consumers in workgroup-id order.  Returns 1 when the plan
 * passes (and *grid = workgroups of the launch), 0 when the forward keeps LayerNorm as a launch of its own for this shape. */
int anyloc_h3_lead_plan_check(int32_t tiles_m, int32_t tiles_n, int32_t group_m, int64_t M, uint32_t* grid);

/* ------------------------------------------------------------ pooling ----
 * One global descriptor per image from its patch tokens, without VLAD:
 *   ANYLOC_POOL_AVG      mean over tokens          (scripts/dino_v2_gp.py:130-131)
 *   ANYLOC_POOL_MAX      max over tokens           (scripts/dino_v2_gp.py:132-133)
 *   ANYLOC_POOL_GEM      x = mean(t^p); |x|^(1/p) * sign(x)   (scripts/dino_v2_gem.py:186-188)
 *   ANYLOC_POOL_GEM_ABS  mean(|t|^p)^(1/p)         (scripts/dino_v2_gem.py:174-175)
 *   tokens  [total_tokens, D]; image i = rows offsets[i] .. offsets[i+1]
 *           (offsets: n_img+1 int64 on the device), or offsets == NULL and every
 *           image has n_tok rows.  out [n_img, D].  p is read by the GeM modes only. */
#define ANYLOC_POOL_AVG 0
#define ANYLOC_POOL_MAX 1
#define ANYLOC_POOL_GEM 2
#define ANYLOC_POOL_GEM_ABS 3
int anyloc_pool_tokens(const float* tokens, const int64_t* offsets, int64_t n_img,
                       int64_t n_tok, int64_t D, int mode, float p, float* out,
                       void* stream);

/* ------------------------------------------------------------ PCA fit ----
 * reference utilities.py:522-586 (called at scripts/dino_v2_vlad.py:357-369): sklearn PCA(lower_dim,
 * svd_solver='full') = a LAPACK SVD of the centred [n, f] descriptor matrix.  The host (anyloc_amd/pca.py)
 * reduces it to the smaller symmetric matrix of the centred data, a symmetric eigenproblem, and -- from
 * the Gram side -- one back-projection; both products are formed in FLOAT64 on v_mfma_f64_16x16x4_f64
 * (they square the condition number), reading the fp32 data where it lies and centring it on the way:
 *   Xc = (double) X[n, f] - mean[f]   (mean: float64, NULL = no centring)
 *   anyloc_pca_gram_f64  side 0: out[n, n] = Xc Xc^T (Gram);  side 1: out[f, f] = Xc^T Xc (scatter)
 *   anyloc_pca_axes_f64  out[k, f] = U^T Xc,  U[c, i] = vec[c * sample_stride + i * axis_stride] (float64): component
 *                        c (a sample) of eigenvector i of the Gram matrix, in either storage order
 * ABI 7. */
int anyloc_pca_gram_f64(const float* X, int64_t n, int64_t f, const double* mean, int side,
                        double* out, void* stream);
int anyloc_pca_axes_f64(const double* vec, int64_t sample_stride, int64_t axis_stride, int64_t k,
                        const float* X, int64_t n, int64_t f, const double* mean, double* out,
                        void* stream);

/* ------------------------------------------------------------- matmul ----
 * C[M,N] = A[M,K] * W[N,K]^T (+ bias[N] when bias != NULL): the exact-fp32 MFMA GEMM
 * (v_mfma_f32_32x32x2_f32) behind retrieval, the VLAD / k-means scores, the patch
 * embedding and the ViT blocks of small batches (torch Linear layout: both
 * operands K-contiguous).  lda/ldw/ldc are row strides in floats.  Requires
 * K % 4 == 0 and 16-byte aligned rows.  Exposed for tests/benchmarks. */
int anyloc_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw,
                   const float* bias, float* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, void* stream);

/* Building blocks of the ViT forward, exposed for unit tests:
 * LayerNorm over the last dim (eps as given; DINOv2 uses 1e-6) and multi-head
 * self-attention softmax((q/8) k^T) v with head_dim 64 on a packed QKV buffer
 * [B*T, 3*D] (q | k | v, heads contiguous) -> out [B*T, D]. */
int anyloc_layernorm(const float* x, float* y, const float* weight, const float* bias,
                     int64_t rows, int64_t dim, float eps, void* stream);
int anyloc_attention(const float* qkv, float* out, int64_t batch, int64_t tokens,
                     int64_t dim, int64_t heads, void* stream);
/* The attention of the two-term fp16 ("h3") forward, exposed for unit tests: the same
 * softmax((q/8) k^T) v from a packed fp32 QKV buffer, computed as three fp16 matrix-core
 * products per contraction on per-(head, 32-row group) scaled operand tiles, and written
 * as the two-plane fp16 image + per-row 2^-e that anyloc_gemm_nt_h3 takes as its A operand
 * (out_img: anyloc_h2_bytes(batch*tokens, dim) bytes, out_inv: batch*tokens floats). */
size_t anyloc_attention_h3_workspace_bytes(int64_t batch, int64_t tokens, int64_t heads);
int anyloc_attention_h3(const float* qkv, void* out_img, float* out_inv, int64_t batch,
                        int64_t tokens, int64_t dim, int64_t heads, void* workspace,
                        size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------- VLAD ----
 * Hard-assignment VLAD of `n_img` images in one call.
 * replaces: VLAD.generate / generate_multi / generate_res_vec (hard mode),
 *   utilities.py:819-861, :888-890, :892-926, :928-972.
 *   tokens   [total_tokens, D]   patch descriptors of all images, packed
 *   offsets  [n_img + 1] int64   image i owns token rows offsets[i]..offsets[i+1]
 *                                 (ragged images allowed, empty images allowed)
 *   centers  [K, D]              raw k-means centroids (c_centers)
 *   out      [n_img, K*D]        normalised VLAD descriptors
 *   labels   [total_tokens] int64 or NULL: hard assignment of every token
 * flags: ANYLOC_VLAD_NORM_DESCS re-normalises tokens before the residual
 *   (utilities.py:959-960); ANYLOC_VLAD_INTRA_NORM normalises each cluster
 *   block (:859-860).  Labels are argmax_k of the fast-pytorch-kmeans cosine
 *   score of the tokens as passed (:849) -- ties -> lowest k; with
 *   ANYLOC_VLAD_EUCLIDEAN, of its euclidean similarity 2ab - a^2 - b^2 (the metric
 *   kmeans.predict uses when the VLAD object was built with dist_mode="euclidean"). */
#define ANYLOC_VLAD_NORM_DESCS 1u
#define ANYLOC_VLAD_INTRA_NORM 2u
#define ANYLOC_VLAD_EUCLIDEAN 4u   /* labels by the fpk euclidean similarity (VLAD(dist_mode="euclidean")) instead of cosine */
/* ABI 6: workgroups per image of the one-pass kernel as the CALLER's choice, p in 1..64 (0 = the library's own choice from
 * the image count, anyloc_vlad_auto_parts).  An image's partial sums are added in part order, so its bits depend on p and
 * on nothing else: a caller that hands one batch over in several calls (the host surface streams a CPU tensor in
 * ~256 MB pieces) passes the count the WHOLE batch would have got and receives the bits of the one-call result.  Ignored
 * where the general two-pass path serves the call.  p must not exceed the count the workspace was sized for
 * (anyloc_vlad_workspace_bytes sizes for the library's choice at the call's n_img; fewer images -> more parts, so the
 * whole batch's count always fits a piece's workspace). */
#define ANYLOC_VLAD_PARTS(p) (((unsigned)(p) & 0x7fu) << 8)
size_t anyloc_vlad_workspace_bytes(int64_t total_tokens, int64_t n_img,
                                   int64_t D, int64_t K);
int anyloc_vlad_auto_parts(int64_t total_tokens, int64_t n_img, int64_t D, int64_t K);   /* the library's choice (1 where the one-pass kernel does not apply) */
/* ABI 8: the workspace of a call that passes ANYLOC_VLAD_PARTS(parts) -- sized for the larger of the library's own count and
 * the caller's (anyloc_vlad_workspace_bytes alone covers a caller's count only while it does not exceed the library's). */
size_t anyloc_vlad_workspace_bytes_parts(int64_t total_tokens, int64_t n_img, int64_t D, int64_t K, int32_t parts);
int anyloc_vlad_hard(const float* tokens, const int64_t* offsets, int64_t n_img,
                     int64_t total_tokens, int64_t D, const float* centers,
                     int64_t K, unsigned flags, float* out, int64_t* labels,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Soft-assignment VLAD with the reference's summation (utilities.py:862-887):
 * block k = sum_q sum_c softmax(temp*cos(x_q,c))[q,k] * (xhat_q - c_c). */
int anyloc_vlad_soft(const float* tokens, const int64_t* offsets, int64_t n_img,
                     int64_t total_tokens, int64_t D, const float* centers,
                     int64_t K, float soft_temp, unsigned flags, float* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The soft-assignment weights alone (what the reference caches as <id>_s.pt, utilities.py:870-878):
 * weights[n, k] = softmax_k(soft_temp * F.cosine_similarity(x_n, c_k)), [n_tok, K] fp32, K <= 64.
 * Workspace: anyloc_vlad_workspace_bytes(n_tok, 1, D, K). */
int anyloc_vlad_soft_weights(const float* tokens, int64_t n_tok, int64_t D, const float* centers,
                             int64_t K, float soft_temp, float* weights, void* workspace,
                             size_t workspace_bytes, void* stream);

/* The residual tensor itself (VLAD.generate_res_vec, utilities.py:928-972; the reference materialises it for every
 * image, this library only on request): out[n,k,:] = normalise(x_n) - centers[k]  ([n_tok, K, D]; flag
 * ANYLOC_VLAD_NORM_DESCS as above; the other flags are ignored). */
int anyloc_vlad_residuals(const float* tokens, int64_t n_tok, int64_t D, const float* centers,
                          int64_t K, unsigned flags, float* out, void* stream);

/* VLAD of ONE image from a GIVEN assignment -- the cache-hit branches of VLAD.generate
 * (utilities.py:843-847 hard: labels int64 [n_tok]; :864-868 soft: weights fp32 [n_tok, K]; pass exactly one,
 * the other NULL): per-cluster sums of normalise(x_n) - centers (hard: own cluster; soft: the reference's
 * all-cluster sum), intra-norm, global norm.  The [n_tok, K, D] residual tensor is never formed: tokens are
 * what a lazy cache stores.  Workspace: anyloc_vlad_workspace_bytes(n_tok, 1, D, K). */
int anyloc_vlad_assigned(const float* tokens, int64_t n_tok, int64_t D, const float* centers,
                         int64_t K, const int64_t* labels, const float* soft_weights,
                         unsigned flags, float* out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ------------------------------------------------------------ k-means ----
 * One fast-pytorch-kmeans iteration (cosine or euclidean mode):
 *   labels[n] = argmax_k sim(x_n, c_k);  sums[k,:] = sum_{labels[n]==k} x_n;
 *   counts[k] = #{n : labels[n]==k}.
 * replaces: fpk KMeans.fit_predict loop body reached from VLAD.fit,
 *   utilities.py:766, :786 (and predict at :849).  The division by counts, the
 *   NaN->0 rule, the tolerance test and (multi-GPU) the all-reduce of
 *   sums/counts are done by the host (anyloc_amd/kmeans.py).
 *   mode 0 = cosine (rows/(norm+1e-8)), 1 = euclidean (2ab - a^2 - b^2). */
size_t anyloc_kmeans_workspace_bytes(int64_t n, int64_t D, int64_t K);
int anyloc_kmeans_step(const float* x, int64_t n, int64_t D, const float* centers,
                       int64_t K, int mode, float* sums, float* counts,
                       int64_t* labels, void* workspace, size_t workspace_bytes,
                       void* stream);

/* The rest of the iteration (ABI 5): centers_new[k,:] = sums[k,:] / counts[k] (empty cluster: 0 / 0 = NaN -> 0, fpk's
 * rule), *err = sum((centers_new - centers_old)^2) as float64 -- the quantity fpk compares with its tolerance
 * (utilities.py:766,786 -> fast_pytorch_kmeans fit_predict).  With several GPUs the host all-reduces sums / counts between
 * anyloc_kmeans_step and this call.  All pointers device; centers_new may not alias centers_old. */
int anyloc_kmeans_update(const float* sums, const float* counts, const float* centers_old, int64_t K, int64_t D,
                         float* centers_new, double* err, void* stream);

/* -------------------------------------------------------------- top-k ----
 * Exact brute-force search of `nq` queries against `ndb` database rows.
 * replaces: faiss IndexFlatIP / IndexFlatL2 add + search,
 *   utilities.py:439-450 (called from get_top_k_recall).
 *   metric 0: inner product, best = largest;  1: squared L2, best = smallest.
 *   dist [nq,k] fp32 and idx [nq,k] int64, best first; idx = index_base + row.
 *   If k > ndb the tail is padded with idx -1 (faiss behaviour).
 *   Ties are broken towards the lower database index.
 *   flags: ANYLOC_TOPK_NORMALIZE_DB -- the database is given RAW and every row is used as
 *   row / max(||row||_2, 1e-12), i.e. the F.normalize(db) of utilities.py:436 is applied to the
 *   scores instead of materialising a normalised copy of the database (queries are taken as given).
 *   With <= 64 queries of >= 4096 dimensions the database is streamed once by a split-K launch
 *   (HBM-bound); otherwise scores are fp32-MFMA GEMM panels (MFMA-bound). */
#define ANYLOC_TOPK_NORMALIZE_DB 1u
size_t anyloc_topk_workspace_bytes(int64_t nq, int64_t ndb, int64_t dim, int64_t k);
int anyloc_topk(const float* queries, int64_t nq, const float* db, int64_t ndb,
                int64_t dim, int64_t k, int metric, unsigned flags, int64_t index_base,
                float* dist, int64_t* idx, void* workspace,
                size_t workspace_bytes, void* stream);

/* ABI 8: the database side prepared ONCE -- faiss' `index.add(db)` (utilities.py:441-442 / :446-447) apart from
 * `index.search(qu, k)` (:450).  anyloc_topk quantises every 8192-row database panel into the two-plane fp16 operand image of
 * its score GEMM on EVERY call (two reads + one write of the database: 14 ms of a 320 ms retrieval on a 125 000 x 49 152
 * shard); a caller that searches one database several times builds those images, the rows' power-of-two scales and their
 * sums of squares once into its own buffer (anyloc_topk_index_bytes: 4 bytes per element -- the size of the fp32 rows, which
 * the search no longer touches) and passes it to anyloc_topk_search_index: same arguments and results as anyloc_topk (the
 * same score kernels on the same operands: bit-identical lists), every query count on the fp16 panels.  dim % 16 == 0;
 * anyloc_topk_index_bytes returns 0 for a shape the panels do not serve (use anyloc_topk). */
int anyloc_topk_path(int64_t nq, int64_t ndb, int64_t dim);   /* which scoring path anyloc_topk takes for this shape: 0 = fp32-MFMA
                                                                  panels, 1 = few queries (database streamed once), 2 = fp16 panels */
size_t anyloc_topk_index_bytes(int64_t ndb, int64_t dim);
int anyloc_topk_index_build(const float* db, int64_t ndb, int64_t dim, void* index, size_t index_bytes, void* stream);
size_t anyloc_topk_index_workspace_bytes(int64_t nq, int64_t ndb, int64_t dim, int64_t k);
int anyloc_topk_search_index(const float* queries, int64_t nq, const void* index, int64_t ndb, int64_t dim, int64_t k,
                             int metric, unsigned flags, int64_t index_base, float* dist, int64_t* idx, void* workspace,
                             size_t workspace_bytes, void* stream);
/* ABI 9: the same search with the database's fp32 rows next to the prepared index (db: [ndb, dim], the rows the index was built
 * from).  With the rows at hand the SCREENED search can run (option topk_screen, csrc/scores_screen.hip; anyloc_topk runs it too):
 * every panel is scored on the LEADING fp16 planes alone (one matrix-core product per k instead of three) under a proven bound
 * |s - s~| <= eps |q| |d|, the k-th best screened value of a query minus twice its bound is a threshold no row of the true list can
 * fall below, and the few tens of rows above it are re-scored from the fp32 rows with float64 sums and ranked (value, then lower
 * index).  One bound per query: its norm and measured residual x the largest row norm the compared value sees (1 with
 * ANYLOC_TOPK_NORMALIZE_DB, the largest raw row norm without it: rows of very different raw norms make it loose).  k <= 128, dim <= 49 152; a query
 * with more than 512 rows inside its bound makes the call run the unscreened search instead (one 4-byte read-back per call decides).
 * Lists: the rows of the exact search; distances within 1e-6 of it (more accurate, not bit-identical).  db == NULL: as
 * anyloc_topk_search_index. */
int anyloc_topk_search_index_rows(const float* queries, int64_t nq, const float* db, const void* index, int64_t ndb, int64_t dim,
                                  int64_t k, int metric, unsigned flags, int64_t index_base, float* dist, int64_t* idx,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- ViT ----
 * DINOv2 ViT forward with early exit at the last tapped layer.
 * replaces: DinoV2ExtractFeatures.__call__, utilities.py:263-285 (the hub
 *   model forward at :269, the hooked facet at :270-281, F.normalize at :283).
 * Weights are caller-owned device tensors in torch Linear layout
 * ([out,in] row-major), handed over once at creation; the handle stores the
 * pointers, it does not copy.  */
typedef struct anyloc_vit anyloc_vit_t;

typedef struct anyloc_vit_block_weights {
  const float *norm1_w, *norm1_b;       /* [D] */
  const float *qkv_w, *qkv_b;           /* [3D,D], [3D] */
  const float *proj_w, *proj_b;         /* [D,D], [D] */
  const float *ls1;                     /* [D] LayerScale gamma */
  const float *norm2_w, *norm2_b;       /* [D] */
  /* ffn_kind 0 (mlp):   fc1 [H,D],[H]; fc2 [D,H],[D]
   * ffn_kind 1 (swiglu): fc1 = w12 with rows pair-interleaved per 64 outputs
   *   (32 gate rows then the 32 matching value rows) [2H,D],[2H]; fc2 = w3 */
  const float *fc1_w, *fc1_b;
  const float *fc2_w, *fc2_b;
  const float *ls2;                     /* [D] */
} anyloc_vit_block_weights;

typedef struct anyloc_vit_config {
  int32_t dim;          /* D: 384 / 768 / 1024 / 1536 */
  int32_t depth;        /* number of blocks supplied */
  int32_t heads;        /* head_dim must be 64 */
  int32_t ffn_kind;     /* 0 = mlp (exact GELU), 1 = swiglu */
  int32_t ffn_hidden;   /* H */
  int32_t patch;        /* 14 */
  int32_t patch_k_pad;  /* row length of patch_w (3*14*14=588 padded to a multiple of 4; zeros) */
} anyloc_vit_config;

int anyloc_vit_create(anyloc_vit_t** out, const anyloc_vit_config* cfg /*host*/,
                      const float* patch_w /*[D,patch_k_pad]*/, const float* patch_b,
                      const float* cls_token /*[D]*/,
                      const anyloc_vit_block_weights* blocks /*host array [depth]*/);
void anyloc_vit_destroy(anyloc_vit_t* h);

/* facets (reference _DINO_FACETS, utilities.py:218) */
#define ANYLOC_FACET_QUERY 0
#define ANYLOC_FACET_KEY 1
#define ANYLOC_FACET_VALUE 2
#define ANYLOC_FACET_TOKEN 3
/* forward flags */
/* Optional split-bf16 execution of the block GEMMs (csrc/gemm_x6.hip): attach the
 * three-plane images (anyloc_split_x3 of qkv.weight [3D,D], proj.weight [D,D], the
 * fc1 / interleaved w12 matrix and fc2 / w3, all caller-owned, alive while attached)
 * and pass ANYLOC_VIT_SPLIT_BF16 to anyloc_vit_forward.  blocks == NULL detaches. */
typedef struct anyloc_vit_block_x3 {
  const void* qkv_w3;
  const void* proj_w3;
  const void* fc1_w3;
  const void* fc2_w3;
} anyloc_vit_block_x3;
int anyloc_vit_attach_x3(anyloc_vit_t* h, const anyloc_vit_block_x3* blocks /*host array [depth]*/);
#define ANYLOC_VIT_SPLIT_BF16 8u     /* block GEMMs on the bf16 matrix cores, fp32-level accuracy */

/* Two-term fp16 execution of the block GEMMs (csrc/gemm_h3.hip: three fp16 matrix-core
 * products per k-step, row-scaled operands): attach anyloc_split_h2 images + row scales
 * of the same four matrices and pass ANYLOC_VIT_SPLIT_FP16 (takes precedence over
 * ANYLOC_VIT_SPLIT_BF16).  blocks == NULL detaches. */
typedef struct anyloc_vit_block_h2 {
  const void* qkv_w2;  const float* qkv_inv;
  const void* proj_w2; const float* proj_inv;
  const void* fc1_w2;  const float* fc1_inv;
  const void* fc2_w2;  const float* fc2_inv;
  /* Bounds that let the fc1 epilogue quantise the FFN activation for fc2 without an fp32 round trip
   * (|fc1_j(y)| <= ||y||_2 max_j ||W_j||_2 + max_j |b_j|):  {max gate-row (mlp: fc1-row) L2 norm, max |gate bias|,
   * max value-row L2 norm, max |value bias|}; the value pair is 0 for the GELU mlp.  All zero = not provided: the
   * activation is then written as fp32 and quantised by a separate pass. */
  float fc1_bound[4];
  /* SwiGLU only (ABI 4).  fc1_layout = 0: the rows of fc1_w2 are interleaved 32 gate / 32 value (the layout of
   * anyloc_vit_block_weights.fc1_w; fc1_b2 may be NULL = use fc1_b).  fc1_layout = 1: every 32-row block of fc1_w2
   * holds 16 consecutive hidden channels c .. c+15 so that the MFMA accumulators of the TRANSPOSED product (weights as
   * the A operand) give one lane a token row and 8 consecutive gate / value channels: block row t = 16 v + 8 q + 4 h + i
   * (v = 0 gate / 1 value, q, h in {0, 1}, i < 4) is channel c + 8 h + 4 q + i -- the SwiGLU epilogue then writes its
   * 16-byte image chunks straight from registers.  fc1_b2: the fc1 bias in the row order of fc1_w2 (device, [2 hidden]). */
  const float* fc1_b2;
  int32_t fc1_layout;
  int32_t reserved;
} anyloc_vit_block_h2;
int anyloc_vit_attach_h2(anyloc_vit_t* h, const anyloc_vit_block_h2* blocks /*host array [depth]*/);

/* FFN-bound telemetry of the two-term fp16 forward (ABI 5; per image and without an extra pass over the image since ABI 8).
 * The fused fc1 epilogue quantises the hidden activation against an UPPER BOUND of its row (fc1_bound above); a bound more
 * than ~2^18 above the row's real maximum would cost low bits.  With a device array set here, every later forward also
 * reports, per executed block, max over token rows of 2^15 / (largest scaled magnitude the row actually holds in the fc2
 * operand image) -- how loose the bound was: the epilogue leaves the rows' maxima by atomicMax (no extra read of the image),
 * one small launch at the end of the forward reduces them.  per_image = 0: ffn_looseness[depth], one figure per block over
 * all rows of the call; per_image = 1: ffn_looseness[depth][batch] (row-major, the call's batch), one figure per block and
 * image, so that a caller can decide per IMAGE -- independent of what else is in the batch and of earlier calls.  A block that
 * did not run fused (exact mode, not executed) reports 0.  NULL switches it off (default).
 * anyloc_vit_block_ffn_exact(h, layer, 1) makes that block write its activation as fp32 and quantise it against the exact
 * row maximum instead (the data flow of fc1_bound = 0).  The Python host checks EVERY call: images with a block above 2^14
 * are run again with exactly their own loose blocks switched, and the switches are cleared after the call. */
int anyloc_vit_set_telemetry(anyloc_vit_t* h, float* ffn_looseness /*device [depth] / [depth][batch] or NULL*/, int32_t per_image);
int anyloc_vit_block_ffn_exact(anyloc_vit_t* h, int32_t layer, int32_t exact);
#define ANYLOC_VIT_SPLIT_FP16 16u    /* block GEMMs as three fp16 products, fp32-level accuracy */

#define ANYLOC_VIT_USE_CLS 1u        /* keep the CLS row (utilities.py:270-273) */
#define ANYLOC_VIT_NORM_TAPS 2u      /* L2-normalise each tap (utilities.py:282-283) */
#define ANYLOC_VIT_NORM_CONCAT 4u    /* L2-normalise the concatenated taps again
                                        (scripts/dino_v2_vlad_viz.py:175-196) */

size_t anyloc_vit_workspace_bytes(const anyloc_vit_t* h, int64_t batch,
                                  int64_t img_h, int64_t img_w);
/*   img  [B,3,H,W]  ImageNet-normalised, H and W multiples of 14
 *   pos  [1+N, D]   positional table already interpolated for (H,W), row 0 = CLS
 *   tap_layers / tap_facets: host arrays of n_taps entries, layers ascending
 *   out  [B, N (+1 with USE_CLS), n_taps*D]  taps concatenated on the feature axis */
int anyloc_vit_forward(anyloc_vit_t* h, const float* img, int64_t batch,
                       int64_t img_h, int64_t img_w, const float* pos,
                       int32_t n_taps, const int32_t* tap_layers,
                       const int32_t* tap_facets, unsigned flags, float* out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Name and average device time (ms, HIP events on the launch stream) of the
 * kernels issued by the most recent call with profiling enabled; used by
 * bench.py for the live roofline figure.  enable: 0/1. */
int anyloc_profile_enable(int enable);
/* ABI 5: bracket only the launches whose profiling tag equals `tag` (NULL or "" = all).  Two HIP events around every launch cost
 * a launch-bound sequence real time (~3 us per event on the queue); bench.py brackets only the kernel whose roofline it
 * reports inside its timed region and takes the per-kernel table from untimed steps. */
int anyloc_profile_filter(const char* tag);
int anyloc_profile_reset(void);
/* writes up to `cap` bytes of a JSON object {"kernel": {"calls":n,"ms":t,"flops":f,"bytes":b}, ...} */
int anyloc_profile_dump(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ANYLOC_HIP_H */
