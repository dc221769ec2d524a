// DINOv2 ViT forward with early exit at the last tapped layer.
//
// replaces: DinoV2ExtractFeatures.__call__ (reference utilities.py:263-285):
// the torch.hub DINOv2 forward at :269, the forward-hook capture of
// blocks[L].attn.qkv / blocks[L] (:245-252, :258-261), CLS drop (:270-273),
// facet slice (:274-281) and F.normalize (:282-283).
//
// The reference runs all blocks, the final norm and the head and throws the
// result away; only the hooked tensor is used.  Here execution stops at the
// last tap, and when a q/k/v tap sits in the last executed block only that
// facet's third of the QKV projection is computed.
//
// Per block (all GEMMs on gemm_f32.hip, fp32 MFMA, fused epilogues):
//   y   = LN1(x)                                    layernorm
//   qkv = y Wqkv^T + b                              EPI_STORE
//   a   = softmax((q/8) k^T) v                      attention.hip
//   x  += ls1 * (a Wproj^T + b)                     EPI_LS_RESID (in place)
//   y   = LN2(x)
//   h   = gelu(y W1^T + b)  |  silu(y Wg^T+b)*(y Wv^T+b)    EPI_GELU | EPI_SWIGLU
//   x  += ls2 * (h W2^T + b)                        EPI_LS_RESID (in place)
#include <cstdlib>
#include <vector>

#include "common.hpp"

struct anyloc_vit {
  anyloc_vit_config cfg;
  const float* patch_w;
  const float* patch_b;
  const float* cls;
  std::vector<anyloc_vit_block_weights> blocks;
  std::vector<anyloc_vit_block_x3> x3;      // optional: three-plane bf16 images of the four weight matrices
  std::vector<anyloc_vit_block_h2> h2;      // optional: two-plane fp16 images + row scales of the same matrices
  std::vector<char> ffn_exact;              // per block: 1 = quantise the FFN activation against the exact row maximum
  float* ffn_looseness = nullptr;           // telemetry target (device [depth] or [depth][batch]) or null
  int telemetry_per_image = 0;              // 1: one figure per (block, image) instead of one per block
  unsigned char* patch_w2 = nullptr;        // fp16 mode: two-plane image + row scales of patch_w, built (and owned) by
  float* patch_inv = nullptr;               // anyloc_vit_attach_h2
  void drop_patch_image() {
    if (patch_w2) (void)hipFree(patch_w2);
    if (patch_inv) (void)hipFree(patch_inv);
    patch_w2 = nullptr;
    patch_inv = nullptr;
  }
};

namespace anyloc {
namespace {

struct VitWs {
  float *x, *y, *qkv, *h;   // qkv doubles as the im2col buffer; attention output aliases y
  unsigned char* a3;        // split-bf16 mode: plane image of a D-wide activation operand (LN output, attention output)
  unsigned char* h3;        //                  plane image of the FFN hidden activation
  float *ainv, *hinv;       // fp16 mode: 2^-e per row of the images in a3 / h3
  float* qinv;              // fp16 mode, fused attention: 2^-e per (part, head, 32-row group) tile of q | k | v (in qkv)
  float* sk_part;           // fp16 mode, small-M plans: partial accumulators of a split-K launch (gemm_h3s.hip)
  unsigned* sk_tickets;     //                           arrival counters per tile, zero between launches
  unsigned* ln_tickets;     // fp16 mode, LayerNorm lead role (gemm_h3_kernel.hpp): [depth][2][ln_tk] rows done per 128-row tile of a lead launch
  size_t ln_tk;             //   words per launch: max(16, ceil(M / 128))
  unsigned* hmax;           // fp16 mode, FFN-bound telemetry: [depth][M] largest scaled magnitude per row (bits); tickets, lead
                            // tickets and these lie back to back so that ONE memset per forward clears them
  size_t bytes;
};

VitWs carve(void* ws, size_t cap, const anyloc_vit_config& c, int64_t batch, int64_t H, int64_t W) {
  Arena a(ws, cap);
  const int64_t np = (H / c.patch) * (W / c.patch), T = np + 1, M = batch * T;
  VitWs w;
  w.x = a.take<float>(M * c.dim);
  w.y = a.take<float>(M * c.dim);
  // fp32 [M, 3D], or the im2col patches, or (fp16 mode) the q | k | v tiles of attention_h3: rows padded to 32
  const int64_t qkv_elems = std::max<int64_t>((M + 31) / 32 * 32 * 3 * c.dim, batch * np * ((c.patch_k_pad + 15) / 16 * 16));
  w.qkv = a.take<float>(qkv_elems);
  w.h = a.take<float>(M * c.ffn_hidden);
  // (fp16 mode also quantises the gathered patches into a3: [batch * np, patch_k_pad rounded up to 16] as two fp16 planes)
  w.a3 = a.take<unsigned char>(std::max(x3_bytes(M, c.dim), h2_bytes(batch * np, (c.patch_k_pad + 15) / 16 * 16)));
  w.h3 = a.take<unsigned char>(x3_bytes(M, c.ffn_hidden));
  w.ainv = a.take<float>(M);
  w.hinv = a.take<float>(M);
  w.qinv = a.take<float>(qkv_inv_count(M, c.heads));
  w.sk_part = a.take<float>(H3_SPLIT_PART_BYTES / sizeof(float));
  w.sk_tickets = a.take<unsigned>(H3_SPLIT_TICKETS);
  w.ln_tk = std::max<size_t>(16, (size_t)((M + 127) / 128));
  w.ln_tickets = a.take<unsigned>((size_t)c.depth * 2 * w.ln_tk);
  w.hmax = a.take<unsigned>((size_t)c.depth * M);
  w.bytes = a.off;
  return w;
}

int linear(const float* A, int64_t lda, const float* Wt, int64_t K, const float* bias, float* C, int64_t ldc, int64_t M,
           int64_t N, int epi, const float* gamma, const char* tag, hipStream_t stream) {
  GemmProblem g{};
  g.A = A; g.lda = lda;
  g.W = Wt; g.ldw = K;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.bias = bias;
  g.gamma = gamma;
  g.resid = C;
  g.tag = tag;
  return gemm_nt(g, epi, stream);
}

// options (common.hpp): x6_fuse / h3_fuse = 0 keep fp32 activations and quantise them in front of every GEMM (the round-1
// data flows; A/B measurements and tests).  The two-term fp16 forward has no row threshold by default: with q | k | v, the
// attention output and the FFN activation kept in fp16 planes it beats the fp32-MFMA kernels at every batch (B=1: 9.6 vs
// 16.5 ms, profiles/r02_extractor_vs_batch.log); the split-bf16 forward switches at 1600 rows.

// y = act(A W^T + b) on the six-product bf16 GEMM.  A is given as fp32 (split into planes here) or, when A == nullptr,
// a3 already holds its plane image (written by the producer).  c3 != nullptr: the activation is written as the plane
// image of the next GEMM instead of fp32 C.
int linear_x6(const float* A, int64_t K, unsigned char* a3, const void* w3, int64_t w_rows, int64_t w_row0,
              const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int epi, const float* gamma,
              const char* tag, hipStream_t stream, unsigned char* c3 = nullptr) {
  if (A) ANYLOC_TRY(split_x3(A, K, M, K, a3, stream));
  X6Problem g{};
  g.C3 = c3; g.RC = M;
  g.A3 = a3; g.RA = M;
  g.W3 = static_cast<const unsigned char*>(w3) + w_row0 * 32; g.RW = w_rows;
  g.w_off = w_row0 * 32;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K16 = (int)((K + 15) / 16);
  g.bias = bias;
  g.gamma = gamma;
  g.resid = C;
  g.tag = tag;
  return gemm_x6(g, epi, stream);
}

// the same linear layer on the row-scaled two-term fp16 GEMM (gemm_h3.hip).  A == nullptr: a2 / ainv already hold the
// quantised operand (written by layernorm_h2); otherwise A (fp32, row-major, width K) is quantised here first.
// a LayerNorm whose output is the GEMM's own operand image (a2 / ainv; with `bound` also the FFN bound into c_inv)
struct LnFront {
  const float *x, *w, *b;
  const float* bound;       // HOST [4] or null
  unsigned* tickets;        // zeroed words of this launch, one per 128-row tile (VitWs::ln_tk)
};

int linear_h3(const float* A, int64_t K, unsigned char* a2, float* ainv, const void* w2, const float* winv, int64_t w_rows,
              int64_t w_row0, const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int epi, const float* gamma,
              const char* tag, hipStream_t stream, unsigned char* c2 = nullptr, const float* c_inv = nullptr,
              unsigned char* qkv_planes = nullptr, float* qkv_inv = nullptr, int heads = 0, const VitWs* ws = nullptr,
              int kind = H3_KIND_OTHER, unsigned* c_max = nullptr, const LnFront* ln = nullptr) {
  if (A) ANYLOC_TRY(split_h2(A, K, M, K, a2, ainv, stream));
  H3Problem g{};
  if (ws) { g.sk_part = ws->sk_part; g.sk_tickets = ws->sk_tickets; }
  g.kind = kind;
  g.C2 = c2; g.RC = M; g.c_inv = c_inv; g.c_max = c_max;
  g.qkv_planes = qkv_planes; g.qkv_inv = qkv_inv; g.heads = heads; g.groups = (M + 31) / 32;
  g.A2 = a2; g.RA = M; g.a_inv = ainv;
  g.W2 = static_cast<const unsigned char*>(w2) + w_row0 * 32; g.RW = w_rows; g.w_inv = winv + w_row0;
  g.w_off = w_row0 * 32;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K16 = (int)(K / 16);
  g.bias = bias;
  g.gamma = gamma;
  g.resid = C;
  g.tag = tag;
  if (ln) {
    // LayerNorm in front of this GEMM: as the lead role of the GEMM's own launch where the small-M plan has it (one image per
    // call: LN1 + qkv, LN2 + w12), as a launch of its own otherwise -- the same arithmetic, the same bits
    if (h3_ln_lead_feasible(g, epi)) {
      g.ln_x = ln->x; g.ln_w = ln->w; g.ln_b = ln->b; g.ln_eps = 1e-6f; g.ln_dim = (int)K;
      g.ln_has_bound = ln->bound != nullptr;
      for (int i = 0; i < 4; ++i) g.ln_bound[i] = ln->bound ? ln->bound[i] : 0.0f;
      g.ln_tickets = ln->tickets;
    } else {
      ANYLOC_TRY(layernorm_h2(ln->x, ln->w, ln->b, M, (int)K, 1e-6f, a2, ainv, stream, ln->bound, ln->bound ? const_cast<float*>(c_inv) : nullptr));
    }
  }
  return gemm_h3(g, epi, stream);
}

}  // namespace
}  // namespace anyloc

using namespace anyloc;

extern "C" {

int anyloc_layernorm(const float* x, float* y, const float* weight, const float* bias, int64_t rows, int64_t dim,
                     float eps, void* stream) {
  ANYLOC_CHECK_ARG(x && y && weight && bias && rows > 0 && rows < (1ll << 31) && dim > 0, "layernorm: bad args");
  return layernorm(x, y, weight, bias, rows, (int)dim, eps, static_cast<hipStream_t>(stream));
}

int anyloc_attention(const float* qkv, float* out, int64_t batch, int64_t tokens, int64_t dim, int64_t heads,
                     void* stream) {
  ANYLOC_CHECK_ARG(qkv && out, "attention: null pointer");
  return attention(qkv, out, batch, (int)tokens, (int)dim, (int)heads, static_cast<hipStream_t>(stream));
}

size_t anyloc_attention_h3_workspace_bytes(int64_t batch, int64_t tokens, int64_t heads) {
  if (batch <= 0 || tokens <= 0 || heads <= 0) return 0;
  const int64_t rows = batch * tokens;
  return align_up(qkv_planes_bytes(rows, (int)heads), 256) + align_up(qkv_inv_count(rows, (int)heads) * sizeof(float), 256) + 256;
}

int anyloc_attention_h3(const float* qkv, void* out_img, float* out_inv, int64_t batch, int64_t tokens, int64_t dim,
                        int64_t heads, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(qkv && out_img && out_inv && workspace, "attention_h3: null pointer");
  ANYLOC_CHECK_ARG(batch > 0 && tokens > 0 && heads > 0 && dim == heads * 64, "attention_h3: bad shape");
  if (workspace_bytes < anyloc_attention_h3_workspace_bytes(batch, tokens, heads)) {
    set_error("attention_h3: workspace %zu < %zu", workspace_bytes, anyloc_attention_h3_workspace_bytes(batch, tokens, heads));
    return ANYLOC_ERR_WORKSPACE;
  }
  Arena a(workspace, workspace_bytes);
  const int64_t rows = batch * tokens;
  unsigned char* planes = a.take<unsigned char>(qkv_planes_bytes(rows, (int)heads));
  float* inv = a.take<float>(qkv_inv_count(rows, (int)heads));
  ANYLOC_TRY(qkv_planes_from_f32(qkv, rows, (int)dim, (int)heads, planes, inv, stream));
  return attention_h3(planes, inv, batch, (int)tokens, (int)dim, (int)heads, static_cast<unsigned char*>(out_img), out_inv, stream);
}

int anyloc_vit_create(anyloc_vit_t** out, const anyloc_vit_config* cfg, const float* patch_w, const float* patch_b,
                      const float* cls_token, const anyloc_vit_block_weights* blocks) {
  ANYLOC_CHECK_ARG(out && cfg && patch_w && patch_b && cls_token && blocks, "vit_create: null pointer");
  ANYLOC_CHECK_ARG(cfg->dim > 0 && cfg->dim % 64 == 0 && cfg->heads * 64 == cfg->dim,
                   "vit_create: dim %d / heads %d (head_dim must be 64)", cfg->dim, cfg->heads);
  ANYLOC_CHECK_ARG(cfg->depth > 0 && cfg->depth <= 256, "vit_create: depth %d", cfg->depth);
  ANYLOC_CHECK_ARG(cfg->ffn_kind == 0 || cfg->ffn_kind == 1, "vit_create: ffn_kind %d", cfg->ffn_kind);
  ANYLOC_CHECK_ARG(cfg->ffn_hidden > 0 && cfg->ffn_hidden % 64 == 0, "vit_create: ffn_hidden %d", cfg->ffn_hidden);
  ANYLOC_CHECK_ARG(cfg->patch > 0 && cfg->patch_k_pad >= 3 * cfg->patch * cfg->patch && cfg->patch_k_pad % 4 == 0,
                   "vit_create: patch %d / patch_k_pad %d", cfg->patch, cfg->patch_k_pad);
  for (int i = 0; i < cfg->depth; ++i) {
    const anyloc_vit_block_weights& b = blocks[i];
    ANYLOC_CHECK_ARG(b.norm1_w && b.norm1_b && b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.ls1 && b.norm2_w &&
                         b.norm2_b && b.fc1_w && b.fc1_b && b.fc2_w && b.fc2_b && b.ls2,
                     "vit_create: block %d has a null weight", i);
  }
  anyloc_vit* h = new (std::nothrow) anyloc_vit();
  if (!h) {
    set_error("vit_create: out of host memory");
    return ANYLOC_ERR_HIP;
  }
  h->cfg = *cfg;
  h->patch_w = patch_w;
  h->patch_b = patch_b;
  h->cls = cls_token;
  h->blocks.assign(blocks, blocks + cfg->depth);
  *out = h;
  return ANYLOC_OK;
}

int anyloc_vit_attach_x3(anyloc_vit_t* h, const anyloc_vit_block_x3* blocks) {
  ANYLOC_CHECK_ARG(h, "vit_attach_x3: null handle");
  if (!blocks) {
    h->x3.clear();
    return ANYLOC_OK;
  }
  for (int i = 0; i < h->cfg.depth; ++i)
    ANYLOC_CHECK_ARG(blocks[i].qkv_w3 && blocks[i].proj_w3 && blocks[i].fc1_w3 && blocks[i].fc2_w3,
                     "vit_attach_x3: block %d has a null plane image", i);
  h->x3.assign(blocks, blocks + h->cfg.depth);
  return ANYLOC_OK;
}

int anyloc_vit_attach_h2(anyloc_vit_t* h, const anyloc_vit_block_h2* blocks) {
  ANYLOC_CHECK_ARG(h, "vit_attach_h2: null handle");
  h->drop_patch_image();
  if (!blocks) {
    h->h2.clear();
    h->ffn_exact.assign(h->cfg.depth, 0);
    return ANYLOC_OK;
  }
  ANYLOC_CHECK_ARG(h->cfg.dim % 16 == 0 && h->cfg.ffn_hidden % 16 == 0 && h->cfg.ffn_hidden <= 4096 && h->cfg.dim <= 2048,
                   "vit_attach_h2: dim %d / ffn_hidden %d outside the fp16 path's limits", h->cfg.dim, h->cfg.ffn_hidden);
  for (int i = 0; i < h->cfg.depth; ++i) {
    const anyloc_vit_block_h2& b = blocks[i];
    ANYLOC_CHECK_ARG(b.qkv_w2 && b.qkv_inv && b.proj_w2 && b.proj_inv && b.fc1_w2 && b.fc1_inv && b.fc2_w2 && b.fc2_inv,
                     "vit_attach_h2: block %d has a null image or scale array", i);
    ANYLOC_CHECK_ARG(b.fc1_layout == 0 || (b.fc1_layout == 1 && h->cfg.ffn_kind == 1 && b.fc1_b2 && h->cfg.ffn_hidden % 64 == 0),
                     "vit_attach_h2: block %d: fc1_layout %d (1 needs a SwiGLU model, fc1_b2 and ffn_hidden %% 64 == 0)", i,
                     b.fc1_layout);
  }
  // The patch-embedding weights [dim, patch_k_pad] as an operand image of the same GEMM, the contraction zero-padded to
  // whole 16-element k-blocks (one-off, on the null stream).  Built FIRST, on the device that owns patch_w (not whatever
  // device is current), every temporary freed on every path; the handle changes only when all of it succeeded -- a failed
  // attach leaves the handle as it was before the call, minus the previous patch image (already dropped above, with the
  // previous h2 blocks detached by the clear() below).
  h->h2.clear();
  h->ffn_exact.assign(h->cfg.depth, 0);      // the exact-quantiser switches belong to the weights that are being replaced
  const int64_t D = h->cfg.dim, K0 = h->cfg.patch_k_pad, Kp = (K0 + 15) / 16 * 16;
  int prev_dev = -1, w_dev = -1;
  hipPointerAttribute_t attr;
  if (hipGetDevice(&prev_dev) == hipSuccess && hipPointerGetAttributes(&attr, h->patch_w) == hipSuccess &&
      attr.type == hipMemoryTypeDevice)
    w_dev = attr.device;
  else
    (void)hipGetLastError();                 // (an unregistered pointer: stay on the current device)
  if (w_dev >= 0 && w_dev != prev_dev) ANYLOC_HIP(hipSetDevice(w_dev));
  float* padded = nullptr;
  unsigned char* w2 = nullptr;
  float* winv = nullptr;
  int rc = ANYLOC_OK;
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == ANYLOC_OK) {
      set_error("vit_attach_h2: %s: %s", what, hipGetErrorString(e));
      rc = ANYLOC_ERR_HIP;
    }
    return e == hipSuccess;
  };
  if (hip_ok(hipMalloc(reinterpret_cast<void**>(&padded), sizeof(float) * D * Kp), "hipMalloc (padded patch weights)") &&
      hip_ok(hipMalloc(reinterpret_cast<void**>(&w2), h2_bytes(D, Kp)), "hipMalloc (patch-embedding image)") &&
      hip_ok(hipMalloc(reinterpret_cast<void**>(&winv), sizeof(float) * D), "hipMalloc (patch-embedding row scales)") &&
      hip_ok(hipMemset(padded, 0, sizeof(float) * D * Kp), "hipMemset") &&
      hip_ok(hipMemcpy2D(padded, sizeof(float) * Kp, h->patch_w, sizeof(float) * K0, sizeof(float) * K0, D, hipMemcpyDeviceToDevice),
             "hipMemcpy2D")) {
    rc = split_h2(padded, Kp, D, Kp, w2, winv, nullptr);
    if (rc == ANYLOC_OK) hip_ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
  }
  if (padded) (void)hipFree(padded);
  if (rc != ANYLOC_OK) {
    if (w2) (void)hipFree(w2);
    if (winv) (void)hipFree(winv);
  }
  if (w_dev >= 0 && w_dev != prev_dev) (void)hipSetDevice(prev_dev);
  if (rc != ANYLOC_OK) return rc;
  h->patch_w2 = w2;
  h->patch_inv = winv;
  h->h2.assign(blocks, blocks + h->cfg.depth);
  return ANYLOC_OK;
}

int anyloc_vit_set_telemetry(anyloc_vit_t* h, float* ffn_looseness, int32_t per_image) {
  ANYLOC_CHECK_ARG(h, "vit_set_telemetry: null handle");
  h->ffn_looseness = ffn_looseness;
  h->telemetry_per_image = per_image ? 1 : 0;
  return ANYLOC_OK;
}

int anyloc_vit_block_ffn_exact(anyloc_vit_t* h, int32_t layer, int32_t exact) {
  ANYLOC_CHECK_ARG(h && layer >= 0 && layer < h->cfg.depth, "vit_block_ffn_exact: bad handle / layer");
  if (h->ffn_exact.size() != (size_t)h->cfg.depth) h->ffn_exact.assign(h->cfg.depth, 0);
  h->ffn_exact[layer] = exact ? 1 : 0;
  return ANYLOC_OK;
}

void anyloc_vit_destroy(anyloc_vit_t* h) {
  if (h) h->drop_patch_image();
  delete h;
}

size_t anyloc_vit_workspace_bytes(const anyloc_vit_t* h, int64_t batch, int64_t img_h, int64_t img_w) {
  if (!h || batch <= 0 || img_h < h->cfg.patch || img_w < h->cfg.patch) return 0;
  return carve(nullptr, 0, h->cfg, batch, img_h, img_w).bytes + 256;
}

// the launch sequence of one forward on `stream` (shape and taps already validated by anyloc_vit_forward)
static int vit_forward_launches(anyloc_vit_t* h, const float* img, int64_t batch, int64_t img_h, int64_t img_w,
                                const float* pos, int32_t n_taps, const int32_t* tap_layers, const int32_t* tap_facets,
                                unsigned flags, float* out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const anyloc_vit_config& c = h->cfg;
  const int D = c.dim, gh = (int)(img_h / c.patch), gw = (int)(img_w / c.patch), np = gh * gw, T = np + 1;
  const int64_t M = batch * T;
  VitWs w = carve(workspace, workspace_bytes, c, batch, img_h, img_w);
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("vit_forward: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  ANYLOC_CHECK_ARG(!(flags & ANYLOC_VIT_SPLIT_BF16) || !h->x3.empty(),
                   "vit_forward: ANYLOC_VIT_SPLIT_BF16 without anyloc_vit_attach_x3");
  // below ~3 images of 530 tokens the GEMMs have too few 128-row tiles to fill 256 CUs twice over: the fp32-MFMA
  // kernel with its 64-row split is faster there (measured B=1: 60 vs 40 images/s), so the split-bf16 request is
  // honoured from option x6_min_rows rows up
  ANYLOC_CHECK_ARG(!(flags & ANYLOC_VIT_SPLIT_FP16) || !h->h2.empty(),
                   "vit_forward: ANYLOC_VIT_SPLIT_FP16 without anyloc_vit_attach_h2");
  const bool h3m = (flags & ANYLOC_VIT_SPLIT_FP16) && M >= option(OPT_H3_MIN_ROWS);
  const bool x6 = !h3m && (flags & ANYLOC_VIT_SPLIT_BF16) && M >= option(OPT_X6_MIN_ROWS);
  const bool fuse_x6 = x6 && option(OPT_X6_FUSE) != 0;
  const bool h3f = h3m && option(OPT_H3_FUSE) != 0;
  const bool use_cls = flags & ANYLOC_VIT_USE_CLS;
  const int rows_per_img = use_cls ? T : np, skip = use_cls ? 0 : 1;
  const int64_t ldo = (int64_t)n_taps * D;
  const int norm_taps = (flags & ANYLOC_VIT_NORM_TAPS) ? 1 : 0;
  const int last_layer = tap_layers[n_taps - 1];
  // split-K arrival counters; with telemetry on also the rows' maxima of every block that will run (adjacent: one memset)
  const bool telem = h3m && h->ffn_looseness != nullptr;
  if (h3m)
    ANYLOC_HIP(hipMemsetAsync(w.sk_tickets, 0,
                              telem ? (size_t)(reinterpret_cast<char*>(w.hmax + (size_t)(last_layer + 1) * M) - reinterpret_cast<char*>(w.sk_tickets))
                                    : (size_t)(reinterpret_cast<char*>(w.hmax) - reinterpret_cast<char*>(w.sk_tickets)),
                              stream));
  // does any tap need the block OUTPUT of the last executed layer?
  bool last_needs_full = false;
  for (int t = 0; t < n_taps; ++t)
    if (tap_layers[t] == last_layer && tap_facets[t] == ANYLOC_FACET_TOKEN) last_needs_full = true;

  // ---- patch embedding: conv 14x14 stride 14 == GEMM over gathered patches, + bias + pos ----
  float* col = w.qkv;
  const bool patch_h3 = h3m && h->patch_w2 && option(OPT_H3_PATCH) != 0;
  const int kp = patch_h3 ? (c.patch_k_pad + 15) / 16 * 16 : c.patch_k_pad;    // fp16 mode: whole 16-element k-blocks
  ANYLOC_TRY(im2col(img, col, batch, (int)img_h, (int)img_w, c.patch, kp, stream));
  if (patch_h3) {
    // fp16 mode: the gathered patches are quantised like every other operand (row maximum -> power-of-two scale)
    ANYLOC_TRY(split_h2(col, kp, batch * np, kp, w.a3, w.ainv, stream));
    H3Problem g{};
    g.A2 = w.a3; g.RA = batch * np; g.a_inv = w.ainv;
    g.W2 = h->patch_w2; g.RW = D; g.w_inv = h->patch_inv;
    g.C = w.x; g.ldc = D;
    g.M = batch * np; g.N = D; g.K16 = kp / 16;
    g.bias = h->patch_b;
    g.pos = pos;
    g.patches = np;
    g.tag = "vit_patch_embed_gemm";
    ANYLOC_TRY(gemm_h3(g, EPI_PATCH, stream));
  } else {
    GemmProblem g{};
    g.A = col; g.lda = c.patch_k_pad;
    g.W = h->patch_w; g.ldw = c.patch_k_pad;
    g.C = w.x; g.ldc = D;
    g.M = batch * np; g.N = D; g.K = c.patch_k_pad;
    g.bias = h->patch_b;
    g.pos = pos;
    g.patches = np;
    g.tag = "vit_patch_embed_gemm";
    ANYLOC_TRY(gemm_nt(g, EPI_PATCH, stream));
  }
  ANYLOC_TRY(cls_rows(w.x, h->cls, pos, batch, T, D, stream));

  for (int l = 0; l <= last_layer; ++l) {
    const anyloc_vit_block_weights& b = h->blocks[l];
    const bool last = (l == last_layer);
    // split-bf16 mode, fused producers: LayerNorm / attention / FFN activation write plane images directly
    const bool fuse = fuse_x6;
    bool qkv_tap0 = false;                      // (decided here already: a q / k / v tap keeps the unfused attention data flow)
    for (int t = 0; t < n_taps; ++t)
      if (tap_layers[t] == l && tap_facets[t] != ANYLOC_FACET_TOKEN) qkv_tap0 = true;
    // fp16 mode, fused attention: LayerNorm 1 travels with the QKV GEMM (linear_h3's `ln`: lead role of that launch for one image
    // per call, a launch of its own otherwise)
    const bool ln1_with_qkv = h3f && !qkv_tap0 && D % 128 == 0 && !(last && !last_needs_full);
    const LnFront ln1{w.x, b.norm1_w, b.norm1_b, nullptr, w.ln_tickets + (size_t)l * 2 * w.ln_tk};
    if (h3m && !ln1_with_qkv) ANYLOC_TRY(layernorm_h2(w.x, b.norm1_w, b.norm1_b, M, D, 1e-6f, w.a3, w.ainv, stream));
    else if (h3m) {}
    else if (fuse) ANYLOC_TRY(layernorm_x3(w.x, b.norm1_w, b.norm1_b, M, D, 1e-6f, w.a3, stream));
    else ANYLOC_TRY(layernorm(w.x, w.y, b.norm1_w, b.norm1_b, M, D, 1e-6f, stream));
    const float* y_in = fuse ? nullptr : w.y;     // nullptr: the plane image is already in w.a3
    if (last && !last_needs_full) {
      // only q/k/v taps remain: compute just the tapped thirds of the QKV projection
      for (int t = 0; t < n_taps; ++t) {
        if (tap_layers[t] != l) continue;
        const int f = tap_facets[t];
        if (h3m)
          ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].qkv_w2, h->h2[l].qkv_inv, 3 * D, (int64_t)f * D,
                               b.qkv_b + (int64_t)f * D, w.qkv, D, M, D, EPI_STORE, nullptr, "vit_facet_gemm", stream, nullptr,
                               nullptr, nullptr, nullptr, 0, &w, H3_KIND_PROJ));
        else if (x6)
          ANYLOC_TRY(linear_x6(y_in, D, w.a3, h->x3[l].qkv_w3, 3 * D, (int64_t)f * D, b.qkv_b + (int64_t)f * D, w.qkv, D,
                               M, D, EPI_STORE, nullptr, "vit_facet_gemm", stream));
        else
          ANYLOC_TRY(linear(w.y, D, b.qkv_w + (int64_t)f * D * D, D, b.qkv_b + (int64_t)f * D, w.qkv, D, M, D,
                            EPI_STORE, nullptr, "vit_facet_gemm", stream));
        ANYLOC_TRY(facet_rows(w.qkv, D, 0, out, ldo, t * D, batch, T, skip, rows_per_img, D, norm_taps, 1e-12f,
                              stream));
      }
      break;
    }
    bool qkv_tap = false;                       // a q / k / v tap of this layer needs the fp32 projection
    for (int t = 0; t < n_taps; ++t)
      if (tap_layers[t] == l && tap_facets[t] != ANYLOC_FACET_TOKEN) qkv_tap = true;
    const bool fuse_attn = h3f && !qkv_tap && D % 128 == 0;
    if (fuse_attn) {
      // fp16 mode, fused: the QKV GEMM writes per-head two-plane fp16 tiles, attention_h3 consumes them by DMA and writes
      // the image of the projection GEMM -- q, k, v and the attention output never exist in fp32
      ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].qkv_w2, h->h2[l].qkv_inv, 3 * D, 0, b.qkv_b, nullptr, 3 * D, M,
                           3 * D, EPI_QKV_PLANES, nullptr, "vit_qkv_gemm", stream, nullptr, nullptr,
                           reinterpret_cast<unsigned char*>(w.qkv), w.qinv, c.heads, &w, H3_KIND_QKV, nullptr, &ln1));
      ANYLOC_TRY(attention_h3(reinterpret_cast<const unsigned char*>(w.qkv), w.qinv, batch, T, D, c.heads, w.a3, w.ainv, stream));
      ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].proj_w2, h->h2[l].proj_inv, D, 0, b.proj_b, w.x, D, M, D,
                           EPI_LS_RESID, b.ls1, "vit_proj_gemm", stream, nullptr, nullptr, nullptr, nullptr, 0, &w, H3_KIND_PROJ));
    } else {
      if (h3m)
        ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].qkv_w2, h->h2[l].qkv_inv, 3 * D, 0, b.qkv_b, w.qkv, 3 * D, M,
                             3 * D, EPI_STORE, nullptr, "vit_qkv_gemm", stream));
      else if (x6)
        ANYLOC_TRY(linear_x6(y_in, D, w.a3, h->x3[l].qkv_w3, 3 * D, 0, b.qkv_b, w.qkv, 3 * D, M, 3 * D, EPI_STORE, nullptr,
                             "vit_qkv_gemm", stream));
      else
        ANYLOC_TRY(linear(w.y, D, b.qkv_w, D, b.qkv_b, w.qkv, 3 * D, M, 3 * D, EPI_STORE, nullptr, "vit_qkv_gemm", stream));
      for (int t = 0; t < n_taps; ++t)
        if (tap_layers[t] == l && tap_facets[t] != ANYLOC_FACET_TOKEN)
          ANYLOC_TRY(facet_rows(w.qkv, 3 * D, tap_facets[t] * D, out, ldo, t * D, batch, T, skip, rows_per_img, D,
                                norm_taps, 1e-12f, stream));
      ANYLOC_TRY(attention(w.qkv, w.y, batch, T, D, c.heads, stream, fuse ? w.a3 : nullptr, x6 || h3m));
      if (h3m)     // the attention output is fp32: its rows span all heads, the row maximum is only known now
        ANYLOC_TRY(linear_h3(w.y, D, w.a3, w.ainv, h->h2[l].proj_w2, h->h2[l].proj_inv, D, 0, b.proj_b, w.x, D, M, D,
                             EPI_LS_RESID, b.ls1, "vit_proj_gemm", stream));
      else if (x6)
        ANYLOC_TRY(linear_x6(y_in, D, w.a3, h->x3[l].proj_w3, D, 0, b.proj_b, w.x, D, M, D, EPI_LS_RESID, b.ls1,
                             "vit_proj_gemm", stream));
      else
        ANYLOC_TRY(linear(w.y, D, b.proj_w, D, b.proj_b, w.x, D, M, D, EPI_LS_RESID, b.ls1, "vit_proj_gemm", stream));
    }
    const float* fb = h3f ? h->h2[l].fc1_bound : nullptr;
    const bool fuse_ffn = fb && (fb[0] > 0.f || fb[1] > 0.f) && !(l < (int)h->ffn_exact.size() && h->ffn_exact[l]);
    // (fused FFN: LayerNorm 2 travels with the fc1 / w12 GEMM the same way)
    const bool ln2_with_fc1 = h3m && fuse_ffn;
    const LnFront ln2{w.x, b.norm2_w, b.norm2_b, fb, w.ln_tickets + ((size_t)l * 2 + 1) * w.ln_tk};
    if (h3m && !ln2_with_fc1) ANYLOC_TRY(layernorm_h2(w.x, b.norm2_w, b.norm2_b, M, D, 1e-6f, w.a3, w.ainv, stream, nullptr, w.hinv));
    else if (h3m) {}
    else if (fuse) ANYLOC_TRY(layernorm_x3(w.x, b.norm2_w, b.norm2_b, M, D, 1e-6f, w.a3, stream));
    else ANYLOC_TRY(layernorm(w.x, w.y, b.norm2_w, b.norm2_b, M, D, 1e-6f, stream));
    const int Hh = c.ffn_hidden;
    if (h3m && fuse_ffn) {
      // the hidden activation is quantised in the epilogue against the row bound LayerNorm 2 left in w.hinv
      if (c.ffn_kind == 0)
        ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].fc1_w2, h->h2[l].fc1_inv, Hh, 0, b.fc1_b, nullptr, Hh, M, Hh,
                             EPI_GELU_H2, nullptr, "vit_fc1_gemm", stream, w.h3, w.hinv, nullptr, nullptr, 0, &w, H3_KIND_FC1,
                             telem ? w.hmax + (size_t)l * M : nullptr, &ln2));
      else
        ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].fc1_w2, h->h2[l].fc1_inv, 2 * Hh, 0,
                             h->h2[l].fc1_b2 ? h->h2[l].fc1_b2 : b.fc1_b, nullptr, Hh, M, 2 * Hh,
                             h->h2[l].fc1_layout == 1 ? EPI_SWIGLU_T_H2 : EPI_SWIGLU_H2, nullptr, "vit_w12_gemm", stream, w.h3,
                             w.hinv, nullptr, nullptr, 0, &w, H3_KIND_FC1, telem ? w.hmax + (size_t)l * M : nullptr, &ln2));
      ANYLOC_TRY(linear_h3(nullptr, Hh, w.h3, w.hinv, h->h2[l].fc2_w2, h->h2[l].fc2_inv, D, 0, b.fc2_b, w.x, D, M, D,
                           EPI_LS_RESID, b.ls2, "vit_fc2_gemm", stream, nullptr, nullptr, nullptr, nullptr, 0, &w, H3_KIND_FC2));
    } else if (h3m) {
      if (c.ffn_kind == 0)
        ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].fc1_w2, h->h2[l].fc1_inv, Hh, 0, b.fc1_b, w.h, Hh, M, Hh,
                             EPI_GELU, nullptr, "vit_fc1_gemm", stream));
      else
        ANYLOC_TRY(linear_h3(nullptr, D, w.a3, w.ainv, h->h2[l].fc1_w2, h->h2[l].fc1_inv, 2 * Hh, 0,
                             h->h2[l].fc1_b2 ? h->h2[l].fc1_b2 : b.fc1_b, w.h, Hh, M, 2 * Hh,
                             h->h2[l].fc1_layout == 1 ? EPI_SWIGLU_T : EPI_SWIGLU, nullptr, "vit_w12_gemm", stream));
      ANYLOC_TRY(linear_h3(w.h, Hh, w.h3, w.hinv, h->h2[l].fc2_w2, h->h2[l].fc2_inv, D, 0, b.fc2_b, w.x, D, M, D,
                           EPI_LS_RESID, b.ls2, "vit_fc2_gemm", stream));
    } else if (x6) {
      unsigned char* c3 = fuse ? w.h3 : nullptr;
      if (c.ffn_kind == 0)
        ANYLOC_TRY(linear_x6(y_in, D, w.a3, h->x3[l].fc1_w3, Hh, 0, b.fc1_b, w.h, Hh, M, Hh, EPI_GELU, nullptr,
                             "vit_fc1_gemm", stream, c3));
      else
        ANYLOC_TRY(linear_x6(y_in, D, w.a3, h->x3[l].fc1_w3, 2 * Hh, 0, b.fc1_b, w.h, Hh, M, 2 * Hh, EPI_SWIGLU, nullptr,
                             "vit_w12_gemm", stream, c3));
      ANYLOC_TRY(linear_x6(fuse ? nullptr : w.h, Hh, w.h3, h->x3[l].fc2_w3, D, 0, b.fc2_b, w.x, D, M, D, EPI_LS_RESID,
                           b.ls2, "vit_fc2_gemm", stream));
    } else {
      if (c.ffn_kind == 0) {
        ANYLOC_TRY(linear(w.y, D, b.fc1_w, D, b.fc1_b, w.h, Hh, M, Hh, EPI_GELU, nullptr, "vit_fc1_gemm", stream));
      } else {
        ANYLOC_TRY(linear(w.y, D, b.fc1_w, D, b.fc1_b, w.h, Hh, M, 2 * Hh, EPI_SWIGLU, nullptr, "vit_w12_gemm", stream));
      }
      ANYLOC_TRY(linear(w.h, Hh, b.fc2_w, Hh, b.fc2_b, w.x, D, M, D, EPI_LS_RESID, b.ls2, "vit_fc2_gemm", stream));
    }
    for (int t = 0; t < n_taps; ++t)
      if (tap_layers[t] == l && tap_facets[t] == ANYLOC_FACET_TOKEN)
        ANYLOC_TRY(facet_rows(w.x, D, 0, out, ldo, t * D, batch, T, skip, rows_per_img, D, norm_taps, 1e-12f, stream));
  }
  if (flags & ANYLOC_VIT_NORM_CONCAT)
    ANYLOC_TRY(l2norm_rows(out, ldo, out, ldo, batch * rows_per_img, ldo, 1e-12f, stream));
  // FFN-bound telemetry: one figure per executed block (and image) from the row maxima the fc1 / w12 epilogues left
  if (telem) ANYLOC_TRY(ffn_looseness(w.hmax, last_layer + 1, M, h->telemetry_per_image ? T : M, h->ffn_looseness, stream));
  return ANYLOC_OK;
}

int anyloc_vit_forward(anyloc_vit_t* h, const float* img, int64_t batch, int64_t img_h, int64_t img_w,
                       const float* pos, int32_t n_taps, const int32_t* tap_layers, const int32_t* tap_facets,
                       unsigned flags, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(h && img && pos && out && tap_layers && tap_facets, "vit_forward: null pointer");
  const anyloc_vit_config& c = h->cfg;
  ANYLOC_CHECK_ARG(batch > 0 && batch < 65536, "vit_forward: batch %lld", (long long)batch);
  ANYLOC_CHECK_ARG(img_h >= c.patch && img_w >= c.patch && img_h % c.patch == 0 && img_w % c.patch == 0,
                   "vit_forward: image %lldx%lld is not a positive multiple of the patch size %d", (long long)img_h,
                   (long long)img_w, c.patch);
  ANYLOC_CHECK_ARG(n_taps >= 1 && n_taps <= 64, "vit_forward: n_taps %d", n_taps);
  for (int t = 0; t < n_taps; ++t) {
    ANYLOC_CHECK_ARG(tap_layers[t] >= 0 && tap_layers[t] < c.depth, "vit_forward: tap layer %d outside [0,%d)",
                     tap_layers[t], c.depth);
    ANYLOC_CHECK_ARG(tap_facets[t] >= 0 && tap_facets[t] <= 3, "vit_forward: facet %d", tap_facets[t]);
    ANYLOC_CHECK_ARG(t == 0 || tap_layers[t] >= tap_layers[t - 1], "vit_forward: tap layers must ascend");
  }
  return vit_forward_launches(h, img, batch, img_h, img_w, pos, n_taps, tap_layers, tap_facets, flags, out, workspace,
                              workspace_bytes, stream);
}

}  // extern "C"
