// Shared host/device helpers for libanyloc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/anyloc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace anyloc {

// ---- error reporting (thread-local, never throws across the ABI) ----------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define ANYLOC_CHECK_ARG(cond, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      ::anyloc::set_error(__VA_ARGS__);        \
      return ANYLOC_ERR_INVALID_ARG;           \
    }                                          \
  } while (0)

#define ANYLOC_HIP(call)                                   \
  do {                                                     \
    hipError_t e__ = (call);                               \
    if (e__ != hipSuccess) return ::anyloc::hip_fail(e__, #call); \
  } while (0)

#define ANYLOC_TRY(call)        \
  do {                          \
    int s__ = (call);           \
    if (s__ != ANYLOC_OK) return s__; \
  } while (0)

// ---- per-kernel profiling with HIP events on the launch stream ------------
// A Scope brackets one kernel launch; when profiling is off it costs one branch.
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(const char* name, hipStream_t s, double flops, double bytes);
  ~ProfScope();
};

bool profiling_enabled();

// ---- tuning options (anyloc_set_option / anyloc_get_option; include/anyloc_hip.h lists the names) ------------------
// Process-wide integers read by the host-side dispatch code; never read from the environment on a call path: the one
// environment variable, ANYLOC_OPTIONS="name=value,...", is parsed once when the first option is looked up.
enum Option {
  OPT_GEMM_F32_CFG,      // gemm_f32.hip tile configuration (micro-benchmarks); 0 = default
  OPT_X6_CFG,            // gemm_x6.hip tile configuration; 0 = default
  OPT_H3_CFG,            // gemm_h3.hip tile configuration; 0 = default (128x256, 3-deep ring)
  OPT_H3_GROUP_M,        // gemm_h3: tile rows per XCD scheduling group
  OPT_H3_TINY_MAX,       // gemm_h3: below this many 128x128 tiles run 64x64 two-wave tiles
  OPT_H3_DEEP_MAX,       // ... and below this many 64x64 tiles a ring stage holds four k-blocks
  OPT_H3_DEEP2_MAX,      // ... below this many, two
  OPT_H3_EPI_LDS,        // gemm_h3 LayerScale-residual epilogue: 1 = 16-byte accesses through LDS, 0 = dword read-modify-write
  OPT_LN_ROWS_PER_WAVE,  // layernorm_h2: 0 = by ln_small_rows; 1 / 2 / 4 = rows per wave at every size (A/B)
  OPT_LN_SMALL_ROWS,     // layernorm_h2: below this many rows one row per wave
  OPT_LN_WAVES,          // layernorm_h2 with two rows per wave: waves per block, 8 (16 rows: 512-byte store runs) or 4
  OPT_LN_DIRECT_ROWS,    // layernorm_h2: below this many rows one single-wave workgroup per row, no LDS tile (0 = never)
  OPT_H3_FUSE,           // h3 forward: 1 = q|k|v, attention output and FFN activation stay in fp16 planes; 0 = fp32 + quantiser passes
  OPT_X6_FUSE,           // x6 forward: the same for the bf16 plane images
  OPT_H3_MIN_ROWS,       // h3 forward: below this many token rows use the fp32-MFMA kernels
  OPT_X6_MIN_ROWS,       // x6 forward: the same
  OPT_ATTN_CFG,          // fp32-MFMA attention: workgroup shape (micro-benchmarks)
  OPT_ATTN_X6,           // anyloc_attention: 1 = split-bf16 products for every call, 0 = never, -1 = as the caller asks
  OPT_VLAD_PARTS,        // workgroups per image of the fused VLAD kernel (0 = chosen from the image count)
  OPT_VLAD_TWO_PASS,     // 1 = force the general two-pass VLAD path
  OPT_VLAD_FUSED_V,      // fused VLAD kernel: 0 = default choice, 1 = exact-score kernel, 3 / 4 = screening kernel with 4 / 8 waves
  OPT_KMEANS_FUSED_V,    // the same for the k-means step
  OPT_KMEANS_MAX_CHUNKS, // k-means: upper limit of row chunks (partial sums); 0 = two per CU
  OPT_H3_MFMA16,         // gemm_h3: 1 = large GEMMs on the 16x16x32 MFMA kernel (gemm_h3m.hip) where it has the epilogue
  OPT_H3_SWIGLU_T,       // Python host: build the SwiGLU fc1 image in the 16-channel block layout (transposed-accumulator epilogue)
  OPT_H3_FAST_SILU,      // fused SwiGLU epilogue of the h3 w12 GEMM: SiLU on v_exp_f32 + v_rcp_f32 instead of expf + IEEE division
  OPT_TOPK_FEWQ_X6,      // few-query retrieval scores: 2 = two fp16 planes under a running row scale (scores_h3.hip), 1 = three bf16
                         // planes (scores_x6.hip), 0 = fp32 MFMA
  OPT_TOPK_H3,           // retrieval score panels on the two-term fp16 GEMM: -1 = where it pays, 0 = never, 1 = wherever possible
  // small-M plans of the two-term fp16 GEMM (gemm_h3s.hip): overrides of the built-in plan table for sweeps / A-B runs
  OPT_H3S_CFG,           // tile configuration id (-1 = the plan table)
  OPT_H3S_KSPLIT,        // split-K factor (0 = the plan table)
  OPT_H3S_KB,            // k-blocks per ring stage: 1, 2 or 4 (0 = the plan table)
  OPT_H3S_STAGES,        // ring depth: 3 or 6 (0 = the plan table)
  OPT_H3S_MASK,          // which GEMMs the three overrides apply to: bit 0 qkv, 1 proj, 2 fc1 / w12, 3 fc2, 4 others
  OPT_H3S_ENABLE,        // 1 = small-M plans (default), 0 = the round-3 small-batch kernels (64x64 two-wave tiles, no split-K)
  OPT_H3_PATCH,          // fp16 mode: patch embedding on the two-term fp16 GEMM (1, default) or the fp32 MFMA GEMM (0)
  OPT_TOPK_FEWQ_QDMA,    // few-query scores on fp16 planes: 1 = queries pre-split once, DMA'd into LDS per slab; 0 = split per slab
  OPT_H3S_W12_TALL,      // small-M plan of the w12 / fc1 GEMM of one image: 1 = 192 x 128 tiles (one workgroup per CU), 0 = 128 x 128
  OPT_ATTN_H3_QG,        // attention_h3: 32-query groups per wave, 1 (four waves of 32 queries, default) or 2 (two waves of 64: A/B)
  OPT_ATTN_H3_KS,        // attention_h3: key splits across the waves of a workgroup, 0 = by grid size (2 when all workgroups are resident), 1, 2
  OPT_H3S_LN_LEAD,       // one image per call: LayerNorm as the lead role of its consumer GEMM's launch (LN1 + qkv, LN2 + w12); 0 = two launches
  OPT_H3_LN_LEAD,        // batched calls: LayerNorm as lead workgroups interleaved with its consumer GEMM's tiles (LN1 + qkv, LN2 + fc1 / w12); 0 = two launches
  OPT_VLAD_GATHER_V,     // one-pass VLAD kernel at D = 1536: variants of the register-indexed gather kept for the hazard study (0 = shipped)
  OPT_TOPK_SCREEN,       // many-query retrieval: score panels on the leading fp16 planes + exact re-scoring of the rows inside the bound (scores_screen.hip): -1 = where it pays, 0 = never, 1 = wherever possible
  OPT_COUNT
};
int64_t option(Option o);

int launch_status(const char* what);

// A kernel that needs more than the default 64 KiB of dynamic LDS must be told so with hipFuncSetAttribute -- a property of
// the (kernel, DEVICE) pair: a process that drives several GPUs (device = "cuda:N", the sharded bench) has to set it on each.
// One DynLds per launch site (a static local: one per kernel instantiation) remembers the devices already done as a bit set;
// a failed call is reported and NOT remembered.
struct DynLds { std::atomic<unsigned long long> seen{0}; };
inline int ensure_dyn_lds(DynLds& st, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDevice");
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev < 64 && (st.seen.load(std::memory_order_relaxed) & bit)) return ANYLOC_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  if (dev < 64) st.seen.fetch_or(bit, std::memory_order_relaxed);
  return ANYLOC_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
  char* base;
  size_t cap, off;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), cap(n), off(0) {}
  template <class T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= cap; }
};

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- exact three-way bf16 split (split-bf16 GEMM / attention): x = x1 + x2 + x3, round-to-nearest-even at every
// step, residuals exact in fp32.  Two values per call: v_cvt_pk_bf16_f32 rounds and packs a pair in one instruction;
// pk[plane] = (plane of a) | (plane of b) << 16.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_x3(float a, float b, unsigned pk[3]) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    f32x2 v;
    v[0] = a; v[1] = b;
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    pk[pl] = u;
    a -= __uint_as_float(u << 16);               // exact
    b -= __uint_as_float(u & 0xffff0000u);
  }
}

// ---- internal kernels shared between translation units ---------------------
enum GemmEpilogue {
  EPI_STORE = 0,      // C = acc (+ bias)
  EPI_GELU = 1,       // C = gelu_erf(acc + bias)
  EPI_LS_RESID = 2,   // C = resid + gamma * (acc + bias)        (C may alias resid)
  EPI_SWIGLU = 3,     // C[:, n/2] = silu(acc_gate + b) * (acc_val + b), rows pair-interleaved
  EPI_PATCH = 4,      // C[b*T + 1 + p, :] = acc + bias + pos[1 + p, :]
  // gemm_h3 only: the result leaves the kernel already quantised for its consumer (no fp32 round trip)
  EPI_QKV_PLANES = 5, // q | k | v written as per-head two-plane fp16 tiles + one 2^-e per (32-row group, head): attention_h3
  EPI_GELU_H2 = 6,    // gelu(acc + bias) written as the h2 image of the next GEMM, rows scaled by the caller's c_inv
  EPI_SWIGLU_H2 = 7,  // silu(gate) * value, the same
  // gemm_h3 only, weights in the 16-channel block layout (anyloc_vit_block_h2.fc1_layout = 1): the product is formed
  // transposed (weights as the MFMA A operand), a lane holds one token and 8 consecutive gate / value channels
  EPI_SWIGLU_T = 8,   // C[:, n/2] = silu(gate) * value as fp32
  EPI_SWIGLU_T_H2 = 9 // ... as the h2 image of the next GEMM, 16-byte chunks straight from the accumulators
};

struct GemmProblem {
  const float* A; int64_t lda;
  const float* W; int64_t ldw;
  float* C; int64_t ldc;
  int64_t M, N, K;
  const float* bias;     // [N] or null
  const float* gamma;    // EPI_LS_RESID
  const float* resid;    // EPI_LS_RESID, leading dim ldc
  const float* pos;      // EPI_PATCH: [T, N]
  int patches;           // EPI_PATCH: patches per image (T = patches + 1)
  float* rowsq;          // optional: rowsq[m] = sum_k A[m,k]^2 (written by the n-block 0 column)
  const char* tag;       // profiling label
  // split-K (gemm_nt_splitk): K is the SLICE length, slice s contracts columns [s K, (s+1) K) of both operands (row
  // strides lda / ldw unchanged) into C + s * c_split_stride (and rowsq + s * M): the caller sums the slices
  int ksplit; int64_t c_split_stride;
};
int gemm_nt(const GemmProblem& p, int epilogue, hipStream_t stream);
// C_s[M, N<=64] = A[:, slice s] W[:, slice s]^T for s < ksplit in ONE launch (grid.y = slice) on 128 x 64 tiles, with the
// partial row sums of squares of A: few-query retrieval, where a plain GEMM would have too few tiles to stream HBM
int gemm_nt_splitk(const GemmProblem& p, hipStream_t stream);
// the same pass with both operands split on the fly into three bf16 planes (six bf16 MFMA products, fp32 accumulate):
// part[s][row][0..63] and rsq_part[s][row] for K slices s < ksplit of length kslice (scores_x6.hip)
int scores_fewq_x6(const float* db, int64_t ldd, int64_t rows, const float* queries, int64_t ldq, int64_t nq, int64_t kslice,
                   int ksplit, float* part, float* rsq_part, hipStream_t stream);
// the same pass on two fp16 planes under a running power-of-two row scale: three fp16 MFMA products (scores_h3.hip); qinv[nq] =
// 2^-e of the query rows (row_scales_h2)
size_t fewq_query_image_bytes(int64_t dim);
int fewq_query_image(const float* queries, int64_t ldq, int64_t nq, const float* qinv, int64_t dim, unsigned char* qimg,
                     hipStream_t stream);
int scores_fewq_h3(const float* db, int64_t ldd, int64_t rows, const float* queries, int64_t ldq, int64_t nq, const float* qinv,
                   const unsigned char* qimg, int64_t kslice, int ksplit, float* part, float* rsq_part, hipStream_t stream);

// split-bf16 GEMM on three-plane bf16 operand images (gemm_x6.hip)
struct X6Problem {
  const unsigned char* A3; int64_t RA;     // plane image of A [M, 16*K16], RA = rows the image was built with
  const unsigned char* W3; int64_t RW;     // plane image of W [N, 16*K16]
  float* C; int64_t ldc;
  unsigned char* C3; int64_t RC;            // optional (GELU / SwiGLU): write the result as a plane image with RC rows
  int64_t M, N;
  int K16;                                  // k-blocks of 16
  int64_t a_off, w_off;                     // byte offset of A3 / W3 inside its image (row sub-range of a larger image)
  const float* bias;
  const float* gamma;                       // EPI_LS_RESID
  const float* resid;                       // EPI_LS_RESID, leading dim ldc
  const char* tag;
};
size_t x3_bytes(int64_t rows, int64_t K);
int split_x3(const float* x, int64_t ldx, int64_t rows, int64_t K, void* x3, hipStream_t stream);
int gemm_x6(const X6Problem& p, int epilogue, hipStream_t stream);
int layernorm_x3(const float* x, const float* w, const float* b, int64_t rows, int dim, float eps, void* x3,
                 hipStream_t stream);

// row-scaled two-term fp16 GEMM on two-plane operand images (gemm_h3.hip)
struct H3Problem {
  const unsigned char* A2; int64_t RA; const float* a_inv;   // image of A [M, 16*K16], rows of the image, 2^-e per row
  const unsigned char* W2; int64_t RW; const float* w_inv;   // image of W [N, 16*K16]
  float* C; int64_t ldc;
  int64_t M, N;
  int K16;
  int64_t a_off, w_off;                     // byte offset of A2 / W2 inside its image (row sub-range)
  const float* bias;
  const float* gamma;                       // EPI_LS_RESID
  const float* resid;                       // EPI_LS_RESID, leading dim ldc
  int accumulate;                           // EPI_STORE: C += A W^T (a long contraction cut into K chunks, one launch each)
  int fast_silu;                            // EPI_SWIGLU_H2: SiLU on v_exp_f32 / v_rcp_f32 (set by gemm_h3 from option h3_fast_silu)
  int epi_lds;                              // EPI_LS_RESID: transposed 16-byte epilogue through LDS (set by gemm_h3)
  int group_m;                              // tile-rows per XCD scheduling group (set by gemm_h3)
  // split-K (small-M plans, gemm_h3s.hip): ksplit > 1 cuts the contraction into ksplit ranges of kper k-blocks, one
  // workgroup each; partial accumulators meet in sk_part, the last arrival of a tile (sk_tickets, zero between launches)
  // sums them in split order and runs the epilogue.  Both buffers: h3_split_workspace().
  int ksplit, kper;
  float* sk_part; unsigned* sk_tickets;
  int kind;                                 // which block GEMM this is (H3_KIND_*: plan table / option h3s_mask); 0 = other
  const float* pos; int patches;            // EPI_PATCH: position table [patches + 1, N]; row b * patches + p -> C row b * (patches + 1) + 1 + p
  // EPI_QKV_PLANES: N = 3 * heads * 64; see QkvPlanes below
  unsigned char* qkv_planes; float* qkv_inv; int heads; int64_t groups;
  // EPI_GELU_H2 / EPI_SWIGLU_H2: output image (RC rows) quantised with the given per-row 2^-e (c_inv[row])
  unsigned char* C2; int64_t RC; const float* c_inv;
  // FFN-bound telemetry (optional): c_max[row] = bits of the largest scaled magnitude the row holds in the output image, merged by
  // atomicMax (one per row and wave: +0.09 ms per one-image ViT-g forward, lost in the noise of a batched one)
  unsigned* c_max;
  // LayerNorm LEAD role (small-M plans, gemm_h3s.hip; round 6): when ln_x is set, the first ln_wgs workgroups of the launch
  // normalise the rows of ln_x (one row per wave, the arithmetic of layernorm_h2) straight into THIS GEMM's operand image (A2 /
  // a_inv, and c_inv when ln_bound is set) with write-through stores and count the rows of every BM-row tile in ln_tickets; a
  // GEMM workgroup waits for its row tile's count before it stages A.  One launch instead of two for LN1 + qkv and LN2 + w12.
  const float* ln_x; const float* ln_w; const float* ln_b; float ln_eps; int ln_dim;
  float ln_bound[4]; int ln_has_bound;
  unsigned* ln_tickets; int ln_wgs;
  const char* tag;
};

// Layout of the Q / K / V operand tiles attention_h3 consumes (written by gemm_h3's EPI_QKV_PLANES epilogue).
// Rows (tokens of all images, M = batch * T) are cut into GLOBAL groups of 32 rows (group g = rows 32g .. 32g+31,
// independent of image boundaries); every (part p in {q,k,v}, head h, group g) owns one power-of-two scale
//   inv[(p * heads + h) * groups + g] = 2^-e,  max |x| * 2^e in [2^14, 2^15)  over the 32 x 64 values of the tile,
// and two fp16 planes (x * 2^e = hi + lo) of 4 KiB each at byte offset (((p * heads + h) * groups + g) * 2 + plane) * 4096:
//   q, k:  [row 0..31][8 chunks of 8 d]   chunk c of row r sits at  r * 128 + ((c ^ ((r >> 1) & 7)) << 4)
//   v:     [d 0..63][4 chunks of 8 rows]  chunk (hh * 2 + s) of column d holds rows (j & 3) + 8 * (2 s + (j >> 2)) + 4 hh,
//          j = 0..7 -- the order the MFMA C/D registers of a 32-row block hold them -- at d * 64 + ((chunk ^ ((d >> 2) & 3)) << 4)
// Both images are what a wave reads conflict-free with ds_read_b128 after a linear global -> LDS DMA of the tile.
inline size_t qkv_planes_bytes(int64_t rows, int heads) { return (size_t)3 * heads * ((rows + 31) / 32) * 2 * 4096; }
inline size_t qkv_inv_count(int64_t rows, int heads) { return (size_t)3 * heads * ((rows + 31) / 32); }
size_t h2_bytes(int64_t rows, int64_t K);
int split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, hipStream_t stream);
// rows of any width (K % 16 == 0; used above 4096 columns): also returns the rows' sums of squares when row_sumsq != nullptr
int split_h2_wide(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, float* row_sumsq,
                  hipStream_t stream);
// the first half of it alone: inv_scale[row] = 2^-e of the row-scaled split (and the rows' sums of squares when asked)
int row_scales_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, float* inv_scale, float* row_sumsq, hipStream_t stream);
int split_h1_wide(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, float* row_sumsq, float* resid_sq,
                  hipStream_t stream);
// bound (HOST array of 4 floats) != nullptr: also writes bound_inv[row] = 2^-e for an upper bound of the FFN hidden activation of that row
// (Cauchy-Schwarz: |fc1 output| <= ||ln(x)||_2 * max_j ||W_j||_2 + max |b|), bound = {gate_norm, gate_bias, val_norm, val_bias}
int layernorm_h2(const float* x, const float* w, const float* b, int64_t rows, int dim, float eps, void* h2,
                 float* inv_scale, hipStream_t stream, const float* bound = nullptr, float* bound_inv = nullptr);
// attention on the tiles of EPI_QKV_PLANES; writes the h2 image (+ per-row 2^-e) the projection GEMM reads
int attention_h3(const unsigned char* planes, const float* inv, int64_t batch, int T, int D, int heads,
                 unsigned char* out2, float* out_inv, hipStream_t stream);
// test / fallback producer of the same tiles from an fp32 [rows, 3D] buffer
int qkv_planes_from_f32(const float* qkv, int64_t rows, int D, int heads, unsigned char* planes, float* inv,
                        hipStream_t stream);
// FFN-bound telemetry: rowmax [nblocks][M] = bits of the largest scaled magnitude every row of a block's fc2 operand image holds
// (atomicMax by the fc1 / w12 epilogue, zero = block not run fused) -> out[l * groups + g] = max over the rows of group g
// (rows_per_group consecutive rows: an image, or all M) of 2^15 / rowmax; 0 for a block that left no maxima
int ffn_looseness(const unsigned* rowmax, int nblocks, int64_t M, int64_t rows_per_group, float* out, hipStream_t stream);
int gemm_h3(const H3Problem& p, int epilogue, hipStream_t stream);
// small-M plans (gemm_h3s.hip): GEMMs of fewer than ~2 workgroups of 128 x 256 per CU -- one or a few images per call
enum { H3_KIND_OTHER = 0, H3_KIND_QKV = 1, H3_KIND_PROJ = 2, H3_KIND_FC1 = 3, H3_KIND_FC2 = 4 };
constexpr size_t H3_SPLIT_PART_BYTES = 48u << 20;      // partial accumulators of one split-K launch (<= 1024 workgroups x 32 KiB + slack)
constexpr size_t H3_SPLIT_TICKETS = 4096;               // tiles of one split-K launch
inline size_t h3_split_workspace_bytes() { return H3_SPLIT_PART_BYTES + H3_SPLIT_TICKETS * sizeof(unsigned) + 512; }
bool h3_small_supported(int epilogue);
// would gemm_h3 run this GEMM with the LayerNorm lead role (H3Problem::ln_x)?  The caller then skips its LayerNorm launch.
bool h3_ln_lead_feasible(const H3Problem& p, int epilogue);       // gemm_h3.hip: small-M rule below or the batched rule
bool h3s_ln_lead_feasible(const H3Problem& p, int epilogue);      // gemm_h3s.hip (one image per call)
int h3_lead_plan_check(int tiles_m, int tiles_n, int group_m, int64_t M, unsigned* grid);
int gemm_h3_small(const H3Problem& p, int epilogue, hipStream_t stream);
// the same GEMM on v_mfma_f32_16x16x32_f16 (gemm_h3m.hip); ANYLOC_ERR_UNSUPPORTED for epilogues it does not have
int gemm_h3m(const H3Problem& p, int epilogue, hipStream_t stream);
// screened retrieval (scores_screen.hip)
float screen_accum(int k16_chunk, int chunks);
int screen_resid(const unsigned char* img, int64_t R, int K16, int64_t rows, const float* inv, const float* ss, float* rho,
                 unsigned* rho_max, hipStream_t stream);
int gemm_screen(const H3Problem& p, hipStream_t stream);
int screen_rho_max(const float* rho, int64_t n, unsigned* rho_max, hipStream_t stream);
int screen_rho_from_resid(const float* resid_sq, const float* inv, const float* ss, int64_t rows, float* rho, unsigned* rho_max,
                          hipStream_t stream);
int screen_margins(const float* qn, const float* rho_q, const unsigned* rho_max, const unsigned* ss_max, int64_t nq, int metric,
                   float accum, float* margin, hipStream_t stream);
int screen_compact(const float* scores, int64_t ld, int64_t ncols, int64_t nq, int k, int metric, const float* qn, const float* dn,
                   const float* dnorm, const float* thr, const float* margin, int cmax, int* cand, int* count, int* overflow,
                   hipStream_t stream);
bool screen_rescore_supported(int64_t dim);
int screen_rescore(const float* queries, const float* db, int64_t dim, int64_t nq, int cmax, const int* cand, const int* count,
                   int metric, const float* qn, const float* dn, const float* dnorm, float* cand_v, hipStream_t stream);
int screen_select(const int* cand, const float* cand_v, const int* count, int cmax, int64_t col_base, int64_t nq, int k, float* run_v,
                  int64_t* run_i, int first, hipStream_t stream);

int l2norm_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows,
                int64_t dim, float eps, hipStream_t stream);
int layernorm(const float* x, float* y, const float* w, const float* b, int64_t rows, int dim,
              float eps, hipStream_t stream);
int im2col(const float* img, float* col, int64_t batch, int H, int W, int P, int kpad,
           hipStream_t stream);
int cls_rows(float* x, const float* cls, const float* pos, int64_t batch, int T, int dim,
             hipStream_t stream);
int facet_rows(const float* src, int64_t lds_, int coff, float* out, int64_t ldo, int ooff,
               int64_t batch, int T, int skip, int rows_per_img, int dim, int normalize, float eps,
               hipStream_t stream);
int attention(const float* qkv, float* out, int64_t batch, int T, int D, int heads,
              hipStream_t stream, unsigned char* out3 = nullptr,    // out3: write the result as a plane image instead
              bool x6 = false);                                      // x6: split-bf16 matrix products

// single-pass fused VLAD / k-means (vlad_fused.hip)
struct FusedArgs {
  const float* x;          // [total, D] tokens / rows
  const int64_t* offsets;  // VLAD: [units+1] token offsets (device); k-means: null
  int64_t chunk_rows;      // k-means: rows per unit
  int64_t total;           // total rows
  int D, K;
  const float* chat;       // [32, D] fpk-normalised centres (rows >= K are zero)
  const float* cbias;      // [32] additive score bias (euclidean mode) or zeros
  const float* centers;    // [K, D] raw centres (VLAD residual) or null
  float* out;              // VLAD: [units, K*D]; k-means: [units, K, D] partial sums
  unsigned* cnt_part;      // k-means: [units, K] label counts; VLAD: null
  int64_t* lab64;          // optional [total] labels
  int norm_descs, intra;
  int metric;              // 0 cosine (||chat_k|| = 1), 1 euclidean (chat = 2 c, cbias = -||c||^2): scales fused3's error bound
  int parts;               // VLAD: workgroups per image (1 = one each); > 1 needs the two buffers below
  float* part_buf;         // [units, parts, K, D] partial sums
  unsigned* part_tickets;  // [units] arrival counters (zeroed by the launcher)
};
bool fused_supported(int64_t D, int64_t K);
int vlad_fused(const FusedArgs& a, int64_t units, bool kmeans, hipStream_t stream);

}  // namespace anyloc
