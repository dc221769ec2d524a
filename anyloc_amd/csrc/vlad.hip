// VLAD aggregation (hard + soft assignment) and the k-means iteration.
//
// replaces (reference utilities.py): VLAD.generate_res_vec :928-972 (the
// [N,K,D] residual tensor is never materialised), VLAD.generate :819-890,
// generate_multi :892-926, and the fast-pytorch-kmeans assign/update loop
// reached from VLAD.fit :766,:786 and predict :849.
//
// Default path (K <= 32 and D in {384, 768, 1024, 1536}): center_prep + ONE fused single-pass
// kernel (vlad_fused.hip).  This file holds the general two-pass path used for every other shape
// (and with ANYLOC_VLAD_TWO_PASS=1), the soft-assignment variant and the k-means wrappers.
//
// Two-pass stages (all on the caller's stream):
//   1. center_prep      chat_k = c_k / (||c_k|| + 1e-8)  (fpk cos_sim), zero-padded to 32 rows
//   2. gemm_nt(+rowsq)  scores[n,k] = x_n . chat_k on fp32 MFMA; the same pass
//                       accumulates ||x_n||^2 while staging the token tiles.
//                       argmax_k is invariant to fpk's positive per-row scale
//                       1/(||x_n|| + 1e-8), so tokens are used as passed.
//   3. assign           label_n = first argmax_k scores[n,k];  nrm_n = max(||x_n||, 1e-12)
//   4. accumulate       one block per (image, 256-column slice): K x 256 accumulators
//                       in LDS, tokens streamed once with coalesced loads,
//                       acc[label_n] += x_n / nrm_n - c_label   (sequential in n: deterministic)
//   5. finalize         intra-norm of each cluster block, then the global L2 norm.
// HBM-bound: algorithmic bytes per image = (N*D + 2*K*D) * 4.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace anyloc {

namespace {

constexpr int SL = 256;   // feature columns per accumulate block

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// mode 0 (cosine): chat = c / (||c|| + 1e-8), cb = 0
// mode 1 (euclid): chat = 2 c,                cb = -||c||^2   (argmax of 2ab - b^2; -a^2 is per-row constant)
__global__ __launch_bounds__(256) void center_prep_kernel(const float* __restrict__ c, float* __restrict__ chat,
                                                          float* __restrict__ cb, int K, int D, int mode) {
  __shared__ float red[4];
  const int k = blockIdx.x;
  float* dst = chat + (int64_t)k * D;
  if (k >= K) {
    for (int i = threadIdx.x; i < D; i += 256) dst[i] = 0.f;
    if (threadIdx.x == 0) cb[k] = 0.f;
    return;
  }
  const float* src = c + (int64_t)k * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) ss += src[i] * src[i];
  ss = block_sum256(ss, red);
  if (mode == 0) {
    const float den = sqrtf(ss) + 1e-8f;
    for (int i = threadIdx.x; i < D; i += 256) dst[i] = src[i] / den;
    if (threadIdx.x == 0) cb[k] = 0.f;
  } else {
    for (int i = threadIdx.x; i < D; i += 256) dst[i] = 2.0f * src[i];
    if (threadIdx.x == 0) cb[k] = -ss;
  }
}

// 32 lanes per token (coalesced score reads): lane j scans columns j, j+32, ... then a 5-step
// shuffle arg-max; ties -> lowest column (torch.max / fpk max_sim return the first maximum)
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ scores, int kpad, int K,
                                                     const float* __restrict__ rowsq, int64_t n,
                                                     int* __restrict__ lab32, int64_t* __restrict__ lab64,
                                                     float* __restrict__ nrm, int norm_descs) {
  const int64_t tok = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (tok < n) {
    const float* s = scores + tok * kpad;
    for (int c = j; c < K; c += 32) {
      const float v = s[c];
      if (v > best) { best = v; bi = c; }
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (tok < n && j == 0) {
    if (bi == 0x7fffffff) bi = 0;      // a row of NaN scores: no comparison ever succeeds
    lab32[tok] = bi;
    if (lab64) lab64[tok] = bi;
    if (nrm) nrm[tok] = norm_descs ? fmaxf(sqrtf(rowsq[tok]), 1e-12f) : 1.0f;
  }
}

// VLAD:   dst[img][k*D + d]  = sum_{n in img, label_n == k} (x[n,d]/nrm_n - c[k,d])
// KMEANS: dst[chunk][k*D + d] = sum_{n in chunk, label_n == k} x[n,d]
template <bool KMEANS>
__global__ __launch_bounds__(SL) void accumulate_kernel(const float* __restrict__ x, const int64_t* __restrict__ offsets,
                                                        int64_t chunk_rows, int64_t total, int D, int K,
                                                        const int* __restrict__ lab, const float* __restrict__ nrm,
                                                        const float* __restrict__ c, float* __restrict__ dst,
                                                        unsigned* __restrict__ cnt_part) {
  extern __shared__ float acc[];   // [K][sl] (+ [K] label histogram for the k-means column-slice 0)
  const int sl = blockDim.x;       // feature columns of this block: SL, or SL/2 when K > 128 (LDS: K * sl floats)
  const int tid = threadIdx.x;
  const int d = blockIdx.x * sl + tid;
  const int64_t g = blockIdx.y;
  int64_t n0, n1;
  if (KMEANS) {
    n0 = g * chunk_rows;
    n1 = min(n0 + chunk_rows, total);
  } else {
    n0 = offsets[g];
    n1 = offsets[g + 1];
  }
  for (int k = 0; k < K; ++k) acc[k * sl + tid] = 0.f;
  const bool live = d < D;
  const float* xp = x + (live ? d : 0);
  int64_t n = n0;
  for (; n + 8 <= n1; n += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = live ? xp[(n + u) * D] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = lab[n + u];
      if (KMEANS) {
        acc[k * sl + tid] += v[u];
      } else {
        const float cv = live ? c[(int64_t)k * D + d] : 0.f;
        acc[k * sl + tid] += v[u] / nrm[n + u] - cv;
      }
    }
  }
  for (; n < n1; ++n) {
    const float v = live ? xp[n * D] : 0.f;
    const int k = lab[n];
    if (KMEANS) {
      acc[k * sl + tid] += v;
    } else {
      const float cv = live ? c[(int64_t)k * D + d] : 0.f;
      acc[k * sl + tid] += v / nrm[n] - cv;
    }
  }
  if (live) {
    float* o = dst + g * (int64_t)K * D + d;
    for (int k = 0; k < K; ++k) o[(int64_t)k * D] = acc[k * sl + tid];
  }
  if (KMEANS && blockIdx.x == 0 && cnt_part) {
    // per-chunk label histogram (LDS atomics), reduced over chunks in a fixed order afterwards
    unsigned* hist = reinterpret_cast<unsigned*>(acc + K * sl);
    for (int k = tid; k < K; k += sl) hist[k] = 0u;
    __syncthreads();
    for (int64_t i = n0 + tid; i < n1; i += sl) atomicAdd(&hist[lab[i]], 1u);
    __syncthreads();
    for (int k = tid; k < K; k += sl) cnt_part[g * K + k] = hist[k];
  }
}

// per image: optional intra-norm of each [D] block, then global norm of the [K*D] vector (in place)
__global__ __launch_bounds__(1024) void vlad_finalize_kernel(float* __restrict__ v, int K, int D, int intra) {
  __shared__ float knorm[256];
  __shared__ float red[16];
  float* p = v + (int64_t)blockIdx.x * K * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < K; k += 16) {
    float ss = 0.f;
    if (intra) {
      for (int i = lane; i < D; i += 64) { const float t = p[(int64_t)k * D + i]; ss += t * t; }
      ss = wave_sum(ss);
    }
    if (lane == 0) knorm[k] = intra ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
  }
  __syncthreads();
  const int total = K * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < total; i += 1024) {
    const float t = p[i] / knorm[i / D];
    ss += t * t;
  }
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  const float gn = fmaxf(sqrtf(tot), 1e-12f);
  for (int i = threadIdx.x; i < total; i += 1024) p[i] = (p[i] / knorm[i / D]) / gn;
}

// sums[k,d] = sum over chunks of partial[chunk][k,d]  (fixed order -> deterministic)
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ part, int64_t n_chunks,
                                                            int64_t kd, float* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= kd) return;
  float s = 0.f;
  int64_t ch = 0;
  for (; ch + 16 <= n_chunks; ch += 16) {                   // 16 independent loads in flight, added in chunk order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(ch + u) * kd + i];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; ch < n_chunks; ++ch) s += part[ch * kd + i];
  sums[i] = s;
}

// counts[k] = sum over chunks of cnt_part[chunk][k]: one wave per cluster, lane l takes chunks l, l + 64, ... (integer
// sums: exact in any order), 64-lane butterfly.  (One thread per cluster walking the chunks serially took 158 us for 512
// chunks -- 3 % of a 5 M-row step.)
__global__ __launch_bounds__(64) void reduce_counts_kernel(const unsigned* __restrict__ cnt_part, int64_t n_chunks, int K,
                                                           float* __restrict__ counts) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= K) return;
  unsigned long long t = 0;
  for (int64_t ch = lane; ch < n_chunks; ch += 64) t += cnt_part[ch * K + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)t, o, 64), hi = __shfl_xor((unsigned)(t >> 32), o, 64);
    t += ((unsigned long long)hi << 32) | lo;
  }
  if (lane == 0) counts[k] = (float)t;
}

// The rest of a fast-pytorch-kmeans iteration on the device (fpk 0.1.6 fit_predict, reached from utilities.py:766,786):
//   c_new[k,:] = sums[k,:] / counts[k]  (an empty cluster divides 0 by 0: NaN -> 0),  err = sum((c_new - c_old)^2)
// One workgroup: K x D is 49 152 values at the headline shape, the launch is latency, not bandwidth; the error is summed in
// float64 in a fixed order (per thread, then a fixed tree): the same bits run to run.  The host reads back 8 bytes.
__global__ __launch_bounds__(1024) void kmeans_update_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                                             const float* __restrict__ c_old, int K, int D,
                                                             float* __restrict__ c_new, double* __restrict__ err) {
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double e = 0.0;
  const int64_t n = (int64_t)K * D;
  for (int64_t i = tid; i < n; i += 1024) {
    const int k = (int)(i / D);
    float v = sums[i] / counts[k];
    if (!(v == v)) v = 0.0f;                              // fpk: c_grad[c_grad != c_grad] = 0
    const float d = v - c_old[i];
    e += (double)(d * d);                                 // (the square in fp32, as torch computes (c_grad - c) ** 2)
    c_new[i] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long u = __double_as_longlong(e);
    const unsigned lo = __shfl_xor((unsigned)u, o, 64), hi = __shfl_xor((unsigned)(u >> 32), o, 64);
    e += __longlong_as_double(((unsigned long long)hi << 32) | lo);
  }
  if (lane == 0) red[wave] = e;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    *err = t;
  }
}

// soft assignment weights: w[n,k] = softmax_k(temp * cos(x_n, c_k)),  F.cosine_similarity eps 1e-8:
//   cos = x.c / (max(||x||,eps) * max(||c||,eps));  scores hold x_n . c_k (raw centres)
__global__ __launch_bounds__(256) void soft_weights_kernel(float* __restrict__ scores, int kpad, int K,
                                                           const float* __restrict__ rowsq,
                                                           const float* __restrict__ csq, int64_t n, float temp,
                                                           float* __restrict__ nrm, int norm_descs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float* s = scores + i * kpad;
  const float xn = fmaxf(sqrtf(rowsq[i]), 1e-8f);
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float cs = temp * (s[k] / (xn * fmaxf(sqrtf(csq[k]), 1e-8f)));
    s[k] = cs;
    m = fmaxf(m, cs);
  }
  float z = 0.f;
  for (int k = 0; k < K; ++k) { const float e = expf(s[k] - m); s[k] = e; z += e; }
  for (int k = 0; k < K; ++k) s[k] /= z;
  nrm[i] = norm_descs ? fmaxf(sqrtf(rowsq[i]), 1e-12f) : 1.0f;
}

__global__ __launch_bounds__(256) void rowsq_kernel(const float* __restrict__ x, int D, float* __restrict__ out) {
  __shared__ float red[4];
  const float* r = x + (int64_t)blockIdx.x * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) ss += r[i] * r[i];
  ss = block_sum256(ss, red);
  if (threadIdx.x == 0) out[blockIdx.x] = ss;
}

// reference quirk (utilities.py:881-884): block k = sum_q sum_c w[q,k] * (xhat_q - c_c).
// One block per (image, SLS-column slice); each thread owns one column d and K accumulators in LDS:
//   acc[k] += w[q,k] * (xhat[q,d] - c[c,d])  for every c -- the reference's own double sum, term by term in fp32
// (residual, then product), accumulated in float64 so that the result is the exact sum of the reference's fp32
// terms: the reference adds its N*K terms with torch's pairwise fp32 summation, and a sequential fp32 chain of
// N*K = 2 000 ... 90 000 adds with heavy cancellation would sit ~5e-5 away from it.
constexpr int SLS = 128;
__global__ __launch_bounds__(SLS) void soft_accumulate_kernel(const float* __restrict__ x,
                                                              const int64_t* __restrict__ offsets, int D, int K,
                                                              int kpad, const float* __restrict__ w,
                                                              const float* __restrict__ nrm,
                                                              const float* __restrict__ c, float* __restrict__ dst) {
  extern __shared__ double acc64[];   // [K][SLS] accumulators (float64), then [K][SLS] centre slice (float32)
  double* acc = acc64;
  float* cs = reinterpret_cast<float*>(acc + K * SLS);
  const int tid = threadIdx.x;
  const int d = blockIdx.x * SLS + tid;
  const bool live = d < D;
  const int64_t g = blockIdx.y;
  const int64_t n0 = offsets[g], n1 = offsets[g + 1];
  for (int k = 0; k < K; ++k) {
    acc[k * SLS + tid] = 0.0;
    cs[k * SLS + tid] = live ? c[(int64_t)k * D + d] : 0.f;
  }
  for (int64_t n = n0; n < n1; ++n) {
    const float xh = live ? x[n * D + d] / nrm[n] : 0.f;
    const float* wr = w + n * kpad;
    for (int cc = 0; cc < K; ++cc) {
      const float r = xh - cs[cc * SLS + tid];
      for (int k = 0; k < K; ++k) acc[k * SLS + tid] += (double)(wr[k] * r);
    }
  }
  if (live) {
    float* o = dst + g * (int64_t)K * D + d;
    for (int k = 0; k < K; ++k) o[(int64_t)k * D] = (float)acc[k * SLS + tid];
  }
}


// residual tensor of VLAD.generate_res_vec (reference utilities.py:959-962): out[n,k,:] = x_n / max(||x_n||, 1e-12) - c_k
// (x_n as given when !norm_descs).  One block per token; a pure HBM write of K*D floats per token.
__global__ __launch_bounds__(256) void residuals_kernel(const float* __restrict__ x, int D, int K, const float* __restrict__ c,
                                                        int norm_descs, float* __restrict__ out) {
  __shared__ float red[4];
  const int64_t n = blockIdx.x;
  const float* r = x + n * D;
  float den = 1.0f;
  if (norm_descs) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) ss += r[i] * r[i];
    den = fmaxf(sqrtf(block_sum256(ss, red)), 1e-12f);
  }
  float* o = out + n * (int64_t)K * D;
  const int d4 = D >> 2;
  for (int i = threadIdx.x; i < d4; i += 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(r)[i];
    v[0] /= den; v[1] /= den; v[2] /= den; v[3] /= den;
    for (int k = 0; k < K; ++k) {
      const f32x4 cv = reinterpret_cast<const f32x4*>(c + (int64_t)k * D)[i];
      f32x4 t;
      t[0] = v[0] - cv[0]; t[1] = v[1] - cv[1]; t[2] = v[2] - cv[2]; t[3] = v[3] - cv[3];
      reinterpret_cast<f32x4*>(o + (int64_t)k * D)[i] = t;
    }
  }
}

// VLAD from a GIVEN assignment (cache restore, reference utilities.py:843-847 / :864-868): per token the row norm
// (F.normalize denominator) and the int32 copy of its label; also the [0, n] offsets of the single image
__global__ __launch_bounds__(256) void assigned_prep_kernel(const float* __restrict__ x, int D, int64_t n, int K,
                                                            const int64_t* __restrict__ labels, int norm_descs,
                                                            float* __restrict__ nrm, int* __restrict__ lab32,
                                                            int64_t* __restrict__ offsets) {
  __shared__ float red[4];
  const int64_t t = blockIdx.x;
  float den = 1.0f;
  if (norm_descs) {
    const float* r = x + t * D;
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) ss += r[i] * r[i];
    den = fmaxf(sqrtf(block_sum256(ss, red)), 1e-12f);
  }
  if (threadIdx.x == 0) {
    nrm[t] = den;
    if (labels) lab32[t] = (int)min<int64_t>(max<int64_t>(labels[t], 0), K - 1);
    if (t == 0) { offsets[0] = 0; offsets[1] = n; }
  }
}

inline int64_t kpad_of(int64_t K) { return (K + 31) / 32 * 32; }

// Workgroups per image of the fused kernel: one workgroup per image cannot fill 256 CUs below ~200 images, so a small
// batch splits every image's token tiles over up to 8 workgroups (the last to finish reduces, vlad_fused.hip); every
// part keeps >= 2 tiles of 16 tokens on average.  Option vlad_parts forces a count (A/B, tests).
inline int fused_parts(int64_t n_img, int64_t total) {
  const int forced = (int)option(OPT_VLAD_PARTS);
  if (n_img <= 0) return 1;
  if (forced > 0) return forced > 64 ? 64 : forced;
  int64_t p = 256 / n_img;
  const int64_t tiles = (total / n_img + 15) / 16;
  if (p > tiles / 2) p = tiles / 2;
  if (p > 8) p = 8;
  return p < 1 ? 1 : (int)p;
}

// option vlad_two_pass = 1 selects the two-pass path even where the fused kernel applies (A/B tests)
inline bool two_pass_forced() { return option(OPT_VLAD_TWO_PASS) != 0; }

struct VladWs {
  float *chat, *cb, *scores, *rowsq, *nrm;
  int* lab32;
  float* part_buf;
  unsigned* tickets;
  size_t bytes;
};
VladWs carve(void* ws, size_t cap, int64_t n, int64_t D, int64_t K, int64_t n_img = 0, int parts = 1) {
  Arena a(ws, cap);
  VladWs w;
  const int64_t kp = kpad_of(K);
  w.chat = a.take<float>(kp * D);
  w.cb = a.take<float>(kp);
  w.scores = a.take<float>((n > 0 ? n : 1) * kp);
  w.rowsq = a.take<float>(n > 0 ? n : 1);
  w.nrm = a.take<float>(n > 0 ? n : 1);
  w.lab32 = a.take<int>(n > 0 ? n : 1);
  w.part_buf = nullptr;
  w.tickets = nullptr;
  if (parts > 1) {
    w.part_buf = a.take<float>(n_img * parts * K * D);
    w.tickets = a.take<unsigned>(n_img);
  }
  w.bytes = a.off;
  return w;
}

int run_scores(const float* x, int64_t n, int64_t D, const VladWs& w, int64_t K, bool with_bias, hipStream_t stream,
               const char* tag) {
  GemmProblem g{};
  g.A = x; g.lda = D;
  g.W = w.chat; g.ldw = D;
  g.C = w.scores; g.ldc = kpad_of(K);
  g.M = n; g.N = kpad_of(K); g.K = D;
  g.bias = with_bias ? w.cb : nullptr;
  g.rowsq = w.rowsq;
  g.tag = tag;
  return gemm_nt(g, EPI_STORE, stream);
}

}  // namespace
}  // namespace anyloc

using namespace anyloc;

extern "C" {

size_t anyloc_vlad_workspace_bytes(int64_t total_tokens, int64_t n_img, int64_t D, int64_t K) {
  const int parts = (fused_supported(D, K) && n_img > 0) ? fused_parts(n_img, total_tokens) : 1;
  return carve(nullptr, 0, total_tokens, D, K, n_img, parts).bytes + 256;
}

// the same for a call that passes ANYLOC_VLAD_PARTS(parts): sized for the larger of the library's count and the caller's
size_t anyloc_vlad_workspace_bytes_parts(int64_t total_tokens, int64_t n_img, int64_t D, int64_t K, int32_t parts) {
  int p = 1;
  if (fused_supported(D, K) && n_img > 0) {
    p = fused_parts(n_img, total_tokens);
    if (parts > p) p = parts > 64 ? 64 : parts;
  }
  return carve(nullptr, 0, total_tokens, D, K, n_img, p).bytes + 256;
}

int anyloc_vlad_auto_parts(int64_t total_tokens, int64_t n_img, int64_t D, int64_t K) {
  return (fused_supported(D, K) && n_img > 0 && !two_pass_forced()) ? fused_parts(n_img, total_tokens) : 1;
}

static int vlad_common_checks(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t total, int64_t D,
                              const float* centers, int64_t K, float* out) {
  ANYLOC_CHECK_ARG(offsets && centers && out, "vlad: null pointer");
  ANYLOC_CHECK_ARG(tokens || total == 0, "vlad: null tokens");
  ANYLOC_CHECK_ARG(n_img >= 0 && total >= 0, "vlad: negative size");
  ANYLOC_CHECK_ARG(K >= 1 && K <= 256, "vlad: num_clusters %lld outside [1,256]", (long long)K);
  ANYLOC_CHECK_ARG(D >= 4 && D % 4 == 0, "vlad: desc_dim %lld must be a positive multiple of 4", (long long)D);
  ANYLOC_CHECK_ARG(n_img < 65536ll * 32768, "vlad: too many images");
  return ANYLOC_OK;
}

int anyloc_vlad_hard(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t total_tokens, int64_t D,
                     const float* centers, int64_t K, unsigned flags, float* out, int64_t* labels, void* workspace,
                     size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_TRY(vlad_common_checks(tokens, offsets, n_img, total_tokens, D, centers, K, out));
  if (n_img == 0) return ANYLOC_OK;
  const bool fused = fused_supported(D, K) && !two_pass_forced();
  const int asked = (int)((flags >> 8) & 0x7fu);                 // ANYLOC_VLAD_PARTS(p): the caller's count (0 = ours)
  ANYLOC_CHECK_ARG(asked <= 64, "vlad_hard: ANYLOC_VLAD_PARTS(%d) outside 1..64", asked);
  const int parts = !fused ? 1 : asked > 0 ? asked : fused_parts(n_img, total_tokens);
  VladWs w = carve(workspace, workspace_bytes, total_tokens, D, K, n_img, parts);
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("vlad_hard: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  const int kp = (int)kpad_of(K);
  // labels = kmeans.predict(tokens) in the metric the vocabulary was built with (reference utilities.py:849 with
  // VLAD(dist_mode=...)): fpk cosine score, or fpk euclidean similarity 2ab - a^2 - b^2 (arg-max = nearest centre)
  const int metric = (flags & ANYLOC_VLAD_EUCLIDEAN) ? 1 : 0;
  // The fused kernel wherever it applies (K <= 32, the ViT widths); option vlad_two_pass = 1 selects the general path.
  if (fused) {
    // single-pass fused kernel (vlad_fused.hip): tokens are read from HBM once
    {
      ProfScope prof("vlad_center_prep", stream, 3.0 * K * D, 8.0 * K * D);
      hipLaunchKernelGGL(center_prep_kernel, dim3(kp), dim3(256), 0, stream, centers, w.chat, w.cb, (int)K, (int)D, metric);
      ANYLOC_TRY(launch_status("center_prep_kernel"));
    }
    FusedArgs fa{};
    fa.x = tokens; fa.offsets = offsets; fa.total = total_tokens;
    fa.D = (int)D; fa.K = (int)K;
    fa.chat = w.chat; fa.cbias = w.cb; fa.centers = centers; fa.metric = metric;
    fa.out = out; fa.lab64 = labels;
    fa.norm_descs = (flags & ANYLOC_VLAD_NORM_DESCS) ? 1 : 0;
    fa.intra = (flags & ANYLOC_VLAD_INTRA_NORM) ? 1 : 0;
    fa.parts = parts; fa.part_buf = w.part_buf; fa.part_tickets = w.tickets;
    return vlad_fused(fa, n_img, false, stream);
  }
  if (total_tokens > 0) {
    {
      ProfScope prof("vlad_center_prep", stream, 3.0 * K * D, 8.0 * K * D);
      hipLaunchKernelGGL(center_prep_kernel, dim3(kp), dim3(256), 0, stream, centers, w.chat, w.cb, (int)K, (int)D, metric);
      ANYLOC_TRY(launch_status("center_prep_kernel"));
    }
    ANYLOC_TRY(run_scores(tokens, total_tokens, D, w, K, metric == 1, stream, "vlad_scores_gemm"));
    {
      ProfScope prof("vlad_assign", stream, 0.0, 4.0 * total_tokens * (kp + 4));
      hipLaunchKernelGGL(assign_kernel, dim3((unsigned)((total_tokens + 7) / 8)), dim3(256), 0, stream, w.scores,
                         kp, (int)K, w.rowsq, total_tokens, w.lab32, labels, w.nrm,
                         (flags & ANYLOC_VLAD_NORM_DESCS) ? 1 : 0);
      ANYLOC_TRY(launch_status("assign_kernel"));
    }
  }
  {
    const int sl = K > 128 ? SL / 2 : SL;           // K up to 256: half-width column slices keep K * sl floats in LDS
    const size_t lds = (size_t)K * sl * sizeof(float);
    static DynLds dyn_lds_once;
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(accumulate_kernel<false>), (int)(256 * SL * (int)sizeof(float) / 2)));
    const double bytes = 4.0 * ((double)total_tokens * D + 2.0 * (double)n_img * K * D);
    ProfScope prof("vlad_accumulate", stream, 2.0 * total_tokens * D, bytes);
    // grid.y is limited to 65535: loop over image groups
    for (int64_t i0 = 0; i0 < n_img; i0 += 65535) {
      const int64_t cnt = std::min<int64_t>(65535, n_img - i0);
      hipLaunchKernelGGL(accumulate_kernel<false>, dim3((unsigned)((D + sl - 1) / sl), (unsigned)cnt), dim3(sl), lds,
                         stream, tokens, offsets + i0, (int64_t)0, total_tokens, (int)D, (int)K, w.lab32, w.nrm,
                         centers, out + i0 * K * D, (unsigned*)nullptr);
      ANYLOC_TRY(launch_status("accumulate_kernel"));
    }
  }
  {
    ProfScope prof("vlad_finalize", stream, 6.0 * n_img * K * D, 12.0 * n_img * K * D);
    hipLaunchKernelGGL(vlad_finalize_kernel, dim3((unsigned)n_img), dim3(1024), 0, stream, out, (int)K, (int)D,
                       (flags & ANYLOC_VLAD_INTRA_NORM) ? 1 : 0);
    ANYLOC_TRY(launch_status("vlad_finalize_kernel"));
  }
  return ANYLOC_OK;
}

int anyloc_vlad_soft(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t total_tokens, int64_t D,
                     const float* centers, int64_t K, float soft_temp, unsigned flags, float* out, void* workspace,
                     size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_TRY(vlad_common_checks(tokens, offsets, n_img, total_tokens, D, centers, K, out));
  if (n_img == 0) return ANYLOC_OK;
  ANYLOC_CHECK_ARG(K <= 64, "vlad_soft: num_clusters %lld > 64 unsupported", (long long)K);
  VladWs w = carve(workspace, workspace_bytes, total_tokens, D, K);
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("vlad_soft: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  const int kp = (int)kpad_of(K);
  if (total_tokens > 0) {
    // raw centres padded with zero rows as the GEMM's W operand; cb holds ||c_k||^2
    ANYLOC_HIP(hipMemsetAsync(w.chat, 0, sizeof(float) * kp * D, stream));
    ANYLOC_HIP(hipMemcpyAsync(w.chat, centers, sizeof(float) * K * D, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)K), dim3(256), 0, stream, centers, (int)D, w.cb);
    ANYLOC_TRY(launch_status("rowsq_kernel"));
    ANYLOC_TRY(run_scores(tokens, total_tokens, D, w, K, false, stream, "vlad_soft_scores_gemm"));
    hipLaunchKernelGGL(soft_weights_kernel, dim3((unsigned)((total_tokens + 255) / 256)), dim3(256), 0, stream,
                       w.scores, kp, (int)K, w.rowsq, w.cb, total_tokens, soft_temp, w.nrm,
                       (flags & ANYLOC_VLAD_NORM_DESCS) ? 1 : 0);
    ANYLOC_TRY(launch_status("soft_weights_kernel"));
  }
  {
    const size_t lds = (size_t)K * SLS * (sizeof(double) + sizeof(float));
    static DynLds dyn_lds_once;
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(soft_accumulate_kernel), (int)(64 * SLS * 12)));
    ProfScope prof("vlad_soft_accumulate", stream, 2.0 * total_tokens * D * K * K, 4.0 * total_tokens * D);
    for (int64_t i0 = 0; i0 < n_img; i0 += 65535) {
      const int64_t cnt = std::min<int64_t>(65535, n_img - i0);
      hipLaunchKernelGGL(soft_accumulate_kernel, dim3((unsigned)((D + SLS - 1) / SLS), (unsigned)cnt), dim3(SLS), lds,
                         stream, tokens, offsets + i0, (int)D, (int)K, kp, w.scores, w.nrm, centers,
                         out + i0 * K * D);
      ANYLOC_TRY(launch_status("soft_accumulate_kernel"));
    }
  }
  hipLaunchKernelGGL(vlad_finalize_kernel, dim3((unsigned)n_img), dim3(1024), 0, stream, out, (int)K, (int)D,
                     (flags & ANYLOC_VLAD_INTRA_NORM) ? 1 : 0);
  return launch_status("vlad_finalize_kernel");
}


int anyloc_vlad_soft_weights(const float* tokens, int64_t n_tok, int64_t D, const float* centers, int64_t K, float soft_temp,
                             float* weights, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(centers && weights && (tokens || n_tok == 0), "vlad_soft_weights: null pointer");
  ANYLOC_CHECK_ARG(n_tok >= 0 && K >= 1 && K <= 64 && D >= 4 && D % 4 == 0, "vlad_soft_weights: bad shape (K <= 64)");
  if (n_tok == 0) return ANYLOC_OK;
  VladWs w = carve(workspace, workspace_bytes, n_tok, D, K);
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("vlad_soft_weights: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  const int kp = (int)kpad_of(K);
  ANYLOC_HIP(hipMemsetAsync(w.chat, 0, sizeof(float) * kp * D, stream));
  ANYLOC_HIP(hipMemcpyAsync(w.chat, centers, sizeof(float) * K * D, hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)K), dim3(256), 0, stream, centers, (int)D, w.cb);
  ANYLOC_TRY(launch_status("rowsq_kernel"));
  ANYLOC_TRY(run_scores(tokens, n_tok, D, w, K, false, stream, "vlad_soft_scores_gemm"));
  hipLaunchKernelGGL(soft_weights_kernel, dim3((unsigned)((n_tok + 255) / 256)), dim3(256), 0, stream, w.scores, kp, (int)K,
                     w.rowsq, w.cb, n_tok, soft_temp, w.nrm, 0);
  ANYLOC_TRY(launch_status("soft_weights_kernel"));
  ANYLOC_HIP(hipMemcpy2DAsync(weights, sizeof(float) * K, w.scores, sizeof(float) * kp, sizeof(float) * K, (size_t)n_tok,
                              hipMemcpyDeviceToDevice, stream));
  return ANYLOC_OK;
}

int anyloc_vlad_residuals(const float* tokens, int64_t n_tok, int64_t D, const float* centers, int64_t K, unsigned flags,
                          float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(centers && out && (tokens || n_tok == 0), "vlad_residuals: null pointer");
  ANYLOC_CHECK_ARG(n_tok >= 0 && n_tok < (1ll << 31) && K >= 1 && D >= 4 && D % 4 == 0, "vlad_residuals: bad shape");
  if (n_tok == 0) return ANYLOC_OK;
  ProfScope prof("vlad_residuals", stream, (double)n_tok * K * D, 4.0 * ((double)n_tok * D * (K + 1) + (double)K * D));
  hipLaunchKernelGGL(residuals_kernel, dim3((unsigned)n_tok), dim3(256), 0, stream, tokens, (int)D, (int)K, centers,
                     (flags & ANYLOC_VLAD_NORM_DESCS) ? 1 : 0, out);
  return launch_status("residuals_kernel");
}

int anyloc_vlad_assigned(const float* tokens, int64_t n_tok, int64_t D, const float* centers, int64_t K,
                         const int64_t* labels, const float* soft_weights, unsigned flags, float* out, void* workspace,
                         size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(centers && out && (tokens || n_tok == 0), "vlad_assigned: null pointer");
  ANYLOC_CHECK_ARG((labels != nullptr) != (soft_weights != nullptr), "vlad_assigned: pass labels (hard) or soft weights, not both");
  ANYLOC_CHECK_ARG(n_tok >= 0 && n_tok < (1ll << 31) && K >= 1 && K <= 256 && D >= 4 && D % 4 == 0, "vlad_assigned: bad shape");
  ANYLOC_CHECK_ARG(labels || K <= 64, "vlad_assigned: soft weights need K <= 64");
  VladWs w = carve(workspace, workspace_bytes, n_tok, D, K);
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("vlad_assigned: workspace %zu < %zu", workspace_bytes, w.bytes);
    return ANYLOC_ERR_WORKSPACE;
  }
  int64_t* offsets = reinterpret_cast<int64_t*>(w.cb);        // 16 bytes of the (>= 128-byte) bias slot
  if (n_tok == 0) {
    ANYLOC_HIP(hipMemsetAsync(out, 0, sizeof(float) * K * D, stream));
    return ANYLOC_OK;
  }
  const int norm = (flags & ANYLOC_VLAD_NORM_DESCS) ? 1 : 0;
  hipLaunchKernelGGL(assigned_prep_kernel, dim3((unsigned)n_tok), dim3(256), 0, stream, tokens, (int)D, n_tok, (int)K, labels, norm,
                     w.nrm, w.lab32, offsets);
  ANYLOC_TRY(launch_status("assigned_prep_kernel"));
  if (labels) {
    const int sl = K > 128 ? SL / 2 : SL;
    const size_t lds = (size_t)K * sl * sizeof(float);
    ANYLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(accumulate_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 256 * SL * (int)sizeof(float) / 2));
    hipLaunchKernelGGL(accumulate_kernel<false>, dim3((unsigned)((D + sl - 1) / sl), 1u), dim3(sl), lds, stream, tokens, offsets,
                       (int64_t)0, n_tok, (int)D, (int)K, w.lab32, w.nrm, centers, out, (unsigned*)nullptr);
    ANYLOC_TRY(launch_status("accumulate_kernel"));
  } else {
    const size_t lds = (size_t)K * SLS * (sizeof(double) + sizeof(float));
    ANYLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(soft_accumulate_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 64 * SLS * 12));
    hipLaunchKernelGGL(soft_accumulate_kernel, dim3((unsigned)((D + SLS - 1) / SLS), 1u), dim3(SLS), lds, stream, tokens, offsets,
                       (int)D, (int)K, (int)K, soft_weights, w.nrm, centers, out);
    ANYLOC_TRY(launch_status("soft_accumulate_kernel"));
  }
  hipLaunchKernelGGL(vlad_finalize_kernel, dim3(1u), dim3(1024), 0, stream, out, (int)K, (int)D,
                     (flags & ANYLOC_VLAD_INTRA_NORM) ? 1 : 0);
  return launch_status("vlad_finalize_kernel");
}

// ------------------------------------------------------------------ k-means
// rows per chunk of the k-means step: one workgroup and one partial [K, D] sum per chunk, the partial sums added in chunk
// order by reduce_chunks_kernel.  At least 1024 rows per chunk and at most two chunks per CU (option kmeans_max_chunks
// overrides): every chunk costs a prologue, 196 KB of partial sums written and re-read, and with one fused workgroup
// filling a CU more chunks only add rounds -- 5 M x 1536 rows: 2048 chunks 6.67 ms, 1024 6.20, 512 5.87, 256 5.87 per
// step on clustered rows (profiles/r02_kmeans_chunks.log); two per CU keeps some slack for CUs of unequal speed.
static int64_t kmeans_chunk_rows(int64_t n) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
  }
  const int64_t forced = option(OPT_KMEANS_MAX_CHUNKS);
  const int64_t max_chunks = forced > 0 ? forced : 2ll * cus;
  int64_t rows = (n + max_chunks - 1) / max_chunks;
  return rows < 1024 ? 1024 : rows;
}

size_t anyloc_kmeans_workspace_bytes(int64_t n, int64_t D, int64_t K) {
  size_t b = carve(nullptr, 0, n, D, K).bytes;
  const int64_t rows = kmeans_chunk_rows(n), chunks = (n + rows - 1) / rows;
  b += align_up((size_t)(chunks > 0 ? chunks : 1) * K * D * sizeof(float), 256);
  b += align_up((size_t)(chunks > 0 ? chunks : 1) * K * sizeof(unsigned), 256);
  return b + 256;
}

int anyloc_kmeans_update(const float* sums, const float* counts, const float* centers_old, int64_t K, int64_t D,
                         float* centers_new, double* err, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(sums && counts && centers_old && centers_new && err, "kmeans_update: null pointer");
  ANYLOC_CHECK_ARG(K >= 1 && D >= 1 && K * D < (1ll << 31), "kmeans_update: bad shape");
  ProfScope prof("kmeans_update", stream, 3.0 * K * D, 12.0 * K * D);
  hipLaunchKernelGGL(kmeans_update_kernel, dim3(1), dim3(1024), 0, stream, sums, counts, centers_old, (int)K, (int)D, centers_new, err);
  return launch_status("kmeans_update_kernel");
}

int anyloc_kmeans_step(const float* x, int64_t n, int64_t D, const float* centers, int64_t K, int mode, float* sums,
                       float* counts, int64_t* labels, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ANYLOC_CHECK_ARG(x && centers && sums && counts, "kmeans_step: null pointer");
  ANYLOC_CHECK_ARG(n > 0, "kmeans_step: no rows");
  ANYLOC_CHECK_ARG(K >= 1 && K <= 256, "kmeans_step: K %lld outside [1,256]", (long long)K);
  ANYLOC_CHECK_ARG(D >= 4 && D % 4 == 0, "kmeans_step: D %lld must be a positive multiple of 4", (long long)D);
  ANYLOC_CHECK_ARG(mode == 0 || mode == 1, "kmeans_step: mode %d", mode);
  const size_t need = anyloc_kmeans_workspace_bytes(n, D, K);
  if (!workspace || need > workspace_bytes) {
    set_error("kmeans_step: workspace %zu < %zu", workspace_bytes, need);
    return ANYLOC_ERR_WORKSPACE;
  }
  VladWs w = carve(workspace, workspace_bytes, n, D, K);
  Arena tail(static_cast<char*>(workspace) + w.bytes, workspace_bytes - w.bytes);
  const int64_t rows = kmeans_chunk_rows(n), chunks = (n + rows - 1) / rows;
  float* part = tail.take<float>(chunks * K * D);
  unsigned* cnt_part = tail.take<unsigned>(chunks * K);
  const int kp = (int)kpad_of(K);

  hipLaunchKernelGGL(center_prep_kernel, dim3(kp), dim3(256), 0, stream, centers, w.chat, w.cb, (int)K, (int)D, mode);
  ANYLOC_TRY(launch_status("center_prep_kernel"));
  if (fused_supported(D, K) && !two_pass_forced()) {
    FusedArgs fa{};
    fa.x = x; fa.chunk_rows = rows; fa.total = n;
    fa.D = (int)D; fa.K = (int)K;
    fa.chat = w.chat; fa.cbias = w.cb; fa.metric = mode;
    fa.out = part; fa.cnt_part = cnt_part; fa.lab64 = labels;
    ANYLOC_TRY(vlad_fused(fa, chunks, true, stream));
    {
      ProfScope prof("kmeans_reduce", stream, 1.0 * chunks * K * D, 4.0 * (chunks + 1.0) * K * D);
      hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((K * D + 255) / 256)), dim3(256), 0, stream, part,
                         chunks, K * D, sums);
      ANYLOC_TRY(launch_status("reduce_chunks_kernel"));
    }
    hipLaunchKernelGGL(reduce_counts_kernel, dim3((unsigned)K), dim3(64), 0, stream, cnt_part, chunks, (int)K, counts);
    return launch_status("reduce_counts_kernel");
  }
  ANYLOC_TRY(run_scores(x, n, D, w, K, mode == 1, stream, "kmeans_scores_gemm"));
  {
    ProfScope prof("kmeans_assign", stream, 0.0, 4.0 * n * (kp + 2));
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, stream, w.scores, kp, (int)K,
                       w.rowsq, n, w.lab32, labels, (float*)nullptr, 0);
    ANYLOC_TRY(launch_status("assign_kernel"));
  }
  {
    const int sl = K > 128 ? SL / 2 : SL;
    const size_t lds = (size_t)K * sl * sizeof(float) + (size_t)K * sizeof(unsigned);
    static DynLds dyn_lds_once;
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(accumulate_kernel<true>), (int)(129 * 1024)));
    ProfScope prof("kmeans_accumulate", stream, 1.0 * n * D, 4.0 * ((double)n * D + (double)chunks * K * D));
    hipLaunchKernelGGL(accumulate_kernel<true>, dim3((unsigned)((D + sl - 1) / sl), (unsigned)chunks), dim3(sl), lds,
                       stream, x, (const int64_t*)nullptr, rows, n, (int)D, (int)K, w.lab32, (const float*)nullptr,
                       (const float*)nullptr, part, cnt_part);
    ANYLOC_TRY(launch_status("accumulate_kernel<kmeans>"));
  }
  {
    ProfScope prof("kmeans_reduce", stream, 1.0 * chunks * K * D, 4.0 * (chunks + 1.0) * K * D);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((K * D + 255) / 256)), dim3(256), 0, stream, part, chunks,
                       K * D, sums);
    ANYLOC_TRY(launch_status("reduce_chunks_kernel"));
  }
  hipLaunchKernelGGL(reduce_counts_kernel, dim3((unsigned)K), dim3(64), 0, stream, cnt_part, chunks, (int)K, counts);
  return launch_status("reduce_counts_kernel");
}

}  // extern "C"
