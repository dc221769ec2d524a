// Global pooling of patch tokens into one descriptor per image: average, max and
// generalised-mean (GeM) -- the aggregations the reference applies to the same
// DINOv2 tokens instead of VLAD (scripts/dino_v2_gp.py:130-133,
// scripts/dino_v2_gem.py:170-188).  HBM-bound: every token is read exactly once
// (N*D*4 bytes per image), nothing but the [n_img, D] result is written.
//
// Layout: grid (ceil(D/256), n_img).  A block owns 256 consecutive feature columns of one image;
// wave w streams rows w, w+4, ... as 1 KiB contiguous segments (one float4 per lane), so every
// global access is a full-line coalesced read.  The four per-wave partials are combined through
// LDS in a fixed order: the result is deterministic and independent of the launch.
#include <cmath>

#include "common.hpp"

namespace anyloc {

namespace {

enum PoolMode { POOL_AVG = 0, POOL_MAX = 1, POOL_GEM = 2, POOL_GEM_ABS = 3 };

template <int MODE>
__device__ __forceinline__ float pool_fold(float acc, float v, float p) {
  if (MODE == POOL_AVG) return acc + v;
  if (MODE == POOL_MAX) return (v > acc || v != v) ? v : acc;        // NaN propagates, as torch.max
  if (MODE == POOL_GEM) return acc + powf(v, p);                      // torch.pow(x, p): NaN for x<0, p non-integer
  return acc + powf(fabsf(v), p);
}

template <int MODE>
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ tokens, const int64_t* __restrict__ offsets,
                                                   int64_t uniform_n, int D, float p, float* __restrict__ out) {
  __shared__ float part[3][256];
  const int img = blockIdx.y;
  const int64_t r0 = offsets ? offsets[img] : img * uniform_n;
  const int64_t r1 = offsets ? offsets[img + 1] : r0 + uniform_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + 4 * lane;
  const float init = MODE == POOL_MAX ? -INFINITY : 0.f;
  float a0 = init, a1 = init, a2 = init, a3 = init;
  if (c0 + 3 < D && (D & 3) == 0) {
    const float* src = tokens + c0;
    int64_t r = r0 + wave;
    for (; r + 12 < r1; r += 16) {            // 4 independent 16-byte loads in flight per lane
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + r * D);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (r + 4) * D);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (r + 8) * D);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (r + 12) * D);
#define ANYLOC_FOLD4(v)                                                                                      \
  a0 = pool_fold<MODE>(a0, v[0], p); a1 = pool_fold<MODE>(a1, v[1], p);                                      \
  a2 = pool_fold<MODE>(a2, v[2], p); a3 = pool_fold<MODE>(a3, v[3], p);
      ANYLOC_FOLD4(v0) ANYLOC_FOLD4(v1) ANYLOC_FOLD4(v2) ANYLOC_FOLD4(v3)
    }
    for (; r < r1; r += 4) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + r * D);
      ANYLOC_FOLD4(v0)
#undef ANYLOC_FOLD4
    }
  } else {
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      if (c0 + 0 < D) a0 = pool_fold<MODE>(a0, tokens[r * D + c0 + 0], p);
      if (c0 + 1 < D) a1 = pool_fold<MODE>(a1, tokens[r * D + c0 + 1], p);
      if (c0 + 2 < D) a2 = pool_fold<MODE>(a2, tokens[r * D + c0 + 2], p);
      if (c0 + 3 < D) a3 = pool_fold<MODE>(a3, tokens[r * D + c0 + 3], p);
    }
  }
  if (wave > 0) {
    float* dst = &part[wave - 1][4 * lane];
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
  }
  __syncthreads();
  if (wave == 0) {
    float acc[4] = {a0, a1, a2, a3};
    const float n = (float)(r1 - r0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const float o = part[w][4 * lane + j];
        acc[j] = MODE == POOL_MAX ? ((o > acc[j] || o != o) ? o : acc[j]) : acc[j] + o;
      }
      float res = acc[j];
      if (MODE == POOL_AVG) res = acc[j] / n;                          // empty image: 0/0 = NaN, as torch.mean
      if (MODE == POOL_GEM) {                                          // |x|^(1/p) * sign(x)   (dino_v2_gem.py:186-188)
        const float x = acc[j] / n;
        res = powf(fabsf(x), 1.0f / p) * (x > 0.f ? 1.f : (x < 0.f ? -1.f : x));
      }
      if (MODE == POOL_GEM_ABS) res = powf(acc[j] / n, 1.0f / p);      // (:174-175)
      if (c0 + j < D) out[(int64_t)img * D + c0 + j] = res;
    }
  }
}

}  // namespace

int pool_tokens(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t uniform_n, int64_t D,
                int mode, float p, float* out, hipStream_t stream) {
  ANYLOC_CHECK_ARG(tokens && out, "pool: null pointer");
  ANYLOC_CHECK_ARG(n_img > 0 && n_img < 65536 && D > 0 && D < (1 << 30), "pool: bad n_img=%lld / D=%lld",
                   (long long)n_img, (long long)D);
  ANYLOC_CHECK_ARG(offsets || uniform_n >= 0, "pool: need offsets or a uniform token count");
  ANYLOC_CHECK_ARG(mode >= POOL_AVG && mode <= POOL_GEM_ABS, "pool: unknown mode %d", mode);
  ANYLOC_CHECK_ARG(mode < POOL_GEM || p != 0.f, "pool: GeM exponent must be non-zero");
  const dim3 grid((unsigned)((D + 255) / 256), (unsigned)n_img);
  ProfScope prof("pool_tokens", stream, 0.0, 0.0);
  switch (mode) {
    case POOL_AVG: hipLaunchKernelGGL(pool_kernel<POOL_AVG>, grid, dim3(256), 0, stream, tokens, offsets, uniform_n, (int)D, p, out); break;
    case POOL_MAX: hipLaunchKernelGGL(pool_kernel<POOL_MAX>, grid, dim3(256), 0, stream, tokens, offsets, uniform_n, (int)D, p, out); break;
    case POOL_GEM: hipLaunchKernelGGL(pool_kernel<POOL_GEM>, grid, dim3(256), 0, stream, tokens, offsets, uniform_n, (int)D, p, out); break;
    default: hipLaunchKernelGGL(pool_kernel<POOL_GEM_ABS>, grid, dim3(256), 0, stream, tokens, offsets, uniform_n, (int)D, p, out); break;
  }
  return launch_status("pool_kernel");
}

}  // namespace anyloc

extern "C" int anyloc_pool_tokens(const float* tokens, const int64_t* offsets, int64_t n_img, int64_t n_tok,
                                  int64_t D, int mode, float p, float* out, void* stream) {
  return anyloc::pool_tokens(tokens, offsets, n_img, n_tok, D, mode, p, out, (hipStream_t)stream);
}
