// Micro-benchmark: HBM read rate of the few-query retrieval's access pattern.  A workgroup streams a tile of 128 database rows
// (row stride 49 152 floats = 192 KiB) along a K slice; per step every row contributes one contiguous SEG-byte segment
// (SEG = 128: the 32-k slabs of scores_fewq_h3_kernel -- one 128-byte line per row and wave-level load covers 8 rows;
// 256 / 512 / 1024: longer runs per row, fewer rows per step, the same 16 KiB per step), two steps in flight.  Loads only (values are
// folded into a checksum), 2 workgroups of 256 threads per CU like the real kernel.  Question: how much of the gap between the
// kernel's ~4 TB/s and a plain stream is the pattern itself?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/row_stream tools/micro/row_stream_pattern.hip && /tmp/row_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// SEG bytes per row and step; a step of the workgroup = 16 KiB = (16384 / SEG) rows x SEG bytes = 4 loads of 16 bytes per
// thread (NLD = 8: 32 KiB, 8 loads), two steps in flight; the tile's 128 rows are walked in 128 / (16384 / SEG) row groups, one after the other
template <int SEG, int NLD = 4>
__global__ __launch_bounds__(256, 2) void stream_rows(const float* __restrict__ db, int64_t ldd, int64_t kslice, float* __restrict__ out) {
  constexpr int LPR = SEG / 16;                  // lanes per row segment
  constexpr int RPP = 256 / LPR;                 // rows per pass (one load instruction of the workgroup)
  constexpr int RPS = RPP * NLD;                 // rows per step
  const int tid = threadIdx.x;
  const int kq = tid % LPR, r0 = tid / LPR;
  const int steps = (int)(kslice * 4 / SEG);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < 128 / RPS; ++g) {
    const float* base = db + ((int64_t)blockIdx.x * 128 + g * RPS) * ldd + (int64_t)blockIdx.y * kslice + 4 * kq;
    f32x4 cur[NLD], nxt[NLD];
    auto fetch = [&](int s, f32x4 (&dst)[NLD]) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        dst[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (int64_t)(r0 + RPP * i) * ldd + (int64_t)s * (SEG / 4)));
    };
    fetch(0, cur);
    for (int s = 0; s < steps; ++s) {
      fetch(s + 1 < steps ? s + 1 : s, nxt);
#pragma unroll
      for (int i = 0; i < NLD; ++i) acc += cur[i];
#pragma unroll
      for (int i = 0; i < NLD; ++i) cur[i] = nxt[i];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;     // keeps the loads
}

// the plain stream: the same bytes read as one contiguous range per workgroup
__global__ __launch_bounds__(256, 2) void stream_flat(const float* __restrict__ db, int64_t per_wg, float* __restrict__ out) {
  const f32x4* p = reinterpret_cast<const f32x4*>(db + (int64_t)blockIdx.x * per_wg) + threadIdx.x;
  const int steps = (int)(per_wg / 4 / 256);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < steps; s += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(s + u) * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

template <typename F>
double time_ms(F launch, int reps) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int64_t rows = 9984, dim = 49152;          // 78 tiles of 128 rows; 1.96 GB
  float *db, *out;
  CHECK(hipMalloc(&db, sizeof(float) * rows * dim));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(db, 0x3c, sizeof(float) * rows * dim));
  const double gb = sizeof(float) * (double)rows * dim * 1e-9;
  for (int S : {4, 6, 8, 12}) {
    const int64_t ks = dim / S;
    const dim3 grid((unsigned)(rows / 128), (unsigned)S);
    const double t128 = time_ms([&] { hipLaunchKernelGGL(stream_rows<128>, grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    const double t256 = time_ms([&] { hipLaunchKernelGGL(stream_rows<256>, grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    const double t512 = time_ms([&] { hipLaunchKernelGGL(stream_rows<512>, grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    const double t1k = time_ms([&] { hipLaunchKernelGGL(stream_rows<1024>, grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    const double u256 = time_ms([&] { hipLaunchKernelGGL((stream_rows<256, 8>), grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    const double u512 = time_ms([&] { hipLaunchKernelGGL((stream_rows<512, 8>), grid, dim3(256), 0, 0, db, dim, ks, out); }, 10);
    printf("K slices %2d (%4u workgroups), TB/s at 16 KiB per step: 128 B per row %.2f (%.3f ms) | 256 B %.2f | 512 B %.2f | 1024 B %.2f || 32 KiB per step: 256 B %.2f | 512 B %.2f\n",
           S, grid.x * grid.y, gb / t128, t128, gb / t256, gb / t512, gb / t1k, gb / u256, gb / u512);
  }
  for (int wgs : {512, 1024, 2048}) {
    const int64_t per = rows * dim / wgs / 4096 * 4096;
    const double t = time_ms([&] { hipLaunchKernelGGL(stream_flat, dim3(wgs), dim3(256), 0, 0, db, per, out); }, 10);
    printf("flat stream, %4d workgroups: %.3f ms = %.2f TB/s\n", wgs, t, sizeof(float) * (double)per * wgs * 1e-9 / t);
  }
  return 0;
}
