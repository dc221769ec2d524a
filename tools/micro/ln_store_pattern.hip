// Micro-benchmark: the memory pattern of layernorm_h2 without its arithmetic.  A block reads RB rows of 1536 floats (flat,
// coalesced) and writes them as the operand image h2[k/16][plane][row][16 fp16]: per (k-block, plane) a run of RB x 32 bytes,
// runs 32 * rows bytes apart (1 MB at 32 330 rows).  Variants: RB = 16 / 32 / 64 rows per block (512-byte / 1-KiB / 2-KiB
// runs), the same bytes written flat, read-only and write-only -- where is the ceiling of LayerNorm's 4.5 TB/s?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ln_store tools/micro/ln_store_pattern.hip && /tmp/ln_store
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                              \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

constexpr int DIM = 1536, KB = DIM / 16;

// MODE 0: read + image-order write; 1: read + flat write; 2: read only; 3: image-order write only
// block = 64 * NW threads, RB rows; a thread handles (k-block, row, half) items = 16 bytes per plane, like h2_store_chunk
template <int RB, int MODE>
__global__ __launch_bounds__(512) void ln_pattern(const float* __restrict__ x, unsigned char* __restrict__ out, int64_t rows,
                                                  float* __restrict__ sink) {
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  constexpr int ITEMS = RB * KB * 2;                       // (k-block, row, half) triples of the block
  float acc = 0.f;
#pragma unroll 4
  for (int item = tid; item < ITEMS; item += 512) {
    // read side: flat order -- item -> (row, 8 consecutive floats)
    const int r_flat = item / (KB * 2), c8 = item % (KB * 2);
    const int64_t rrow = row0 + r_flat < rows ? row0 + r_flat : rows - 1;
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {5.f, 6.f, 7.f, 8.f};
    if (MODE != 3) {
      const f32x4* src = reinterpret_cast<const f32x4*>(x + rrow * DIM + c8 * 8);
      a = src[0];
      b = src[1];
    }
    if (MODE == 2) {
      acc += a[0] + b[3];
      continue;
    }
    const u32x4 p0 = {__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
    const u32x4 p1 = {__float_as_uint(b[0]), __float_as_uint(b[1]), __float_as_uint(b[2]), __float_as_uint(b[3])};
    if (MODE == 1) {                                       // flat: the same 32 bytes where they came from
      u32x4* dst = reinterpret_cast<u32x4*>(out + (rrow * DIM + c8 * 8) * 4);
      if (row0 + r_flat < rows) {
        dst[0] = p0;
        dst[1] = p1;
      }
    } else {                                               // image order: item -> (k-block, row, half); lanes walk rows fastest
      const int kbl = item / (2 * RB), r = (item >> 1) % RB, half = item & 1;
      const int64_t row = row0 + r;
      if (row < rows) {
        unsigned char* dst = out + (((int64_t)kbl * 2) * rows + row) * 32 + (half << 4);
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + rows * 32) = p1;
      }
    }
  }
  if (MODE == 2 && acc == 123.456f) sink[0] = acc;
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

template <int RB, int MODE>
double run(const float* x, unsigned char* out, int64_t rows, float* sink) {
  const dim3 grid((unsigned)((rows + RB - 1) / RB));
  return time_ms([&] { hipLaunchKernelGGL((ln_pattern<RB, MODE>), grid, dim3(512), 0, 0, x, out, rows, sink); }, 20);
}

int main() {
  const int64_t rows = 32330;
  float *x, *sink;
  unsigned char* out;
  CHECK(hipMalloc(&x, sizeof(float) * rows * DIM));
  CHECK(hipMalloc(&out, sizeof(float) * rows * DIM + 4096));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(x, 0x3c, sizeof(float) * rows * DIM));
  const double mb = sizeof(float) * (double)rows * DIM * 1e-6;   // one direction
  for (int rep = 0; rep < 2; ++rep) {
    const double r16 = run<16, 0>(x, out, rows, sink), r32 = run<32, 0>(x, out, rows, sink), r64 = run<64, 0>(x, out, rows, sink);
    const double f16 = run<16, 1>(x, out, rows, sink), ro = run<16, 2>(x, out, rows, sink);
    const double w16 = run<16, 3>(x, out, rows, sink), w64 = run<64, 3>(x, out, rows, sink);
    printf("read + image write (%.0f + %.0f MB): 16 rows per block (512-B runs) %.1f us = %.2f TB/s | 32 rows %.1f us = %.2f | 64 rows %.1f us = %.2f\n",
           mb, mb, r16 * 1e3, 2 * mb / r16 * 1e-3, r32 * 1e3, 2 * mb / r32 * 1e-3, r64 * 1e3, 2 * mb / r64 * 1e-3);
    printf("read + flat write %.1f us = %.2f TB/s | read only %.1f us = %.2f TB/s | image write only: 512-B runs %.1f us = %.2f TB/s, 2-KiB runs %.1f us = %.2f\n",
           f16 * 1e3, 2 * mb / f16 * 1e-3, ro * 1e3, mb / ro * 1e-3, w16 * 1e3, mb / w16 * 1e-3, w64 * 1e3, mb / w64 * 1e-3);
  }
  return 0;
}
