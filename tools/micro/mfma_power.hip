// Micro-benchmark: sustained fp16 MFMA rate of the whole chip on RANDOM operands, register-resident (no LDS, no HBM in the
// loop), for the two instruction shapes -- does one of them deliver more TFLOP/s inside the power limit?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/micro/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(t * 8 + i) & 0xfffff];
    b[i] = src[(t * 8 + 4 + i) & 0xfffff];
  }
  float sum = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][2];                       // 8 independent accumulators of 16 registers
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  } else {
    f32x4 acc[4][4];                        // 16 independent accumulators of 4 registers: the same flops per iteration
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  }
  out[t] = sum;
}

int main() {
  const int n = 1 << 20;
  std::vector<_Float16> h(n * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 64.0f);
  f16x8* src;
  float* out;
  hipMalloc(&src, n * 16);
  hipMalloc(&out, 4 << 20);
  hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
  std::vector<_Float16> z(n * 8, (_Float16)0.f);
  const int blocks = 256 * 2, iters = 20000;
  for (int data = 0; data < 2; ++data) {
    if (data == 1) hipMemcpy(src, z.data(), n * 16, hipMemcpyHostToDevice);
    for (int shape : {32, 16, 32, 16}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      // ~0.25 s of back-to-back launches first (the power controller needs far longer than one 6-ms kernel to settle), then
      // the average of the next 20
      for (int rep = 0; rep < 40; ++rep) {
        if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; ++rep) {
        if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20.0f;
      // per wave and iteration: 8 x (32*32*16*2) = 16 x (16*16*32*2) = 262144 flops
      const double fl = (double)blocks * 4 * iters * 262144.0;
      printf("%s operands  mfma_%s  %.3f ms  %.1f TFLOP/s\n", data ? "zero  " : "random", shape == 32 ? "32x32x16" : "16x16x32", ms,
             fl / ms / 1e9);
    }
  }
  return 0;
}
