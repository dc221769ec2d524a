// Micro-benchmark: how are the waves of small workgroups spread over the four SIMDs of a CU?  24 MFMA-bound waves per CU
// (6 per SIMD if spread evenly), as workgroups of 1, 2, 3, 4, 6 or 8 waves; the same total work in every case.  Even
// spreading gives the same time for every workgroup size; "always start at SIMD 0" makes the 1-, 2- and 3-wave cases slower.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wp tools/micro/wave_placement.hip && /tmp/wp
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_waves(float* __restrict__ out, int iters) {
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)0.f, b[i] = (_Float16)0.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a), "v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 24 * 64 * 4);
  const int iters = 20000;
  const int sizes[6] = {1, 2, 3, 4, 6, 8};
  for (int rep = 0; rep < 2; ++rep)
    for (int k = 0; k < 6; ++k) {
      const int nw = sizes[k], wgs = 256 * 24 / nw;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(mfma_waves, dim3(wgs), dim3(64 * nw), 0, 0, out, iters);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_waves, dim3(wgs), dim3(64 * nw), 0, 0, out, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      // evenly spread: 6 waves per SIMD x iters x 2 MFMAs x 32 cycles
      printf("%d waves per workgroup, %5d workgroups: %7.3f ms  (even spreading at 2.4 GHz: %.3f ms)\n", nw, wgs, ms / 3,
             6.0 * iters * 2 * 32 / 2.4e6);
    }
  return 0;
}
