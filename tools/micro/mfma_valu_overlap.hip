// Micro-benchmark: do the matrix pipe and the vector ALU of a SIMD run side by side when they are fed by DIFFERENT waves
// (and by the same wave)?  One workgroup of 8 waves per CU = 2 waves per SIMD (wave i sits on SIMD i % 4):
//   mode 0  all 8 waves: MFMA loop only                  mode 1  all 8 waves: VALU loop only (v_fma_f32)
//   mode 2  waves 0-3 MFMA loop, waves 4-7 VALU loop     (different waves of one SIMD feed the two pipes)
//   mode 3  every wave: MFMA loop and VALU loop INTERLEAVED in one instruction stream (independent registers)
//   mode 4  as 1 with v_exp_f32 (quarter rate)           mode 5  as 2 with v_exp_f32
// Operands are zero (no power throttling in the picture).  If the pipes overlap, t(2) ~ max(t(0), t(1)) / 2 ... ; if they
// serialise, t(2) ~ (t(0) + t(1)) / 2.  Times are per launch; the per-wave work is the same in every mode.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/micro/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 128 matrix-pipe cycles per call: 4 MFMAs of 32 x 32 x 16 (8 passes each) or 8 of 16 x 16 x 32 (4 passes each)
template <int SHAPE>
__device__ __forceinline__ void mfma_body(f32x16 (&acc)[4], const f16x8& a, const f16x8& b) {
  if constexpr (SHAPE == 1) {           // ONE dependent chain: every MFMA accumulates into the result of the one before
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
  } else if constexpr (SHAPE == 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  } else {
    f32x4* q = reinterpret_cast<f32x4*>(&acc[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b));
  }
}

template <bool EXP>
__device__ __forceinline__ void valu_body(float (&x)[16], float c) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if constexpr (EXP)
      asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    else
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
  }
}

// per iteration: 4 MFMAs of 32 x 32 x 16 (8 passes: 32 cycles each = 128 matrix-pipe cycles) and / or 32 VALU instructions
// (fma: 4 cycles each = 128 VALU cycles; exp: 16 cycles each)
template <int MODE, int SHAPE>
__global__ __launch_bounds__(512) void overlap_kernel(float* __restrict__ out, int iters, float c) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  float x[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = c * (float)(i + threadIdx.x);
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)0.f, b[i] = (_Float16)0.f;
  constexpr bool EXP = MODE >= 4;
  const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 5) && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 4 || MODE == 3 || ((MODE == 2 || MODE == 5) && wave >= 4);
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(8 * i + j) & 15]) : "v"(c));
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) mfma_body<SHAPE>(acc, a, b);
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
      valu_body<EXP>(x, c);
      valu_body<EXP>(x, c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int SHAPE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((overlap_kernel<MODE, SHAPE>), dim3(256), dim3(512), 0, 0, out, iters, 0.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((overlap_kernel<MODE, SHAPE>), dim3(256), dim3(512), 0, 0, out, iters, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* what[6] = {"all waves MFMA", "all waves v_fma", "4 waves MFMA + 4 waves v_fma (2 per SIMD, one each)",
                         "all waves MFMA and v_fma interleaved", "all waves v_exp", "4 waves MFMA + 4 waves v_exp"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int shape = 32; shape >= 0; shape -= 16) {
      float t[6];
      if (shape == 0) {
        const float u[6] = {run<0, 1>(out, iters), run<1, 1>(out, iters), run<2, 1>(out, iters), run<3, 1>(out, iters), run<4, 1>(out, iters), run<5, 1>(out, iters)};
        for (int m = 0; m < 6; ++m) t[m] = u[m];
      } else if (shape == 32) {
        const float u[6] = {run<0, 32>(out, iters), run<1, 32>(out, iters), run<2, 32>(out, iters), run<3, 32>(out, iters), run<4, 32>(out, iters), run<5, 32>(out, iters)};
        for (int m = 0; m < 6; ++m) t[m] = u[m];
      } else {
        const float u[6] = {run<0, 16>(out, iters), run<1, 16>(out, iters), run<2, 16>(out, iters), run<3, 16>(out, iters), run<4, 16>(out, iters), run<5, 16>(out, iters)};
        for (int m = 0; m < 6; ++m) t[m] = u[m];
      }
      printf("---- MFMA shape %s (mode 3 always interleaves the 32 x 32 x 16 one)\n", shape == 32 ? "32 x 32 x 16, 8 passes, 4 independent accumulators" : shape == 16 ? "16 x 16 x 32, 4 passes, 8 independent accumulators" : "32 x 32 x 16, ONE dependent chain");
      for (int m = 0; m < 6; ++m)
        printf("mode %d  %-52s %8.3f ms  %7.1f cycles per iteration (2.4 GHz)\n", m, what[m], t[m], t[m] * 1e-3 * 2.4e9 / iters);
      printf("  MFMA + v_fma on one SIMD from two waves: perfect overlap %.3f, no overlap %.3f, measured %.3f ms\n",
             (t[0] > t[1] ? t[0] : t[1]) / 2, (t[0] + t[1]) / 2, t[2]);
      printf("  MFMA + v_exp on one SIMD from two waves: perfect overlap %.3f, no overlap %.3f, measured %.3f ms\n",
             (t[0] > t[4] ? t[0] : t[4]) / 2, (t[0] + t[4]) / 2, t[5]);
    }
  }
  return 0;
}
