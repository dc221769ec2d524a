// Stand-alone probe of the GPR-index-mode hazard behind DESIGN.md 4.3 (round 6): register-resident accumulators, one per
// (cluster, column), indexed by a wave-uniform cluster id in an SGPR through ONE s_set_gpr_idx_on ... s_set_gpr_idx_off region
// around three v_add_f32 (mode 0x9: src0 and dst relative), as the gather of csrc/vlad_fused.hip does.  Variants:
//   A  residual as ONE fma, no wait state behind the mode switch      (the form that was not reproducible inside the library)
//   B  the same + `s_nop 3` between s_set_gpr_idx_on and the first indexed VALU
//   C  residual as mul, sub (the shipped arithmetic), no wait state
//   R  the compiler's own lowering of `acc[j][k] += r` (reference; fma residual)
//   D  as A with `s_nop 7` IN FRONT of s_set_gpr_idx_on: the wave's issue slots are idle when the switch issues, the first
//      indexed v_add follows it immediately (inside the library this made EVERY image wrong, profiles/r05_vlad_stress_bisect.log)
//   E  as D + `s_nop 0` (one wait state) behind the switch
// Every variant runs REPS times on the same inputs; reported: results that differ bitwise from the variant's first run, and
// from the reference R (A, B, R compute the same arithmetic; C differs from R by the fma's single rounding only).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gih tools/micro/gpr_idx_hazard.hip && /tmp/gih [workgroups] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x32 __attribute__((ext_vector_type(32)));
constexpr int D = 1536, K = 32, SW = 8, CW = 3, TOK = 512;

template <int VAR>
__global__ __launch_bounds__(64 * SW) void gather(const float* __restrict__ x, const float* __restrict__ cen,
                                                  const int* __restrict__ lab, const float* __restrict__ nrm,
                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = wave * (D / SW) + CW * lane;
  const float* xb = x + (size_t)blockIdx.x * TOK * D + col;
  const int* lb = lab + (size_t)blockIdx.x * TOK;
  const float* nb = nrm + (size_t)blockIdx.x * TOK;
  f32x32 acc[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j)
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[j][k] = 0.f;
#pragma unroll 1
  for (int t0 = 0; t0 < TOK; t0 += 8) {
    int kk[8];
    float c[8][CW], v[8][CW], nq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) kk[e] = __builtin_amdgcn_readfirstlane(lb[t0 + e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      nq[e] = nb[t0 + e];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        c[e][j] = cen[(size_t)(kk[e] < 0 ? 0 : kk[e]) * D + col + j];
        v[e][j] = xb[(size_t)(t0 + e) * D + j];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kk[e];
      float r[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const float t = VAR == 2 ? v[e][j] * nq[e] - c[e][j] : __builtin_fmaf(v[e][j], nq[e], -c[e][j]);
        r[j] = k < 0 ? 0.0f : t;
      }
      const int ks = k < 0 ? 0 : k;
      if constexpr (VAR == 3) {
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[j][ks] += r[j];
      } else if constexpr (VAR == 1) {
        asm volatile("s_set_gpr_idx_on %6, 0x9\n\ts_nop 3\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                     "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                     : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "s"(ks));
      } else if constexpr (VAR == 4) {
        asm volatile("s_nop 7\n\ts_set_gpr_idx_on %6, 0x9\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                     "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                     : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "s"(ks));
      } else if constexpr (VAR == 5) {
        asm volatile("s_nop 7\n\ts_set_gpr_idx_on %6, 0x9\n\ts_nop 0\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                     "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                     : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "s"(ks));
      } else {
        asm volatile("s_set_gpr_idx_on %6, 0x9\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                     "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                     : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "s"(ks));
      }
    }
  }
  float* ob = out + (size_t)blockIdx.x * K * D + col;
#pragma unroll
  for (int k = 0; k < 32; ++k)
#pragma unroll
    for (int j = 0; j < CW; ++j) ob[(size_t)k * D + j] = acc[j][k];
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 600, reps = argc > 2 ? atoi(argv[2]) : 50;
  const size_t nx = (size_t)wgs * TOK * D, nout = (size_t)wgs * K * D;
  std::vector<float> hx(nx), hc((size_t)K * D), hn((size_t)wgs * TOK);
  std::vector<int> hl((size_t)wgs * TOK);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hc) v = rnd();
  for (auto& v : hn) v = 1.0f + 0.5f * rnd();
  for (size_t i = 0; i < hl.size(); ++i) { s = s * 1664525u + 1013904223u; hl[i] = (int)((s >> 10) % 33) - 1; }   // -1 .. 31
  float *x, *c, *n, *o;
  int* l;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&c, hc.size() * 4); (void)hipMalloc(&n, hn.size() * 4);
  (void)hipMalloc(&l, hl.size() * 4); (void)hipMalloc(&o, nout * 4);
  (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(n, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(nout), first(nout), cur(nout);
  hipLaunchKernelGGL(gather<3>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
  (void)hipMemcpy(ref.data(), o, nout * 4, hipMemcpyDeviceToHost);
  const char* names[6] = {"A  fma, no wait state", "B  fma, s_nop 3 behind s_set_gpr_idx_on", "C  mul + sub, no wait state",
                          "R  compiler lowering (reference)", "D  s_nop 7 in front of the switch, none behind",
                          "E  s_nop 7 in front, s_nop 0 behind"};
  for (int var = 0; var < 6; ++var) {
    long differ_first = 0, differ_ref = 0;
    for (int r = 0; r <= reps; ++r) {
      (void)hipMemset(o, 0xff, nout * 4);
      if (var == 0) hipLaunchKernelGGL(gather<0>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      if (var == 1) hipLaunchKernelGGL(gather<1>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      if (var == 2) hipLaunchKernelGGL(gather<2>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      if (var == 3) hipLaunchKernelGGL(gather<3>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      if (var == 4) hipLaunchKernelGGL(gather<4>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      if (var == 5) hipLaunchKernelGGL(gather<5>, dim3(wgs), dim3(64 * SW), 0, 0, x, c, l, n, o);
      (void)hipMemcpy(cur.data(), o, nout * 4, hipMemcpyDeviceToHost);
      if (r == 0) first = cur;
      for (int w = 0; w < wgs; ++w) {
        const size_t off = (size_t)w * K * D;
        if (r > 0 && memcmp(&cur[off], &first[off], (size_t)K * D * 4)) ++differ_first;
        if (var != 2 && memcmp(&cur[off], &ref[off], (size_t)K * D * 4)) ++differ_ref;
      }
    }
    printf("%-44s %d workgroups x %d repeats: %ld results differ from the first run", names[var], wgs, reps, differ_first);
    if (var != 2) printf(", %ld of %d from the reference", differ_ref, wgs * (reps + 1));
    printf("\n");
  }
  return 0;
}
