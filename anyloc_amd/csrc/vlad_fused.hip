// Single-pass fused VLAD / k-means kernel: tokens are read from HBM exactly once.
//
// replaces (reference utilities.py): kmeans.predict :849 + generate_res_vec :959-962 + the
// per-cluster sums / intra-norm / global norm of VLAD.generate :854-861,:889 (VLAD mode), and the
// assign + update body of fast-pytorch-kmeans' fit loop reached from VLAD.fit :786 (k-means mode).
//
// Two structures live here (vlad_fused() picks; options vlad_fused_v / kmeans_fused_v select one for A/B runs and tests):
//   vlad_fused_kernel     exact fp32-MFMA scores, centres streamed from L2, one owner thread per (cluster, column)
//   fused3_kernel         fp16 screening scores from register-resident centres + exact fp32 resolution of close calls,
//                         balanced register-indexed gather -- the default
// The first structure, described: one 1024-thread workgroup (16 waves, one per CU) owns a *unit*: an image (VLAD) or a chunk
// of rows (k-means).  It walks the unit's tokens in tiles of 16:
//   stage   16 x D tile: coalesced buffer loads (rows past the unit read as 0) -> VGPR -> LDS,
//           issued one tile ahead of its use
//   score   waves 0-7: S[16 x 32] partials over their D/8 slice on v_mfma_f32_16x16x4_f32
//           (A = tile from LDS, B = fpk-normalised centres from L2); the A fragments also give the
//           row norms ||x_n||
//   assign  512 threads: sum of the 8 partials in fixed order, first-arg-max over k by shuffles
//   gather  all 1024 threads: thread (k = tid/32, j = tid%32) owns columns 4j + 128m of cluster
//           k in REGISTERS; for every token of the tile assigned to k it adds x/||x|| - c_k
//           (VLAD) or x (k-means).  Every (k, d) has exactly one owner and tokens are visited
//           in order: no atomics, deterministic.
// VLAD mode finishes in the same launch: per-cluster L2 norm by a 32-lane shuffle reduction
// (one cluster = half a wave), global norm by a block reduction, one coalesced store.
// HBM bound: algorithmic bytes per image = (N*D + 2*K*D) * 4; LDS: 16*(D+4)*4 + 16 KiB + small.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace anyloc {

namespace {

constexpr int TT = 16;       // tokens per tile
constexpr int NTH = 1024;    // threads per workgroup
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes of one half wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), which would make every wave wait at the barrier for the HBM loads of the NEXT tile it has just
// issued -- the prefetch would overlap nothing (measured with s_memtime stamps: 25 % of the tile time).
// a pointer rebuilt from readfirstlane'd halves: tells the compiler it is wave-uniform (buffer descriptors, T20 of the guide)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// min over int64 in SALU-friendly integer ops.  (`min<int64_t>(a, b)` resolved to a floating-point overload here: the
// result came out of v_cvt_i32_f64 in a VGPR, the readfirstlane around it was folded away as "already uniform", and every
// buffer load through the descriptor it went into got a waterfall loop.)
__device__ __forceinline__ int64_t imin64(int64_t x, int64_t y) { return x < y ? x : y; }

// NV = D / 128: float4 columns per owner thread (and 16-wide k-groups per scoring wave)
template <int NV, bool KMEANS>
__global__ __launch_bounds__(NTH) void vlad_fused_kernel(FusedArgs a) {
  constexpr int D = NV * 128;
  constexpr int LD = D + 4;                  // padded LDS row: conflict-free ds_read_b128 of A fragments
  constexpr int NF = (NV + 1) / 2;           // staged float4 per thread per tile (4*D/1024 = NV/2)
  constexpr int SW = 8;                      // scoring waves 0-7: each contracts a D/8 slice (all 16 waves on D/16 slices
                                             // measured no faster: the tile loop is bound by its chain of phases, not by MFMA)
  constexpr int NG = NV * 8 / SW;            // 16-wide k-groups per scoring wave
  // vmcnt retires in order PER WAVE: a wave that has the HBM loads of the next tile in flight
  // stalls on them at its next L2 load.  So the scoring waves (0-7) issue their share of the
  // next tile only AFTER their scoring loop, the other waves (8-15) at the top of the iteration;
  // all 16 waves stage 1/16 of the tile and are owners in the gather phase.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;                         // [TT][LD]
  float* part = tile + TT * LD;              // [SW][TT][32] score partials
  float* rsqp = part + SW * TT * 32;         // [SW][TT] row sum-of-squares partials
  float* nrm = rsqp + SW * TT;               // [TT]
  int* lab = reinterpret_cast<int*>(nrm + TT);   // [TT]
  float* red = reinterpret_cast<float*>(lab + TT);   // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t unit = blockIdx.x;
  int64_t n0, n1;
  int part_id = 0;
  if (KMEANS) {
    n0 = unit * a.chunk_rows;
    n1 = min(n0 + a.chunk_rows, a.total);
  } else {
    // a.parts > 1 (few images): a.parts workgroups share an image, each walking a contiguous run of its token tiles;
    // the last one to finish adds the partial sums in part order and normalises (see the epilogue)
    if (a.parts > 1) {
      part_id = (int)(unit % a.parts);
      unit /= a.parts;
    }
    n0 = a.offsets[unit];
    n1 = a.offsets[unit + 1];
    if (a.parts > 1) {
      const int64_t per = (((n1 - n0 + TT - 1) / TT) + a.parts - 1) / a.parts * TT;   // rows per part, whole tiles
      n0 = min(n0 + part_id * per, n1);
      n1 = min(n0 + per, n1);
    }
  }
  const int64_t nrows = n1 - n0;
  const int ntiles = (int)((nrows + TT - 1) / TT);

  // ---- staging: thread owns float4 slots f = tid + 1024*i of the [16][D/4] tile ----
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + n0 * D), 0, (int)imin64(nrows * D * 4, 0x7fffffff), 0x00020000);
  f32x4 stg[NF];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // thread owns float4 slots f = tid + 1024*i of the [16][D/4] tile; slot -> (row, column) is
  // recomputed where needed (constant divisor) to save registers
  const bool scorer = wave < SW;
  auto fetch = [&](int t) {
    const unsigned so = (unsigned)t * (unsigned)(TT * D * 4);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = tid + NTH * i, row = f / (D / 4), c4 = f - row * (D / 4);
      stg[i] = (f < TT * (D / 4)) ? bload16(x_rsrc, (unsigned)((row * D + 4 * c4) * 4), so) : zero4;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = tid + NTH * i, row = f / (D / 4), c4 = f - row * (D / 4);
      if (f < TT * (D / 4)) *reinterpret_cast<f32x4*>(tile + row * LD + 4 * c4) = stg[i];
    }
  };

  // ---- owner coordinates: cluster k = tid / 32, columns 4*j + 128*m ----
  const int ok_ = tid >> 5, oj = tid & 31;
  f32x4 acc[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) acc[m] = zero4;
  unsigned my_count = 0;
  const bool k_live = ok_ < a.K;
  // raw centres of this owner's cluster (VLAD residual): descriptor over [K, D], constant lane offset
  const __amdgpu_buffer_rsrc_t cen_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(KMEANS ? a.chat : a.centers), 0, (KMEANS ? 32 : a.K) * D * 4, 0x00020000);
  const unsigned cen_off = (unsigned)((ok_ * D + 4 * oj) * 4);

  // ---- scoring coordinates (waves < SW): D/SW slice, 16x16x4 MFMA fragments ----
  const int sw = wave & (SW - 1);
  const int fr = lane & 15, fq = lane >> 4;            // fragment row / k-quad
  const __amdgpu_buffer_rsrc_t c_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.chat), 0, 32 * D * 4, 0x00020000);
  const unsigned b_off0 = (unsigned)(((fr)*D + sw * (D / SW) + 4 * fq) * 4);
  const unsigned b_off1 = (unsigned)(((16 + fr) * D + sw * (D / SW) + 4 * fq) * 4);
  const float* a_frag = tile + fr * LD + sw * (D / SW) + 4 * fq;

  const float my_bias = a.cbias[tid & 31];    // score bias of centre tid % 32 (assign phase): loaded once, not per tile
  if (ntiles > 0) {
    fetch(0);
    stash();
  }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    if (!scorer && t + 1 < ntiles) fetch(t + 1);
    const int valid = (int)imin64(TT, nrows - (int64_t)t * TT);

    if (wave < SW) {
      // ---- scores: S[16 tokens][32 centres] over this wave's D/SW slice; the same A fragments
      //      give the row sum-of-squares for free ----
      f32x4 s0 = zero4, s1 = zero4;
      float rs = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a_frag + 16 * g);
        const f32x4 b0 = bload16(c_rsrc, b_off0, (unsigned)(64 * g));
        const f32x4 b1 = bload16(c_rsrc, b_off1, (unsigned)(64 * g));
        if (!KMEANS) rs += av[0] * av[0] + av[1] * av[1] + av[2] * av[2] + av[3] * av[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b0[e], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b1[e], s1, 0, 0, 0);
        }
      }
      // C/D layout of 16x16: col = lane&15, row = 4*(lane>>4) + reg
      float* p = part + sw * (TT * 32) + (4 * fq) * 32 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r * 32] = s0[r];
        p[r * 32 + 16] = s1[r];
      }
      if (!KMEANS) {
        rs += __shfl_xor(rs, 16, 64);      // the four k-quads of token row fr
        rs += __shfl_xor(rs, 32, 64);
        if (fq == 0) rsqp[sw * TT + fr] = rs;
      }
      if (t + 1 < ntiles) fetch(t + 1);    // only now: the B-operand loads above are retired
    }
    lds_barrier();

    if (tid < 512) {
      // ---- assign: fixed-order sum of the SW partials, first arg-max over k < K ----
      const int row = tid >> 5, k = tid & 31;
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < SW; ++w2) s += part[w2 * (TT * 32) + row * 32 + k];
      if (!KMEANS && k == 0) {
        float q = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < SW; ++w2) q += rsqp[w2 * TT + row];
        nrm[row] = a.norm_descs ? fmaxf(sqrtf(q), 1e-12f) : 1.0f;
      }
      s += my_bias;
      float best = (k < a.K) ? s : -INFINITY;
      int bi = (k < a.K && best == best) ? k : 0x7fffffff;
      if (!(best == best)) best = -INFINITY;          // NaN score never wins
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (k == 0) {
        if (bi == 0x7fffffff) bi = 0;
        const bool live = row < valid;
        lab[row] = live ? bi : -1;
        if (live && a.lab64) a.lab64[n0 + (int64_t)t * TT + row] = bi;
      }
    }
    lds_barrier();

    // ---- gather: owners add their tokens (in order) ----
    if (k_live) {
      const float* tp = tile + 4 * oj;
      unsigned mine = 0;                                // bit n: token n of the tile is assigned to my cluster
#pragma unroll
      for (int n = 0; n < TT; n += 4) {
        const i32x4_t q = *reinterpret_cast<const i32x4_t*>(lab + n);   // int vector: may alias lab[]
#pragma unroll
        for (int e = 0; e < 4; ++e) mine |= (q[e] == ok_ ? 1u : 0u) << (n + e);
      }
      for (int n = 0; n < TT; ++n) {
        if (!((mine >> n) & 1u)) continue;
        const float* rp = tp + n * LD;
        if (KMEANS) {
#pragma unroll
          for (int m = 0; m < NV; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 128 * m);
            acc[m][0] += v[0]; acc[m][1] += v[1]; acc[m][2] += v[2]; acc[m][3] += v[3];
          }
          if (oj == 0) ++my_count;
        } else {
          const float inv = 1.0f / nrm[n];          // x * (1/||x||): within 1 ulp of F.normalize's x / ||x||
#pragma unroll
          for (int m = 0; m < NV; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 128 * m);
            const f32x4 c = bload16(cen_rsrc, cen_off, (unsigned)(512 * m));
            acc[m][0] += v[0] * inv - c[0];
            acc[m][1] += v[1] * inv - c[1];
            acc[m][2] += v[2] * inv - c[2];
            acc[m][3] += v[3] * inv - c[3];
            // keep at most four column groups in flight: the 128-VGPR budget of a 1024-thread
            // workgroup cannot hold all NV loads at once next to the accumulators
            if ((m & 1) == 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    lds_barrier();
    if (t + 1 < ntiles) stash();
    lds_barrier();
  }

  // ---- epilogue ----
  if (KMEANS) {
    if (k_live) {
      float* o = a.out + (unit * a.K + ok_) * (int64_t)D + 4 * oj;
#pragma unroll
      for (int m = 0; m < NV; ++m) *reinterpret_cast<f32x4*>(o + 128 * m) = acc[m];
      if (oj == 0) a.cnt_part[unit * a.K + ok_] = my_count;
    }
    return;
  }
  if (a.parts > 1) {
    // Partial sums of this part -> workspace; the workgroup that takes the last ticket of its image (a relaxed atomic
    // between an agent-scope release and acquire: the partials of every earlier part are visible to it, whichever XCD
    // wrote them) reduces them in PART ORDER, so the result does not depend on which workgroup arrives last.  Nobody
    // waits on anybody: no spinning, no co-residency requirement.
    const int64_t kd = (int64_t)a.K * D;
    if (k_live) {
      float* pb = a.part_buf + (unit * a.parts + part_id) * kd + (int64_t)ok_ * D + 4 * oj;
#pragma unroll
      for (int m = 0; m < NV; ++m) *reinterpret_cast<f32x4*>(pb + 128 * m) = acc[m];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores have left
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned ticket = __hip_atomic_fetch_add(a.part_tickets + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(a.parts - 1);
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      lab[0] = last;
    }
    __syncthreads();
    if (!lab[0]) return;
    if (k_live) {
      const float* pb = a.part_buf + unit * a.parts * kd + (int64_t)ok_ * D + 4 * oj;
#pragma unroll
      for (int m = 0; m < NV; ++m) acc[m] = zero4;
      for (int q = 0; q < a.parts; ++q) {
#pragma unroll
        for (int m = 0; m < NV; ++m) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(pb + q * kd + 128 * m);
          acc[m][0] += v[0]; acc[m][1] += v[1]; acc[m][2] += v[2]; acc[m][3] += v[3];
        }
      }
    }
  }
  // VLAD: intra-norm of each cluster block (its 32 owners = half a wave), then the global norm
  float ss = 0.f;
#pragma unroll
  for (int m = 0; m < NV; ++m) ss += acc[m][0] * acc[m][0] + acc[m][1] * acc[m][1] + acc[m][2] * acc[m][2] + acc[m][3] * acc[m][3];
  ss = half_wave_sum(ss);
  if (a.intra) {
    const float kn = fmaxf(sqrtf(ss), 1e-12f);
    ss = 0.f;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      acc[m][0] /= kn; acc[m][1] /= kn; acc[m][2] /= kn; acc[m][3] /= kn;
      ss += acc[m][0] * acc[m][0] + acc[m][1] * acc[m][1] + acc[m][2] * acc[m][2] + acc[m][3] * acc[m][3];
    }
    ss = half_wave_sum(ss);
  }
  if (oj == 0) red[ok_] = k_live ? ss : 0.f;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) tot += red[k];
  const float gn = fmaxf(sqrtf(tot), 1e-12f);
  if (k_live) {
    float* o = a.out + (unit * a.K + ok_) * (int64_t)D + 4 * oj;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      f32x4 v = acc[m];
      v[0] /= gn; v[1] /= gn; v[2] /= gn; v[3] /= gn;
      *reinterpret_cast<f32x4*>(o + 128 * m) = v;
    }
  }
}


// ---- third structure: fp16 screening scores from register-resident centres + exact resolution of close calls ----
// What bounds the kernel above is the centre stream: 196 KB of normalised centres from L2 per 16-token tile and CU
// (scoring 12 000 of 22 000 cycles per tile).  Here every wave keeps its D/8 slice of all 32 centres in REGISTERS as one
// fp16 plane (48 VGPRs at D = 1536) and scores the tile on v_mfma_f32_16x16x32_f16 with the tokens split into two fp16
// terms on the fly (x = hi + lo to ~2^-19): a screening score with a PROVEN error bound
//     |s~_k - s_k| <= e = max_k||chat_k|| (||x|| 5.4e-4 + 2e-6)
// (2^-11 from rounding the centre, ~2^-19 from the token, fp32 accumulation, fp16 subnormal quanta).  Every centre with
// s~_k >= max s~ - 2e is a candidate -- the true arg-max is always among them.  One candidate (the usual case: top-2
// cosine gaps of real tokens are ~10^-2, 2e is ~10^-3): done.  Several: their scores are recomputed exactly in fp32 by one
// wave per token (x from the LDS tile, the candidate's fp32 centre from L2) and the first maximum wins, as in the exact
// kernels.  Rows with a non-finite score or a norm beyond the fp16 range take the exact path with all centres.
// Labels therefore equal the fp32 arg-max except at fp32-rounding ties, like the kernels above.
// 512 threads; thread (cluster tid/16, columns 4 (tid%16) + 64 m) owns 2 NV float4 accumulators (both modes).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

__device__ __forceinline__ h16x8 pack_h16x8(const f32x4 a, const f32x4 b) {
  u32x4s r;
  r[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[0], a[1]));
  r[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[2], a[3]));
  r[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b[0], b[1]));
  r[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b[2], b[3]));
  return __builtin_bit_cast(h16x8, r);
}

// compile-time loop: the accumulator array below must only ever see constant indices (or it is demoted to scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// SW = waves per workgroup: 8 (two per SIMD, 256 registers each) or 4 (one per SIMD: the wave may use the whole 512-entry
// register file of its SIMD lane, 256 VGPRs + 256 AGPRs).  Per wave at D = 1536: 96 / 192 accumulators + 48 / 96 registers
// of fp16 centres + 48 / 96 registers of the next tile in flight.  Nothing may spill: a scratch reload waits -- the
// vector-memory counter retires in order -- for the HBM loads of the next tile.
constexpr int f3_cw(int slice) {
  int cw = (slice + 63) / 64;
  while (slice % cw) ++cw;
  return cw;
}
constexpr int f3_gcd(int x, int y) { return y == 0 ? x : f3_gcd(y, x % y); }

// CW consecutive fp32 columns of a lane as ONE memory instruction (CW = 3: buffer_load_dwordx3 -- a lane's columns start at a
// multiple of 12 bytes, dword alignment is all a buffer access needs); other widths fall back to dwords
typedef unsigned f3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned f3_u32x3 __attribute__((ext_vector_type(3)));
template <int CW>
__device__ __forceinline__ void f3_load_cols(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float (&c)[CW]) {
  if constexpr (CW == 2) {
    const f3_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
    c[0] = __uint_as_float(r[0]); c[1] = __uint_as_float(r[1]);
  } else if constexpr (CW == 3) {
    const f3_u32x3 r = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff, 0);
    c[0] = __uint_as_float(r[0]); c[1] = __uint_as_float(r[1]); c[2] = __uint_as_float(r[2]);
  } else if constexpr (CW == 4) {
    const u32x4s r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    c[0] = __uint_as_float(r[0]); c[1] = __uint_as_float(r[1]); c[2] = __uint_as_float(r[2]); c[3] = __uint_as_float(r[3]);
  } else {
#pragma unroll
    for (int j = 0; j < CW; ++j) c[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 4 * j, soff, 0));
  }
}
template <int CW>
__device__ __forceinline__ void f3_store_cols(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, const float (&c)[CW]) {
  if constexpr (CW == 2) {
    __builtin_amdgcn_raw_buffer_store_b64(f3_u32x2{__float_as_uint(c[0]), __float_as_uint(c[1])}, rsrc, voff, soff, 0);
  } else if constexpr (CW == 3) {
    __builtin_amdgcn_raw_buffer_store_b96(f3_u32x3{__float_as_uint(c[0]), __float_as_uint(c[1]), __float_as_uint(c[2])}, rsrc, voff, soff, 0);
  } else if constexpr (CW == 4) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4s{__float_as_uint(c[0]), __float_as_uint(c[1]), __float_as_uint(c[2]), __float_as_uint(c[3])}, rsrc, voff, soff, 0);
  } else {
#pragma unroll
    for (int j = 0; j < CW; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c[j]), rsrc, voff + 4 * j, soff, 0);
  }
}

// GV (hazard study, round 6; option vlad_gather_v, D = 1536 VLAD mode only): 0 = the shipped gather arithmetic (mul, sub, select);
// 1 = the round-5 form that was NOT reproducible run to run (one fma + select); 2 / 3 / 4 = 1 with `s_nop 3` behind
// s_set_gpr_idx_on / in front of s_set_gpr_idx_off / behind it; 5 = all three; 7 = 1 with the compiler's own lowering of the
// dynamic subscript; 8 = 1 with LLVM's usage pattern in our asm (indexed v_mov out, plain adds, indexed v_mov back); 9 = 1 with the
// accumulator read through src1 (mode 0xA); 10 / 11 = 2 with `s_nop 0` / `s_nop 1` (how many wait states the switch needs)
template <int NV, int SW, bool KMEANS, int GV = 0>
__global__ __launch_bounds__(64 * SW) void fused3_kernel(FusedArgs a) {
  constexpr int D = NV * 128;
  constexpr int LD = D + 4;
  constexpr int NT3 = 64 * SW;
  constexpr int NF = 16 * (D / 4) / NT3;      // staged float4 per thread per tile
  constexpr int SLICE = D / SW;               // columns of a wave: its scoring slice AND the columns it accumulates
  constexpr int NKB = SLICE / 32;             // 32-wide k-blocks per wave slice
  static_assert(SLICE % 32 == 0 && 16 * (D / 4) % NT3 == 0, "a wave's slice must be whole 32-wide k-blocks");
  constexpr int NX = (D + 255) / 256;         // float4 per lane of one row (exact resolution)
  // accumulators: wave w owns columns [w SLICE, (w+1) SLICE) of ALL 32 clusters, CW consecutive columns per lane ->
  // 32 CW registers per lane, held as CW vectors of 32 (one element per cluster).  Every wave visits every token of the
  // tile; the token's label is wave-uniform and indexes the vectors DYNAMICALLY IN REGISTERS (s_set_gpr_idx / v_movrel:
  // a 32-way branch instead made the allocator keep two copies of every accumulator).  The gather is balanced whatever
  // the label mix -- with one owner wave per cluster the tile waited for the wave whose clusters got the most tokens
  // (measured: gather 4 400 cycles + 3 700 of imbalance per tile).
  constexpr int CW = f3_cw(SLICE);
  constexpr int GL = SLICE / CW;              // lanes of a wave that own columns (64 or 48)
  static_assert(GL * CW == SLICE && GL <= 64, "column ownership must tile the slice");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;                          // [TT][LD]
  float* part = tile + TT * LD;               // [SW][TT][32]
  float* rsqp = part + SW * TT * 32;          // [SW][TT]
  float* nrm = rsqp + SW * TT;                // [TT]
  int* lab = reinterpret_cast<int*>(nrm + TT);            // [TT]
  unsigned* amb = reinterpret_cast<unsigned*>(lab + TT);  // [TT] candidate masks of the rows to resolve exactly (0: none)
  float* red = reinterpret_cast<float*>(amb + TT);        // [SW][32] epilogue reductions
  int* npairs = reinterpret_cast<int*>(red + SW * 32);    // [1] (row, centre) pairs queued for exact scoring in this tile
  int* pairs = npairs + 4;                                // [16 * 32] queue of (row << 5 | centre)
  float* exs = part;                          // after barrier B the score partials are dead: exact-score table [16][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t unit = blockIdx.x;
  int64_t n0, n1;
  int part_id = 0;
  if (KMEANS) {
    n0 = unit * a.chunk_rows;
    n1 = min(n0 + a.chunk_rows, a.total);
  } else {
    if (a.parts > 1) {
      part_id = (int)(unit % a.parts);
      unit /= a.parts;
    }
    n0 = a.offsets[unit];
    n1 = a.offsets[unit + 1];
    if (a.parts > 1) {
      const int64_t per = (((n1 - n0 + TT - 1) / TT) + a.parts - 1) / a.parts * TT;
      n0 = min(n0 + part_id * per, n1);
      n1 = min(n0 + per, n1);
    }
  }
  const int64_t nrows = n1 - n0;
  const int ntiles = (int)((nrows + TT - 1) / TT);

  // The descriptor of the unit's rows is rebuilt from two scalars at every use: built once here, the compiler kept half
  // of it in VGPRs across the tile loop and wrapped every load of the next tile in a waterfall loop.
  const unsigned long long x_base = reinterpret_cast<unsigned long long>(uniform_ptr(const_cast<float*>(a.x + n0 * D)));
  const int x_bytes = __builtin_amdgcn_readfirstlane((int)imin64(nrows * D * 4, 0x7fffffff));
#define x_rsrc __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(x_base), 0, x_bytes, 0x00020000)
  f32x4 stg[NF];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // slot f = tid + NT3 i of the [16][D/4] tile is (row, float4 column) = (f / (D/4), f % (D/4)); NT3 i / (D/4) = 2 SW i / NV
  // repeats with period P = NV / gcd(2 SW, NV), so only P (row, column) pairs are computed and the others are RSTEP rows
  // further down: constant offsets instead of NF precomputed addresses per thread
  constexpr int P = NV / f3_gcd(2 * SW, NV);
  constexpr int RSTEP = 2 * SW * P / NV;
  static_assert(NF % P == 0, "staging pattern");
  auto fetch = [&](int t) {
    const unsigned so = (unsigned)t * (unsigned)(TT * D * 4);
#pragma unroll
    for (int r = 0; r < P; ++r) {
      const int f = tid + NT3 * r, row = f / (D / 4), c4 = f - row * (D / 4);
      const unsigned vo = (unsigned)((row * D + 4 * c4) * 4);
#pragma unroll
      for (int j = 0; j < NF / P; ++j) stg[j * P + r] = bload16(x_rsrc, vo, so + (unsigned)(j * RSTEP * D * 4));
    }
  };
  // two staged float4 per k-block of the scoring loop (NF = 2 NKB): the loads of the next tile are spread through the
  // scoring phase, so the wave's VALU / MFMA work runs while the texture path accepts them (98 KB per tile and CU at
  // 64 B/clk is ~1 500 cycles; issued in one burst, every wave sat in that queue before it scored)
  static_assert(NF == 2 * NKB, "two staged float4 per k-block");
  // (VLAD mode: the staging addresses are rebuilt from an opaque copy of the thread id at every use -- hoisted out of the tile
  // loop they are six registers these variants do not have, and a spilled loop invariant is reloaded behind the HBM loads in
  // flight)
  auto opaque_tid = [&]() {
    int tt = tid;
    if constexpr (!KMEANS) asm volatile("" : "+v"(tt));
    return tt;
  };
  // (the tile is contiguous in memory: float4 slot f = tid + NT3 r sits at byte 16 f -- (row D + 4 c4) 4 = 16 f -- so the load
  // address is one shift of the thread id plus constants: no division, nothing to hoist or to recompute)
  auto fetch_pair = [&](int t, auto kbc) {
    const unsigned so = (unsigned)t * (unsigned)(TT * D * 4);
    static_for<2>([&](auto h) {
      constexpr int i = 2 * decltype(kbc)::value + decltype(h)::value, r = i % P, j = i / P;
      stg[j * P + r] = bload16(x_rsrc, (unsigned)(tid * 16), so + (unsigned)(NT3 * r * 16 + j * RSTEP * D * 4));
    });
  };
  auto stash = [&]() {
    const int ft = opaque_tid();
#pragma unroll
    for (int r = 0; r < P; ++r) {
      // row = (ft + NT3 r) / (D / 4) by compares: ft < NT3, so the quotient is the constant (NT3 r) / (D / 4) plus the number
      // of row boundaries below ft -- one or two compile-time thresholds instead of a division sequence per address
      constexpr int Q = D / 4;
      const int f = ft + NT3 * r;
      int row = (NT3 * r) / Q;
#pragma unroll
      for (int m = 1; ((NT3 * r) / Q + m) * Q - NT3 * r < NT3; ++m) row += ft >= ((NT3 * r) / Q + m) * Q - NT3 * r ? 1 : 0;
      float* dst = tile + 4 * f + 4 * row;        // = tile + row LD + 4 (f - row Q): the padded row adds 4 floats per row
#pragma unroll
      for (int j = 0; j < NF / P; ++j) *reinterpret_cast<f32x4*>(dst + j * RSTEP * LD) = stg[j * P + r];
    }
  };

  f32x32 acc[CW];
  static_for<CW>([&](auto j) {
    static_for<32>([&](auto k) { acc[j][(int)k] = 0.f; });
  });
  unsigned my_count = 0;                       // wave 0, lane k: tokens labelled k
  const int gcol = wave * SLICE + CW * (lane < GL ? lane : 0);
  const __amdgpu_buffer_rsrc_t cen_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<float*>(KMEANS ? a.chat : a.centers)), 0, (KMEANS ? 32 : a.K) * D * 4, 0x00020000);

  // scoring coordinates: 16x16x32 fragments -- token / centre = lane & 15, 8 consecutive k at 8 (lane >> 4) of the k-block
  const int fr = lane & 15, fq = lane >> 4;
  const __amdgpu_buffer_rsrc_t c_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float*>(a.chat)), 0, 32 * D * 4, 0x00020000);
  const float* a_frag = tile + fr * LD + wave * SLICE + 8 * fq;
  h16x8 bh[2][NKB];                            // this wave's slice of the 32 centres, fp16, for the whole kernel
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const unsigned off = (unsigned)(((16 * h + fr) * D + wave * SLICE + 32 * kb + 8 * fq) * 4);
      // round to nearest (the error bound counts on half an fp16 ulp per centre element; the tokens' own split below may
      // truncate, its second term picks up what the first one dropped)
      const f32x4 c0 = bload16(c_rsrc, off, 0), c1 = bload16(c_rsrc, off, 16);
      f32x8 cc;
#pragma unroll
      for (int e = 0; e < 4; ++e) { cc[e] = c0[e]; cc[4 + e] = c1[e]; }
      bh[h][kb] = __builtin_convertvector(cc, h16x8);
    }
  const float my_bias = a.cbias[tid & 31];
  // ||chat_k||: 1 (cosine) or 2 ||c_k|| = 2 sqrt(-bias_k) (euclidean); its maximum scales the error bound
  float cnmax = a.metric ? 2.0f * sqrtf(fmaxf(-my_bias, 0.0f)) : 1.0f;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) cnmax = fmaxf(cnmax, __shfl_xor(cnmax, o, 64));

  // (register-indexed access to the accumulators of a RUNTIME cluster id, wave-uniform in an SGPR: the epilogue and the fold)
  auto acc_get = [&](int k, float (&v)[CW]) {
    if constexpr (CW == 1) {
      asm volatile("s_set_gpr_idx_on %1, 0x1\n\ts_nop 3\n\tv_mov_b32 %0, v224\n\ts_set_gpr_idx_off" : "=&v"(v[0]) : "s"(k), "{v[224:255]}"(acc[0]));
    } else if constexpr (CW == 2) {
      asm volatile("s_set_gpr_idx_on %2, 0x1\n\ts_nop 3\n\tv_mov_b32 %0, v192\n\tv_mov_b32 %1, v224\n\ts_set_gpr_idx_off"
                   : "=&v"(v[0]), "=&v"(v[1]) : "s"(k), "{v[192:223]}"(acc[0]), "{v[224:255]}"(acc[1]));
    } else if constexpr (CW == 3) {
      asm volatile("s_set_gpr_idx_on %3, 0x1\n\ts_nop 3\n\tv_mov_b32 %0, v160\n\tv_mov_b32 %1, v192\n\tv_mov_b32 %2, v224\n\ts_set_gpr_idx_off"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                   : "s"(k), "{v[160:191]}"(acc[0]), "{v[192:223]}"(acc[1]), "{v[224:255]}"(acc[2]));
    } else if constexpr (CW == 4) {
      asm volatile("s_set_gpr_idx_on %4, 0x1\n\ts_nop 3\n\tv_mov_b32 %0, v128\n\tv_mov_b32 %1, v160\n\tv_mov_b32 %2, v192\n\tv_mov_b32 %3, v224\n\t"
                   "s_set_gpr_idx_off"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                   : "s"(k), "{v[128:159]}"(acc[0]), "{v[160:191]}"(acc[1]), "{v[192:223]}"(acc[2]), "{v[224:255]}"(acc[3]));
    } else {
      static_for<CW>([&](auto j) { v[(int)j] = acc[j][k]; });
    }
  };
  auto acc_set = [&](int k, const float (&v)[CW]) {
    if constexpr (CW == 1) {
      asm volatile("s_set_gpr_idx_on %2, 0x8\n\ts_nop 3\n\tv_mov_b32 v224, %1\n\ts_set_gpr_idx_off" : "+{v[224:255]}"(acc[0]) : "v"(v[0]), "s"(k));
    } else if constexpr (CW == 2) {
      asm volatile("s_set_gpr_idx_on %4, 0x8\n\ts_nop 3\n\tv_mov_b32 v192, %2\n\tv_mov_b32 v224, %3\n\ts_set_gpr_idx_off"
                   : "+{v[192:223]}"(acc[0]), "+{v[224:255]}"(acc[1]) : "v"(v[0]), "v"(v[1]), "s"(k));
    } else if constexpr (CW == 3) {
      asm volatile("s_set_gpr_idx_on %6, 0x8\n\ts_nop 3\n\tv_mov_b32 v160, %3\n\tv_mov_b32 v192, %4\n\tv_mov_b32 v224, %5\n\ts_set_gpr_idx_off"
                   : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                   : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(k));
    } else if constexpr (CW == 4) {
      asm volatile("s_set_gpr_idx_on %8, 0x8\n\ts_nop 3\n\tv_mov_b32 v128, %4\n\tv_mov_b32 v160, %5\n\tv_mov_b32 v192, %6\n\tv_mov_b32 v224, %7\n\t"
                   "s_set_gpr_idx_off"
                   : "+{v[128:159]}"(acc[0]), "+{v[160:191]}"(acc[1]), "+{v[192:223]}"(acc[2]), "+{v[224:255]}"(acc[3])
                   : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(k));
    } else {
      static_for<CW>([&](auto j) { acc[j][k] = v[(int)j]; });
    }
  };

  if (ntiles > 0) {
    fetch(0);
    stash();
  }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int valid = (int)imin64(TT, nrows - (int64_t)t * TT);
    if (tid == 0) *npairs = 0;                 // (read after barrier B; barrier A orders this store before the atomics)
    // The HBM loads of the next tile go out during scoring: scoring and assign wait for no vector memory, so by the time
    // the exact resolution or the VLAD gather wait for their own (L2) loads -- the counter retires in order -- these
    // have landed
    {
      // ---- screening scores of this wave's slice (and the row sums of squares from the same fragments) ----
      f32x4 s0 = zero4, s1 = zero4;
      float rs = 0.f;
      static_for<NKB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        fetch_pair(t + 1, kbc);          // (past the last tile: out of the descriptor's range, returns zeros, never stashed)
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(a_frag + 32 * kb);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(a_frag + 32 * kb + 4);
        rs += (x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3]);
        rs += (x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]);
        const h16x8 hi = pack_h16x8(x0, x1);
        f32x4 r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r0[e] = x0[e] - (float)hi[e];
          r1[e] = x1[e] - (float)hi[4 + e];
        }
        const h16x8 lo = pack_h16x8(r0, r1);
        // two accumulator chains per wave (x 2 waves per SIMD); more would cost registers this kernel does not have
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, bh[0][kb], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, bh[1][kb], s1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, bh[0][kb], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, bh[1][kb], s1, 0, 0, 0);
        if (kb % 2 == 1) __builtin_amdgcn_sched_barrier(0);          // two k-blocks' fragments in flight at most
      });
      // C/D layout: centre = lane & 15, token = 4 (lane >> 4) + reg
      float* p = part + wave * (TT * 32) + (4 * fq) * 32 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r * 32] = s0[r];
        p[r * 32 + 16] = s1[r];
      }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
      if (fq == 0) rsqp[wave * TT + fr] = rs;
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < 8 / SW; ++it) {
      // ---- assign: fixed-order sum of the SW partials; arg-max; candidates within the error bound ----
      // (addresses are rebuilt from an opaque copy of the thread id every tile: hoisted out of the tile loop they are
      // registers this kernel does not have, and a spilled one is reloaded behind the HBM loads in flight)
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int row = (tl >> 5) + 2 * SW * it, k = tl & 31;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < SW; ++w2) {
        s += part[w2 * (TT * 32) + row * 32 + k];
        q += rsqp[w2 * TT + row];
      }
      s += my_bias;
      const float xn = sqrtf(q);
      const bool kin = k < a.K;
      float best = kin ? s : -INFINITY;
      int bi = (kin && best == best) ? k : 0x7fffffff;
      if (!(best == best)) best = -INFINITY;
      // first arg-max over the 32 lanes of the row: four DPP rotations inside each 16-lane row (every lane ends up with
      // its row's winner), then one cross-row exchange
      auto take = [&](float ov, int oi) {
        const bool tk = (ov > best) | ((ov == best) & (oi < bi));
        best = tk ? ov : best;
        bi = tk ? oi : bi;
      };
      take(dpp_f32<0x128>(best), dpp_i32<0x128>(bi));     // row_ror:8
      take(dpp_f32<0x124>(best), dpp_i32<0x124>(bi));     // row_ror:4
      take(dpp_f32<0x122>(best), dpp_i32<0x122>(bi));     // row_ror:2
      take(dpp_f32<0x121>(best), dpp_i32<0x121>(bi));     // row_ror:1
      take(__shfl_xor(best, 16, 64), __shfl_xor(bi, 16, 64));
      const float tau = 2.0f * cnmax * (xn * 5.4e-4f + 2e-6f);
      const bool odd = kin && !(fabsf(s) < INFINITY);               // NaN / inf score
      const unsigned long long oddm = __ballot(odd || !(xn < 6.0e4f));
      const unsigned sh = 32u * (unsigned)(lane >> 5);
      const bool wild = ((unsigned)(oddm >> sh)) != 0u;              // this row cannot be screened in fp16
      const unsigned long long cm = __ballot(kin && (wild || s >= best - tau));
      const unsigned mask = (unsigned)(cm >> sh);
      const bool live = row < valid;
      // (a row that is BITWISE all zero has EXACT screening scores -- its matrix-core part is 0, the bias is fp32 -- so its
      // 32-way tie needs no resolution: first index, as the exact kernels give; without this rule zero rows cost 13x a
      // normal row.  A zero sum of squares alone does not say that: the squares of |x| < ~1e-19 underflow, and such a row
      // still orders its tiny scores -- so on xn == 0 (rare) the row's 32 lanes OR its bits from the LDS tile)
      bool zero_row = false;
      if (xn == 0.f) {
        int tl2 = tl;
        asm volatile("" : "+v"(tl2));
        const float* xr = tile + ((tl2 >> 5) + 2 * SW * it) * LD + (tl2 & 31);
        unsigned ob = 0;
#pragma unroll 4
        for (int i = 0; i < D / 32; ++i) ob |= __float_as_uint(xr[32 * i]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) ob |= (unsigned)__shfl_xor((int)ob, o, 64);
        zero_row = (ob & 0x7fffffffu) == 0u;
      }
      const bool close = live && !zero_row && __builtin_popcount(mask) > 1;
      if (close && ((mask >> k) & 1u)) pairs[atomicAdd(npairs, 1)] = (row << 5) | k;   // LDS atomic; order is irrelevant
      if (k == 0) {
        if (bi == 0x7fffffff) bi = 0;
        amb[row] = close ? mask : 0u;
        lab[row] = live ? bi : -1;
        // (the INVERSE: one IEEE division per row here instead of one per row and LANE in the gather -- the same value)
        if (!KMEANS) nrm[row] = a.norm_descs ? 1.0f / fmaxf(xn, 1e-12f) : 1.0f;
        if (live && !close && a.lab64) a.lab64[n0 + (int64_t)t * TT + row] = bi;
      }
    }
    lds_barrier();
    // ---- exact resolution: the queued (row, centre) pairs are dealt round-robin to the waves (a close row has 2-3
    //      candidates; dealing whole rows left most waves idle behind the one that had a row), each scored exactly in
    //      fp32 by a whole wave; then the rows' arg-max over their candidates' exact scores ----
    const int np = __builtin_amdgcn_readfirstlane(*npairs);
    if (np > 0) {
      int ll = lane;
      asm volatile("" : "+v"(ll));
      for (int pi = wave; pi < np; pi += SW) {
        const int pr = __builtin_amdgcn_readfirstlane(pairs[pi]);
        const int rr = pr >> 5, k = pr & 31;
        const float* xrow = tile + rr * LD + 4 * ll;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          if (4 * ll + 256 * i < D) {
            const f32x4 c = bload16(c_rsrc, (unsigned)(16 * ll), (unsigned)((k * D + 256 * i) * 4));
            const f32x4 x = *reinterpret_cast<const f32x4*>(xrow + 256 * i);
            d = fmaf(x[0], c[0], d); d = fmaf(x[1], c[1], d);
            d = fmaf(x[2], c[2], d); d = fmaf(x[3], c[3], d);
          }
        }
        d = wave_sum(d) + a.cbias[k];
        if (lane == 0) exs[pr] = d;
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 8 / SW; ++it) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int row = (tl >> 5) + 2 * SW * it, k = tl & 31;
        const unsigned m = amb[row];
        const bool cand = (m >> k) & 1u;
        const float d = cand ? exs[row * 32 + k] : 0.f;
        float best = (cand && d == d) ? d : -INFINITY;
        int bi = (cand && d == d) ? k : 0x7fffffff;
        auto take = [&](float ov, int oi) {
          const bool tk = (ov > best) | ((ov == best) & (oi < bi));
          best = tk ? ov : best;
          bi = tk ? oi : bi;
        };
        take(dpp_f32<0x128>(best), dpp_i32<0x128>(bi));
        take(dpp_f32<0x124>(best), dpp_i32<0x124>(bi));
        take(dpp_f32<0x122>(best), dpp_i32<0x122>(bi));
        take(dpp_f32<0x121>(best), dpp_i32<0x121>(bi));
        take(__shfl_xor(best, 16, 64), __shfl_xor(bi, 16, 64));
        if (k == 0 && m != 0u) {
          if (bi == 0x7fffffff) bi = 0;
          lab[row] = bi;
          if (a.lab64) a.lab64[n0 + (int64_t)t * TT + row] = bi;
        }
      }
      lds_barrier();                           // (np is workgroup-uniform: with nothing queued, lab[] is final already)
    }
    {
      // ---- gather: every wave adds every token (in order) to its columns of the token's cluster ----
      const float* tp = tile + gcol;
      // acc[j][k] += v[j] with the cluster index in an SGPR: ONE region of GPR-index mode (source 0 and destination
      // indexed) around CW adds.  The compiler's own lowering of the dynamic subscript switches the mode on and off
      // around a v_mov for every read and every write (7 instructions per element); the vectors are pinned to the top
      // of the register file so the instructions can name their base registers.
      auto add_token = [&](int k, const float* v) {
        if (k < 0) return;
        if constexpr (CW == 1) {
          asm volatile("s_set_gpr_idx_on %2, 0x9\n\ts_nop 3\n\tv_add_f32 v224, v224, %1\n\ts_set_gpr_idx_off"
                       : "+{v[224:255]}"(acc[0]) : "v"(v[0]), "s"(k));
        } else if constexpr (CW == 2) {
          asm volatile("s_set_gpr_idx_on %4, 0x9\n\ts_nop 3\n\tv_add_f32 v192, v192, %2\n\tv_add_f32 v224, v224, %3\n\ts_set_gpr_idx_off"
                       : "+{v[192:223]}"(acc[0]), "+{v[224:255]}"(acc[1]) : "v"(v[0]), "v"(v[1]), "s"(k));
        } else if constexpr (CW == 3 && GV == 7) {
          static_for<CW>([&](auto j) { acc[j][(int)k] += v[j]; });
        } else if constexpr (CW == 3 && GV == 8) {
          float t0, t1, t2;
          asm volatile("s_set_gpr_idx_on %6, 0x1\n\tv_mov_b32 %0, v160\n\tv_mov_b32 %1, v192\n\tv_mov_b32 %2, v224\n\ts_set_gpr_idx_off"
                       : "=&v"(t0), "=&v"(t1), "=&v"(t2)
                       : "{v[160:191]}"(acc[0]), "{v[192:223]}"(acc[1]), "{v[224:255]}"(acc[2]), "s"(k));
          t0 += v[0]; t1 += v[1]; t2 += v[2];
          asm volatile("s_set_gpr_idx_on %6, 0x8\n\tv_mov_b32 v160, %3\n\tv_mov_b32 v192, %4\n\tv_mov_b32 v224, %5\n\ts_set_gpr_idx_off"
                       : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                       : "v"(t0), "v"(t1), "v"(t2), "s"(k));
        } else if constexpr (CW == 3 && GV == 9) {
          // the accumulator read through src1 (mode 0xA = SRC1_REL | DST_REL) instead of src0
          asm volatile("s_set_gpr_idx_on %6, 0xa\n\tv_add_f32 v160, %3, v160\n\tv_add_f32 v192, %4, v192\n\t"
                       "v_add_f32 v224, %5, v224\n\ts_set_gpr_idx_off"
                       : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                       : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(k));
        } else if constexpr (CW == 3 && ((GV >= 2 && GV <= 5) || GV == 10 || GV == 11)) {
#define ANYLOC_GV_ASM(PRE, MID, POST)                                                                                  \
  asm volatile("s_set_gpr_idx_on %6, 0x9\n\t" PRE "v_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"             \
               "v_add_f32 v224, v224, %5\n\t" MID "s_set_gpr_idx_off" POST                                             \
               : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])                              \
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(k))
          if constexpr (GV == 2) ANYLOC_GV_ASM("s_nop 3\n\t", "", "");
          else if constexpr (GV == 3) ANYLOC_GV_ASM("", "s_nop 3\n\t", "");
          else if constexpr (GV == 4) ANYLOC_GV_ASM("", "", "\n\ts_nop 3");
          else if constexpr (GV == 10) ANYLOC_GV_ASM("s_nop 0\n\t", "", "");
          else if constexpr (GV == 11) ANYLOC_GV_ASM("s_nop 1\n\t", "", "");
          else ANYLOC_GV_ASM("s_nop 3\n\t", "s_nop 3\n\t", "\n\ts_nop 3");
#undef ANYLOC_GV_ASM
        } else if constexpr (CW == 3 && GV == 1) {
          // (hazard study: NO wait state behind the mode switch -- with the one-fma residual this is the irreproducible form)
          asm volatile("s_set_gpr_idx_on %6, 0x9\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                       "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                       : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                       : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(k));
        } else if constexpr (CW == 3) {
          asm volatile("s_set_gpr_idx_on %6, 0x9\n\ts_nop 3\n\tv_add_f32 v160, v160, %3\n\tv_add_f32 v192, v192, %4\n\t"
                       "v_add_f32 v224, v224, %5\n\ts_set_gpr_idx_off"
                       : "+{v[160:191]}"(acc[0]), "+{v[192:223]}"(acc[1]), "+{v[224:255]}"(acc[2])
                       : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(k));
        } else if constexpr (CW == 4) {
          asm volatile("s_set_gpr_idx_on %8, 0x9\n\ts_nop 3\n\tv_add_f32 v128, v128, %4\n\tv_add_f32 v160, v160, %5\n\t"
                       "v_add_f32 v192, v192, %6\n\tv_add_f32 v224, v224, %7\n\ts_set_gpr_idx_off"
                       : "+{v[128:159]}"(acc[0]), "+{v[160:191]}"(acc[1]), "+{v[192:223]}"(acc[2]), "+{v[224:255]}"(acc[3])
                       : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(k));
        } else {
          static_for<CW>([&](auto j) { acc[j][(int)k] += v[j]; });
        }
      };
      if constexpr (KMEANS) {
        // four tokens per round: one LDS round trip for their labels and columns, then the register-indexed adds
#pragma unroll 1
        for (int n4 = 0; n4 < TT; n4 += 4) {
          const i32x4_t lq = *reinterpret_cast<const i32x4_t*>(lab + n4);
          float v[4][CW];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < CW; ++j) v[e][j] = tp[(n4 + e) * LD + j];
          int kk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) kk[e] = __builtin_amdgcn_readfirstlane(lq[e]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // rows past the unit carry label -1 and are all zero (the loads were out of the descriptor's range): adding
            // them to cluster 0 changes nothing and saves a branch per token
            add_token(kk[e] < 0 ? 0 : kk[e], v[e]);
            if (wave == 0) my_count += (lane == kk[e]) ? 1u : 0u;
          }
        }
      } else {
        // x / ||x|| - c_k.  The centres' CW columns of TG tokens are requested at once -- one CW-wide load per token, TT / TG
        // L2 round trips per tile -- and the tokens are then added in order while their columns come from the LDS tile.  The
        // order of additions per (cluster, column) is the token order.
        // Round 5: the tile columns come from LDS two tokens at a time, one pair ahead of the adds -- rounds 3-4 read each
        // token's three floats right before its adds, sixteen exposed LDS round trips per tile and wave.  Tokens-per-image
        // sweeps (tools/probe_vlad_fixed.py) put VLAD mode at 6.2 - 6.3 us per tile against 5.2 us for the k-means loop; two
        // structures WITHOUT this gather (8-bit / 7-bit centre tables) measured 6.7 and 7.4 us.
        // TG tokens per round (16 = the whole tile needs 16 CW registers more than this kernel has at D = 1536)
        constexpr int TG = 8;
#pragma unroll 1
        for (int g0 = 0; g0 < TT; g0 += TG) {
          int kk[TG];
#pragma unroll
          for (int n4 = 0; n4 < TG; n4 += 4) {
            const i32x4_t lq = *reinterpret_cast<const i32x4_t*>(lab + g0 + n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) kk[n4 + e] = __builtin_amdgcn_readfirstlane(lq[e]);
          }
          float c[TG][CW];
#pragma unroll
          for (int e = 0; e < TG; ++e)
            f3_load_cols<CW>(cen_rsrc, (unsigned)(gcol * 4), (unsigned)((kk[e] < 0 ? 0 : kk[e]) * D * 4), c[e]);
          // (software-pipelined by token pairs: the LDS reads of pair p + 1 are in flight while pair p is added -- one exposed
          // LDS round trip per round instead of one per token; four tokens at once spill four loop invariants)
          float v[2][2][CW];
          f32x2 nq[2];
          auto read_pair = [&](int buf, int p2) {
            nq[buf] = *reinterpret_cast<const f32x2*>(nrm + g0 + 2 * p2);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int j = 0; j < CW; ++j) v[buf][e][j] = tp[(g0 + 2 * p2 + e) * LD + j];
          };
          read_pair(0, 0);
#pragma unroll
          for (int p2 = 0; p2 < TG / 2; ++p2) {
            const int buf = p2 & 1;
            if (p2 + 1 < TG / 2) read_pair(buf ^ 1, p2 + 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              // A row past the unit (label -1) adds an exact zero to cluster 0 -- a select, not a branch.  Round 5 found two
              // forms of this gather (a fused multiply-add here; the branch form of add_token) NOT reproducible run to run, and
              // shipped whichever arithmetic passed the stress.  Round 6 found the mechanism (DESIGN.md 4.3,
              // profiles/r06_vlad_gather_hazard.log, tools/micro/gpr_idx_hazard.hip): in GPR-index mode the first indexed VALU
              // behind s_set_gpr_idx_on can execute before the mode switch has taken effect unless wait states separate them --
              // `s_nop 3` behind the switch makes the fma form bitwise reproducible too, idle issue slots IN FRONT of the switch
              // make every image wrong.  Every s_set_gpr_idx_on of this file is followed by `s_nop 3` now; the arithmetic stays
              // mul, sub, select (the bits of rounds 3-5).  tests/test_gpu_vlad_topk.py::test_vlad_reproducible_under_load and
              // tests/test_gpu_round6.py::test_gather_hazard_variants_of_the_one_pass_vlad_kernel keep watching.
              const int k = kk[2 * p2 + e];
#pragma unroll
              for (int j = 0; j < CW; ++j) {
                const float r = GV == 0 ? v[buf][e][j] * nq[buf][e] - c[2 * p2 + e][j]
                                        : __builtin_fmaf(v[buf][e][j], nq[buf][e], -c[2 * p2 + e][j]);
                v[buf][e][j] = k < 0 ? 0.0f : r;
              }
              add_token(k < 0 ? 0 : k, v[buf][e]);
            }
          }
        }
      }
    }
    lds_barrier();
    if (t + 1 < ntiles) stash();
    lds_barrier();
  }

  const bool g_live = lane < GL;
  // a lane's CW columns of cluster k travel as one CW-wide buffer access (idle lanes of a 48-lane slice point past the
  // descriptor: their stores are dropped, their loads return zeros)
  const int64_t kd = (int64_t)a.K * D;
  const unsigned col_off = g_live ? (unsigned)(gcol * 4) : 0x7fffff00u;
  if constexpr (KMEANS) {
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.out + unit * kd), 0, (int)(kd * 4), 0x00020000);
    static_for<32>([&](auto k) {
      if (k < a.K) {
        float v[CW];
        static_for<CW>([&](auto j) { v[(int)j] = acc[j][(int)k]; });
        f3_store_cols<CW>(o_rsrc, col_off, (unsigned)((int)k * D * 4), v);
      }
    });
    if (wave == 0 && lane < a.K) a.cnt_part[unit * a.K + lane] = my_count;
    return;
  } else {
    // ---- VLAD epilogue (round 5): RUNTIME loops over the clusters, the accumulators read and written through the GPR index
    // (cluster id in an SGPR), a few hundred instructions.  Rounds 2-4 unrolled every step over the 32 compile-time cluster
    // ids -- 10 300 of the kernel's 12 800 instructions, ~80 KB of straight-line code that every workgroup executes ONCE, cold
    // (the tile loop is 2 000 instructions; the k-means kernel is 1 950 in all).  Measured before this change: 34 tiles cost
    // 250 us and 9 tiles 122 us -- 5.1 us per tile, the k-means kernel's rate, plus ~75 us per workgroup that is not tiles.
    // The arithmetic per element (operations and their order) is what it was: the same bits.
    const int Kc = __builtin_amdgcn_readfirstlane(a.K);
    if (a.parts > 1) {
      // (the hand-off of vlad_fused_kernel: partial sums -> workspace, last ticket reduces in part order)
      {
        const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(a.part_buf + (unit * a.parts + part_id) * kd), 0, (int)(kd * 4), 0x00020000);
#pragma unroll 1
        for (int k = 0; k < Kc; ++k) {
          float v[CW];
          acc_get(k, v);
          f3_store_cols<CW>(p_rsrc, col_off, (unsigned)(k * D * 4), v);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(a.part_tickets + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (unsigned)(a.parts - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        lab[0] = last;
      }
      __syncthreads();
      if (!lab[0]) return;
      // the parts' sums of a cluster, added in part order; four clusters x up to eight parts = 32 CW-wide loads per lane in
      // flight per round (one workgroup pulls the parts x K x D x 4 bytes -- 784 KB at 61 images x 4 parts -- through its CU's
      // load path, so what counts is how many requests are outstanding)
      const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          uniform_ptr(a.part_buf + unit * a.parts * kd), 0, (int)imin64((int64_t)a.parts * kd * 4, 0x7fffffff), 0x00020000);
#pragma unroll 1
      for (int k0 = 0; k0 < Kc; k0 += 4) {
        float sum[4][CW];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
          for (int j = 0; j < CW; ++j) sum[c4][j] = 0.f;
        for (int q0 = 0; q0 < a.parts; q0 += 8) {              // (more than 8 parts only through option vlad_parts)
          float pv[4][8][CW];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (k0 + c4 < Kc && q0 + q < a.parts)
                f3_load_cols<CW>(q_rsrc, col_off, (unsigned)(((q0 + q) * kd + (k0 + c4) * D) * 4), pv[c4][q]);
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (k0 + c4 < Kc && q0 + q < a.parts)
#pragma unroll
                for (int j = 0; j < CW; ++j) sum[c4][j] += pv[c4][q][j];
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
          if (k0 + c4 < Kc) acc_set(k0 + c4, sum[c4]);
      }
    }
    // intra-norm of each cluster block (its columns are spread over the SW waves), then the global norm
    if (a.intra) {
#pragma unroll 1
      for (int k = 0; k < Kc; ++k) {
        float v[CW];
        acc_get(k, v);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) ss += v[j] * v[j];
        ss = wave_sum(g_live ? ss : 0.f);
        if (lane == 0) red[wave * 32 + k] = ss;
      }
      __syncthreads();
#pragma unroll 1
      for (int k = 0; k < Kc; ++k) {
        float tot = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < SW; ++w2) tot += red[w2 * 32 + k];
        const float kn = fmaxf(sqrtf(tot), 1e-12f);
        float v[CW];
        acc_get(k, v);
#pragma unroll
        for (int j = 0; j < CW; ++j) v[j] /= kn;
        acc_set(k, v);
      }
      __syncthreads();
    }
    float ss = 0.f;
#pragma unroll 1
    for (int k = 0; k < Kc; ++k) {
      float v[CW];
      acc_get(k, v);
#pragma unroll
      for (int j = 0; j < CW; ++j) ss += v[j] * v[j];
    }
    ss = wave_sum(g_live ? ss : 0.f);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < SW; ++w2) tot += red[w2];
    const float gn = fmaxf(sqrtf(tot), 1e-12f);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.out + unit * kd), 0, (int)(kd * 4), 0x00020000);
#pragma unroll 1
    for (int k = 0; k < Kc; ++k) {
      float v[CW];
      acc_get(k, v);
#pragma unroll
      for (int j = 0; j < CW; ++j) v[j] = v[j] / gn;
      f3_store_cols<CW>(o_rsrc, col_off, (unsigned)(k * D * 4), v);
    }
  }
}

#undef x_rsrc

template <int NV, int SW, bool KMEANS, int GV = 0>
int launch_fused3(const FusedArgs& a, int64_t units, hipStream_t stream) {
  constexpr int D = NV * 128;
  const size_t lds = sizeof(float) * (TT * (D + 4) + SW * TT * 32 + SW * TT + TT + TT + TT + SW * 32 + 4 + TT * 32);
  if constexpr (NV == 12 && SW == 8 && !KMEANS && GV == 0) {
    switch ((int)option(OPT_VLAD_GATHER_V)) {                 // hazard study (tools/stress_vlad.py): see fused3_kernel
      case 1: return launch_fused3<NV, SW, KMEANS, 1>(a, units, stream);
      case 2: return launch_fused3<NV, SW, KMEANS, 2>(a, units, stream);
      case 3: return launch_fused3<NV, SW, KMEANS, 3>(a, units, stream);
      case 4: return launch_fused3<NV, SW, KMEANS, 4>(a, units, stream);
      case 5: return launch_fused3<NV, SW, KMEANS, 5>(a, units, stream);
      case 7: return launch_fused3<NV, SW, KMEANS, 7>(a, units, stream);
      case 8: return launch_fused3<NV, SW, KMEANS, 8>(a, units, stream);
      case 9: return launch_fused3<NV, SW, KMEANS, 9>(a, units, stream);
      case 10: return launch_fused3<NV, SW, KMEANS, 10>(a, units, stream);
      case 11: return launch_fused3<NV, SW, KMEANS, 11>(a, units, stream);
      default: break;
    }
  }
  auto kern = fused3_kernel<NV, SW, KMEANS, GV>;
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(kern), (int)((int)lds)));
  ProfScope prof(KMEANS ? "kmeans_fused" : "vlad_fused", stream, 2.0 * a.total * D * 32,
                 4.0 * ((double)a.total * D + 2.0 * (double)units * a.K * D));
  unsigned grid = (unsigned)units;
  if (!KMEANS && a.parts > 1) {
    ANYLOC_CHECK_ARG(a.part_buf && a.part_tickets, "vlad_fused: parts without a partials buffer");
    ANYLOC_HIP(hipMemsetAsync(a.part_tickets, 0, sizeof(unsigned) * units, stream));
    grid = (unsigned)(units * a.parts);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * SW), lds, stream, a);
  return launch_status("fused3_kernel");
}

template <int NV, bool KMEANS>
int launch_fused(const FusedArgs& a, int64_t units, hipStream_t stream) {
  constexpr int D = NV * 128;
  constexpr int SW = 8;
  const size_t lds = sizeof(float) * (TT * (D + 4) + SW * TT * 32 + SW * TT + TT + TT + 32);
  auto kern = vlad_fused_kernel<NV, KMEANS>;
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(kern), (int)((int)lds)));
  const double bytes = 4.0 * ((double)a.total * D + 2.0 * (double)units * a.K * D);
  ProfScope prof(KMEANS ? "kmeans_fused" : "vlad_fused", stream, 2.0 * a.total * D * 32, bytes);
  unsigned grid = (unsigned)units;
  if (!KMEANS && a.parts > 1) {
    ANYLOC_CHECK_ARG(a.part_buf && a.part_tickets, "vlad_fused: parts without a partials buffer");
    ANYLOC_HIP(hipMemsetAsync(a.part_tickets, 0, sizeof(unsigned) * units, stream));
    grid = (unsigned)(units * a.parts);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds, stream, a);
  return launch_status("vlad_fused_kernel");
}

}  // namespace

bool fused_supported(int64_t D, int64_t K) {
  return K >= 1 && K <= 32 && (D == 384 || D == 768 || D == 1024 || D == 1536);
}

int vlad_fused(const FusedArgs& a, int64_t units, bool kmeans, hipStream_t stream) {
  if (units <= 0) return ANYLOC_OK;
  ANYLOC_CHECK_ARG(units < (1ll << 31), "vlad_fused: too many units");
  // options kmeans_fused_v / vlad_fused_v (A/B, tests): 0 (default) = fused3_kernel -- 8 waves where D / 128 is even, 4 waves
  // otherwise -- at every parts count (round 4: with the CW-wide hand-off 0.135 vs 0.154 ms at 61 images x 4 parts);
  // 1 = vlad_fused_kernel (both modes), 3 = fused3 with 4 waves, 4 = fused3 with 8
  const int ver = (int)option(kmeans ? OPT_KMEANS_FUSED_V : OPT_VLAD_FUSED_V);
  const bool f3 = ver == 0 ? true : ver >= 3;
#define ANYLOC_FUSED_CASE(NV)                                                                         \
  case NV * 128:                                                                                      \
    if (f3) {                                                                                         \
      if constexpr (NV % 2 == 0) {                                                                    \
        if (ver != 3)                                                                                 \
          return kmeans ? launch_fused3<NV, 8, true>(a, units, stream) : launch_fused3<NV, 8, false>(a, units, stream); \
      }                                                                                               \
      return kmeans ? launch_fused3<NV, 4, true>(a, units, stream) : launch_fused3<NV, 4, false>(a, units, stream);     \
    }                                                                                                 \
    return kmeans ? launch_fused<NV, true>(a, units, stream) : launch_fused<NV, false>(a, units, stream);
  switch (a.D) {
    ANYLOC_FUSED_CASE(3)
    ANYLOC_FUSED_CASE(6)
    ANYLOC_FUSED_CASE(8)
    ANYLOC_FUSED_CASE(12)
    default:
      set_error("vlad_fused: unsupported D=%d", a.D);
      return ANYLOC_ERR_UNSUPPORTED;
  }
#undef ANYLOC_FUSED_CASE
}

}  // namespace anyloc
