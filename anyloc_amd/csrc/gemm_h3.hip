// Row-scaled two-term fp16 split ("h3") GEMM: C = A W^T with fp32-level accuracy from THREE fp16
// matrix-core products per k-step (half the passes of gemm_x6.hip, which is limited by the chip's power budget).
//
// Every operand row is scaled by a power of two so that its largest magnitude lies in [2^14, 2^15):
//     x * 2^e = h + l,   h = fp16_rne(x * 2^e),   l = fp16_rne(x * 2^e - h)      (22 mantissa bits)
// l needs no extra scaling: the row scale keeps it a normal fp16 for every element within 2^16 of the row maximum,
// and below that its absolute error is 2^-39 of the row maximum.  a*b = (hh + hl + lh + ll) 2^-(ea+eb); ll is below
// 2^-24 and dropped, the other three accumulate in ONE fp32 accumulator (they have their natural magnitudes) and the
// epilogue multiplies by inv_a[row] * inv_w[col] (powers of two: exact).  CPU emulation of the whole ViT:
// tools/split_fp16_study.py.  In the ViT forward (ANYLOC_GEMM=h3) LayerNorm quantises its own output (the row is in
// registers); the attention output and the FFN hidden activation are written as fp32 and quantised by split_h2_kernel,
// because their rows are produced by different workgroups and the exact row maximum must be known first.
//
// Operand image ("h2"): [k/16][plane 0..1][row][16] fp16, 32 bytes per (k-block, plane, row), 16-byte halves swapped
// when (row >> 3) & 1 -- the x3 image of gemm_x6.hip with two planes; same DMA staging, same fragment reads.
#include <cstdlib>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));




// KB = k-blocks (of 16) per ring stage: one barrier and one counted wait per KB k-blocks instead of per k-block
template <int MI, int NI, int WM, int WN, int STAGES, int KB = 1>
struct H3Cfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  static constexpr int A_PLANE = BM * 32, W_PLANE = BN * 32;
  static constexpr int A_OP = 2 * A_PLANE, KSTAGE = A_OP + 2 * W_PLANE;   // one k-block: A planes, then W planes
  static constexpr int STAGE = KB * KSTAGE;
  static constexpr int LDS = STAGES * STAGE;
  static constexpr int A_DMA = BM / (32 * NW), W_DMA = BN / (32 * NW);
  static constexpr int NDMA = KB * 2 * (A_DMA + W_DMA);
  static_assert(BM % (32 * NW) == 0 && BN % (32 * NW) == 0, "each wave stages whole 32-row pieces");
  static_assert((STAGES - 2) * NDMA <= 63, "the counted vmcnt wait must be encodable");
};


// scale / inverse scale of a row (or tile) from its largest magnitude: amax * 2^e in [2^14, 2^15).  An all-zero (or
// denormal) row gets the LARGEST scale, 2^100: its planes are zero either way, and its 2^-100 never wins where the
// scales of several tiles are compared (attention_h3 takes the maximum over an image's V tiles as the output's bound)
__device__ __forceinline__ float h2_row_scale(float amax, float& inv) {
  const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
  const int e = ex == 0 ? 100 : max(-100, min(100, 14 - (ex - 127)));
  inv = __uint_as_float((unsigned)(127 - e) << 23);
  return __uint_as_float((unsigned)(127 + e) << 23);
}
// 2^e from a stored 2^-e (both normal powers of two: exact)
__device__ __forceinline__ float h2_scale_of_inv(float inv) { return __uint_as_float((254u << 23) - __float_as_uint(inv)); }
// two scaled values -> packed fp16 pair of the leading plane and of the residual plane
__device__ __forceinline__ void h2_pack2(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 pr;
  pr[0] = a; pr[1] = b;
  const f16x2 h = __builtin_convertvector(pr, f16x2);
  f32x2 res;
  res[0] = pr[0] - (float)h[0];
  res[1] = pr[1] - (float)h[1];
  const f16x2 l = __builtin_convertvector(res, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ float h3_gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float h3_silu(float v) { return v / (1.0f + expf(-v)); }
// the same on the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each -- ~5 instructions instead of ~25
// (expf's range reduction + an IEEE division).  |error| <= ~3 ulp of silu(v), i.e. below the 2^-22 quantisation the value
// gets on its way into the fc2 operand image.  v -> -inf: exp2 -> +inf, rcp -> 0, v * 0 = -0; v -> +inf: exp2 -> 0, v * 1.
__device__ __forceinline__ float h3_silu_fast(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

template <int MI, int NI, int WM, int WN, int STAGES, int OCC, int EPI, int KB = 1>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_h3_kernel(H3Problem p, int tiles_m, int tiles_n) {
  using Cfg = H3Cfg<MI, NI, WM, WN, STAGES, KB>;
  constexpr bool TR = EPI == EPI_SWIGLU_T || EPI == EPI_SWIGLU_T_H2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  xcd_grouped_tile(blockIdx.x, tiles_m, tiles_n, p.group_m, tm, tn);
  const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;

  const unsigned a_slab = (unsigned)(2 * p.RA * 32), w_slab = (unsigned)(2 * p.RW * 32);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.A2), 0, (int)((int64_t)p.K16 * a_slab - p.a_off), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.W2), 0, (int)((int64_t)p.K16 * w_slab - p.w_off), 0x00020000);
  unsigned a_voff[2], w_voff[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    a_voff[pl] = (unsigned)(((int64_t)pl * p.RA + m0 + 32 * wave) * 32 + lane * 16);
    w_voff[pl] = (unsigned)(((int64_t)pl * p.RW + n0 + 32 * wave) * 32 + lane * 16);
  }
  // stage step `ks` = k-blocks ks * KB ... ks * KB + KB - 1 (past the last k-block: out of the descriptor's range, zero-fills)
  auto issue = [&](int ks, int stage) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      unsigned char* st = smem + stage * Cfg::STAGE + kb * Cfg::KSTAGE + wave * 1024;
      const unsigned ao = (unsigned)(ks * KB + kb) * a_slab, wo = (unsigned)(ks * KB + kb) * w_slab;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < Cfg::A_DMA; ++c)
          dma16_to_lds(a_rsrc, st + pl * Cfg::A_PLANE + c * (1024 * Cfg::NW), a_voff[pl] + c * (1024 * Cfg::NW), ao);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < Cfg::W_DMA; ++c)
          dma16_to_lds(w_rsrc, st + Cfg::A_OP + pl * Cfg::W_PLANE + c * (1024 * Cfg::NW), w_voff[pl] + c * (1024 * Cfg::NW), wo);
    }
  };

  const int fr = lane & 31, fh = lane >> 5;
  const unsigned char* frag = smem + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int nk = (p.K16 + KB - 1) / KB;                    // stage steps
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s);

  auto slab = [&](int kt, int stage) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned char* sa = frag + stage * Cfg::STAGE + kb * Cfg::KSTAGE + (wm * 32 * MI) * 32;
      const unsigned char* sw = frag + stage * Cfg::STAGE + kb * Cfg::KSTAGE + Cfg::A_OP + (wn * 32 * NI) * 32;
      f16x8 a[MI][2], b[NI][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi][pl] = *reinterpret_cast<const f16x8*>(sa + pl * Cfg::A_PLANE + mi * 1024);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni][pl] = *reinterpret_cast<const f16x8*>(sw + pl * Cfg::W_PLANE + ni * 1024);
      }
      if (kb == 0) issue(kt + STAGES - 1, (stage + STAGES - 1) % STAGES);
      // TR (EPI_SWIGLU_T*): the weight fragment is the MFMA's A operand, so acc[mi][ni] holds the TRANSPOSED 32 x 32 block --
      // lane = token, registers = 16 weight rows -- same products, same sums, other owner of each element
#define ANYLOC_H3_TERM(pa, pb)                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
      acc[mi][ni] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni][pb], a[mi][pa], acc[mi][ni], 0, 0, 0) \
                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
      ANYLOC_H3_TERM(1, 0) ANYLOC_H3_TERM(0, 1) ANYLOC_H3_TERM(0, 0)
#undef ANYLOC_H3_TERM
    }
    if constexpr (KB == 1) {
      constexpr int PIECES = Cfg::NDMA, G = (3 * MI * NI) / (PIECES + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };
  for (int kt = 0; kt < nk; kt += STAGES) {
    slab(kt, 0);
    if (kt + 1 < nk) slab(kt + 1, 1);
    if (STAGES > 2 && kt + 2 < nk) slab(kt + 2, 2);
    if (STAGES > 3 && kt + 3 < nk) slab(kt + 3, 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (TR) {
    // SwiGLU on transposed accumulators (weights in the 16-channel block layout, include/anyloc_hip.h): lane (token n =
    // lane & 31, half hl) holds, of the 32-row weight block ni, rows 4 hl + {0..3}, 8 + 4 hl + {0..3} = the GATES of hidden
    // channels c .. c+7 (c = 16 * block + 8 hl) in registers 0..7 and rows 16 + ..., 24 + ... = their VALUES in registers
    // 8..15.  silu(g) * v of one token and 8 consecutive channels is exactly one 16-byte chunk per plane of the fc2 operand
    // image: no LDS transposition, two 16-byte stores per block and lane, 32 rows x 32 B = whole 1-KiB runs per instruction.
    const int hl = lane >> 5;
    const int64_t wr0 = n0 + wn * 32 * NI;               // first weight row of this wave (N % (32 NI) == 0: all-in or all-out)
    if (wr0 >= p.N) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t row = m0 + wm * 32 * MI + mi * 32 + (lane & 31);
      const bool rok = row < p.M;
      const float ai = rok ? p.a_inv[row] : 0.0f;
      float cs = 1.0f;
      if constexpr (EPI == EPI_SWIGLU_T_H2) cs = rok ? h2_scale_of_inv(p.c_inv[row]) : 0.0f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int64_t wr = wr0 + ni * 32 + 4 * hl;
        f32x4 sc[4], bi[4];                              // [gate rows 0..3, gate rows 4..7, value rows 0..3, value rows 4..7]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sc[q] = *reinterpret_cast<const f32x4*>(p.w_inv + wr + 8 * q);
          bi[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wr + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = acc[mi][ni][j] * (ai * sc[j >> 2][j & 3]) + bi[j >> 2][j & 3];
          const float v = acc[mi][ni][8 + j] * (ai * sc[2 + (j >> 2)][j & 3]) + bi[2 + (j >> 2)][j & 3];
          o[j] = (p.fast_silu ? h3_silu_fast(g) : h3_silu(g)) * v;
        }
        const int64_t blk = (wr0 + ni * 32) >> 5;         // 16 hidden channels per weight block: channels 16 blk + 8 hl + j
        if (rok) {
          if constexpr (EPI == EPI_SWIGLU_T_H2) {
            unsigned qh[4], ql[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h2_pack2(o[2 * j] * cs, o[2 * j + 1] * cs, qh[j], ql[j]);
            hu32x4 ph, plo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
            unsigned char* dst = p.C2 + ((blk * 2) * p.RC + row) * 32 + ((hl ^ (int)((row >> 3) & 1)) << 4);
            *reinterpret_cast<hu32x4*>(dst) = ph;
            *reinterpret_cast<hu32x4*>(dst + p.RC * 32) = plo;
          } else {
            float* dst = p.C + row * p.ldc + 16 * blk + 8 * hl;
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
          }
        }
      }
    }
    return;
  }
  // ---- epilogue: acc * 2^-(e_row + e_col), then the same fused forms as gemm_x6.hip / gemm_f32.hip ----
  const int64_t wrow0 = m0 + wm * 32 * MI + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * 32 * NI + (lane & 31);
  if constexpr (EPI == EPI_LS_RESID) {
    if (p.epi_lds) {
      // x += gamma * (acc * 2^-(e_row+e_col) + bias) with 16-byte global accesses: the C/D layout gives a lane one
      // column of 16 rows (dword read-modify-write, 64 + 64 memory instructions per 32x64 block pair); instead each
      // wave transposes 32 x 64 sub-blocks through its own 8.5 KiB of the (now idle) LDS ring and every lane handles
      // four consecutive columns of a row: 8 + 8 memory instructions per sub-block.
      __builtin_amdgcn_s_barrier();                       // nobody reads fragments from the ring any more
      float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
      const int lr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float ai[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
          ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
        }
#pragma unroll
        for (int np = 0; np < NI / 2; ++np) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int ni = 2 * np + nb;
            const int64_t col = wcol0 + ni * 32;
            const bool cok = col < p.N;
            const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
            const float sw_ = cok ? p.w_inv[col] : 0.0f, gam = cok ? p.gamma[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 68 + nb * 32 + (lane & 31)] =
                  (acc[mi][ni][r] * (ai[r] * sw_) + bv) * gam;
          }
          const int64_t colb = n0 + wn * 32 * NI + np * 64 + c4;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + lr0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c4]);
            const int64_t row = m0 + wm * 32 * MI + mi * 32 + lr;
            if (row < p.M && colb < p.N) {
              const int64_t o = row * p.ldc + colb;
              f32x4 x4 = *reinterpret_cast<const f32x4*>(&p.resid[o]);
              x4[0] += t[0]; x4[1] += t[1]; x4[2] += t[2]; x4[3] += t[3];
              *reinterpret_cast<f32x4*>(&p.C[o]) = x4;
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_QKV_PLANES) {
    // q | k | v leave the kernel as the per-(head, 32-row group) two-plane fp16 tiles of attention_h3 (layout: common.hpp).
    // A wave owns 32 MI rows x 32 NI columns = MI row groups x NI/2 heads of ONE part (D % (32 NI) == 0).  v tiles are
    // written straight from the C/D layout (a lane holds one d and the 16 rows of its half in exactly the order the
    // consumer's B fragments want); q / k tiles are row-major, so they are transposed through the idle LDS ring.
    __builtin_amdgcn_s_barrier();
    float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const int Dm = p.heads * 64;
    const int64_t colw = n0 + wn * 32 * NI;
    const int part = (int)(colw / Dm);
    const int head0 = (int)((colw - (int64_t)part * Dm) >> 6);
    const int hl = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t rb = m0 + wm * 32 * MI + mi * 32;
      if (rb >= p.M || colw >= p.N) continue;
      float ai[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
      }
#pragma unroll
      for (int hh = 0; hh < NI / 2; ++hh) {
        float v[2][16];
        float amax = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int64_t col = wcol0 + (2 * hh + nb) * 32;
          const float bv = p.bias ? p.bias[col] : 0.0f, sw_ = p.w_inv[col];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            v[nb][r] = ai[r] != 0.0f ? acc[mi][2 * hh + nb][r] * (ai[r] * sw_) + bv : 0.0f;
            amax = fmaxf(amax, fabsf(v[nb][r]));
          }
        }
        amax = wave_max(amax);
        float inv;
        const float scale = h2_row_scale(amax, inv);
        const int64_t tile = (int64_t)(part * p.heads + head0 + hh) * p.groups + (rb >> 5);
        if (lane == 0) p.qkv_inv[tile] = inv;
        unsigned char* dst = p.qkv_planes + tile * 8192;
        if (part == 2) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int d = nb * 32 + (lane & 31);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              unsigned qh[4], ql[4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                h2_pack2(v[nb][8 * s2 + 2 * j] * scale, v[nb][8 * s2 + 2 * j + 1] * scale, qh[j], ql[j]);
              hu32x4 ph, plo;
#pragma unroll
              for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
              unsigned char* o = dst + d * 64 + (((hl * 2 + s2) ^ ((d >> 2) & 3)) << 4);
              *reinterpret_cast<hu32x4*>(o) = ph;
              *reinterpret_cast<hu32x4*>(o + 4096) = plo;
            }
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + nb * 32 + (lane & 31)] = v[nb][r] * scale;
          const int lr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + lr0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c4]);
            uint2 ph, plo;
            h2_pack2(t[0], t[1], ph.x, plo.x);
            h2_pack2(t[2], t[3], ph.y, plo.y);
            unsigned char* o = dst + lr * 128 + (((c4 >> 3) ^ ((lr >> 1) & 7)) << 4) + (c4 & 4) * 2;
            *reinterpret_cast<uint2*>(o) = ph;
            *reinterpret_cast<uint2*>(o + 4096) = plo;
          }
        }
      }
    }
    return;
  }
  if constexpr (EPI == EPI_GELU_H2 || EPI == EPI_SWIGLU_H2) {
    // the FFN hidden activation leaves the kernel as the h2 image of the fc2 GEMM, every row scaled by the caller's
    // power of two (1 / c_inv[row], an upper bound of the row: layernorm_h2).  32 x 64 sub-blocks are transposed through
    // the idle LDS ring so that a lane owns eight consecutive k of a row = one 16-byte half of an image row per plane.
    __builtin_amdgcn_s_barrier();
    float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    constexpr bool SW = EPI == EPI_SWIGLU_H2;
    constexpr int OC = SW ? 16 * NI : 64;                  // output columns per transposed sub-block
    constexpr int NSUB = SW ? 1 : NI / 2;
    constexpr int LPR = OC / 8;                            // lanes per row when reading back
    constexpr int RPI = 64 / LPR;                          // rows per read-back iteration
    const int hl = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t rb = m0 + wm * 32 * MI + mi * 32;
      if (rb >= p.M) continue;
      float ai[16], cs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        ai[r] = row < p.M ? p.a_inv[row] : 0.0f;
        cs[r] = row < p.M ? h2_scale_of_inv(p.c_inv[row]) : 0.0f;
      }
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        int64_t ocol0;                                     // first output column of this sub-block
        if constexpr (SW) {
          ocol0 = (n0 + wn * 32 * NI) / 2;
#pragma unroll
          for (int nj = 0; nj < NI; nj += 2) {
            const int64_t colg = wcol0 + nj * 32, colv = colg + 32;
            const bool cok = colv < p.N;
            const float bg = (cok && p.bias) ? p.bias[colg] : 0.0f, bv = (cok && p.bias) ? p.bias[colv] : 0.0f;
            const float sg = cok ? p.w_inv[colg] : 0.0f, sv = cok ? p.w_inv[colv] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float g = acc[mi][nj][r] * (ai[r] * sg) + bg;
              const float v = acc[mi][nj + 1][r] * (ai[r] * sv) + bv;
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + (nj / 2) * 32 + (lane & 31)] =
                  (p.fast_silu ? h3_silu_fast(g) : h3_silu(g)) * v * cs[r];
            }
          }
        } else {
          ocol0 = n0 + wn * 32 * NI + sub * 64;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int64_t col = wcol0 + (2 * sub + nb) * 32;
            const bool cok = col < p.N;
            const float bv = (cok && p.bias) ? p.bias[col] : 0.0f, sw_ = cok ? p.w_inv[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[((r & 3) + 8 * (r >> 2) + 4 * hl) * 68 + nb * 32 + (lane & 31)] =
                  h3_gelu_erf(acc[mi][2 * sub + nb][r] * (ai[r] * sw_) + bv) * cs[r];
          }
        }
        const int lr0 = lane / LPR, c8 = (lane % LPR) * 8;
        const int64_t kcol = ocol0 + c8;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int lr = it * RPI + lr0;
          const int64_t row = rb + lr;
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c8]);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(&st[lr * 68 + c8 + 4]);
          unsigned qh[4], ql[4];
          h2_pack2(t0[0], t0[1], qh[0], ql[0]);
          h2_pack2(t0[2], t0[3], qh[1], ql[1]);
          h2_pack2(t1[0], t1[1], qh[2], ql[2]);
          h2_pack2(t1[2], t1[3], qh[3], ql[3]);
          hu32x4 ph, plo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ph[j] = qh[j]; plo[j] = ql[j]; }
          if (row < p.M && kcol < (SW ? p.N / 2 : p.N)) {
            unsigned char* o = p.C2 + (((kcol >> 4) * 2) * p.RC + row) * 32 + ((((kcol >> 3) & 1) ^ (int)((row >> 3) & 1)) << 4);
            *reinterpret_cast<hu32x4*>(o) = ph;
            *reinterpret_cast<hu32x4*>(o + p.RC * 32) = plo;
          }
        }
      }
    }
    return;
  }
  if constexpr (EPI == EPI_SWIGLU) {
    float bg[NI / 2], bv[NI / 2], sg[NI / 2], sv[NI / 2];
    bool cok[NI / 2];
#pragma unroll
    for (int nj = 0; nj < NI; nj += 2) {
      const int64_t colg = wcol0 + nj * 32, colv = colg + 32;
      cok[nj / 2] = colv < p.N;
      bg[nj / 2] = (cok[nj / 2] && p.bias) ? p.bias[colg] : 0.0f;
      bv[nj / 2] = (cok[nj / 2] && p.bias) ? p.bias[colv] : 0.0f;
      sg[nj / 2] = cok[nj / 2] ? p.w_inv[colg] : 0.0f;
      sv[nj / 2] = cok[nj / 2] ? p.w_inv[colv] : 0.0f;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          const float ai = p.a_inv[row];
#pragma unroll
          for (int nj = 0; nj < NI; nj += 2)
            if (cok[nj / 2]) {
              const int64_t ocol = (n0 + wn * 32 * NI + nj * 32) / 2 + (lane & 31);
              const float g = acc[mi][nj][r] * (ai * sg[nj / 2]) + bg[nj / 2];
              const float v = acc[mi][nj + 1][r] * (ai * sv[nj / 2]) + bv[nj / 2];
              p.C[row * p.ldc + ocol] = h3_silu(g) * v;
            }
        }
      }
  } else {
    float bv[NI], sw_[NI], gam[NI];
    bool cok[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t col = wcol0 + ni * 32;
      cok[ni] = col < p.N;
      bv[ni] = (cok[ni] && p.bias) ? p.bias[col] : 0.0f;
      sw_[ni] = cok[ni] ? p.w_inv[col] : 0.0f;
      gam[ni] = 0.0f;
      if constexpr (EPI == EPI_LS_RESID) gam[ni] = cok[ni] ? p.gamma[col] : 0.0f;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          const float ai = p.a_inv[row];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if (cok[ni]) {
              const float v = acc[mi][ni][r] * (ai * sw_[ni]) + bv[ni];
              const int64_t o = row * p.ldc + wcol0 + ni * 32;
              if constexpr (EPI == EPI_STORE) p.C[o] = p.accumulate ? p.C[o] + v : v;
              else if constexpr (EPI == EPI_GELU) p.C[o] = h3_gelu_erf(v);
              else p.C[o] = p.resid[o] + v * gam[ni];
            }
        }
      }
  }
}

// ---- quantisers: rows held in registers (NV float4 per lane and row, 4 rows per wave, 16 rows per block) ----
// the scaled values of 16 rows go through a 16 x 256 LDS tile, chunk by chunk, and are stored in IMAGE order
// (thread = (k-block, row, half): whole 512-byte runs per store instruction, 16 bytes per lane and plane)
// store chunk i (256 columns) of 16 rows, already scaled and sitting in the LDS tile, in IMAGE order
// (thread = (k-block, row, half): whole 512-byte runs per store instruction, 16 bytes per lane and plane)
template <int RB = 16>
__device__ __forceinline__ void h2_store_chunk(float (*tile)[256 + 4], int i, int dim, int64_t row0, int64_t rows,
                                               unsigned char* out, int64_t R) {
  static_assert(RB == 16 || RB == 8 || RB == 4, "16 rows per block (4 per wave), 8 (2 per wave) or 4 (1 per wave)");
  const int tid = threadIdx.x;
  constexpr int ITEMS = RB * 32;                           // (k-block, row, half) triples of one 256-column chunk
#pragma unroll
  for (int u = 0; u < (ITEMS + 255) / 256; ++u) {
    const int item = tid + 256 * u;
    const int kbl = item / (2 * RB), r = (item >> 1) & (RB - 1), half = item & 1;
    const int k0 = 256 * i + 16 * kbl + 8 * half;
    const int64_t row = row0 + r;
    if (item < ITEMS && k0 < dim && row < rows) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half]);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(&tile[r][16 * kbl + 8 * half + 4]);
      hu32x4 ph, plo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 pr;
        pr[0] = j < 2 ? lo[2 * j] : hi[2 * j - 4];
        pr[1] = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
        const f16x2 h = __builtin_convertvector(pr, f16x2);
        f32x2 res;
        res[0] = pr[0] - (float)h[0];
        res[1] = pr[1] - (float)h[1];
        const f16x2 l = __builtin_convertvector(res, f16x2);
        ph[j] = __builtin_bit_cast(unsigned, h);
        plo[j] = __builtin_bit_cast(unsigned, l);
      }
      unsigned char* dst = out + (((int64_t)(k0 >> 4) * 2) * R + row) * 32 + ((half ^ (int)((row >> 3) & 1)) << 4);
      *reinterpret_cast<hu32x4*>(dst) = ph;
      *reinterpret_cast<hu32x4*>(dst + R * 32) = plo;
    }
  }
}

// rows held in registers: the scaled values of 16 rows go through the LDS tile chunk by chunk
template <int NV, int RPW = 4>
__device__ __forceinline__ void h2_store_rows(const f32x4 (&v)[RPW][NV], const float (&scale)[RPW], float (*tile)[256 + 4],
                                              int dim, int64_t row0, int64_t rows, unsigned char* out, int64_t R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (i > 0) __syncthreads();
    if (idx < n4) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[q][i][j] * scale[q];
        *reinterpret_cast<f32x4*>(&tile[wave * RPW + q][4 * lane]) = o;
      }
    }
    __syncthreads();
    h2_store_chunk<4 * RPW>(tile, i, dim, row0, rows, out, R);
  }
}

// wide rows (K > 2048): two passes over the row instead of holding it in registers -- pass 1 finds the maximum,
// pass 2 re-reads the 16 rows (L2-resident: 16 x 4 K floats) and quantises; 8x the occupancy of the register version
__global__ __launch_bounds__(256) void split_h2_stream_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                              unsigned char* __restrict__ out, float* __restrict__ inv,
                                                              int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const f32x4* xr[4];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    xr[q] = reinterpret_cast<const f32x4*>(x + row * ldx);
    float amax = 0.f;
    for (int idx = lane; idx < n4; idx += 64) {
      const f32x4 t = xr[q][idx];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))));
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * 4 + q < rows) inv[row] = iv;
  }
  const int nchunks = (dim + 255) / 256;
  for (int i = 0; i < nchunks; ++i) {
    const int idx = lane + 64 * i;
    if (i > 0) __syncthreads();
    if (idx < n4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o = xr[q][idx];
        o[0] *= scale[q]; o[1] *= scale[q]; o[2] *= scale[q]; o[3] *= scale[q];
        *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
      }
    }
    __syncthreads();
    h2_store_chunk(tile, i, dim, row0, rows, out, R);
  }
}

// ---- very wide rows (retrieval: K = 49 152 VLAD columns; any K % 16 == 0 above 4096) -- two kernels, both with one
// workgroup per small unit of work so that thousands of them stream HBM: (1) one workgroup per row finds the row's largest
// magnitude (-> the power-of-two scale) and its sum of squares (the caller's F.normalize / L2 terms come out of the
// same read); (2) one workgroup per (16 rows, 2048 columns) quantises with the scales of (1).
__global__ __launch_bounds__(256) void row_amax_sq_kernel(const float* __restrict__ x, int64_t ldx, int64_t dim,
                                                          float* __restrict__ inv, float* __restrict__ ss) {
  __shared__ float red_m[4], red_s[4];
  const f32x4* r = reinterpret_cast<const f32x4*>(x + (int64_t)blockIdx.x * ldx);
  const int64_t n4 = dim >> 2;
  float amax = 0.f, sq = 0.f;
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {        // four 16-byte loads in flight per thread
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + 256 * u;
      if (j < n4) t[u] = r[j];
      else t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(t[u][0]), fabsf(t[u][1])), fmaxf(fabsf(t[u][2]), fabsf(t[u][3]))));
      sq += (t[u][0] * t[u][0] + t[u][1] * t[u][1]) + (t[u][2] * t[u][2] + t[u][3] * t[u][3]);
    }
  }
  amax = wave_max(amax);
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) {
    red_m[threadIdx.x >> 6] = amax;
    red_s[threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float iv;
    h2_row_scale(fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3])), iv);
    inv[blockIdx.x] = iv;
    if (ss) ss[blockIdx.x] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
  }
}

constexpr int WIDE_CHUNKS = 8;      // 256-column chunks per workgroup of the quantising kernel

__global__ __launch_bounds__(256) void split_h2_wide_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                            unsigned char* __restrict__ out, const float* __restrict__ inv,
                                                            int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int c0 = blockIdx.y * WIDE_CHUNKS;
  const int nchunks = min(WIDE_CHUNKS, (dim + 255) / 256 - c0);
  const int n4 = dim >> 2;
  const f32x4* xr[4];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    xr[q] = reinterpret_cast<const f32x4*>(x + row * ldx);
    scale[q] = h2_scale_of_inv(inv[row]);
  }
  f32x4 cur[4], nxt[4];
  auto load = [&](int i, f32x4 (&dst)[4]) {
    const int idx = lane + 64 * (c0 + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = idx < n4 ? xr[q][idx] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  load(0, cur);
  for (int i = 0; i < nchunks; ++i) {
    if (i + 1 < nchunks) load(i + 1, nxt);                 // the next chunk's loads fly over this chunk's LDS round trip
    if (i > 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = cur[q];
      o[0] *= scale[q]; o[1] *= scale[q]; o[2] *= scale[q]; o[3] *= scale[q];
      *reinterpret_cast<f32x4*>(&tile[wave * 4 + q][4 * lane]) = o;
    }
    __syncthreads();
    h2_store_chunk(tile, c0 + i, dim, row0, rows, out, R);
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
}

// fp32 row-major [rows, K] -> h2 image + inv[row] = 2^-e
template <int NV>
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, int64_t ldx, int dim, int64_t rows,
                                                       unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R) {
  __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  f32x4 v[4][NV];
  float scale[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = min(row0 + wave * 4 + q, rows - 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ldx);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        v[q][i] = xr[idx];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[q][i][0]), fabsf(v[q][i][1])), fmaxf(fabsf(v[q][i][2]), fabsf(v[q][i][3]))));
      }
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * 4 + q < rows) inv[row] = iv;
  }
  h2_store_rows<NV>(v, scale, tile, dim, row0, rows, out, R);
}

// LayerNorm (torch semantics, biased variance) whose output is quantised straight into the h2 image
// bound_inv != nullptr (LN2 of a block whose FFN activation is quantised in the fc1 epilogue): also writes
// bound_inv[row] = 2^-e, where 2^e scales an UPPER BOUND of |act(fc1(y_row))| into [2^14, 2^15).  Cauchy-Schwarz:
// |fc1_j(y)| <= ||y||_2 max_j ||W_j||_2 + max_j |b_j|, |gelu(t)| <= |t|, |silu(g) v| <= |g| |v|; bound4 = {gate (or fc1)
// row-norm maximum, gate bias maximum, value row-norm maximum, value bias maximum} (value pair 0 / 0: GELU MLP).
// RPW rows per wave (4 waves per block): 4 by default; 1 when there are few rows (one or two images: 530 rows are 34
// blocks of 16 rows on 256 CUs -- 25 us of latency per launch; 133 blocks of 4 rows spread them).  Per-row arithmetic does
// not depend on RPW, so the results are bitwise the same.
template <int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_h2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, int dim, int64_t rows, float eps,
                                                           unsigned char* __restrict__ out, float* __restrict__ inv, int64_t R,
                                                           const f32x4 bound4, float* __restrict__ bound_inv) {
  __shared__ __attribute__((aligned(16))) float tile[4 * RPW][256 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = dim >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * (4 * RPW);
  f32x4 v[RPW][NV];
  float scale[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int64_t row = min(row0 + wave * RPW + q, rows - 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * dim);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        v[q][i] = xr[idx];
        s += (v[q][i][0] + v[q][i][1]) + (v[q][i][2] + v[q][i][3]);
      }
    }
    const float mean = wave_sum(s) / (float)dim;
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < n4) {
        const float d0 = v[q][i][0] - mean, d1 = v[q][i][1] - mean, d2 = v[q][i][2] - mean, d3 = v[q][i][3] - mean;
        qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(qs) / (float)dim + eps);
    float amax = 0.f, ysq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        const f32x4 wv = reinterpret_cast<const f32x4*>(w)[idx], bv = reinterpret_cast<const f32x4*>(b)[idx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[q][i][j] = (v[q][i][j] - mean) * rstd * wv[j] + bv[j];
          amax = fmaxf(amax, fabsf(v[q][i][j]));
          ysq += v[q][i][j] * v[q][i][j];
        }
      }
    }
    float iv;
    scale[q] = h2_row_scale(wave_max(amax), iv);
    if (lane == 0 && row0 + wave * RPW + q < rows) inv[row] = iv;
    if (bound_inv) {
      const float yn = sqrtf(wave_sum(ysq)) * 1.001f;                 // 0.1 % head room for the fp32 roundings
      const float bg = yn * bound4[0] + bound4[1];
      const float bd = (bound4[2] > 0.f || bound4[3] > 0.f) ? bg * (yn * bound4[2] + bound4[3]) : bg;
      float biv;
      h2_row_scale(fmaxf(bd * 1.001f, 1e-30f), biv);
      if (lane == 0 && row0 + wave * RPW + q < rows) bound_inv[row] = biv;
    }
  }
  h2_store_rows<NV, RPW>(v, scale, tile, dim, row0, rows, out, R);
}

}  // namespace

size_t h2_bytes(int64_t rows, int64_t K) { return (size_t)((K + 15) / 16) * 2 * (size_t)rows * 32; }

int split_h2_wide(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, float* row_sumsq,
                  hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && rows > 0 && K > 0 && ldx >= K, "split_h2_wide: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && ldx % 4 == 0 && K < (1ll << 31) && rows < (1ll << 31) && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                   "split_h2_wide: K must be a multiple of 16, rows 16-byte aligned (K=%lld)", (long long)K);
  ProfScope prof("split_h2_wide", stream, 0.0, 12.0 * rows * K);
  hipLaunchKernelGGL(row_amax_sq_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, K, inv_scale, row_sumsq);
  ANYLOC_TRY(launch_status("row_amax_sq_kernel"));
  const int chunks = (int)((K + 255) / 256);
  const dim3 grid((unsigned)((rows + 15) / 16), (unsigned)((chunks + WIDE_CHUNKS - 1) / WIDE_CHUNKS));
  ANYLOC_CHECK_ARG(grid.y < 65536, "split_h2_wide: K too large");
  hipLaunchKernelGGL(split_h2_wide_kernel, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, static_cast<unsigned char*>(h2),
                     inv_scale, rows);
  return launch_status("split_h2_wide_kernel");
}

int split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, hipStream_t stream) {
  ANYLOC_CHECK_ARG(x && h2 && inv_scale && rows > 0 && K > 0 && ldx >= K, "split_h2: bad arguments");
  ANYLOC_CHECK_ARG(K % 16 == 0 && ldx % 4 == 0, "split_h2: K must be a multiple of 16 (got %lld)", (long long)K);
  if (K > 4096) return split_h2_wide(x, ldx, rows, K, h2, inv_scale, nullptr, stream);
  ProfScope prof("split_h2", stream, 0.0, 8.0 * rows * K);
  const dim3 grid((unsigned)((rows + 15) / 16));
  unsigned char* out = static_cast<unsigned char*>(h2);
  const int nv = (int)((K / 4 + 63) / 64);
#define ANYLOC_SPLIT_H2(NVV) \
  hipLaunchKernelGGL(split_h2_kernel<NVV>, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, out, inv_scale, rows)
  if (nv <= 1) ANYLOC_SPLIT_H2(1);
  else if (nv <= 2) ANYLOC_SPLIT_H2(2);
  else if (nv <= 4) ANYLOC_SPLIT_H2(4);
  else if (nv <= 6) ANYLOC_SPLIT_H2(6);
  else if (nv <= 8) ANYLOC_SPLIT_H2(8);
  else hipLaunchKernelGGL(split_h2_stream_kernel, grid, dim3(256), 0, stream, x, ldx, (int)K, rows, out, inv_scale, rows);
#undef ANYLOC_SPLIT_H2
  return launch_status("split_h2_kernel");
}

int layernorm_h2(const float* x, const float* w, const float* b, int64_t rows, int dim, float eps, void* h2,
                 float* inv_scale, hipStream_t stream, const float* bound, float* bound_inv) {
  ANYLOC_CHECK_ARG(dim % 16 == 0 && dim <= 2048, "layernorm_h2: dim %d (needs a multiple of 16, at most 2048)", dim);
  ProfScope prof("layernorm_h2", stream, 8.0 * rows * dim, 8.0 * rows * dim);
  unsigned char* out = static_cast<unsigned char*>(h2);
  const int nv = (dim / 4 + 63) / 64;
  f32x4 b4;                                                  // bound: HOST array of 4 floats (or null)
  for (int i = 0; i < 4; ++i) b4[i] = bound ? bound[i] : 0.0f;
  // few rows (option ln_small_rows, default 4096 = seven 322 x 322 images): one row per wave, four per block; otherwise two
  // rows per wave (48 instead of 96 data registers: more waves in flight; B=61: 6.48 -> 5.65 ms per step, four rows per wave
  // was the round-2 kernel, one row per wave at this size 11.4 ms -- profiles/r03_ab_attn_kbatch_ln_rpw.log).  Option
  // ln_rows_per_wave (0 = that rule) forces 1, 2 or 4 at every size (A/B; same per-row arithmetic, same bits)
  const int64_t forced = option(OPT_LN_ROWS_PER_WAVE);
  const int rpw = (forced == 1 || forced == 2 || forced == 4) ? (int)forced : (rows < option(OPT_LN_SMALL_ROWS) ? 1 : 2);
  const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
#define ANYLOC_LN_H2_R(NVV, RPWV)                                                                                        \
  hipLaunchKernelGGL((layernorm_h2_kernel<NVV, RPWV>), grid, dim3(256), 0, stream, x, w, b, dim, rows, eps, out, inv_scale, \
                     rows, b4, bound ? bound_inv : nullptr)
#define ANYLOC_LN_H2(NVV)              \
  do {                                 \
    if (rpw == 1) ANYLOC_LN_H2_R(NVV, 1);      \
    else if (rpw == 2) ANYLOC_LN_H2_R(NVV, 2); \
    else ANYLOC_LN_H2_R(NVV, 4);               \
  } while (0)
  if (nv <= 1) ANYLOC_LN_H2(1);
  else if (nv <= 2) ANYLOC_LN_H2(2);
  else if (nv <= 3) ANYLOC_LN_H2(3);
  else if (nv <= 4) ANYLOC_LN_H2(4);
  else if (nv <= 6) ANYLOC_LN_H2(6);
  else ANYLOC_LN_H2(8);
#undef ANYLOC_LN_H2_R
#undef ANYLOC_LN_H2
  return launch_status("layernorm_h2_kernel");
}

template <int EPI>
int dispatch_h3(const H3Problem& p, hipStream_t stream) {
  // option h3_cfg (micro-benchmarks): 0 = 128x256 tile, 3-deep ring (default; 128x128 when there are few tiles); 1 = 2-deep;
  // 2-5 = 256x256 tiles (see the switch)
  const int cfg = (int)option(OPT_H3_CFG);
#define ANYLOC_LAUNCH_H3(MI, NI, WM, WN, ST, OCC) ANYLOC_LAUNCH_H3K(MI, NI, WM, WN, ST, OCC, 1)
#define ANYLOC_LAUNCH_H3K(MI, NI, WM, WN, ST, OCC, KB)                                                                \
  do {                                                                                                                \
    using Cfg = H3Cfg<MI, NI, WM, WN, ST, KB>;                                                                        \
    const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);         \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      ANYLOC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<MI, NI, WM, WN, ST, OCC, EPI, KB>), \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS));                          \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((gemm_h3_kernel<MI, NI, WM, WN, ST, OCC, EPI, KB>), dim3((unsigned)(tiles_m * tiles_n)),        \
                       dim3(64 * WM * WN), Cfg::LDS, stream, p, tiles_m, tiles_n);                                    \
  } while (0)
  const bool small = ((p.M + 127) / 128) * ((p.N + 255) / 256) < 512;
  if (small && cfg == 0) {
    // one or two images (the reference's scripts call the extractor per image): with 128x128 tiles proj / fc2 of ViT-g
    // are 60 workgroups on 256 CUs -- below option h3_tiny_max (default 256) such tiles the GEMM runs 64x64 tiles on
    // two-wave workgroups instead
    const int64_t tiny_max = option(OPT_H3_TINY_MAX);
    const int64_t deep_max = option(OPT_H3_DEEP_MAX), deep2_max = option(OPT_H3_DEEP2_MAX);
    if (((p.M + 127) / 128) * ((p.N + 127) / 128) < tiny_max) {
      // fewer 64x64 tiles than ~1.25 per CU (ViT-g proj / fc2 of one image: 216): a workgroup is alone on its CU, nothing
      // hides its per-k-block barrier and LDS round trip (measured 625 cycles per k-block for 192 cycles of MFMA), so
      // the ring stage holds FOUR k-blocks -- one barrier and one counted wait per 64 k.  Same k order per output element:
      // bitwise the result of the one-k-block kernel (option h3_deep_max = 0 restores it).  B=1: fc2 80 -> 49 us, proj 36 -> 24 us
      // per launch; with 320 ... 500 tiles (two images) two k-blocks per stage (profiles/r02_small_batch_kernels.log).
      if (((p.M + 63) / 64) * ((p.N + 63) / 64) < deep_max) {
        ANYLOC_LAUNCH_H3K(1, 2, 2, 1, 3, 2, 4);             // 64x64, 2 waves, 3 stages of 4 k-blocks (96 KiB)
        return launch_status("gemm_h3_kernel");
      }
      if (((p.M + 63) / 64) * ((p.N + 63) / 64) < deep2_max) {
        ANYLOC_LAUNCH_H3K(1, 2, 2, 1, 3, 2, 2);             // the same with 2 k-blocks per stage (48 KiB: three workgroups per CU)
        return launch_status("gemm_h3_kernel");
      }
      ANYLOC_LAUNCH_H3(1, 2, 2, 1, 3, 2);                   // 64x64, 2 waves
      return launch_status("gemm_h3_kernel");
    }
    ANYLOC_LAUNCH_H3(2, 2, 2, 2, 3, 2);                     // few tiles: 128x128
    return launch_status("gemm_h3_kernel");
  }
  switch (cfg) {
    case 1: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 2, 2); break;
    case 2: ANYLOC_LAUNCH_H3(2, 4, 4, 2, 3, 2); break;     // 256x256, 8 waves (2 per SIMD, one workgroup per CU), 96 KiB ring
    case 3: ANYLOC_LAUNCH_H3(2, 4, 4, 2, 4, 2); break;     // the same, 4-deep ring (128 KiB)
    case 4: ANYLOC_LAUNCH_H3(4, 4, 2, 2, 4, 1); break;     // 256x256, 4 waves of 128x128 (one per SIMD), 4-deep ring
    case 5: ANYLOC_LAUNCH_H3(4, 4, 2, 2, 3, 1); break;     // the same, 3-deep ring
    default: ANYLOC_LAUNCH_H3(2, 4, 2, 2, 3, 2); break;
  }
#undef ANYLOC_LAUNCH_H3
#undef ANYLOC_LAUNCH_H3K
  return launch_status("gemm_h3_kernel");
}

int gemm_h3(const H3Problem& p_in, int epilogue, hipStream_t stream) {
  H3Problem p = p_in;
  ANYLOC_CHECK_ARG(p.A2 && p.a_inv && p.W2 && p.w_inv && (p.C || epilogue >= EPI_QKV_PLANES), "gemm_h3: null operand");
  ANYLOC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K16 > 0 && p.RA >= p.M && p.RW >= p.N, "gemm_h3: bad shape");
  ANYLOC_CHECK_ARG((size_t)p.K16 * 2 * (size_t)p.RA * 32 < (1ull << 31) && (size_t)p.K16 * 2 * (size_t)p.RW * 32 < (1ull << 31),
                   "gemm_h3: operand image exceeds the 2 GiB buffer-addressing range");
  const int64_t K = 16ll * p.K16;
  ProfScope prof(p.tag ? p.tag : "gemm_h3", stream, 2.0 * p.M * p.N * K, 4.0 * (p.M + p.N) * K + 4.0 * p.M * p.N);
  // tile-rows per scheduling group (co-resident workgroups of an XCD share A / W panels through its L2): option
  // h3_group_m, default 8
  p.group_m = (int)std::max<int64_t>(1, option(OPT_H3_GROUP_M));
  // plain-store GEMMs of >= 256 tiles of 256 x 256 run on the 16 x 16 x 32 MFMA kernel (gemm_h3m.hip; option h3_mfma16: -1 =
  // when the contraction is >= 4096 long -- the retrieval panels, +3.6 % -- 0 never, 1 whatever the length)
  const int64_t m16 = option(OPT_H3_MFMA16);
  if (epilogue == EPI_STORE && m16 != 0 && (m16 > 0 || p.K16 >= 256) && ((p.M + 255) / 256) * ((p.N + 255) / 256) >= 256) {
    const int rc = gemm_h3m(p, epilogue, stream);
    if (rc != ANYLOC_ERR_UNSUPPORTED) return rc;
  }
  switch (epilogue) {
    case EPI_STORE: return dispatch_h3<EPI_STORE>(p, stream);
    case EPI_GELU: return dispatch_h3<EPI_GELU>(p, stream);
    case EPI_LS_RESID: {
      ANYLOC_CHECK_ARG(p.gamma && p.resid, "gemm_h3: LS_RESID needs gamma and resid");
      H3Problem q = p;
      // option h3_epi_lds = 0: the dword read-modify-write epilogue (also the fallback for unaligned C)
      q.epi_lds = option(OPT_H3_EPI_LDS) != 0 && p.N % 4 == 0 && p.ldc % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
      return dispatch_h3<EPI_LS_RESID>(q, stream);
    }
    case EPI_SWIGLU:
      ANYLOC_CHECK_ARG(p.N % 64 == 0, "gemm_h3: SWIGLU needs N %% 64 == 0");
      return dispatch_h3<EPI_SWIGLU>(p, stream);
    case EPI_QKV_PLANES:
      ANYLOC_CHECK_ARG(p.qkv_planes && p.qkv_inv && p.heads > 0 && p.N == 3ll * p.heads * 64 && (p.heads * 64) % 128 == 0 &&
                           p.groups == (p.M + 31) / 32,
                       "gemm_h3: QKV_PLANES needs N = 3 * heads * 64, D %% 128 == 0 and groups = ceil(M / 32)");
      return dispatch_h3<EPI_QKV_PLANES>(p, stream);
    case EPI_GELU_H2:
      ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M && p.N % 64 == 0, "gemm_h3: GELU_H2 needs an output image, c_inv and N %% 64 == 0");
      return dispatch_h3<EPI_GELU_H2>(p, stream);
    case EPI_SWIGLU_H2:
      ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M && p.N % 128 == 0, "gemm_h3: SWIGLU_H2 needs an output image, c_inv and N %% 128 == 0");
      p.fast_silu = option(OPT_H3_FAST_SILU) != 0;
      return dispatch_h3<EPI_SWIGLU_H2>(p, stream);
    case EPI_SWIGLU_T:
    case EPI_SWIGLU_T_H2:
      ANYLOC_CHECK_ARG(p.N % 128 == 0 && (reinterpret_cast<uintptr_t>(p.w_inv) & 15) == 0 &&
                           (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0),
                       "gemm_h3: SWIGLU_T needs N %% 128 == 0 and 16-byte aligned w_inv / bias");
      if (epilogue == EPI_SWIGLU_T_H2)
        ANYLOC_CHECK_ARG(p.C2 && p.c_inv && p.RC >= p.M, "gemm_h3: SWIGLU_T_H2 needs an output image and c_inv");
      else
        ANYLOC_CHECK_ARG(p.C && p.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0, "gemm_h3: SWIGLU_T needs an aligned fp32 output");
      p.fast_silu = option(OPT_H3_FAST_SILU) != 0;
      return epilogue == EPI_SWIGLU_T ? dispatch_h3<EPI_SWIGLU_T>(p, stream) : dispatch_h3<EPI_SWIGLU_T_H2>(p, stream);
    default: set_error("gemm_h3: unsupported epilogue %d", epilogue); return ANYLOC_ERR_INVALID_ARG;
  }
}

}  // namespace anyloc

using namespace anyloc;

extern "C" size_t anyloc_h2_bytes(int64_t rows, int64_t K) { return h2_bytes(rows, K); }

extern "C" int anyloc_split_h2(const float* x, int64_t ldx, int64_t rows, int64_t K, void* h2, float* inv_scale, void* stream) {
  return split_h2(x, ldx, rows, K, h2, inv_scale, (hipStream_t)stream);
}

extern "C" int anyloc_gemm_nt_h3(const void* a2, const float* a_inv, const void* w2, const float* w_inv, const float* bias,
                                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
  ANYLOC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 16 == 0 && ldc >= N, "gemm_nt_h3: bad shape");
  H3Problem p{};
  p.A2 = static_cast<const unsigned char*>(a2); p.RA = M; p.a_inv = a_inv;
  p.W2 = static_cast<const unsigned char*>(w2); p.RW = N; p.w_inv = w_inv;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K16 = (int)(K / 16); p.bias = bias;
  return gemm_h3(p, EPI_STORE, (hipStream_t)stream);
}
