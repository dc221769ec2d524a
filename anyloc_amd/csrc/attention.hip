// Multi-head self-attention, fp32, head_dim 64, flash-style (scores never leave
// registers).  Replaces the attention of every DINOv2 block executed by the
// hub model forward the reference triggers at utilities.py:269
// (facebookresearch/dinov2 layers/attention.py: softmax((q*64^-0.5) k^T) v).
//
// Design (CDNA4): one wave owns 32 queries; a 256-thread block = 4 waves = 128
// queries of one (image, head) and shares K/V tiles of 32 keys staged in LDS
// (double-buffered, register-prefetched).  The score block is computed
// TRANSPOSED, S^T = K_tile * Q^T, with v_mfma_f32_32x32x2_f32, so that in the
// MFMA C/D layout (col = lane&31 = query, rows = keys in registers) each lane
// holds 16 keys of ONE query: the online-softmax max/sum are in-register plus
// one lane^32 exchange, the running rescale of O^T is a per-lane scalar, and
// the probabilities are already in B-operand layout for O^T += V^T * P^T.
#include <cstdlib>

#include <type_traits>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

constexpr int HD = 64;        // head dim
constexpr int KT = 32;        // keys per tile
constexpr int KLD = HD + 4;   // padded K row (ds_read_b128 conflict-free)

// WB = waves (32-query tiles) per workgroup; FASTEXP = v_exp_f32-based exponential
// PRELOAD: read the whole K fragment set and the whole V column set of a tile into registers before
//          the MFMA chains that consume them (2 waves/SIMD instead of 3, but no LDS wait inside a chain)
// OUT3: the output is written as the three-plane bf16 image (x3 layout of gemm_x6.hip, R = batch*T rows) that the
// projection GEMM reads, instead of fp32
template <int WB, bool FASTEXP, bool PRELOAD, bool OUT3 = false>
__global__ __launch_bounds__(64 * WB, PRELOAD ? 2 : 1) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int T, int D, float scale, unsigned char* __restrict__ out3,
                                                            int64_t R3) {
  constexpr int NT = 64 * WB;
  constexpr int RPP = NT / 16;          // K/V rows staged per pass (16 lanes per 256-byte row)
  constexpr int NLD = KT / RPP;         // staging passes per tile
  __shared__ __attribute__((aligned(16))) float Ks[2][KT][KLD];
  __shared__ __attribute__((aligned(16))) float Vs[2][KT][HD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t ld = 3 * (int64_t)D;
  const float* base = qkv + b * T * ld;
  const int q0 = blockIdx.x * (32 * WB) + wave * 32;
  const int ql = lane & 31, h2 = lane >> 5;
  const bool wave_active = q0 < T;

  // ---- Q fragment (B operand): lane holds Q[q][8s + 4*h2 + j] * scale ----
  f32x4 qf[8];
  {
    const int qr = min(q0 + ql, T - 1);
    const float* qp = base + (int64_t)qr * ld + h * HD + 4 * h2;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * s);
      v[0] *= scale; v[1] *= scale; v[2] *= scale; v[3] *= scale;
      qf[s] = v;
    }
  }

  // ---- K/V staging: 16 lanes cover one 256-byte head row.  Buffer loads: descriptor = this image's
  // QKV rows (num_records = T rows, so keys past T read as zeros), constant per-thread byte offset,
  // key-tile offset as the scalar offset ----
  const int sr = tid >> 4, sc = tid & 15;
  const __amdgpu_buffer_rsrc_t kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(base), 0, (int)((int64_t)T * ld * 4), 0x00020000);
  unsigned kv_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) kv_off[i] = (unsigned)(((int64_t)(sr + RPP * i) * ld + h * HD + 4 * sc) * 4);
  const unsigned k_col = (unsigned)D * 4, v_col = (unsigned)D * 8, tile_bytes = (unsigned)(KT * ld * 4);
  f32x4 rk[NLD], rv[NLD];
  auto fetch = [&](int kt) {
    const unsigned so = (unsigned)kt * tile_bytes;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_off[i] + k_col, so, 0));
      rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_off[i] + v_col, so, 0));
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<f32x4*>(&Ks[buf][sr + RPP * i][4 * sc]) = rk[i];
      *reinterpret_cast<f32x4*>(&Vs[buf][sr + RPP * i][4 * sc]) = rv[i];
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (T + KT - 1) / KT;
  fetch(0);
  stash(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    if (wave_active) {
      // S^T = K_tile (A: rows = keys) x Q^T (B: cols = queries)
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const float* kp = &Ks[buf][ql][4 * h2];
      const float* vp = &Vs[buf][4 * h2][ql];
      float vr[PRELOAD ? 32 : 1];
      if constexpr (PRELOAD) {
        f32x4 kf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) kf[s] = *reinterpret_cast<const f32x4*>(kp + 8 * s);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int krow = (r & 3) + 8 * (r >> 2);
          vr[2 * r] = vp[krow * HD];
          vr[2 * r + 1] = vp[krow * HD + 32];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s][j], qf[s][j], sacc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * s);
#pragma unroll
          for (int j = 0; j < 4; ++j) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[s][j], sacc, 0, 0, 0);
        }
      }
      // lane (ql, h2) now holds S[q = q0+ql][key = kt*32 + (r&3) + 8*(r>>2) + 4*h2]
      const int kbase = kt * KT + 4 * h2;
      if (kbase + 28 + 3 >= T) {   // tile reaches past the last key: mask
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + (r & 3) + 8 * (r >> 2) >= T) sacc[r] = -INFINITY;
      }
      float mloc = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = FASTEXP ? __expf(m_run - m_new) : expf(m_run - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = FASTEXP ? __expf(sacc[r] - m_new) : expf(sacc[r] - m_new);
        lsum += sacc[r];
      }
      lsum += __shfl_xor(lsum, 32, 64);
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
      // O^T[d][q] += sum_key V[key][d] * P[q][key]:  A = V^T (rows = d), B = P^T (sacc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = (r & 3) + 8 * (r >> 2);
        const float v0 = PRELOAD ? vr[2 * r] : vp[krow * HD], v1 = PRELOAD ? vr[2 * r + 1] : vp[krow * HD + 32];
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sacc[r], oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sacc[r], oacc[1], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) stash(buf ^ 1);
    __syncthreads();
  }

  // oacc[db][r] = O[q0+ql][db*32 + (r&3) + 8*(r>>2) + 4*h2]  ->  float4 per (db, r>>2)
  if (wave_active && q0 + ql < T) {
    const float inv = 1.0f / l_run;
    const int64_t row = b * T + q0 + ql;
    float* op = out + row * (int64_t)D + h * HD + 4 * h2;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = oacc[db][4 * g + 0] * inv;
        v[1] = oacc[db][4 * g + 1] * inv;
        v[2] = oacc[db][4 * g + 2] * inv;
        v[3] = oacc[db][4 * g + 3] * inv;
        if constexpr (OUT3) {
          // 4 consecutive k = h*64 + db*32 + 8g + 4*h2 .. +3  ->  8 bytes in each plane
          const int k0 = h * HD + db * 32 + 8 * g + 4 * h2, e = k0 & 15;
          unsigned char* dst = out3 + (((int64_t)(k0 >> 4) * 3) * R3 + row) * 32 +
                               (((e >> 3) ^ (int)((row >> 3) & 1)) << 4) + (e & 7) * 2;
          unsigned t0[3], t1[3];
          split_pair_x3(v[0], v[1], t0);
          split_pair_x3(v[2], v[3], t1);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            uint2 o;
            o.x = t0[pl]; o.y = t1[pl];
            *reinterpret_cast<uint2*>(dst + pl * R3 * 32) = o;
          }
        } else {
          *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g) = v;
        }
      }
  }
}

// ---- split-bf16 ("x6") attention: both matrix products on v_mfma_f32_32x32x16_bf16 with fp32-level accuracy ----
// Same structure as attention_kernel (S^T = K Q^T, online softmax per lane, O^T += V^T P^T), but every operand is the
// exact sum of three bf16 planes and each product is the six leading plane products (see gemm_x6.hip):
//   * Q is split once per wave into registers; K and V are split ONCE PER WORKGROUP while a 32-key tile is staged
//     into LDS (K as [plane][key][64 d], V transposed as [plane][d][32 keys] so that the A fragment of V^T P^T is
//     two 8-byte reads); P = exp(S - m) is split in registers right after the softmax.
//   * 24 + 24 MFMAs of 8 passes per key tile instead of 32 + 32 of 16 passes: 1536 instead of 4096 matrix-core
//     cycles per wave and tile.
constexpr int KROW = 144;     // bytes per key row of one K plane (128 + 16: ds_read_b128 conflict-free)
constexpr int VROW = 72;      // bytes per d row of one V^T plane (64 + 8: ds_read_b64 conflict-free)

typedef __bf16 attn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned attn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));

// four floats -> the packed bf16 pairs of their three planes: pk[plane][0] = (x0, x1), pk[plane][1] = (x2, x3)
__device__ __forceinline__ void split4(const f32x4 v, unsigned pk[3][2]) {
  unsigned t0[3], t1[3];
  split_pair_x3(v[0], v[1], t0);
  split_pair_x3(v[2], v[3], t1);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) { pk[pl][0] = t0[pl]; pk[pl][1] = t1[pl]; }
}

#define ANYLOC_MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(attn_bf16x8, a), __builtin_bit_cast(attn_bf16x8, b), c, 0, 0, 0)

template <bool OUT3>
__global__ __launch_bounds__(256, 2) void attention_x6_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                              int D, float scale, unsigned char* __restrict__ out3,
                                                              int64_t R3) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][3][KT * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[2][3][HD * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t ld = 3 * (int64_t)D;
  const float* base = qkv + b * T * ld;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int ql = lane & 31, h2 = lane >> 5;
  const bool wave_active = q0 < T;

  // ---- Q fragments (B operand): lane (query ql, half h2), k-step s: d = 16 s + 8 h2 + j, three planes ----
  attn_u32x4 qf[3][4];
  {
    const int qr = min(q0 + ql, T - 1);
    const float* qp = base + (int64_t)qr * ld + h * HD + 8 * h2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unsigned lo[3][2], hi[3][2];
      split4(*reinterpret_cast<const f32x4*>(qp + 16 * s), lo);
      split4(*reinterpret_cast<const f32x4*>(qp + 16 * s + 4), hi);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        qf[pl][s][0] = lo[pl][0]; qf[pl][s][1] = lo[pl][1]; qf[pl][s][2] = hi[pl][0]; qf[pl][s][3] = hi[pl][1];
      }
    }
  }

  // ---- K/V staging: 16 lanes cover one 256-byte head row of a key; two passes of 16 keys ----
  const int sr = tid >> 4, sc = tid & 15;
  const __amdgpu_buffer_rsrc_t kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(base), 0, (int)((int64_t)T * ld * 4), 0x00020000);
  unsigned kv_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) kv_off[i] = (unsigned)(((int64_t)(sr + 16 * i) * ld + h * HD + 4 * sc) * 4);
  const unsigned k_col = (unsigned)D * 4, v_col = (unsigned)D * 8, tile_bytes = (unsigned)(KT * ld * 4);
  f32x4 rk[2], rv[2];
  auto fetch = [&](int kt) {
    const unsigned so = (unsigned)kt * tile_bytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_off[i] + k_col, so, 0));
      rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_off[i] + v_col, so, 0));
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = sr + 16 * i;
      unsigned pk[3][2];
      split4(rk[i], pk);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        attn_u32x2 o;
        o[0] = pk[pl][0]; o[1] = pk[pl][1];
        *reinterpret_cast<attn_u32x2*>(&Ks[buf][pl][key * KROW + sc * 8]) = o;
      }
      split4(rv[i], pk);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<unsigned short*>(&Vs[buf][pl][(4 * sc + j) * VROW + key * 2]) =
              (unsigned short)(pk[pl][j >> 1] >> (16 * (j & 1)));
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (T + KT - 1) / KT;
  fetch(0);
  stash(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    if (wave_active) {
      // S^T = K_tile (A: rows = keys) x Q^T (B: cols = queries), 4 k-steps of 16 d, six plane products each
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        attn_u32x4 kf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          kf[pl] = *reinterpret_cast<const attn_u32x4*>(&Ks[buf][pl][ql * KROW + s * 32 + h2 * 16]);
        sacc = ANYLOC_MFMA_BF16(kf[2], qf[0][s], sacc);
        sacc = ANYLOC_MFMA_BF16(kf[0], qf[2][s], sacc);
        sacc = ANYLOC_MFMA_BF16(kf[1], qf[1][s], sacc);
        sacc = ANYLOC_MFMA_BF16(kf[1], qf[0][s], sacc);
        sacc = ANYLOC_MFMA_BF16(kf[0], qf[1][s], sacc);
        sacc = ANYLOC_MFMA_BF16(kf[0], qf[0][s], sacc);
      }
      // V^T fragments (A operand of O^T += V^T P^T): lane (d = db*32 + ql, h2); k-step s2, element j <-> key
      // (j&3) + 8*(2*s2 + (j>>2)) + 4*h2 -- the keys register r = 8*s2 + j of the score block holds
      attn_u32x4 vf[3][2][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned char* vp = &Vs[buf][pl][(db * 32 + ql) * VROW + (16 * s2 + 4 * h2) * 2];
            const attn_u32x2 a0 = *reinterpret_cast<const attn_u32x2*>(vp);
            const attn_u32x2 a1 = *reinterpret_cast<const attn_u32x2*>(vp + 16);
            vf[pl][db][s2][0] = a0[0]; vf[pl][db][s2][1] = a0[1]; vf[pl][db][s2][2] = a1[0]; vf[pl][db][s2][3] = a1[1];
          }
      // lane (ql, h2) holds S[q = q0+ql][key = kt*32 + (r&3) + 8*(r>>2) + 4*h2], still unscaled
      const int kbase = kt * KT + 4 * h2;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] *= scale;                 // power of two: exact
      if (kbase + 28 + 3 >= T) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + (r & 3) + 8 * (r >> 2) >= T) sacc[r] = -INFINITY;
      }
      float mloc = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __expf(m_run - m_new);
      float lsum = 0.f;
      attn_u32x4 pf[3][2];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __expf(sacc[r] - m_new), p1 = __expf(sacc[r + 1] - m_new);
        lsum += p0;
        lsum += p1;
        unsigned t[3];
        split_pair_x3(p0, p1, t);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pf[pl][r >> 3][(r & 7) >> 1] = t[pl];
      }
      lsum += __shfl_xor(lsum, 32, 64);
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          oacc[db] = ANYLOC_MFMA_BF16(vf[2][db][s2], pf[0][s2], oacc[db]);
          oacc[db] = ANYLOC_MFMA_BF16(vf[0][db][s2], pf[2][s2], oacc[db]);
          oacc[db] = ANYLOC_MFMA_BF16(vf[1][db][s2], pf[1][s2], oacc[db]);
          oacc[db] = ANYLOC_MFMA_BF16(vf[1][db][s2], pf[0][s2], oacc[db]);
          oacc[db] = ANYLOC_MFMA_BF16(vf[0][db][s2], pf[1][s2], oacc[db]);
          oacc[db] = ANYLOC_MFMA_BF16(vf[0][db][s2], pf[0][s2], oacc[db]);
        }
    }
    if (kt + 1 < nkt) stash(buf ^ 1);
    __syncthreads();
  }

  // oacc[db][r] = O[q0+ql][db*32 + (r&3) + 8*(r>>2) + 4*h2]
  if (wave_active && q0 + ql < T) {
    const float inv = 1.0f / l_run;
    const int64_t row = b * T + q0 + ql;
    float* op = out + row * (int64_t)D + h * HD + 4 * h2;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = oacc[db][4 * g + 0] * inv;
        v[1] = oacc[db][4 * g + 1] * inv;
        v[2] = oacc[db][4 * g + 2] * inv;
        v[3] = oacc[db][4 * g + 3] * inv;
        if constexpr (OUT3) {
          const int k0 = h * HD + db * 32 + 8 * g + 4 * h2, e = k0 & 15;
          unsigned char* dst = out3 + (((int64_t)(k0 >> 4) * 3) * R3 + row) * 32 +
                               (((e >> 3) ^ (int)((row >> 3) & 1)) << 4) + (e & 7) * 2;
          unsigned pk[3][2];
          split4(v, pk);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            uint2 o;
            o.x = pk[pl][0]; o.y = pk[pl][1];
            *reinterpret_cast<uint2*>(dst + pl * R3 * 32) = o;
          }
        } else {
          *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g) = v;
        }
      }
  }
}


// ---- "h3" attention: both matrix products as THREE fp16 matrix-core products of two-term fp16 operands ----
// The arithmetic of gemm_h3.hip applied to softmax((q/8) k^T) v.  Q, K and V arrive already quantised: the QKV GEMM's
// epilogue (EPI_QKV_PLANES) wrote them as per-(head, 32-row group) tiles of two fp16 planes with one power-of-two scale
// per tile, in exactly the images this kernel wants in LDS (layout: common.hpp) -- so a key tile is staged by four
// 1-KiB global -> LDS DMA instructions per wave, with no conversion, no VGPR round trip and no transposing store, and
// nothing is re-quantised per workgroup.  Structure as attention_kernel: S^T = K Q^T per 32-key tile, online softmax
// per lane, O^T += V^T P^T; P = exp2(.) * 2^14 is split into fp16 hi + lo in registers.  Key tiles are the GLOBAL
// 32-row groups that intersect the image (first / last tile masked), the same groups the producer scaled.
// The output goes straight into the h2 image of the projection GEMM: every row of an image is scaled by the same power
// of two, 1 / max(V tile scales of the image) -- an attention output is a convex combination of V rows, so
// |o| <= max |V| < 2^15 in scaled units -- hence no fp32 round trip and no separate quantiser pass.
typedef _Float16 ah_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ah_f16x2 __attribute__((ext_vector_type(2)));
#define ANYLOC_MFMA_F16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ah_f16x8, a), __builtin_bit_cast(ah_f16x8, b), c, 0, 0, 0)

__device__ __forceinline__ void ah_pack2(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 pr;
  pr[0] = a; pr[1] = b;
  const ah_f16x2 h = __builtin_convertvector(pr, ah_f16x2);
  f32x2 res;
  res[0] = pr[0] - (float)h[0];
  res[1] = pr[1] - (float)h[1];
  const ah_f16x2 l = __builtin_convertvector(res, ah_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float ah_pow2_recip(float inv) { return __uint_as_float((254u << 23) - __float_as_uint(inv)); }


constexpr int AH_STAGE = 16384;       // K hi | K lo | V hi | V lo, 4 KiB each

// QG = 32-query groups per wave (1 or 2), NW = waves per workgroup, KS = key splits: the waves of a workgroup are NW / KS
// query waves x KS key waves; a workgroup covers QG * NW / KS consecutive 32-query groups of one (image, head).
// QG = 2: a wave's K / V fragments serve 64 queries, i.e. half the LDS reads, DMA issues and barriers per unit of work.
// KS = 2 (few images per call): every step stages KS consecutive key tiles, key wave kw takes tile KS * step + kw, and the
// key waves' partial (O, m, l) meet in LDS at the end in split order -- twice the workgroups and half the serial chain of
// key tiles per wave: one ViT-g image is 24 heads x 5 workgroups x 17 tiles with KS = 1, 24 x 9 x 9 with KS = 2.
template <int QG, int NW, int KS = 1>
__global__ __launch_bounds__(64 * NW, 2) void attention_h3_kernel(const unsigned char* __restrict__ planes,
                                                                  const float* __restrict__ inv, int T, int heads, int64_t G,
                                                                  unsigned char* __restrict__ out2, float* __restrict__ out_inv,
                                                                  int64_t R, int QB) {
  static_assert((QG == 1 || QG == 2) && (KS == 1 || KS == 2) && NW % KS == 0 && 16 * KS % NW == 0, "unsupported shape");
  constexpr int NT = 64 * NW;
  constexpr int QW = NW / KS;           // query waves
  constexpr int PIECES = 16 * KS / NW;  // 1-KiB DMA pieces per wave and step
  extern __shared__ __attribute__((aligned(16))) unsigned char ah_smem[];   // 2 x KS stages + 16 bytes for the block reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware: the QB workgroups of one (image, head) -- which stream the same K / V tiles, 271 KB at T = 530 -- get
  // consecutive ids on ONE XCD, so the tiles come from HBM / the fabric into that XCD's L2 once instead of once per XCD
  const int logical = xcd_contiguous_id((int)blockIdx.x, (int)gridDim.x);
  const int qb = logical % QB, h = (logical / QB) % heads;
  const int64_t b = logical / (QB * heads);
  const int64_t r0 = b * T, r1 = r0 + T;
  const int64_t g_first = r0 >> 5, g_last = (r1 - 1) >> 5;
  const int ng = (int)(g_last - g_first + 1);
  const int qw = KS == 1 ? wave : wave % QW, kw = KS == 1 ? 0 : wave / QW;   // (KS = 1 compiles to the unsplit kernel)
  const int64_t gq0 = g_first + (qb * QW + qw) * QG;
  const bool wave_active = gq0 <= g_last;
  const int ql = lane & 31, h2 = lane >> 5;
  const int64_t tile_bytes = 8192;

  // ---- Q fragments (B operand): lane (query ql, half h2), k-step s: d = 16 s + 8 h2 + j = chunk 2 s + h2 of row ql ----
  attn_u32x4 qf[QG][2][4];
  float fq[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int64_t gq = min(gq0 + qg, g_last);
    const int64_t q_tile = ((int64_t)h) * G + gq;
    fq[qg] = inv[q_tile];
    const unsigned char* qb = planes + q_tile * tile_bytes + ql * 128;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        qf[qg][pl][s] = *reinterpret_cast<const attn_u32x4*>(qb + pl * 4096 + (((2 * s + h2) ^ ((ql >> 1) & 7)) << 4));
  }

  // ---- K / V staging: pure DMA of the four 4-KiB plane tiles [K hi | K lo | V hi | V lo], 1 KiB per wave instruction ----
  const int64_t part_bytes = (int64_t)heads * G * tile_bytes;
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(planes + part_bytes), 0, (int)part_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(planes + 2 * part_bytes), 0, (int)part_bytes, 0x00020000);
  auto issue = [&](int step, int stage) {                 // the KS key tiles of a step, 16 pieces each
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int sub = (i * NW) / 16;                       // which tile of the step (compile-time: wave < NW <= 16)
      const int piece = (i * NW) % 16 + wave;              // 0..7: K planes, 8..15: V planes
      const int t = step * KS + sub;
      if (KS == 1 || t < ng) {                             // wave-uniform (KS = 1: the caller only asks for tiles that exist)
        const unsigned soff = (unsigned)((((int64_t)h) * G + g_first + t) * tile_bytes);
        unsigned char* st = ah_smem + (stage * KS + sub) * AH_STAGE + piece * 1024;
        const unsigned voff = (unsigned)((piece & 7) * 1024 + lane * 16);
        if ((i * NW) % 16 < 8) dma16_to_lds(k_rsrc, st, voff, soff);
        else dma16_to_lds(v_rsrc, st, voff, soff);
      }
    }
  };

  issue(0, 0);                          // the first step's tiles: in flight under everything the prologue still loads

  // per-tile K / V scales: lane t keeps those of key tile t (ng <= 64 for T <= 1984; longer images fall back to loads)
  float fk_lane = 1.0f, fv_lane = 1.0f;
  if (lane < ng) {
    fk_lane = inv[((int64_t)heads + h) * G + g_first + lane];
    fv_lane = inv[((int64_t)2 * heads + h) * G + g_first + lane];
  }

  f32x16 oacc[QG][2];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[qg][0][r] = 0.f; oacc[qg][1][r] = 0.f; }
    m_run[qg] = -INFINITY;
    l_run[qg] = 0.f;
  }
  float fv_run = 1.0f;

  // ---- scale of this image's output rows: the largest V-tile scale over all heads and key groups (used by the epilogue) ----
  // its loads travel with the first tile's DMA and the Q fragments; the block maximum is published by the same barrier as stage 0
  float fm = 0.f;
  for (int i = tid; i < heads * ng; i += NT) {
    const int hh = i / ng, gg = i - hh * ng;
    fm = fmaxf(fm, inv[((int64_t)2 * heads + hh) * G + g_first + gg]);
  }
  fm = wave_max(fm);
  float* red = reinterpret_cast<float*>(ah_smem + 2 * KS * AH_STAGE);
  if (lane == 0) red[wave] = fm;
  __syncthreads();                      // drains the DMA counter (fence) and publishes stage 0 (and red[])
  fm = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) fm = fmaxf(fm, red[w]);

  // fragment addresses: lane-dependent part once (the swizzled 16-byte slot of each k-step), the stage / plane / d-block part
  // as the ds_read's immediate offset -- the tile body is instantiated per stage so that the stage is a compile-time constant
  unsigned k_lane[4], v_lane[2];
#pragma unroll
  for (int s = 0; s < 4; ++s) k_lane[s] = (unsigned)(kw * AH_STAGE + ql * 128 + (((2 * s + h2) ^ ((ql >> 1) & 7)) << 4));
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) v_lane[s2] = (unsigned)(kw * AH_STAGE + ql * 64 + (((h2 * 2 + s2) ^ ((ql >> 2) & 3)) << 4));

  const int nsteps = (ng + KS - 1) / KS;
  bool first = true;                    // this wave has not processed a tile yet
  auto tile = [&](const int step, auto stagec) {
    constexpr int stage = decltype(stagec)::value;
    if (step + 1 < nsteps) issue(step + 1, stage ^ 1);
    const int t = step * KS + kw;
    if (wave_active && (KS == 1 || t < ng)) {
      const int64_t gk = g_first + t;
      const unsigned char* Ks = ah_smem + stage * KS * AH_STAGE;
      const unsigned char* Vs = Ks + 8192;
      float fk, fv;
      if (ng <= 64) {                   // wave-uniform lane index: v_readlane, no memory access in the loop
        fk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fk_lane), t));
        fv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fv_lane), t));
      } else {
        fk = inv[((int64_t)heads + h) * G + gk];
        fv = inv[((int64_t)2 * heads + h) * G + gk];
      }
      // S^T = K_tile (A: rows = keys) x Q^T (B: cols = queries): 4 k-steps of 16 d, lo*hi + hi*lo + hi*hi
      f32x16 sacc[QG];
#pragma unroll
      for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qg][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        attn_u32x4 kf[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          kf[pl] = *reinterpret_cast<const attn_u32x4*>(Ks + pl * 4096 + k_lane[s]);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
          sacc[qg] = ANYLOC_MFMA_F16(kf[1], qf[qg][0][s], sacc[qg]);
          sacc[qg] = ANYLOC_MFMA_F16(kf[0], qf[qg][1][s], sacc[qg]);
          sacc[qg] = ANYLOC_MFMA_F16(kf[0], qf[qg][0][s], sacc[qg]);
        }
      }
      // V^T fragments: lane (d = db*32 + ql, half h2), k-step s2: the 8 keys register r = 8 s2 + j of the score block holds
      attn_u32x4 vf[2][2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)                    // d = 32 db + ql: (d >> 2) & 3 does not depend on db
            vf[pl][db][s2] = *reinterpret_cast<const attn_u32x4*>(Vs + pl * 4096 + db * 2048 + v_lane[s2]);
        }
      const bool edge = gk == g_first || gk == g_last;      // wave-uniform: only the image's first / last key group
      // O is kept in units of the current V tile's scale: moving to a tile with another scale is a power-of-two factor
      const float vratio = (KS == 1 ? t == 0 : first) ? 1.0f : fv_run * ah_pow2_recip(fv);
      fv_run = fv;
      if (KS > 1) first = false;
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) {
        // scores in the exp2 domain: t = S_true * log2(e) = sacc * c, c = fq * fk * log2(e) / 8 > 0 -- the maximum is
        // taken on the raw accumulators (a positive scale commutes with max), the scale rides in the exponent's FMA
        const float c = fq[qg] * fk * (0.125f * 1.44269504088896340736f);
        if (edge) {
          asm volatile("" ::: "memory");                    // (keeps the compiler from if-converting the block into selects)
          const int klo = (int)(r0 - gk * 32) - 4 * h2, khi = (int)(r1 - gk * 32) - 4 * h2;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ko = (r & 3) + 8 * (r >> 2);
            if (ko < klo || ko >= khi) sacc[qg][r] = -INFINITY;
          }
        }
        float mloc = sacc[qg][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[qg][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * c;
        const float m_new = fmaxf(m_run[qg], mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qg] - m_new);
        const float moff = 14.0f - m_new;                  // P * 2^14: hi + lo in fp16, the factor cancels against l
        // (measured and dropped, profiles/r03_attn_experiments.log: the same arithmetic on v_pk_fma_f32 / v_pk_add_f32 and the
        // hi / lo split on v_fma_mixlo_f16 / v_fma_mixhi_f16 -- 97 instead of 130 vector instructions per tile, same bits for
        // P -- ran 3 % SLOWER: this loop is not short of vector issue slots)
        float lsum = 0.f;
        attn_u32x4 pf[2][2];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qg][r], c, moff));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qg][r + 1], c, moff));
          lsum += p0;
          lsum += p1;
          unsigned hi, lo;
          ah_pack2(p0, p1, hi, lo);
          pf[0][r >> 3][(r & 7) >> 1] = hi;
          pf[1][r >> 3][(r & 7) >> 1] = lo;
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run[qg] = l_run[qg] * alpha + lsum;
        m_run[qg] = m_new;
        const float resc = alpha * vratio;
        if (!__all(resc == 1.0f)) {                         // rare after the first tiles: keep it a real branch
          asm volatile("" ::: "memory");
#pragma unroll
          for (int r = 0; r < 16; ++r) { oacc[qg][0][r] *= resc; oacc[qg][1][r] *= resc; }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            oacc[qg][db] = ANYLOC_MFMA_F16(vf[1][db][s2], pf[0][s2], oacc[qg][db]);
            oacc[qg][db] = ANYLOC_MFMA_F16(vf[0][db][s2], pf[1][s2], oacc[qg][db]);
            oacc[qg][db] = ANYLOC_MFMA_F16(vf[0][db][s2], pf[0][s2], oacc[qg][db]);
          }
      }
    }
    __syncthreads();                    // everyone is done with this stage; the next tile's DMA has landed
  };
  for (int s = 0; s < nsteps; s += 2) {
    tile(s, std::integral_constant<int, 0>{});
    if (s + 1 < nsteps) tile(s + 1, std::integral_constant<int, 1>{});
  }

  if constexpr (KS > 1) {
    // the key waves' partial sums meet in LDS (the stages are free after the loop's last barrier), in units that do not
    // depend on a wave's own tiles: O * fv_run (a power of two), m and l as they are; key wave 0 adds them in split order
    float* mb = reinterpret_cast<float*>(ah_smem);          // [(KS - 1) * QW * QG][34][64] floats: 17 KiB at KS = 2, QW = 2
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[qg][0][r] *= fv_run; oacc[qg][1][r] *= fv_run; }
    fv_run = 1.0f;
    if (kw > 0) {
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) {
        float* dst = mb + (((kw - 1) * QW + qw) * QG + qg) * 34 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dst[r * 64] = oacc[qg][0][r]; dst[(16 + r) * 64] = oacc[qg][1][r]; }
        dst[32 * 64] = m_run[qg];
        dst[33 * 64] = l_run[qg];
      }
    }
    __syncthreads();
    if (kw > 0) return;
#pragma unroll
    for (int k = 1; k < KS; ++k)
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) {
        const float* src = mb + (((k - 1) * QW + qw) * QG + qg) * 34 * 64 + lane;
        const float m_o = src[32 * 64], l_o = src[33 * 64];
        const float m_new = fmaxf(m_run[qg], m_o);            // key wave 0 always owns tile 0: m_run is finite
        const float a0 = __builtin_amdgcn_exp2f(m_run[qg] - m_new), a1 = __builtin_amdgcn_exp2f(m_o - m_new);   // exp2(-inf) = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          oacc[qg][0][r] = oacc[qg][0][r] * a0 + src[r * 64] * a1;
          oacc[qg][1][r] = oacc[qg][1][r] * a0 + src[(16 + r) * 64] * a1;
        }
        l_run[qg] = l_run[qg] * a0 + l_o * a1;
        m_run[qg] = m_new;
      }
  }

  // oacc[db][r] = O[q][db*32 + (r&3) + 8*(r>>2) + 4*h2] * l_run / fv_run (in P * 2^14 units, which cancel against l_run)
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int64_t row = (gq0 + qg) * 32 + ql;
    if (wave_active && gq0 + qg <= g_last && row >= r0 && row < r1) {
      const float f = fv_run * ah_pow2_recip(fm) / l_run[qg];        // -> value * 2^e_img, |.| < 2^15
      if (h == 0 && h2 == 0) out_inv[row] = fm;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k0 = h * HD + db * 32 + 8 * g + 4 * h2, e = k0 & 15;
          unsigned char* dst = out2 + (((int64_t)(k0 >> 4) * 2) * R + row) * 32 + (((e >> 3) ^ (int)((row >> 3) & 1)) << 4) + (e & 7) * 2;
          uint2 ph, plo;
          ah_pack2(oacc[qg][db][4 * g + 0] * f, oacc[qg][db][4 * g + 1] * f, ph.x, plo.x);
          ah_pack2(oacc[qg][db][4 * g + 2] * f, oacc[qg][db][4 * g + 3] * f, ph.y, plo.y);
          *reinterpret_cast<uint2*>(dst) = ph;
          *reinterpret_cast<uint2*>(dst + R * 32) = plo;
        }
    }
  }
}

// fp32 [rows, 3D] (q | k | v, heads contiguous) -> the tiles of common.hpp (what gemm_h3's EPI_QKV_PLANES writes).
// One block per (32-row group, head, part); used by the kernel tests and by callers that hold fp32 projections.
__global__ __launch_bounds__(256) void qkv_planes_kernel(const float* __restrict__ qkv, int64_t rows, int D, int heads, int64_t G,
                                                         unsigned char* __restrict__ planes, float* __restrict__ inv) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g = blockIdx.x;
  const int h = blockIdx.y, part = blockIdx.z;
  const int r = tid >> 3, c8 = (tid & 7) * 8;
  const int64_t row = g * 32 + r;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = row < rows ? qkv[row * 3 * (int64_t)D + (int64_t)part * D + h * 64 + c8 + j] : 0.f;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  amax = wave_max(amax);
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
  const int e = ex == 0 ? 100 : max(-100, min(100, 14 - (ex - 127)));     // as h2_row_scale (gemm_h3.hip)
  const float scale = __uint_as_float((unsigned)(127 + e) << 23);
  const int64_t tile = ((int64_t)part * heads + h) * G + g;
  if (tid == 0) inv[tile] = __uint_as_float((unsigned)(127 - e) << 23);
  unsigned char* dst = planes + tile * 8192;
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ah_pack2(v[2 * j] * scale, v[2 * j + 1] * scale, hi[j], lo[j]);
  if (part < 2) {
    unsigned char* o = dst + r * 128 + (((c8 >> 3) ^ ((r >> 1) & 7)) << 4);
    attn_u32x4 a, bq;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = hi[j]; bq[j] = lo[j]; }
    *reinterpret_cast<attn_u32x4*>(o) = a;
    *reinterpret_cast<attn_u32x4*>(o + 4096) = bq;
  } else {
    // row r = (j & 3) + 8 * (2 s2 + (j >> 2)) + 4 hh  ->  hh = (r >> 2) & 1, s2 = r >> 4, j = (r & 3) + 4 * ((r >> 3) & 1)
    const int hh = (r >> 2) & 1, s2 = r >> 4, j = (r & 3) + 4 * ((r >> 3) & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = c8 + i;
      unsigned char* o = dst + d * 64 + (((hh * 2 + s2) ^ ((d >> 2) & 3)) << 4) + j * 2;
      *reinterpret_cast<unsigned short*>(o) = (unsigned short)(hi[i >> 1] >> (16 * (i & 1)));
      *reinterpret_cast<unsigned short*>(o + 4096) = (unsigned short)(lo[i >> 1] >> (16 * (i & 1)));
    }
  }
}

}  // namespace

int qkv_planes_from_f32(const float* qkv, int64_t rows, int D, int heads, unsigned char* planes, float* inv, hipStream_t stream) {
  ANYLOC_CHECK_ARG(qkv && planes && inv && rows > 0 && D == heads * HD, "qkv_planes: bad arguments");
  const int64_t G = (rows + 31) / 32;
  ANYLOC_CHECK_ARG(G < (1ll << 31), "qkv_planes: too many rows");
  hipLaunchKernelGGL(qkv_planes_kernel, dim3((unsigned)G, heads, 3), dim3(256), 0, stream, qkv, rows, D, heads, G, planes, inv);
  return launch_status("qkv_planes_kernel");
}

int attention_h3(const unsigned char* planes, const float* inv, int64_t batch, int T, int D, int heads, unsigned char* out2,
                 float* out_inv, hipStream_t stream) {
  ANYLOC_CHECK_ARG(planes && inv && out2 && out_inv, "attention_h3: null pointer");
  ANYLOC_CHECK_ARG(D == heads * HD, "attention_h3: head_dim must be 64 (D=%d heads=%d)", D, heads);
  ANYLOC_CHECK_ARG(T > 0 && batch > 0 && batch < 65536, "attention_h3: bad T/batch");
  const int64_t R = batch * T, G = (R + 31) / 32;
  ANYLOC_CHECK_ARG((int64_t)heads * G * 8192 < (1ll << 31), "attention_h3: operand tiles exceed the 2 GiB buffer-addressing range");
  const double flops = 4.0 * (double)batch * heads * (double)T * T * HD;
  ProfScope prof("attention", stream, flops, 8.0 * batch * T * D * 2);
  const int qgroups = (T + 31) / 32 + 1;                    // an image intersects at most this many 32-row groups
  const size_t lds = 2 * AH_STAGE + 64;
  // four waves of 32 queries per workgroup.  Measured: two waves of 64 queries (13.9 against 13.1 ms per step at B = 61,
  // 34 against 25 us per launch at B = 1: kept as option attn_h3_qg = 2, profiles/r05_attention_qg2.log); measured and
  // removed: a software-pipelined loop with a 3-stage ring (15.0 ms), and (round 3) all K fragments of a tile read first with
  // the twelve score MFMAs issued back to back (13.2 vs 13.2 ms) and the V fragments read per 32-column half (126 instead of
  // 152 VGPRs: four waves per SIMD; 13.4 vs 13.5 ms): neither the dependent-chain gaps nor the occupancy is what holds this
  // kernel at 40 % matrix-core utilisation -- DESIGN.md 4.2b
  // few images per call: two query waves x two key waves per workgroup (KS = 2) -- twice the workgroups, half the serial
  // chain of key tiles per wave -- as long as all of them are resident at once (two per CU: 64 KiB of LDS each);
  // option attn_h3_ks: 0 = this rule, 1 / 2 force.  One ViT-g image (322 x 322): 27.7 -> 20.8 us per launch, two images 31.8 -> 26.3;
  // three and more (648+ workgroups) lose 3 - 12 us, as does the same split with 64-query waves (profiles/r05_attention_ks.log)
  const int QB2 = (qgroups + 1) / 2;
  const int64_t ks_opt = option(OPT_ATTN_H3_KS);
  if (ks_opt == 2 || (ks_opt == 0 && (int64_t)QB2 * heads * batch <= 512)) {
    const dim3 grid2((unsigned)((int64_t)QB2 * heads * batch));
    const size_t lds2 = 4 * AH_STAGE + 64;
    static DynLds dyn_lds_once;                          // (per device: a process may drive several GPUs)
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(attention_h3_kernel<1, 4, 2>), (int)lds2));
    hipLaunchKernelGGL((attention_h3_kernel<1, 4, 2>), grid2, dim3(256), lds2, stream, planes, inv, T, heads, G, out2, out_inv, R, QB2);
    return launch_status("attention_h3_kernel<1,4,2>");
  }
  const int QB = (qgroups + 3) / 4;                         // workgroups (of four 32-query groups) per image and head
  ANYLOC_CHECK_ARG((int64_t)QB * heads * batch < (1ll << 31), "attention_h3: grid too large");
  const dim3 grid((unsigned)((int64_t)QB * heads * batch));
  // option attn_h3_qg = 2 (A/B): two waves of 64 queries per workgroup -- a wave's K / V fragments serve 64 queries (half the
  // LDS reads, DMA issues and barriers per unit of work) at one wave per SIMD and workgroup
  if (option(OPT_ATTN_H3_QG) == 2) {
    hipLaunchKernelGGL((attention_h3_kernel<2, 2>), grid, dim3(128), lds, stream, planes, inv, T, heads, G, out2, out_inv, R, QB);
    return launch_status("attention_h3_kernel<2,2>");
  }
  hipLaunchKernelGGL((attention_h3_kernel<1, 4>), grid, dim3(256), lds, stream, planes, inv, T, heads, G, out2, out_inv, R, QB);
  return launch_status("attention_h3_kernel");
}

// qkv [B*T, 3D] (q | k | v, each head-major 64-wide), out [B*T, D]
int attention(const float* qkv, float* out, int64_t batch, int T, int D, int heads, hipStream_t stream,
              unsigned char* out3, bool x6) {
  ANYLOC_CHECK_ARG(D == heads * HD, "attention: head_dim must be 64 (D=%d heads=%d)", D, heads);
  ANYLOC_CHECK_ARG(T > 0 && batch > 0 && batch < 65536, "attention: bad T/batch");
  const double flops = 4.0 * (double)batch * heads * (double)T * T * HD;
  ProfScope prof("attention", stream, flops, 16.0 * batch * T * D);
  // option attn_cfg (micro-benchmarks): 0 = default (fast exp + operand preload), 1 = neither, 2 = fast exp, 3 = preload
  const int cfg = (int)option(OPT_ATTN_CFG);
  const dim3 g4((T + 127) / 128, heads, (unsigned)batch), g2((T + 63) / 64, heads, (unsigned)batch);
  (void)g2;
  const int64_t R3 = batch * T;
  // option attn_x6: 1 = split-bf16 kernel for every call (kernel tests), 0 = never, -1 (default) = when the caller asks
  if (option(OPT_ATTN_X6) >= 0) x6 = option(OPT_ATTN_X6) != 0;
  if (x6) {
    if (out3) hipLaunchKernelGGL((attention_x6_kernel<true>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3);
    else hipLaunchKernelGGL((attention_x6_kernel<false>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3);
    return launch_status("attention_x6_kernel");
  }
  if (out3) {
    hipLaunchKernelGGL((attention_kernel<4, true, true, true>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3);
    return launch_status("attention_kernel");
  }
  switch (cfg) {
    case 1: hipLaunchKernelGGL((attention_kernel<4, false, false>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3); break;
    case 2: hipLaunchKernelGGL((attention_kernel<4, true, false>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3); break;
    case 3: hipLaunchKernelGGL((attention_kernel<4, false, true>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3); break;
    default: hipLaunchKernelGGL((attention_kernel<4, true, true>), g4, dim3(256), 0, stream, qkv, out, T, D, 0.125f, out3, R3); break;
  }
  return launch_status("attention_kernel");
}

}  // namespace anyloc
