// fp32 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
//
// Every dense contraction of the hot path runs on this kernel: the ViT's
// patch-embed / QKV / attention-out / MLP projections (reference: the hub
// model forward triggered at utilities.py:269), the VLAD and k-means cosine
// scores (fast-pytorch-kmeans max_sim reached from utilities.py:786,:849) and
// the retrieval inner products (faiss IndexFlatIP.search, utilities.py:450).
//
// Design (CDNA4, wave64):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain) at 64
//     FLOP/clk/SIMD = 157.3 TFLOP/s chip peak -- the governing roofline.
//   * block tile BM x BN x BK, WM x WN waves; each wave owns (BM/WM) x (BN/WN)
//     as 32x32 MFMA blocks held in accumulator registers.
//   * both operands are K-contiguous (torch Linear layout).  Global -> register
//     -> LDS staging with coalesced 128-byte row segments (8 lanes x float4),
//     issued as buffer loads (descriptor at the tile origin in SGPRs, constant
//     per-thread byte offset, K-slab offset as the scalar offset: no per-slab
//     64-bit address arithmetic -- worth +10 % over flat global loads here);
//     loads of K-slab t+1 are issued before the MFMAs of slab t and written to
//     the other LDS buffer afterwards (one barrier per slab).
//   * LDS rows are padded to BK+4 floats: the per-lane ds_read_b128 fragment
//     reads (row = lane&31, 16-byte column = lane>>5) hit 16 distinct 16-byte
//     slots per 16-lane group -> conflict-free.
//   * fragments of k-step s+1 are read from LDS into a second register set
//     before the MFMAs of step s are issued (LDS latency hidden behind 8-32
//     MFMAs of 64 cycles each).
//   * k-permutation: a lane's float4 holds k = 8s + 4*(lane>>5) + j, and MFMA
//     number j consumes element j of both operands, so A and W see the same k
//     in the same lane half; every k is consumed exactly once.
//   * block id -> tile: XCD-aware (block b runs on XCD b % 8, each XCD has a
//     private 4 MiB L2) + grouped ordering so co-resident blocks of one XCD
//     share A row-panels and W column-panels.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned voff, unsigned soff) {
  dma16_to_lds(rsrc, reinterpret_cast<unsigned char*>(lds_dst), voff, soff);
}

__device__ __forceinline__ float gelu_erf(float v) {
  return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
}
__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

// KFULL: K is a multiple of BK (no tail predicate in the staging loads)
// ABL: timing-only ablations for tools/microbench_gemm.py (results are WRONG when != 0):
//      bit 0 = no global->LDS staging inside the K loop, bit 1 = no barrier, bit 2 = s_setprio around MFMAs
template <int BM, int BN, int WM, int WN, int BK, int OCC, int EPI, bool ROWSQ, bool KFULL, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_nt_kernel(GemmProblem p, int tiles_m, int tiles_n) {
  constexpr int NT = 64 * WM * WN;
  // DMA (ABL bit 6): operands go global -> LDS directly (buffer_load ... lds), no VGPR round trip and
  // no ds_write pass.  The LDS image is lane-linear (1 KiB per wave instruction = 8 rows of 128 B), so
  // instead of padding rows the 16-byte chunks of a row are XOR-swizzled with ((row >> 1) & 7): applied
  // to the per-lane SOURCE address when staging and to the column when reading fragments.
  constexpr bool DMA = (ABL & 64) != 0;
  static_assert(!DMA || (KFULL && !ROWSQ && BK == 32 && NT == 256), "DMA staging: K % 32 == 0, 4 waves, no row norms");
  constexpr int LDS_LD = DMA ? BK : BK + 4;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int LPR = BK / 4;            // lanes per staged row (float4 each)
  constexpr int RPP = NT / LPR;          // rows staged per pass
  constexpr int A_LD4 = BM / RPP;        // float4 per thread per K-slab
  constexpr int W_LD4 = BN / RPP;
  constexpr int KS = BK / 8;             // k-steps per slab (8 k per step: 4 MFMAs per block pair)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  static_assert(EPI != EPI_SWIGLU || NI == 2, "swiglu pairs the two 32-col blocks of a wave");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDS_LD]
  float* Ws = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  if (p.ksplit > 1) {                     // split-K: this block contracts K-slice blockIdx.y into its own output slab
    const int64_t sl = blockIdx.y;
    p.A += sl * p.K;
    p.W += sl * p.K;
    p.C += sl * p.c_split_stride;
    if (ROWSQ) p.rowsq += sl * p.M;
  }
  int tile_m, tile_n;
  xcd_grouped_tile(blockIdx.x, tiles_m, tiles_n, 8, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;

  // ---- staging coordinates: LPR lanes cover one BK*4-byte row segment ----
  const int kq = tid % LPR, r0 = tid / LPR;
  // Operands are read with buffer loads: wave-uniform descriptor at the tile origin (SGPRs), a
  // per-thread 32-bit byte offset that never changes (VGPR) and the K-slab offset as the scalar
  // offset -- no per-slab 64-bit address arithmetic, nothing for the waitcnt pass to trip over.
  unsigned a_off[A_LD4], w_off[W_LD4];
#pragma unroll
  for (int i = 0; i < A_LD4; ++i) {
    int64_t row = m0 + r0 + RPP * i;
    row = (row < p.M ? row : p.M - 1) - m0;
    a_off[i] = (unsigned)((row * p.lda + 4 * kq) * 4);
  }
#pragma unroll
  for (int i = 0; i < W_LD4; ++i) {
    int64_t row = n0 + r0 + RPP * i;
    row = (row < p.N ? row : p.N - 1) - n0;
    w_off[i] = (unsigned)((row * p.ldw + 4 * kq) * 4);
  }
  if constexpr (DMA) {
    // instruction ii = 4*i + wave stages tile rows 8*ii .. 8*ii+7; lane -> (row 8*ii + lane/8, chunk lane%8)
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
      const int trow = 8 * (4 * i + wave) + (lane >> 3);
      const int chunk = (lane & 7) ^ ((trow >> 1) & 7);
      int64_t row = m0 + trow;
      row = (row < p.M ? row : p.M - 1) - m0;
      a_off[i] = (unsigned)((row * p.lda + 4 * chunk) * 4);
    }
#pragma unroll
    for (int i = 0; i < W_LD4; ++i) {
      const int trow = 8 * (4 * i + wave) + (lane >> 3);
      const int chunk = (lane & 7) ^ ((trow >> 1) & 7);
      int64_t row = n0 + trow;
      row = (row < p.N ? row : p.N - 1) - n0;
      w_off[i] = (unsigned)((row * p.ldw + 4 * chunk) * 4);
    }
  }
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W + n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  const int st_off = r0 * LDS_LD + 4 * kq;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  float rsq[A_LD4];
#pragma unroll
  for (int i = 0; i < A_LD4; ++i) rsq[i] = 0.0f;

  const int nk = (int)((p.K + BK - 1) / BK);
  // two staging register sets: slab j travels in set j&1.  With PF2 (ABL bit 3) the loads of slab
  // t+2 are issued while slab t is computed (slab t+1 is still in flight / in registers), which
  // doubles the tolerated global-load latency; otherwise slab t+1 is fetched during slab t.
  constexpr bool PF2 = (ABL & 8) != 0;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  f32x4 ra[2][A_LD4], rw[2][W_LD4];
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

  auto fetch = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const unsigned kb = (unsigned)kt * (BK * 4);          // byte offset of the slab (scalar)
    if constexpr (DMA) {
#pragma unroll
      for (int i = 0; i < A_LD4; ++i) dma16(a_rsrc, As + S * BM * LDS_LD + (4 * i + wave) * 256, a_off[i], kb);
#pragma unroll
      for (int i = 0; i < W_LD4; ++i) dma16(w_rsrc, Ws + S * BN * LDS_LD + (4 * i + wave) * 256, w_off[i], kb);
    } else if constexpr (KFULL) {
#pragma unroll
      for (int i = 0; i < A_LD4; ++i) ra[S][i] = buf_load16(a_rsrc, a_off[i], kb);
#pragma unroll
      for (int i = 0; i < W_LD4; ++i) rw[S][i] = buf_load16(w_rsrc, w_off[i], kb);
    } else {
      const bool ok = ((int64_t)kt * BK + 4 * kq) < p.K;   // K % 4 == 0: a float4 is all-in or all-out
#pragma unroll
      for (int i = 0; i < A_LD4; ++i) ra[S][i] = ok ? buf_load16(a_rsrc, a_off[i], kb) : zero4;
#pragma unroll
      for (int i = 0; i < W_LD4; ++i) rw[S][i] = ok ? buf_load16(w_rsrc, w_off[i], kb) : zero4;
    }
  };
  auto stash = [&](auto setc) {           // set S holds the slab whose LDS buffer is S
    constexpr int S = decltype(setc)::value;
    if constexpr (DMA) return;
    float* ad = As + S * BM * LDS_LD + st_off;
    float* wd = Ws + S * BN * LDS_LD + st_off;
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
      *reinterpret_cast<f32x4*>(ad + RPP * i * LDS_LD) = ra[S][i];
      if (ROWSQ)
        rsq[i] += ra[S][i][0] * ra[S][i][0] + ra[S][i][1] * ra[S][i][1] + ra[S][i][2] * ra[S][i][2] +
                  ra[S][i][3] * ra[S][i][3];
    }
#pragma unroll
    for (int i = 0; i < W_LD4; ++i) *reinterpret_cast<f32x4*>(wd + RPP * i * LDS_LD) = rw[S][i];
  };

  fetch(0, I0{});
  stash(I0{});
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (PF2 && nk > 1) fetch(1, I1{});

  // fragment address of this lane inside a 32-row block; DMA image: chunk (2s + kq) ^ swz
  const int frag_off = DMA ? (lane & 31) * LDS_LD : (lane & 31) * LDS_LD + 4 * (lane >> 5);
  const int swz = ((lane & 31) >> 1) & 7, kq2 = lane >> 5;
  f32x4 af[2][MI], bf[2][NI];
  auto load_frags = [&](const float* Ab, const float* Wb, int s, int set) {
    const int col = DMA ? 4 * ((2 * s + kq2) ^ swz) : 8 * s;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[set][mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * LDS_LD + col);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[set][ni] = *reinterpret_cast<const f32x4*>(Wb + ni * 32 * LDS_LD + col);
  };

  auto slab = [&](int kt, auto curc) {
    constexpr int CUR = decltype(curc)::value;          // LDS buffer (and register set) of slab kt
    using Cur = std::integral_constant<int, CUR>;
    using Nxt = std::integral_constant<int, CUR ^ 1>;
    const float* Ab = As + CUR * BM * LDS_LD + wm * TM * LDS_LD + frag_off;
    const float* Wb = Ws + CUR * BN * LDS_LD + wn * TN * LDS_LD + frag_off;
    load_frags(Ab, Wb, 0, 0);
    if (!(ABL & 1)) {
      if (PF2) {
        if (kt + 2 < nk) fetch(kt + 2, Cur{});
      } else {
        if (kt + 1 < nk) fetch(kt + 1, Nxt{});
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) load_frags(Ab, Wb, s + 1, (s + 1) & 1);
      if (ABL & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][mi][j], bf[s & 1][ni][j], acc[mi][ni], 0, 0, 0);
      if (ABL & 4) __builtin_amdgcn_s_setprio(0);
    }
    if (!(ABL & 1) && kt + 1 < nk) {
      stash(Nxt{});
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    if (!(ABL & 2)) __syncthreads();
  };
  if constexpr ((ABL & 32) != 0) {
    // Chunked summation for very long contractions (retrieval: K = 49 152): every 64 slabs (2048 k)
    // the running accumulators are folded into a second set, so the fp32 rounding error grows with
    // sqrt(2048) + sqrt(K/2048) instead of sqrt(K) (measured 1e-5 -> 2e-6 on unit-norm VLADs).
    f32x16 tot[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[mi][ni][r] = 0.0f;
    for (int kt = 0; kt < nk; kt += 2) {
      slab(kt, I0{});
      if (kt + 1 < nk) slab(kt + 1, I1{});
      if ((kt & 63) == 62) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              tot[mi][ni][r] += acc[mi][ni][r];
              acc[mi][ni][r] = 0.0f;
            }
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += tot[mi][ni][r];
  } else {
    for (int kt = 0; kt < nk; kt += 2) {
      slab(kt, I0{});
      if (kt + 1 < nk) slab(kt + 1, I1{});
    }
  }

  if (ROWSQ) {
    if (tile_n == 0) {
#pragma unroll
      for (int i = 0; i < A_LD4; ++i) {
        float v = rsq[i];
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
        const int64_t row = m0 + r0 + RPP * i;
        if (kq == 0 && row < p.M) p.rowsq[row] = v;
      }
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA:
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 ----
  const int64_t wrow0 = m0 + wm * TM + 4 * (lane >> 5);
  const int64_t wcol0 = n0 + wn * TN + (lane & 31);
  if constexpr (EPI == EPI_SWIGLU) {
    const int64_t colg = wcol0, colv = wcol0 + 32;
    const int64_t ocol = (n0 + wn * TN) / 2 + (lane & 31);
    const bool cok = colv < p.N;
    const float bg = (cok && p.bias) ? p.bias[colg] : 0.0f;
    const float bv = (cok && p.bias) ? p.bias[colv] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M && cok) {
          const float g = acc[mi][0][r] + bg, v = acc[mi][1][r] + bv;
          p.C[row * p.ldc + ocol] = silu(g) * v;
        }
      }
  } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t col = wcol0 + ni * 32;
      const bool cok = col < p.N;
      const float b = (cok && p.bias) ? p.bias[col] : 0.0f;
      float gam = 0.0f;
      if constexpr (EPI == EPI_LS_RESID) gam = cok ? p.gamma[col] : 0.0f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = wrow0 + mi * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M && cok) {
            float v = acc[mi][ni][r] + b;
            if constexpr (EPI == EPI_STORE) {
              p.C[row * p.ldc + col] = v;
            } else if constexpr (EPI == EPI_GELU) {
              p.C[row * p.ldc + col] = gelu_erf(v);
            } else if constexpr (EPI == EPI_LS_RESID) {
              const int64_t o = row * p.ldc + col;
              p.C[o] = p.resid[o] + v * gam;
            } else if constexpr (EPI == EPI_PATCH) {
              const int64_t img = row / p.patches, pi = row - img * p.patches;
              const int64_t orow = img * (p.patches + 1) + 1 + pi;
              p.C[orow * p.ldc + col] = v + p.pos[(1 + pi) * p.N + col];
            }
          }
        }
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK, int OCC, int EPI, bool ROWSQ, bool KFULL, int ABL = 0>
int launch_cfg(const GemmProblem& p, hipStream_t stream) {
  const int tiles_m = (int)((p.M + BM - 1) / BM), tiles_n = (int)((p.N + BN - 1) / BN);
  const size_t lds = (size_t)2 * (BM + BN) * ((ABL & 64) ? BK : BK + 4) * sizeof(float);
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, BK, OCC, EPI, ROWSQ, KFULL, ABL>;
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(kern), (int)((int)lds)));
  const double ks = p.ksplit > 1 ? p.ksplit : 1;
  const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K * ks;
  const double bytes = 4.0 * (((double)p.M * p.K + (double)p.N * p.K) * ks + (double)p.M * p.N * ks);
  ProfScope prof(p.tag ? p.tag : "gemm_nt", stream, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, p.ksplit > 1 ? p.ksplit : 1), dim3(64 * WM * WN), lds, stream, p, tiles_m,
                     tiles_n);
  return launch_status("gemm_nt_kernel");
}

// tile configuration of the wide (ViT / retrieval) GEMMs; option gemm_f32_cfg selects an
// alternative at run time (micro-benchmarks only)
int gemm_cfg() { return (int)option(OPT_GEMM_F32_CFG); }

template <int EPI, bool KFULL>
int launch_wide(const GemmProblem& p, hipStream_t stream) {
  switch (gemm_cfg()) {
    case 1:  // 256x128, 4 waves of 128x64, one block per CU
      return launch_cfg<256, 128, 2, 2, 32, 1, EPI, false, KFULL>(p, stream);
    case 2:  // 256x128, 8 waves of 64x64, one block per CU
      return launch_cfg<256, 128, 4, 2, 32, 2, EPI, false, KFULL>(p, stream);
    case 3:  // 128x128, BK=16, three blocks per CU
      return launch_cfg<128, 128, 2, 2, 16, 3, EPI, false, KFULL>(p, stream);
    case 4:  // 256x256, 8 waves of 128x64 (swiglu-compatible), one block per CU
      return launch_cfg<256, 256, 2, 4, 16, 2, EPI, false, KFULL>(p, stream);
    // 5..8: timing-only ablations of the default tile (plain-store epilogue only; results are wrong)
    case 5:
    case 6:
    case 7:
    case 8:
      if constexpr (EPI == EPI_STORE && KFULL) {
        const int c = gemm_cfg();
        if (c == 5) return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 1>(p, stream);
        if (c == 6) return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 2>(p, stream);
        if (c == 7) return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 3>(p, stream);
        return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 4>(p, stream);
      }
      return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL>(p, stream);
    case 20:  // default tile, LDS-DMA staging
      if constexpr (KFULL) return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 64>(p, stream);
      return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL>(p, stream);
    case 10:  // default tile, loads issued two slabs ahead
      return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL, 8>(p, stream);
    case 11:  // 256x256 tile (8 waves of 128x64), loads two slabs ahead
      return launch_cfg<256, 256, 2, 4, 16, 2, EPI, false, KFULL, 8>(p, stream);
    default:  // 128x128, 4 waves of 64x64, two blocks per CU
      return launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, KFULL>(p, stream);
  }
}

// Row split of a GEMM into a 128-row-tile part [0, m1) and a 64-row-tile part [m1, M).
// 512 blocks are resident (2 per CU), so a grid runs in "rounds" of 512 tiles; a tile count just
// above a multiple of 512 pays a whole extra round for a few tiles (B = 32: 133 x 12 = 1596 tiles
// = 3.1 rounds).  Cost model in units of one full round of 128x128 tiles: a full round of 64x128
// tiles costs 0.55 (half the work, ~10 % less efficient); a last partial round costs 0.6 of its
// kind when it leaves every CU at most one block (<= 256 tiles), else a full one.
double rounds_cost(int64_t tiles, double unit) {
  const int64_t full = tiles / 512, last = tiles % 512;
  return unit * ((double)full + (last == 0 ? 0.0 : (last <= 256 ? 0.6 : 1.0)));
}
int64_t plan_row_split(int64_t M, int64_t N) {
  const int64_t tn = (N + 127) / 128;
  const int64_t tm = (M + 127) / 128;
  double best = 1e30;
  int64_t best_m1 = M;
  for (int64_t k = tm; k >= 0; --k) {      // ties -> the larger 128-row part
    const int64_t m1 = k * 128 < M ? k * 128 : M;
    const int64_t t128 = k * 128 < M ? k * tn : tm * tn;
    const int64_t t64 = ((M - m1 + 63) / 64) * tn;
    const double cost = rounds_cost(t128, 1.0) + rounds_cost(t64, 0.55) +
                        ((t128 > 0 && t64 > 0) ? 0.02 : 0.0);     // the second launch is not free
    if (cost < best - 1e-9) {
      best = cost;
      best_m1 = m1;
    }
  }
  // the model is coarse: only leave the plain 128-row grid for a predicted gain of >= 4 %
  return best < 0.96 * rounds_cost(tm * tn, 1.0) ? best_m1 : M;
}

template <int EPI>
int launch_wide_k(const GemmProblem& p, hipStream_t stream) {
  if (gemm_cfg() != 0 || p.K % 32 != 0 || EPI == EPI_PATCH)
    return (p.K % 32 == 0) ? launch_wide<EPI, true>(p, stream) : launch_wide<EPI, false>(p, stream);
  const int64_t m1 = plan_row_split(p.M, p.N);
  if (m1 > 0) {
    GemmProblem a = p;
    a.M = m1;
    ANYLOC_TRY((launch_cfg<128, 128, 2, 2, 32, 2, EPI, false, true>(a, stream)));
  }
  if (m1 < p.M) {
    GemmProblem b = p;                      // rows [m1, M) on 64x128 tiles (2x2 waves of 32x64)
    b.M = p.M - m1;
    b.A = p.A + m1 * p.lda;
    b.C = p.C + m1 * p.ldc;
    if (p.resid) b.resid = p.resid + m1 * p.ldc;
    ANYLOC_TRY((launch_cfg<64, 128, 2, 2, 32, 2, EPI, false, true>(b, stream)));
  }
  return ANYLOC_OK;
}

}  // namespace

int gemm_nt_splitk(const GemmProblem& p, hipStream_t stream) {
  ANYLOC_CHECK_ARG(p.A && p.W && p.C && p.rowsq, "gemm_nt_splitk: null operand");
  ANYLOC_CHECK_ARG(p.M > 0 && p.N > 0 && p.N <= 64 && p.K > 0 && p.K % 32 == 0 && p.ksplit >= 1 && p.ksplit < 65536,
                   "gemm_nt_splitk: needs N <= 64, a K slice that is a multiple of 32 and 1 <= ksplit < 65536");
  ANYLOC_CHECK_ARG(p.lda % 4 == 0 && p.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(p.W) & 15) == 0,
                   "gemm_nt_splitk: operands must be 16-byte aligned with row strides that are multiples of 4");
  return launch_cfg<128, 64, 4, 1, 32, 2, EPI_STORE, true, true>(p, stream);
}

int gemm_nt(const GemmProblem& p, int epilogue, hipStream_t stream) {
  ANYLOC_CHECK_ARG(p.A && p.W && p.C, "gemm_nt: null operand");
  ANYLOC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%lld N=%lld K=%lld",
                   (long long)p.M, (long long)p.N, (long long)p.K);
  ANYLOC_CHECK_ARG(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0,
                   "gemm_nt: K, lda, ldw must be multiples of 4 (K=%lld lda=%lld ldw=%lld)",
                   (long long)p.K, (long long)p.lda, (long long)p.ldw);
  ANYLOC_CHECK_ARG((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0,
                   "gemm_nt: operands must be 16-byte aligned");
  ANYLOC_CHECK_ARG((p.M + 127) / 128 * ((p.N + 31) / 32) < (1ll << 31), "gemm_nt: grid too large");
  const bool narrow = p.N <= 32;
  switch (epilogue) {
    case EPI_STORE:
      if (narrow) {
        return p.rowsq ? launch_cfg<128, 32, 4, 1, 32, 2, EPI_STORE, true, false>(p, stream)
                       : launch_cfg<128, 32, 4, 1, 32, 2, EPI_STORE, false, false>(p, stream);
      }
      if (p.rowsq) return launch_cfg<128, 128, 2, 2, 32, 2, EPI_STORE, true, false>(p, stream);
      if (p.K >= 8192 && p.K % 32 == 0 && gemm_cfg() == 0)      // long contraction: chunked summation
        return launch_cfg<128, 128, 2, 2, 32, 2, EPI_STORE, false, true, 32>(p, stream);
      return launch_wide_k<EPI_STORE>(p, stream);
    case EPI_GELU:
      return launch_wide_k<EPI_GELU>(p, stream);
    case EPI_LS_RESID:
      ANYLOC_CHECK_ARG(p.gamma && p.resid, "gemm_nt: LS_RESID needs gamma and resid");
      return launch_wide_k<EPI_LS_RESID>(p, stream);
    case EPI_SWIGLU:
      ANYLOC_CHECK_ARG(p.N % 64 == 0, "gemm_nt: SWIGLU needs N %% 64 == 0");
      return launch_wide_k<EPI_SWIGLU>(p, stream);
    case EPI_PATCH:
      ANYLOC_CHECK_ARG(p.pos && p.patches > 0, "gemm_nt: PATCH needs pos and patches");
      return launch_wide_k<EPI_PATCH>(p, stream);
    default:
      set_error("gemm_nt: unknown epilogue %d", epilogue);
      return ANYLOC_ERR_INVALID_ARG;
  }
}

}  // namespace anyloc

extern "C" int anyloc_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
                              int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
  anyloc::GemmProblem g{};
  g.A = A; g.lda = lda;
  g.W = W; g.ldw = ldw;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.bias = bias;
  g.tag = "gemm_nt";
  return anyloc::gemm_nt(g, anyloc::EPI_STORE, static_cast<hipStream_t>(stream));
}
