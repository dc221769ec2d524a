// Two-term fp16 GEMM on v_mfma_f32_16x16x32_f16: the arithmetic and operand images of gemm_h3.hip (three fp16 matrix-core
// products per k of row-scaled 22-bit operands, fp32 accumulate) on the OTHER fp16 MFMA shape.
//
// Why: the block GEMMs run at the chip's power limit, and inside that limit the 16 x 16 x 32 instruction delivers 13 % more
// flops than the 32 x 32 x 16 one on random operands -- register-resident, no LDS, no HBM: 1 931 vs 1 700-1 718 TFLOP/s,
// both 2 370-2 424 on all-zero operands (tools/micro/mfma_power.hip, profiles/r03_mfma_shape_power.log).  It is the shape the
// vendor's fp16 kernel uses on these shapes (hipBLASLt "MT256x256x64_MI16x16x1", profiles/r02_calib_h3_hipblaslt_zero_data.log).
//
// Structure: 256 x 256 tile, 8 waves (4 x 2) of 64 x 128 = 4 x 8 blocks of 16 x 16 (128 accumulator registers, as before),
// ONE workgroup per CU.  The MFMA contracts 32 k, the h2 image is blocked by 16 k, so a ring stage holds a PAIR of k-blocks:
// [k-block][A hi | A lo | W hi | W lo] = 2 x 32 KiB; two stages = 128 KiB.  Staging is the pure DMA of gemm_h3.hip (1-KiB
// pieces, eight per wave and stage, issued right after the barrier that frees the stage and flying over the 96 MFMAs =
// 1 536 matrix-core cycles of the other stage).
//
// Fragments: lane (row = lane & 15, kg = lane >> 4) supplies 8 consecutive k of its row -- one 16-byte chunk.  WHICH chunk of
// the 32-k pair a k-group reads is free as long as A and B agree; kg -> (k-block kg & 1, half kg >> 1) makes every
// ds_read_b128 service group of 16 lanes hit 16 distinct 16-byte slots of the EXISTING image (rows of 32 B, halves swapped
// when (row >> 3) & 1): same images, same producers, no bank conflicts.
//
// C/D layout of a 16 x 16 block: lane holds column (lane & 15) of B's rows and rows 4 (lane >> 4) + r, r < 4, of A's rows.
// With the WEIGHTS as the A operand (TR) a lane holds one token and four consecutive output columns: 16 bytes of fp32.
#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

typedef _Float16 hm_f16x8 __attribute__((ext_vector_type(8)));

// WM x WN waves of 16 MB rows x 16 NB columns; built: <4, 2, 4, 8> = 256 x 256 tile, 8 waves of 64 x 128, 128 KiB of LDS,
// one workgroup per CU (two waves per SIMD).  Measured and dropped (profiles/r03_h3m_sweep.log): <2, 2, 4, 6> = 128 x 192
// tiles, two workgroups per CU (-2 ... -5 % against gemm_h3_kernel); <2, 2, 8, 8> = four waves of 128 x 128 with the 256
// accumulators pinned in AGPRs (inline-asm MFMAs; the structure of the vendor's MT256x256x64_MI16x16 kernel), all fragment
// reads of a stage issued up front and the next stage's DMA spread between the MFMAs: -5 % at K = 1536, -1.5 % at K = 4096 --
// with one wave per SIMD nothing covers a tile's prologue and epilogue
template <int WM, int WN, int MB, int NB>
struct HmCfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN;
  static constexpr int A_PLANE = BM * 32, W_PLANE = BN * 32;     // one plane of one operand of one k-block
  static constexpr int KBLK = 2 * A_PLANE + 2 * W_PLANE;         // A hi | A lo | W hi | W lo of one k-block
  static constexpr int STAGE = 2 * KBLK;                         // a pair of k-blocks
  static constexpr int LDS = 2 * STAGE;
  static constexpr int A_PIECES = BM / 32, W_PIECES = BN / 32;   // 1-KiB DMA pieces per plane and k-block
  static_assert(NB % 2 == 0, "B fragments are read in two halves");
};

template <int WM, int WN, int MB, int NB, int OCC, int EPI>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_h3m_kernel(H3Problem p, int tiles_m, int tiles_n) {
  using Cfg = HmCfg<WM, WN, MB, NB>;
  constexpr int HM_BM = Cfg::BM, HM_BN = Cfg::BN, HM_KBLK = Cfg::KBLK, HM_STAGE = Cfg::STAGE, A_PLANE = Cfg::A_PLANE,
                W_PLANE = Cfg::W_PLANE, NW = Cfg::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char hm_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  xcd_grouped_tile(blockIdx.x, tiles_m, tiles_n, p.group_m, tm, tn);
  const int64_t m0 = (int64_t)tm * HM_BM, n0 = (int64_t)tn * HM_BN;

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
  // pair `ks` = k-blocks 2 ks, 2 ks + 1 (past the last k-block: outside the descriptor, zero-fills); wave w stages the
  // 1-KiB pieces w, w + NW, ... of every plane
  auto issue = [&](int ks, int stage) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      unsigned char* st = hm_smem + stage * HM_STAGE + kb * HM_KBLK + wave * 1024;
      const unsigned ao = (unsigned)(2 * ks + kb) * a_slab, wo = (unsigned)(2 * ks + kb) * w_slab;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < (Cfg::A_PIECES + NW - 1) / NW; ++c)
          if (Cfg::A_PIECES % NW == 0 || wave + c * NW < Cfg::A_PIECES)
            dma16_to_lds(a_rsrc, st + pl * A_PLANE + c * (1024 * NW), a_voff[pl] + c * (1024 * NW), ao);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < (Cfg::W_PIECES + NW - 1) / NW; ++c)
          if (Cfg::W_PIECES % NW == 0 || wave + c * NW < Cfg::W_PIECES)
            dma16_to_lds(w_rsrc, st + 2 * A_PLANE + pl * W_PLANE + c * (1024 * NW), w_voff[pl] + c * (1024 * NW), wo);
    }
  };

  // fragment address of this lane inside a 16-row block: k-group kg reads (k-block kg & 1, half kg >> 1)
  const int fr = lane & 15, kg = lane >> 4;
  const unsigned char* frag = hm_smem + (kg & 1) * HM_KBLK + fr * 32 + (((kg >> 1) ^ ((fr >> 3) & 1)) << 4);

  f32x4 acc[MB][NB];
#pragma unroll
  for (int ma = 0; ma < MB; ++ma)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K16 + 1) / 2;                       // pairs of k-blocks
  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    const int stage = ks & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of pair ks have landed ...
    __builtin_amdgcn_s_barrier();                       // ... and everybody's; nobody reads the other stage any more
    const unsigned char* sa = frag + stage * HM_STAGE + (wm * 16 * MB) * 32;
    const unsigned char* sw = frag + stage * HM_STAGE + 2 * A_PLANE + (wn * 16 * NB) * 32;
    issue(ks + 1, stage ^ 1);
    hm_f16x8 a[MB][2];
#pragma unroll
    for (int ma = 0; ma < MB; ++ma)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) a[ma][pl] = *reinterpret_cast<const hm_f16x8*>(sa + pl * A_PLANE + ma * 512);
    constexpr int NH = NB / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      hm_f16x8 b[NH][2];
#pragma unroll
      for (int nb = 0; nb < NH; ++nb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[nb][pl] = *reinterpret_cast<const hm_f16x8*>(sw + pl * W_PLANE + (half * NH + nb) * 512);
      // term-major: MB NH MFMAs between two visits of the same accumulator
#define ANYLOC_HM_TERM(pa, pb)                                                                     \
  _Pragma("unroll") for (int ma = 0; ma < MB; ++ma) _Pragma("unroll") for (int nb = 0; nb < NH; ++nb) \
      acc[ma][half * NH + nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ma][pa], b[nb][pb], acc[ma][half * NH + nb], 0, 0, 0);
      ANYLOC_HM_TERM(1, 0) ANYLOC_HM_TERM(0, 1) ANYLOC_HM_TERM(0, 0)
#undef ANYLOC_HM_TERM
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue (EPI_STORE): lane = output column (lane & 15) of each 16-column block, rows 4 (lane >> 4) + r ----
  const int64_t col0 = n0 + wn * 16 * NB + fr;
  float bv[NB], sw_[NB];
  bool cok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int64_t col = col0 + nb * 16;
    cok[nb] = col < p.N;
    bv[nb] = (cok[nb] && p.bias) ? p.bias[col] : 0.0f;
    sw_[nb] = cok[nb] ? p.w_inv[col] : 0.0f;
  }
#pragma unroll
  for (int ma = 0; ma < MB; ++ma)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + wm * 16 * MB + ma * 16 + 4 * kg + r;
      if (row < p.M) {
        const float ai = p.a_inv[row];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (cok[nb]) {
            const float v = acc[ma][nb][r] * (ai * sw_[nb]) + bv[nb];
            const int64_t o = row * p.ldc + col0 + nb * 16;
            p.C[o] = p.accumulate ? p.C[o] + v : v;
          }
      }
    }
}

template <int WM, int WN, int MB, int NB, int OCC, int EPI>
int launch_h3m(const H3Problem& p, hipStream_t stream) {
  using Cfg = HmCfg<WM, WN, MB, NB>;
  const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_h3m_kernel<WM, WN, MB, NB, OCC, EPI>), (int)(Cfg::LDS)));
  hipLaunchKernelGGL((gemm_h3m_kernel<WM, WN, MB, NB, OCC, EPI>), dim3((unsigned)(tiles_m * tiles_n)), dim3(64 * Cfg::NW), Cfg::LDS,
                     stream, p, tiles_m, tiles_n);
  return launch_status("gemm_h3m_kernel");
}

}  // namespace

// EPI_STORE only (the retrieval score panels, anyloc_gemm_nt_h3); returns ANYLOC_ERR_UNSUPPORTED for the fused epilogues of the
// ViT blocks, which stay on gemm_h3_kernel (the caller falls back)
int gemm_h3m(const H3Problem& p, int epilogue, hipStream_t stream) {
  if (epilogue != EPI_STORE) return ANYLOC_ERR_UNSUPPORTED;
  return launch_h3m<4, 2, 4, 8, 2, EPI_STORE>(p, stream);
}

}  // namespace anyloc
