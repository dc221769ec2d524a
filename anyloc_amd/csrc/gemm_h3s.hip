// Small-M plans of the row-scaled two-term fp16 GEMM (kernel: gemm_h3_kernel.hpp; arithmetic: gemm_h3.hip).
//
// The reference's scripts call the extractor with ONE image (scripts/dino_v2_vlad.py:164-188, demo/anyloc_vlad_generate.py
// :163-186): 530 token rows at 322 x 322.  A block GEMM then has a handful of row tiles and the 128 x 256 tiling of the
// batched forward leaves most of the 256 CUs idle: proj / fc2 of ViT-g on 64 x 64 tiles are 216 two-wave workgroups = 432
// waves for 1024 SIMDs.  A plan = (tile shape, k-blocks per ring stage, ring depth, split-K factor) picked per GEMM from its
// shape -- the table in choose() holds the measured winners; what the sweeps showed about this regime is written there:
//   * split-K: the contraction is cut into `ksplit` ranges, one workgroup each, so a K = 4096 GEMM of 216 tiles becomes 864
//     workgroups of 64 k-blocks; partial accumulators meet in a workspace and the LAST arrival of a tile (ticket) sums them
//     in split order -- deterministic -- and runs the fused epilogue (LayerScale-residual, q|k|v planes, SwiGLU + quantise);
//   * wider tiles (64 x 128, 64 x 256) where N is large (qkv, w12): twice the flops per staged byte of a 64 x 64 tile.
// A one-image forward is therefore no longer bitwise the same image inside a batch (other summation order over k); both
// meet the oracle bar, and they agree to ~1e-7 (tests/test_gpu_vit.py).
#include "gemm_h3_kernel.hpp"

namespace anyloc {

namespace {

struct Plan {
  int cfg, kb, ksplit;
  int stages = 3;      // ring depth: 3 or 6 stages
};

// tile configurations: id -> (MI, NI, WM, WN); BM = 32 MI WM, BN = 32 NI WN
constexpr int NCFG = 8;
const int kCfgBM[NCFG] = {64, 64, 64, 64, 128, 64, 64, 192};
const int kCfgBN[NCFG] = {64, 128, 128, 128, 128, 256, 256, 128};

// the plans the LayerNorm lead role is compiled for: the one-image qkv plan (128 x 128, 6-deep ring) and the one-image w12 plan
// (192 x 128, 6-deep ring), with the epilogues those two GEMMs have in the fused forward
template <int EPI, int MI, int NI, int WM, int WN, int KB, int ST>
constexpr bool ln_lead_compiled() {
  return KB == 1 && ST == 6 && WM == 2 && WN == 2 && NI == 2 &&
         ((MI == 2 && EPI == EPI_QKV_PLANES) || (MI == 3 && (EPI == EPI_SWIGLU_T_H2 || EPI == EPI_SWIGLU_H2)));
}
constexpr int LN_LEAD_MAX_TILES = 200;        // GEMM workgroups of a lead launch: fewer than CUs, so the lead workgroups always find one

template <int EPI, int MI, int NI, int WM, int WN, int KB, int ST = 3>
int launch_small(const H3Problem& p, hipStream_t stream) {
  using Cfg = H3Cfg<MI, NI, WM, WN, ST, KB>;
  const int tiles_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM), tiles_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
  if constexpr (ln_lead_compiled<EPI, MI, NI, WM, WN, KB, ST>()) {
    if (p.ln_x) {
      ANYLOC_CHECK_ARG(p.ksplit <= 1 && tiles_m * tiles_n <= LN_LEAD_MAX_TILES && p.ln_tickets && p.ln_w && p.ln_b &&
                           (!p.ln_has_bound || p.c_inv) && p.ln_dim == 16 * p.K16 && Cfg::BM % Cfg::NW == 0,
                       "gemm_h3_small: LayerNorm lead role asked for a launch it does not fit (h3s_ln_lead_feasible)");
      H3Problem q = p;
      q.ln_wgs = (int)(((p.M + Cfg::NW - 1) / Cfg::NW + 7) / 8 * 8);        // one row per wave; a multiple of 8 (XCD mapping of the GEMM ids)
      static DynLds dyn_lds_once;
      ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_h3_kernel<MI, NI, WM, WN, ST, 2, EPI, KB, 1>), (int)(Cfg::LDS)));
      hipLaunchKernelGGL((gemm_h3_kernel<MI, NI, WM, WN, ST, 2, EPI, KB, 1>), dim3((unsigned)(q.ln_wgs + tiles_m * tiles_n)),
                         dim3(64 * WM * WN), Cfg::LDS, stream, q, tiles_m, tiles_n);
      return launch_status("gemm_h3_kernel (small-M plan, LayerNorm lead role)");
    }
  } else {
    ANYLOC_CHECK_ARG(!p.ln_x, "gemm_h3_small: LayerNorm lead role asked for a plan it is not compiled for (h3s_ln_lead_feasible)");
  }
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(&gemm_h3_kernel<MI, NI, WM, WN, ST, 2, EPI, KB>), (int)(Cfg::LDS)));
  hipLaunchKernelGGL((gemm_h3_kernel<MI, NI, WM, WN, ST, 2, EPI, KB>), dim3((unsigned)(tiles_m * tiles_n * std::max(1, p.ksplit))),
                     dim3(64 * WM * WN), Cfg::LDS, stream, p, tiles_m, tiles_n);
  return launch_status("gemm_h3_kernel (small-M plan)");
}

// a 6-deep ring where it fits the CU's 160 KiB and the counted wait's 6 bits
template <int MI, int NI, int WM, int WN, int KB>
constexpr bool deep_ok() {
  using C6 = H3Cfg<MI, NI, WM, WN, 3, KB>;      // (LDS / NDMA scale linearly with the stage count)
  return 2 * C6::LDS <= 160 * 1024 && 4 * C6::NDMA <= 63;
}

template <int EPI, int MI, int NI, int WM, int WN, int KB>
int launch_st(const H3Problem& p, int stages, hipStream_t stream) {
  if constexpr (deep_ok<MI, NI, WM, WN, KB>()) {
    if (stages >= 6) return launch_small<EPI, MI, NI, WM, WN, KB, 6>(p, stream);
  }
  return launch_small<EPI, MI, NI, WM, WN, KB, 3>(p, stream);
}

template <int EPI, int MI, int NI, int WM, int WN>
int launch_kb(const H3Problem& p, int kb, int stages, hipStream_t stream) {
  if (kb >= 4) {
    if constexpr (H3Cfg<MI, NI, WM, WN, 3, 4>::LDS <= 160 * 1024) return launch_st<EPI, MI, NI, WM, WN, 4>(p, stages, stream);
    kb = 2;
  }
  if (kb == 2) return launch_st<EPI, MI, NI, WM, WN, 2>(p, stages, stream);
  return launch_st<EPI, MI, NI, WM, WN, 1>(p, stages, stream);
}

template <int EPI>
int launch_cfg(const H3Problem& p, const Plan& pl, hipStream_t stream) {
  switch (pl.cfg) {
    case 0: return launch_kb<EPI, 1, 2, 2, 1>(p, pl.kb, pl.stages, stream);    // 64 x 64, two waves of 32 x 64
    case 1: return launch_kb<EPI, 2, 2, 1, 2>(p, pl.kb, pl.stages, stream);    // 64 x 128, two waves of 64 x 64
    case 2: return launch_kb<EPI, 1, 2, 2, 2>(p, pl.kb, pl.stages, stream);    // 64 x 128, four waves of 32 x 64
    case 3: return launch_kb<EPI, 1, 4, 2, 1>(p, pl.kb, pl.stages, stream);    // 64 x 128, two waves of 32 x 128
    case 4: return launch_kb<EPI, 2, 2, 2, 2>(p, pl.kb, pl.stages, stream);    // 128 x 128, four waves of 64 x 64
    case 5: return launch_kb<EPI, 2, 2, 1, 4>(p, pl.kb, pl.stages, stream);    // 64 x 256, four waves of 64 x 64
    case 7: return launch_kb<EPI, 3, 2, 2, 2>(p, pl.kb, pl.stages, stream);    // 192 x 128, four waves of 96 x 64 (530 rows = 3 row tiles)
    default: return launch_kb<EPI, 1, 4, 2, 2>(p, pl.kb, pl.stages, stream);   // 64 x 256, four waves of 32 x 128
  }
}

int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The plan table.  Starting point: the round-3 small-batch rules (64 x 64 two-wave tiles with four / two k-blocks per ring
// stage while a workgroup is alone on its CU, 128 x 128 four-wave tiles from 256 such tiles up).  Up to two images (M <= 1100
// rows) the measured winners of the plan sweeps replace them (tools/sweep_b1.py on one MI355X, ViT-G/14 322 x 322, time per
// launch inside a B = 1 / B = 2 forward: profiles/r04_b1_plan_sweep.log, r04_b1_plan_sweep_depth.log):
//   fc2   (K = 4096, N = 1536)  64 x 128 four-wave tiles, 6-deep ring; split-K 2 at B=1: 49.9 -> 39.7 us;  B=2 unsplit: 75.7 -> 56.8
//   proj  (K = 1536, N = 1536)  B=1: 64 x 64, 6-deep ring 25.0 -> 22.7 us;  B=2: 64 x 128, 6-deep ring 34.8 -> 28.3
//   qkv   (N = 4608)            B=1: 128 x 128, 6-deep ring 39.1 -> 37.8 us;   B=2: 64 x 128 four-wave 56.1 -> 53.4
//   w12   (N = 8192)            among the 64- / 128-row shapes the round-3 choice stays the fastest (128 x 128 at B=1: 59 us by HIP
//                               events, 55 us by dispatch timestamps) -- but its 5 x 64 = 320 workgroups put TWO on 64 of the 256 CUs
//                               while 192 CUs hold one: the launch lasts as long as two tiles on one CU (qkv on the same tile,
//                               180 workgroups = one per CU, takes 33 us).  One image is 530 rows = 3 row tiles of 192 (576 rows:
//                               8 % padding instead of 21 %): 192 x 128 tiles are 3 x 64 = 192 equal workgroups, one per CU, each
//                               1.5 x the work of a 128 x 128 tile instead of 2 x (option h3s_w12_tall, default 1; 0 = 128 x 128):
//                               61.0 -> 51.1 us per launch, 49.2 with the 6-deep ring (120 KiB: one workgroup per CU is all there
//                               is), 6.21 -> 5.83 ms per forward, the SAME bits (an unsplit plan keeps the order over k);
//                               192-row tiles forced on qkv / proj / fc2 lose (108 / 72 / 36 tiles; profiles/r04_b1_w12_tall.log)
// What the sweeps say about this regime: split-K pays only for the long contraction (a split workgroup's ticket hand-off and
// the last arrival's slab reads cost what the shorter k-loop saves at K = 1536); a deeper ring (bytes in flight) helps the
// GEMMs with the fewest workgroups; and with the weights resident on-die (a 2-block model) the same launches are no faster
// (profiles/r04_b1_weight_residency_probe.log) -- a one-image GEMM waits for its own fill / barrier / MFMA chain, not for HBM.
Plan choose(const H3Problem& p, int epilogue) {
  Plan pl{0, 1, 1};
  const int64_t t64 = cdiv(p.M, 64) * cdiv(p.N, 64);
  if (cdiv(p.M, 128) * cdiv(p.N, 128) >= option(OPT_H3_TINY_MAX)) pl = Plan{4, 1, 1};
  else pl = Plan{0, t64 < option(OPT_H3_DEEP_MAX) ? 4 : t64 < option(OPT_H3_DEEP2_MAX) ? 2 : 1, 1};
  if (option(OPT_H3S_ENABLE) == 0) return pl;              // the round-3 small-batch kernels, exactly
  if (p.M <= 1100) {
    const bool one = p.M <= 600;                           // one 322 x 322 image (530 rows) / two
    if (p.N <= 2048 && p.K16 >= 192) {                     // fc2-like: long contraction, narrow output
      pl = Plan{2, 1, one ? 2 : 1};
      pl.stages = 6;
    } else if (p.N <= 2048) {                              // proj-like
      pl = one ? Plan{0, 1, 1} : Plan{2, 1, 1};
      pl.stages = 6;
    } else if (p.N < 8192) {                               // qkv-like
      pl = one ? Plan{4, 1, 1} : Plan{2, 1, 1};
      pl.stages = one ? 6 : 3;
    } else if (one && p.M > 384 && option(OPT_H3S_W12_TALL)) {   // w12-like, 385 ... 576 rows: three 192-row tiles, one workgroup per CU
      pl = Plan{7, 1, 1};
      pl.stages = 6;
    }
  } else if (p.M <= 1700) {
    // One 476 x 630 image = 1531 token rows: the reference scripts' DEFAULT shape (configs.py:141 resize [480, 640], centre crop
    // scripts/dino_v2_vlad.py:173-176) -- and three 322 x 322 images.  Round 5 sweep at that shape (tools/sweep_b1.py 1 ... 476x630,
    // profiles/r05_b1_480x640_plan_sweep.log; time per launch inside a B = 1 forward): the 128 x 128 default is within 1 - 3 % of
    // the best plan for qkv (70.6 us) and w12 (103.7 us); the two narrow GEMMs are not --
    //   fc2  (K = 4096, N = 1536): 144 tiles of 128 x 128 on 256 CUs; 64 x 128 four-wave tiles with split-K 2: 88.6 -> 76.3 us
    //   proj (K = 1536, N = 1536): 64 x 128 four-wave tiles, 6-deep ring:                                   40.1 -> 37.4 us
    if (p.N <= 2048 && p.K16 >= 192) {
      pl = Plan{2, 1, 2};
    } else if (p.N <= 2048) {
      pl = Plan{2, 1, 1};
      pl.stages = 6;
    }
  }
  const int64_t mask = option(OPT_H3S_MASK);
  const int bit = p.kind == H3_KIND_QKV ? 1 : p.kind == H3_KIND_PROJ ? 2 : p.kind == H3_KIND_FC1 ? 4 : p.kind == H3_KIND_FC2 ? 8 : 16;
  if (mask & bit) {
    const int64_t c = option(OPT_H3S_CFG), s = option(OPT_H3S_KSPLIT), k = option(OPT_H3S_KB);
    if (c >= 0 && c < NCFG) pl.cfg = (int)c;
    if (s > 0) pl.ksplit = (int)s;
    if (k == 1 || k == 2 || k == 4) pl.kb = (int)k;
    const int64_t st = option(OPT_H3S_STAGES);
    if (st == 3 || st == 6) pl.stages = (int)st;
  }
  return pl;
}

}  // namespace

// the plan gemm_h3_small will run (shared by it and by h3s_ln_lead_feasible)
static Plan final_plan(H3Problem& p, int epilogue) {
  Plan pl = choose(p, epilogue);
  const int64_t tiles = cdiv(p.M, kCfgBM[pl.cfg]) * cdiv(p.N, kCfgBN[pl.cfg]);
  if (!p.sk_part || !p.sk_tickets || p.accumulate) pl.ksplit = 1;
  pl.ksplit = (int)std::min<int64_t>(pl.ksplit, p.K16);
  while (pl.ksplit > 1 && ((size_t)pl.ksplit * tiles * kCfgBM[pl.cfg] * kCfgBN[pl.cfg] * sizeof(float) > H3_SPLIT_PART_BYTES ||
                           tiles > (int64_t)H3_SPLIT_TICKETS))
    --pl.ksplit;
  return pl;
}

bool h3s_ln_lead_feasible(const H3Problem& p_in, int epilogue) {
  if (option(OPT_H3S_LN_LEAD) == 0 || option(OPT_H3_CFG) != 0) return false;
  if (epilogue != EPI_QKV_PLANES && epilogue != EPI_SWIGLU_T_H2 && epilogue != EPI_SWIGLU_H2) return false;
  if (((p_in.M + 127) / 128) * ((p_in.N + 255) / 256) >= 512) return false;        // (dispatch_h3's small-M rule)
  H3Problem p = p_in;
  const Plan pl = final_plan(p, epilogue);
  if (pl.ksplit != 1 || pl.kb != 1 || pl.stages != 6) return false;
  if (!((pl.cfg == 4 && epilogue == EPI_QKV_PLANES) || (pl.cfg == 7 && epilogue != EPI_QKV_PLANES))) return false;
  return cdiv(p.M, kCfgBM[pl.cfg]) * cdiv(p.N, kCfgBN[pl.cfg]) <= LN_LEAD_MAX_TILES;
}

bool h3_small_supported(int epilogue) {
  switch (epilogue) {
    case EPI_STORE: case EPI_LS_RESID: case EPI_QKV_PLANES: case EPI_GELU_H2: case EPI_SWIGLU_H2: case EPI_SWIGLU_T_H2: return true;
    default: return false;
  }
}

int gemm_h3_small(const H3Problem& p_in, int epilogue, hipStream_t stream) {
  H3Problem p = p_in;
  // the epilogues that write q|k|v tiles need whole heads per wave column block: NI even (all configurations have it)
  // split-K needs the workspace, a plain (non-accumulating) epilogue input and enough k-blocks
  Plan pl = final_plan(p, epilogue);
  p.ksplit = pl.ksplit;
  // k-blocks per split: a multiple of the ring stage's k-blocks, so that only the LAST split can end inside a stage
  // (its missing k-blocks lie beyond the buffer descriptors and read as zeros)
  if (pl.ksplit > 1) {
    p.kper = (int)(cdiv(cdiv(p.K16, pl.ksplit), pl.kb) * pl.kb);
    p.ksplit = (int)cdiv(p.K16, p.kper);                   // no empty split
    if (p.ksplit <= 1) { p.ksplit = 1; p.kper = p.K16; }
  } else {
    p.kper = p.K16;
  }
  switch (epilogue) {
    case EPI_STORE: return launch_cfg<EPI_STORE>(p, pl, stream);
    case EPI_LS_RESID: return launch_cfg<EPI_LS_RESID>(p, pl, stream);
    case EPI_QKV_PLANES: return launch_cfg<EPI_QKV_PLANES>(p, pl, stream);
    case EPI_GELU_H2: return launch_cfg<EPI_GELU_H2>(p, pl, stream);
    case EPI_SWIGLU_H2: return launch_cfg<EPI_SWIGLU_H2>(p, pl, stream);
    case EPI_SWIGLU_T_H2: return launch_cfg<EPI_SWIGLU_T_H2>(p, pl, stream);
    default: set_error("gemm_h3_small: epilogue %d has no small-M plan", epilogue); return ANYLOC_ERR_UNSUPPORTED;
  }
}

}  // namespace anyloc
