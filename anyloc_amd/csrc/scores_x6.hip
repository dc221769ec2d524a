// Few-query retrieval scores with the database split ON THE FLY into bf16 planes: HBM-bound instead of fp32-MFMA-bound.
//
// replaces (with topk.hip): faiss IndexFlatIP / IndexFlatL2 .search for a handful of queries against a long database
// (get_top_k_recall, reference utilities.py:439-450) -- the per-step retrieval of bench.py (61 query VLADs x 10 000 rows
// x 49 152 columns), where the database is read once and every other cost should hide behind that read.
//
// The fp32-MFMA version of this pass (gemm_nt_splitk, gemm_f32.hip) is limited by the matrix cores: <= 64 queries padded to
// 64 columns on v_mfma_f32_32x32x2_f32 cost 2048 matrix-core cycles per wave and 32-k slab, 0.40 ms at peak for the bench
// shape against 0.30 ms of HBM time (measured 0.70 ms = 2.8 TB/s).  Here every fp32 operand is the EXACT sum of three bf16
// terms (gemm_x6.hip's arithmetic: x = x1 + x2 + x3, round-to-nearest-even, residuals exact), the six leading plane
// products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 768 matrix-core cycles per wave and slab, fp32-GEMM
// accuracy, and, unlike the two-term fp16 split of gemm_h3.hip, no row scale (a power-of-two row scale needs the row's
// maximum over all 49 152 columns BEFORE the first column is quantised: a second pass over HBM).
//
// Same decomposition and outputs as gemm_nt_splitk: database rows are the M operand (128-row tiles), the queries the N
// operand (64 columns, zero-padded), K is cut into S slices (grid.y) so that tiles x S fills the chip; slice s writes its
// partial scores part[s][row][0..63] and the partial row sums of squares rsq[s][row] (from the fp32 values as staged: the
// F.normalize / L2 terms of the database come out of the same read).  splitk_combine_kernel (topk.hip) adds the slices.
//
// Per 32-k slab and workgroup (4 waves, wave w owns rows 32 w .. 32 w + 31 of the tile and all 64 queries):
//   database  global -> registers (4 x 16 B per thread, issued TWO slabs ahead) -> LDS as fp32, rows padded to 36 floats
//             -> fragment reads (8 consecutive k per lane: 2 x ds_read_b128, conflict-free) -> split in registers
//   queries   global (L2-resident) -> registers -> split ONCE per workgroup -> LDS as three bf16 planes, rows padded to 80 B
//             -> fragment reads (ds_read_b128, conflict-free)
//   12 MFMAs per k-step of 16 (6 plane products x 2 query blocks); two accumulators per block (leading / 2^-16 products).
#include <type_traits>

#include "common.hpp"

namespace anyloc {

namespace {

typedef __bf16 sx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned sx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sx_u32x2 __attribute__((ext_vector_type(2)));

constexpr int SX_BM = 128, SX_BN = 64;
// BK = k per slab: 32 (128 B per database row and slab, two workgroups per CU).  A 64-k slab (256 B runs, one workgroup per
// CU) was measured slower: 0.71 vs 0.61 ms at the bench shape (profiles/r03_fewq_x6.log)
template <int BK>
struct SxCfg {
  static constexpr int ALD = BK + 4;                     // floats per database row in LDS (BK = 32: 36, BK = 64: 68 -> 16-byte slots 9 r / 17 r mod 16: conflict-free)
  static constexpr int BROW = 2 * BK + 16;               // bytes per query row and plane in LDS (80 / 144 -> slots 5 r / 9 r mod 16)
  static constexpr int A_BYTES = SX_BM * ALD * 4;
  static constexpr int B_PLANE = SX_BN * BROW;
  static constexpr int STAGE = A_BYTES + 3 * B_PLANE;    // 33 792 / 62 464 bytes
  static constexpr int LPR = BK / 4;                     // staging lanes per row (16 B each)
  static constexpr int RPP = 256 / LPR;                  // rows per staging pass
  static constexpr int A_LD = SX_BM / RPP, B_LD = SX_BN / RPP;
};

__device__ __forceinline__ f32x4 sx_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

template <int BK>
__global__ __launch_bounds__(256, BK == 32 ? 2 : 1) void scores_fewq_x6_kernel(const float* __restrict__ db, int64_t ldd, int64_t rows,
                                                                const float* __restrict__ qu, int64_t ldq, int nq,
                                                                int64_t kslice, float* __restrict__ part,
                                                                float* __restrict__ rsq_part) {
  using Cfg = SxCfg<BK>;
  constexpr int SX_BK = BK, SX_ALD = Cfg::ALD, SX_BROW = Cfg::BROW, SX_A_BYTES = Cfg::A_BYTES, SX_B_PLANE = Cfg::B_PLANE,
                SX_STAGE = Cfg::STAGE, LPR = Cfg::LPR, RPP = Cfg::RPP, A_LD = Cfg::A_LD, B_LD = Cfg::B_LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char sx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * SX_BM;
  const int64_t sl = blockIdx.y;
  const int kq = tid % LPR, r0 = tid / LPR;              // staging: LPR lanes x 16 B cover one row segment of the slab

  // ---- staging coordinates (buffer loads: descriptor at the tile origin, constant per-thread offset, slab offset scalar) ----
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(db + m0 * ldd + sl * kslice), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(qu + sl * kslice), 0, 0x7fffffff, 0x00020000);
  unsigned a_off[A_LD], b_off[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    int64_t row = m0 + r0 + RPP * i;
    row = (row < rows ? row : rows - 1) - m0;            // rows past the end re-read the last row; never stored
    a_off[i] = (unsigned)((row * ldd + 4 * kq) * 4);
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int row = r0 + RPP * i;
    b_ok[i] = row < nq;
    b_off[i] = (unsigned)(((int64_t)(b_ok[i] ? row : 0) * ldq + 4 * kq) * 4);
  }
  const int nk = (int)(kslice / SX_BK);
  // TWO slabs travel in registers ahead of the one being contracted (sets 0 / 1): with one, a workgroup has 16 KB of HBM
  // loads in flight -- 32 KB per CU at two workgroups -- and the pass ran at 3.1 TB/s, the latency-bandwidth product of that
  // much data (profiles/r03_fewq_x6.log); two sets double it
  f32x4 ra[2][A_LD], rb[2][B_LD];
  float rsq[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) rsq[i] = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  auto fetch = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const unsigned kb = (unsigned)kt * (SX_BK * 4);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) ra[S][i] = sx_load16(a_rsrc, a_off[i], kb);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) rb[S][i] = sx_load16(b_rsrc, b_off[i], kb);   // (padding rows re-read row 0; zeroed in stash)
  };
  // `real` = 1.0f for a slab of the slice, 0.0f for the copy of the last slab the unconditional prefetch brings in past the
  // end (its squares must not be counted a second time; the LDS image it leaves is never contracted)
  auto stash = [&](auto setc, int stage, float real) {   // register set S -> LDS stage
    constexpr int S = decltype(setc)::value;
    unsigned char* st = sx_smem + stage * SX_STAGE;
    float* ad = reinterpret_cast<float*>(st) + r0 * SX_ALD + 4 * kq;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      *reinterpret_cast<f32x4*>(ad + RPP * i * SX_ALD) = ra[S][i];
      rsq[i] += real * (ra[S][i][0] * ra[S][i][0] + ra[S][i][1] * ra[S][i][1] + ra[S][i][2] * ra[S][i][2] + ra[S][i][3] * ra[S][i][3]);
    }
    unsigned char* bd = st + SX_A_BYTES + r0 * SX_BROW + kq * 8;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      unsigned p01[3], p23[3];
      const f32x4 q = b_ok[i] ? rb[S][i] : zero4;          // a select, not a branch around the load: the counted waits survive
      split_pair_x3(q[0], q[1], p01);
      split_pair_x3(q[2], q[3], p23);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        sx_u32x2 w;
        w[0] = p01[pl];
        w[1] = p23[pl];
        *reinterpret_cast<sx_u32x2*>(bd + pl * SX_B_PLANE + RPP * i * SX_BROW) = w;
      }
    }
  };

  // Two accumulators per 32 x 32 block: the three leading products (a1 b1, a1 b2, a2 b1) and the three of relative size
  // 2^-16 (a2 b2, a1 b3, a3 b1).  In ONE accumulator the small ones are below half an ulp of the running sum after a few
  // hundred k and round away -- harmless when their signs are random, but a query that IS (nearly) a database row makes
  // a2 b2 = x2^2 >= 0 for every k: a systematic -4e-6 on a cosine of 1 (measured on the 131 072-column ViT-L VLADs).
  // Summed among themselves they keep their bits; the two sums are added once at the end.
  f32x16 acc[2], accl[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[ni][r] = 0.0f; accl[ni][r] = 0.0f; }

  const int fr = lane & 31, fh = lane >> 5;
  auto contract = [&](int stage) {                       // the 32-k slab sitting in LDS stage `stage`
    const unsigned char* st = sx_smem + stage * SX_STAGE;
    const float* ap = reinterpret_cast<const float*>(st) + (wave * 32 + fr) * SX_ALD + 8 * fh;
    const unsigned char* bp = st + SX_A_BYTES + fr * SX_BROW + fh * 16;
#pragma unroll
    for (int s2 = 0; s2 < SX_BK / 16; ++s2) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + 16 * s2);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 16 * s2 + 4);
      sx_u32x4 bw[2][3];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bw[ni][pl] = *reinterpret_cast<const sx_u32x4*>(bp + pl * SX_B_PLANE + ni * 32 * SX_BROW + s2 * 32);
      unsigned q0[3], q1[3], q2[3], q3[3];
      split_pair_x3(a0[0], a0[1], q0);
      split_pair_x3(a0[2], a0[3], q1);
      split_pair_x3(a1[0], a1[1], q2);
      split_pair_x3(a1[2], a1[3], q3);
      sx_bf16x8 af[3], bf[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        sx_u32x4 w;
        w[0] = q0[pl]; w[1] = q1[pl]; w[2] = q2[pl]; w[3] = q3[pl];
        af[pl] = __builtin_bit_cast(sx_bf16x8, w);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[ni][pl] = __builtin_bit_cast(sx_bf16x8, bw[ni][pl]);
      }
#define ANYLOC_SX_TERM(A, pa, pb) \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) A[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa], bf[ni][pb], A[ni], 0, 0, 0);
      ANYLOC_SX_TERM(accl, 2, 0) ANYLOC_SX_TERM(accl, 0, 2) ANYLOC_SX_TERM(accl, 1, 1)
      ANYLOC_SX_TERM(acc, 1, 0) ANYLOC_SX_TERM(acc, 0, 1) ANYLOC_SX_TERM(acc, 0, 0)
#undef ANYLOC_SX_TERM
    }
  };

  // Every fetch is issued unconditionally (past the last slab it re-reads the last one and the result is dropped): a load
  // inside a branch makes the compiler's vector-memory wait conservative at the join -- vmcnt(0), i.e. the newest prefetch
  // is drained with the one that is needed and the second set buys nothing.
  const int last = nk - 1;
  fetch(0, S0{});
  fetch(min(1, last), S1{});
  stash(S0{}, 0, 1.0f);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // slab kt is in LDS stage 0, slab kt + 1 in register set 1 (loads issued one iteration ago), set 0 is free
    fetch(min(kt + 2, last), S0{});
    __builtin_amdgcn_sched_barrier(0);                   // (the loads stay in front of the contraction they fly over)
    contract(0);
    __builtin_amdgcn_sched_barrier(0);
    stash(S1{}, 1, kt + 1 < nk ? 1.0f : 0.0f);           // (kt + 1 == nk: a copy of the last slab nobody contracts)
    __syncthreads();
    if (kt + 1 < nk) {
      fetch(min(kt + 3, last), S1{});
      __builtin_amdgcn_sched_barrier(0);
      contract(1);
      __builtin_amdgcn_sched_barrier(0);
      stash(S0{}, 0, kt + 2 < nk ? 1.0f : 0.0f);
      __syncthreads();
    }
  }

  // ---- partial row sums of squares: the 8 staging lanes of a row hold its pieces ----
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    float v = rsq[i];
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    const int64_t row = m0 + r0 + RPP * i;
    if (kq == 0 && row < rows) rsq_part[sl * rows + row] = v;
  }
  // ---- partial scores: C/D layout of the 32 x 32 block -- lane = query column, 16 rows of its half ----
  float* out = part + sl * rows * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    if (row < rows) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) out[row * 64 + ni * 32 + fr] = acc[ni][r] + accl[ni][r];
    }
  }
}

template <int BK>
int launch_fewq(const float* db, int64_t ldd, int64_t rows, const float* queries, int64_t ldq, int64_t nq, int64_t kslice, int ksplit,
                float* part, float* rsq_part, hipStream_t stream) {
  const int64_t tiles = (rows + SX_BM - 1) / SX_BM;
  static DynLds dyn_lds_once;
  ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(scores_fewq_x6_kernel<BK>), (int)(2 * SxCfg<BK>::STAGE)));
  hipLaunchKernelGGL((scores_fewq_x6_kernel<BK>), dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 2 * SxCfg<BK>::STAGE, stream, db,
                     ldd, rows, queries, ldq, (int)nq, kslice, part, rsq_part);
  return launch_status("scores_fewq_x6_kernel");
}

}  // namespace

int scores_fewq_x6(const float* db, int64_t ldd, int64_t rows, const float* queries, int64_t ldq, int64_t nq, int64_t kslice,
                   int ksplit, float* part, float* rsq_part, hipStream_t stream) {
  ANYLOC_CHECK_ARG(db && queries && part && rsq_part, "scores_fewq_x6: null operand");
  ANYLOC_CHECK_ARG(rows > 0 && nq > 0 && nq <= 64 && kslice > 0 && kslice % 32 == 0 && ksplit >= 1 && ksplit < 65536,
                   "scores_fewq_x6: needs <= 64 queries, a K slice that is a multiple of 32 and 1 <= ksplit < 65536");
  ANYLOC_CHECK_ARG(ldd % 4 == 0 && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(db) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(queries) & 15) == 0 && kslice % 4 == 0,
                   "scores_fewq_x6: operands must be 16-byte aligned with row strides that are multiples of 4");
  ANYLOC_CHECK_ARG(127 * ldd * 4 + kslice * 4 < (1ll << 31) && 63 * ldq * 4 + kslice * 4 < (1ll << 31),
                   "scores_fewq_x6: a tile's rows must stay inside 2 GiB of buffer addressing");
  ANYLOC_CHECK_ARG((rows + SX_BM - 1) / SX_BM < (1ll << 31), "scores_fewq_x6: grid too large");
  ProfScope prof("topk_scores_gemm", stream, 2.0 * rows * 64 * kslice * ksplit, 4.0 * (rows + 64.0) * kslice * ksplit);
  return launch_fewq<32>(db, ldd, rows, queries, ldq, nq, kslice, ksplit, part, rsq_part, stream);
}

}  // namespace anyloc
