// Few-query retrieval scores with the database split ON THE FLY into TWO fp16 planes under a RUNNING power-of-two row scale:
// three fp16 MFMA products per k-step instead of the six bf16 products of scores_x6.hip, same fp32-GEMM accuracy.
//
// replaces (with topk.hip): faiss IndexFlatIP / IndexFlatL2 .search for a handful of queries against a long database
// (get_top_k_recall, reference utilities.py:439-450) -- the per-step retrieval of bench.py (61 query VLADs x 10 000 rows
// x 49 152 columns).
//
// Why not the row-scaled split of gemm_h3.hip: its scale 2^e (row maximum into [2^14, 2^15)) needs the maximum over all
// 49 152 columns BEFORE the first column is quantised -- a second pass over HBM.  Here the scale of a database row RUNS with
// the slabs: a 32-k slab is quantised with 2^e_run, e_run = the exponent that fits the largest magnitude seen SO FAR in this
// row (of this K slice).  While e_run stays, products of all slabs are in the same units and add up in one fp32 accumulator
// like gemm_h3's.  When a slab brings a new maximum, e_run drops and the accumulator rows of that database row are multiplied
// by 2^(e_new - e_old) -- a power of two, exact -- before the slab is added: a wave-uniform branch taken a handful of times
// per row (new maxima of a 49 152-long row follow the harmonic series), never in the steady state.  Every element is thus
// quantised relative to a maximum that is AT MOST the row's global one: errors <= those of the row-scaled split (22 bits
// relative to the row maximum, the dropped lo x lo product below 2^-24).  Queries are few: their row scales come from a
// pre-pass over the (L2-sized) query block (row_scales_h2).
//
// Decomposition, staging and outputs as scores_x6.hip: database rows are the M operand (128-row tiles), the queries the N
// operand (64 columns, zero padded), K cut into S slices (grid.y); slice s writes part[s][row][0..63] in TRUE units (the
// scales are applied in the epilogue) and the partial row sums of squares rsq[s][row] from the fp32 values as staged.
// Per 32-k slab and workgroup: database values global -> registers (two slabs ahead) -> amax over the row's 8 staging lanes
// (three DPP max) -> scale, split ONCE by the staging lane (the bf16 kernel splits at every fragment read: 88 vector
// instructions per wave and slab) -> LDS as two fp16 planes -> 4 + 8 fragment reads, 12 MFMAs.  The QUERIES are split once per
// call (fewq_query_image_kernel: per slab the exact bytes of a stage's B region) and reach LDS by DMA -- 36 of the ~240 vector
// instructions per wave and slab and 16 registers less, same bits: 0.413-0.418 -> 0.400-0.403 ms at the bench shape,
// 4.9 TB/s (option topk_fewq_qdma = 0: the lanes split them per slab; profiles/r04_fewq_qdma.log); with the row maxima taken on the
// bit patterns (integer max: no canonicalising v_max_f32, the DPP moves fold into it) 0.381-0.385 ms = 5.1 TB/s.
#include <type_traits>

#include "common.hpp"
#include "tile_order.hpp"

namespace anyloc {

namespace {

typedef _Float16 sh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sh_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned sh_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sh_u32x2 __attribute__((ext_vector_type(2)));

constexpr int SH_BM = 128, SH_BN = 64, SH_BK = 32;
constexpr int SH_ROW = 2 * SH_BK + 16;                   // bytes per row and plane in LDS (80: 16-byte slots 5 r mod 16, conflict-free)
constexpr int SH_A_PLANE = SH_BM * SH_ROW, SH_B_PLANE = SH_BN * SH_ROW;
constexpr int SH_FAC = 2 * SH_A_PLANE + 2 * SH_B_PLANE;  // per-row rescale factors of the slab (128 floats), then 4 flags
constexpr int SH_STAGE = SH_FAC + SH_BM * 4 + 16;        // 31 248 bytes

__device__ __forceinline__ f32x4 sh_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
template <int CTRL>
__device__ __forceinline__ float sh_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned sh_dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float sh_pow2(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }   // e in [-126, 127]
__device__ __forceinline__ void sh_pack2(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2 pr;
  pr[0] = a; pr[1] = b;
  const sh_f16x2 h = __builtin_convertvector(pr, sh_f16x2);
  f32x2 res;
  res[0] = pr[0] - (float)h[0];
  res[1] = pr[1] - (float)h[1];
  const sh_f16x2 l = __builtin_convertvector(res, sh_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// QDMA: the queries arrive pre-split (fewq_query_image_kernel: per 32-k slab the two fp16 planes of the 64 query rows in
// exactly the bytes of a stage's B region, 10 KiB) and go global -> LDS by ten 1-KiB DMA pieces per slab -- no registers, no
// vector instructions; !QDMA: the round-4 first version, queries split by the staging lanes like the database
template <bool QDMA>
__global__ __launch_bounds__(256, 2) void scores_fewq_h3_kernel(const float* __restrict__ db, int64_t ldd, int64_t rows,
                                                                const float* __restrict__ qu, int64_t ldq, int nq,
                                                                const float* __restrict__ qinv, int64_t kslice,
                                                                const unsigned char* __restrict__ qimg,
                                                                float* __restrict__ part, float* __restrict__ rsq_part) {
  constexpr int LPR = SH_BK / 4, RPP = 256 / LPR, A_LD = SH_BM / RPP, B_LD = SH_BN / RPP;   // 8 lanes per row, 32 rows per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char sh_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * SH_BM;
  const int64_t sl = blockIdx.y;
  const int kq = tid % LPR, r0 = tid / LPR;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(db + m0 * ldd + sl * kslice), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = QDMA
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(qimg + sl * (kslice / SH_BK) * (int64_t)(2 * SH_B_PLANE)), 0,
                                          0x7fffffff, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qu + sl * kslice), 0, 0x7fffffff, 0x00020000);
  unsigned a_off[A_LD], b_off[B_LD];
  float b_scale[B_LD];                                   // 2^e of the query row (0 for the padding rows: their planes are zero)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    int64_t row = m0 + r0 + RPP * i;
    row = (row < rows ? row : rows - 1) - m0;            // rows past the end re-read the last row; never stored
    a_off[i] = (unsigned)((row * ldd + 4 * kq) * 4);
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int row = r0 + RPP * i;
    const bool ok = row < nq;
    b_off[i] = (unsigned)(((int64_t)(ok ? row : 0) * ldq + 4 * kq) * 4);
    b_scale[i] = ok ? __uint_as_float((254u << 23) - __float_as_uint(qinv[row])) : 0.0f;   // 2^e from the stored 2^-e
  }
  const int nk = (int)(kslice / SH_BK);
  f32x4 ra[2][A_LD], rb[2][B_LD];
  float rsq[A_LD];
  int e_run[A_LD];                                       // running scale exponent of this thread's rows (same in the row's 8 lanes)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    rsq[i] = 0.f;
    e_run[i] = 100;                                      // the largest scale: nothing seen yet (as h2_row_scale's all-zero row)
  }
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  auto fetch = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const unsigned kb = (unsigned)kt * (SH_BK * 4);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) ra[S][i] = sh_load16(a_rsrc, a_off[i], kb);
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      if constexpr (!QDMA) rb[S][i] = sh_load16(b_rsrc, b_off[i], kb);
  };
  // register set S -> LDS stage: amax of the row's slab, running scale, split, planes; `real` = 0.0f for the copy of the last
  // slab the unconditional prefetch brings in past the end (same values: no new maximum; its squares are not counted again)
  auto stash = [&](auto setc, int stage, float real) {
    constexpr int S = decltype(setc)::value;
    unsigned char* st = sh_smem + stage * SH_STAGE;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const f32x4 v = ra[S][i];
      rsq[i] += real * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
      // largest magnitude of the row's slab on the BIT PATTERNS (|x| as unsigned integers order like the floats; no
      // canonicalising v_max_f32 per operand, and the DPP moves fold into the integer max).  Row by row: taking the four rows
      // of a lane through every reduction step together removes the DPP wait states (72 -> 12 nop cycles per two slabs) but
      // measures 7 % SLOWER -- every row then waits for the lane's last load
      constexpr unsigned MAG = 0x7fffffffu;
      unsigned am = max(max(__float_as_uint(v[0]) & MAG, __float_as_uint(v[1]) & MAG),
                        max(__float_as_uint(v[2]) & MAG, __float_as_uint(v[3]) & MAG));
      am = max(am, sh_dpp_u<0xB1>(am));                  // quad_perm [1,0,3,2]
      am = max(am, sh_dpp_u<0x4E>(am));                  // quad_perm [2,3,0,1]
      am = max(am, sh_dpp_u<0x141>(am));                 // row_half_mirror: the other quad of the 8 lanes
      const int ex = (int)(am >> 23);
      const int e_slab = ex == 0 ? 100 : max(-100, min(100, 14 - (ex - 127)));
      const int e_new = min(e_run[i], e_slab);
      const int row = r0 + RPP * i;
      if (kq == 0) {
        reinterpret_cast<float*>(st + SH_FAC)[row] = sh_pow2(max(e_new - e_run[i], -126));
        if (e_new != e_run[i]) reinterpret_cast<int*>(st + SH_FAC + SH_BM * 4)[row >> 5] = 1;
      }
      e_run[i] = e_new;
      const float sc = sh_pow2(e_new);
      unsigned h0, l0, h1, l1;
      sh_pack2(v[0] * sc, v[1] * sc, h0, l0);
      sh_pack2(v[2] * sc, v[3] * sc, h1, l1);
      unsigned char* ad = st + row * SH_ROW + kq * 8;
      *reinterpret_cast<sh_u32x2*>(ad) = sh_u32x2{h0, h1};
      *reinterpret_cast<sh_u32x2*>(ad + SH_A_PLANE) = sh_u32x2{l0, l1};
    }
    if constexpr (!QDMA)
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const f32x4 q = rb[S][i];
      unsigned h0, l0, h1, l1;
      sh_pack2(q[0] * b_scale[i], q[1] * b_scale[i], h0, l0);
      sh_pack2(q[2] * b_scale[i], q[3] * b_scale[i], h1, l1);
      unsigned char* bd = st + 2 * SH_A_PLANE + (r0 + RPP * i) * SH_ROW + kq * 8;
      *reinterpret_cast<sh_u32x2*>(bd) = sh_u32x2{h0, h1};
      *reinterpret_cast<sh_u32x2*>(bd + SH_B_PLANE) = sh_u32x2{l0, l1};
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;

  const int fr = lane & 31, fh = lane >> 5;
  auto contract = [&](int stage) {
    unsigned char* st = sh_smem + stage * SH_STAGE;
    // a new row maximum in this wave's 32 rows: bring their accumulators to the new (smaller) scale first
    int* flag = reinterpret_cast<int*>(st + SH_FAC + SH_BM * 4) + wave;
    if (__builtin_amdgcn_readfirstlane(*flag)) {
      const float* fac = reinterpret_cast<const float*>(st + SH_FAC) + wave * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 f = *reinterpret_cast<const f32x4*>(fac + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][4 * g + j] *= f[j];
          acc[1][4 * g + j] *= f[j];
        }
      }
      if (lane == 0) *flag = 0;
    }
    const unsigned char* ap = st + (wave * 32 + fr) * SH_ROW + fh * 16;
    const unsigned char* bp = st + 2 * SH_A_PLANE + fr * SH_ROW + fh * 16;
#pragma unroll
    for (int s2 = 0; s2 < SH_BK / 16; ++s2) {
      sh_f16x8 af[2], bf[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        af[pl] = __builtin_bit_cast(sh_f16x8, *reinterpret_cast<const sh_u32x4*>(ap + pl * SH_A_PLANE + s2 * 32));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bf[ni][pl] = __builtin_bit_cast(sh_f16x8, *reinterpret_cast<const sh_u32x4*>(bp + pl * SH_B_PLANE + ni * 32 * SH_ROW + s2 * 32));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[ni][0], acc[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[ni][1], acc[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[ni][0], acc[ni], 0, 0, 0);
    }
  };

  // (every fetch is issued unconditionally -- past the last slab it re-reads the last one -- so that the compiler's counted
  // vector-memory waits survive: scores_x6.hip)
  if (tid < 8) reinterpret_cast<int*>(sh_smem + (tid >> 2) * SH_STAGE + SH_FAC + SH_BM * 4)[tid & 3] = 0;
  __syncthreads();
  const int last = nk - 1;
  // QDMA: slab kt's query planes -> the B region of `stage`: pieces w, w + 4, w + 8 (< 10) of 1 KiB by wave w.  Issued BEFORE
  // the half-step's database fetch, so "at most A_LD vector-memory operations outstanding" means the pieces have landed
  // while the fetch stays in flight (the counter retires in order); meet() = that wait + the LDS writes + the barrier.
  auto dma_b = [&](int kt, int stage) {
    if constexpr (QDMA) {
      unsigned char* dst = sh_smem + stage * SH_STAGE + 2 * SH_A_PLANE;
      const unsigned so = (unsigned)kt * (unsigned)(2 * SH_B_PLANE);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int piece = __builtin_amdgcn_readfirstlane(wave) + 4 * c;   // (scalar: a real branch, not an exec mask)
        if (piece < (2 * SH_B_PLANE) / 1024) dma16_to_lds(b_rsrc, dst + piece * 1024, (unsigned)(piece * 1024 + lane * 16), so);
      }
    }
  };
  auto meet = [&]() {
    if constexpr (QDMA) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_LD) : "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
  };
  static_assert((2 * SH_B_PLANE) % 1024 == 0 && (2 * SH_B_PLANE) / 1024 <= 12, "query planes of a slab: whole 1-KiB pieces, <= 3 per wave");
  dma_b(0, 0);
  fetch(0, S0{});
  fetch(min(1, last), S1{});
  stash(S0{}, 0, 1.0f);
  if constexpr (QDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    dma_b(min(kt + 1, last), 1);
    fetch(min(kt + 2, last), S0{});
    __builtin_amdgcn_sched_barrier(0);
    contract(0);
    __builtin_amdgcn_sched_barrier(0);
    stash(S1{}, 1, kt + 1 < nk ? 1.0f : 0.0f);
    meet();
    if (kt + 1 < nk) {
      dma_b(min(kt + 2, last), 0);
      fetch(min(kt + 3, last), S1{});
      __builtin_amdgcn_sched_barrier(0);
      contract(1);
      __builtin_amdgcn_sched_barrier(0);
      stash(S0{}, 0, kt + 2 < nk ? 1.0f : 0.0f);
      meet();
    }
  }
  if constexpr (QDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- partial row sums of squares (the 8 staging lanes of a row hold its pieces) and the rows' final 2^-e ----
  float* einv = reinterpret_cast<float*>(sh_smem + SH_FAC);     // (stage 0's factor table: nobody contracts any more)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    float v = rsq[i];
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    const int64_t row = m0 + r0 + RPP * i;
    if (kq == 0) {
      if (row < rows) rsq_part[sl * rows + row] = v;
      einv[r0 + RPP * i] = sh_pow2(-e_run[i]);
    }
  }
  __syncthreads();
  // ---- partial scores in true units: C/D layout of the 32 x 32 block -- lane = query column, 16 rows of its half ----
  float* out = part + sl * rows * 64;
  float qi[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) qi[ni] = ni * 32 + fr < nq ? qinv[ni * 32 + fr] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    const int64_t row = m0 + lr;
    if (row < rows) {
      const float ai = einv[lr];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) out[row * 64 + ni * 32 + fr] = acc[ni][r] * (ai * qi[ni]);
    }
  }
}

// queries [nq <= 64, dim] fp32 + their 2^-e -> per 32-k slab the bytes of a stage's B region: [plane][64 rows][SH_ROW] (rows
// past nq: zeros; the 16 pad bytes of a row are never read)
__global__ __launch_bounds__(256) void fewq_query_image_kernel(const float* __restrict__ qu, int64_t ldq, int nq,
                                                               const float* __restrict__ qinv, int64_t dim,
                                                               unsigned char* __restrict__ qimg) {
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (slab, row, k-quad of 4)
  const int kq = (int)(item & 7), row = (int)((item >> 3) & 63);
  const int64_t slab = item >> 9;
  if (slab >= dim / SH_BK) return;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  float sc = 0.f;
  if (row < nq) {
    v = *reinterpret_cast<const f32x4*>(qu + (int64_t)row * ldq + slab * SH_BK + 4 * kq);
    sc = __uint_as_float((254u << 23) - __float_as_uint(qinv[row]));      // 2^e from the stored 2^-e
  }
  unsigned h0, l0, h1, l1;
  sh_pack2(v[0] * sc, v[1] * sc, h0, l0);
  sh_pack2(v[2] * sc, v[3] * sc, h1, l1);
  unsigned char* dst = qimg + slab * (int64_t)(2 * SH_B_PLANE) + row * SH_ROW + kq * 8;
  *reinterpret_cast<sh_u32x2*>(dst) = sh_u32x2{h0, h1};
  *reinterpret_cast<sh_u32x2*>(dst + SH_B_PLANE) = sh_u32x2{l0, l1};
}

}  // namespace

size_t fewq_query_image_bytes(int64_t dim) { return (size_t)(dim / SH_BK) * (size_t)(2 * SH_B_PLANE); }

int fewq_query_image(const float* queries, int64_t ldq, int64_t nq, const float* qinv, int64_t dim, unsigned char* qimg,
                     hipStream_t stream) {
  ANYLOC_CHECK_ARG(queries && qinv && qimg && nq > 0 && nq <= 64 && dim % SH_BK == 0 && ldq % 4 == 0, "fewq_query_image: bad arguments");
  const int64_t items = dim / SH_BK * 512;
  hipLaunchKernelGGL(fewq_query_image_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, queries, ldq, (int)nq, qinv,
                     dim, qimg);
  return launch_status("fewq_query_image_kernel");
}

int scores_fewq_h3(const float* db, int64_t ldd, int64_t rows, const float* queries, int64_t ldq, int64_t nq, const float* qinv,
                   const unsigned char* qimg, int64_t kslice, int ksplit, float* part, float* rsq_part, hipStream_t stream) {
  ANYLOC_CHECK_ARG(db && queries && qinv && part && rsq_part, "scores_fewq_h3: null operand");
  ANYLOC_CHECK_ARG(rows > 0 && nq > 0 && nq <= 64 && kslice > 0 && kslice % 32 == 0 && ksplit >= 1 && ksplit < 65536,
                   "scores_fewq_h3: needs <= 64 queries, a K slice that is a multiple of 32 and 1 <= ksplit < 65536");
  ANYLOC_CHECK_ARG(ldd % 4 == 0 && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(db) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(queries) & 15) == 0,
                   "scores_fewq_h3: operands must be 16-byte aligned with row strides that are multiples of 4");
  ANYLOC_CHECK_ARG(127 * ldd * 4 + kslice * 4 < (1ll << 31) && 63 * ldq * 4 + kslice * 4 < (1ll << 31),
                   "scores_fewq_h3: a tile's rows must stay inside 2 GiB of buffer addressing");
  const int64_t tiles = (rows + SH_BM - 1) / SH_BM;
  ANYLOC_CHECK_ARG(tiles < (1ll << 31), "scores_fewq_h3: grid too large");
  ProfScope prof("topk_scores_gemm", stream, 2.0 * rows * 64 * kslice * ksplit, 4.0 * (rows + 64.0) * kslice * ksplit);
  // qimg != nullptr: the queries' pre-split planes (fewq_query_image) go to LDS by DMA; nullptr: split by the staging lanes
#define ANYLOC_FEWQ_H3(QD)                                                                                                       \
  do {                                                                                                                           \
    static DynLds dyn_lds_once; \
    ANYLOC_TRY(ensure_dyn_lds(dyn_lds_once, reinterpret_cast<const void*>(scores_fewq_h3_kernel<QD>), (int)(2 * SH_STAGE)));                                                                                                                            \
    hipLaunchKernelGGL(scores_fewq_h3_kernel<QD>, dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 2 * SH_STAGE, stream, db,  \
                       ldd, rows, queries, ldq, (int)nq, qinv, kslice, qimg, part, rsq_part);                                    \
  } while (0)
  if (qimg) ANYLOC_FEWQ_H3(true);
  else ANYLOC_FEWQ_H3(false);
#undef ANYLOC_FEWQ_H3
  return launch_status("scores_fewq_h3_kernel");
}

}  // namespace anyloc
