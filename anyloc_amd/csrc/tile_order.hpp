// Device helpers shared by the three GEMM families (gemm_f32 / gemm_x6 / gemm_h3).
#pragma once
#include <hip/hip_runtime.h>

namespace anyloc {

// Workgroup id -> output tile.  Workgroup b is dispatched to XCD b % 8, and every XCD has a private 4 MiB L2: the ids of
// one XCD (b, b+8, b+16, ...) are first made contiguous ("logical"), then walked in groups of GM tile rows x all tile
// columns, m fastest -- so the workgroups resident on an XCD at the same time share a few A row panels and W column
// panels through its L2 instead of each XCD touching every panel.
__device__ __forceinline__ void xcd_grouped_tile(int bid, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
  const int nb = tiles_m * tiles_n;
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int group_size = GM * tiles_n;
  const int g = logical / group_size;
  const int first_m = g * GM;
  const int gm = min(tiles_m - first_m, GM);
  const int within = logical - g * group_size;
  tm = first_m + within % gm;
  tn = within / gm;
}

// The first step alone: ids of one XCD made contiguous -- consecutive logical ids run on the SAME XCD at about the same time
__device__ __forceinline__ int xcd_contiguous_id(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ---- LayerNorm LEAD role inside a batched GEMM launch (gemm_h3_kernel<..., LNL = 2>) -------------------------------------------
// The launch holds two kinds of workgroups: GEMM tiles (the xcd_grouped_tile order above, unchanged) and LEAD workgroups that
// normalise LEAD_ROWS rows each into the GEMM's own operand image.  Workgroup b runs on XCD b % 8 and the dispatcher hands
// workgroups out in id order, so the launch is described per XCD as a SEQUENCE (position = b / 8):
//   [lead workgroups of the XCD's first tile-row group]
//   then, per group j the XCD has tiles of:  its tiles of group j, with the lead workgroups of group j + 1 spread evenly among them
// where the lead work of a group belongs to the XCD that holds the group's LAST tile (the XCD that starts inside a group needs it at
// once; the XCD before it needs it at the very end).  Invariant (checked on the host for every shape before the plan is used,
// lead_plan_ok): every lead workgroup of a tile row has a SMALLER workgroup id than every GEMM tile of that row -- in-order
// dispatch then means a waiting GEMM workgroup's producers are already running or done, and lead workgroups never wait.
constexpr int LEAD_ROWS = 16;
struct LeadPlan {
  int tiles_m, tiles_n, group_m, bm;     // the GEMM's tiles, its scheduling group, rows per tile
  long long M;                           // rows
};
struct LeadRole {
  int kind;                              // 0 = nothing (past the XCD's sequence), 1 = GEMM tile, 2 = lead workgroup
  int tm, tn;                            // kind 1
  long long row0;                        // kind 2: first of LEAD_ROWS rows
};
__host__ __device__ inline int lead_group_wgs(const LeadPlan& lp, int g) {
  const long long r0 = (long long)g * lp.group_m * lp.bm;
  const long long r1 = r0 + (long long)lp.group_m * lp.bm < lp.M ? r0 + (long long)lp.group_m * lp.bm : lp.M;
  return r1 > r0 ? (int)((r1 - r0 + LEAD_ROWS - 1) / LEAD_ROWS) : 0;
}
// role of position `loc` in XCD `xcd`'s sequence; *len (optional) = the sequence's length
__host__ __device__ inline LeadRole lead_decode(const LeadPlan& lp, int xcd, int loc, int* len = nullptr) {
  const int nb = lp.tiles_m * lp.tiles_n;
  const int q = nb >> 3, r = nb & 7;
  const int s = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, e = s + q + (xcd < r ? 1 : 0);
  const int gs = lp.group_m * lp.tiles_n;
  const int groups = (lp.tiles_m + lp.group_m - 1) / lp.group_m;
  LeadRole out{0, 0, 0, 0};
  int pos = loc, total = 0;
  bool found = false;
  auto owns = [&](int g) {                       // does this XCD hold the last tile of group g?
    if (g >= groups) return false;
    const int last = (g + 1 < groups ? (g + 1) * gs : nb) - 1;
    return last >= s && last < e;
  };
  auto tile_of = [&](int logical) {
    const int g = logical / gs, first_m = g * lp.group_m;
    const int gm = lp.tiles_m - first_m < lp.group_m ? lp.tiles_m - first_m : lp.group_m;
    const int within = logical - g * gs;
    out.kind = 1;
    out.tm = first_m + within % gm;
    out.tn = within / gm;
  };
  auto lead_of = [&](int g, int i) {
    out.kind = 2;
    out.row0 = (long long)g * lp.group_m * lp.bm + (long long)i * LEAD_ROWS;
  };
  if (e > s) {
    const int g0 = s / gs;
    if (owns(g0)) {
      const int l0 = lead_group_wgs(lp, g0);
      if (!found && pos < l0) { lead_of(g0, pos); found = true; }
      pos -= l0;
      total += l0;
    }
    for (int j = g0; j < groups && j * gs < e; ++j) {
      const int lo = j * gs > s ? j * gs : s;
      const int ge = j + 1 < groups ? (j + 1) * gs : nb;
      const int hi = ge < e ? ge : e;
      const int c = hi - lo;
      const int l = owns(j + 1) && (j + 1) * gs >= s ? lead_group_wgs(lp, j + 1) : 0;
      if (!found && pos >= 0 && pos < c + l) {
        // the leads are spread over the FIRST HALF of the XCD's tiles of group j (element p of that part: leads before it =
        // ceil(p * l / n1)), so that group j + 1's rows are ready long before its first tile is handed out -- spread to the end,
        // every group boundary stalled the XCD's slots for one lead workgroup's latency (profiles/r06_batched_ln_lead.log)
        const int c1 = l > 0 ? (c + 1) / 2 : 0;
        const long long n1 = c1 + l;
        if (pos < n1) {
          const int before = (int)(((long long)pos * l + n1 - 1) / n1), after = (int)(((long long)(pos + 1) * l + n1 - 1) / n1);
          if (after > before) lead_of(j + 1, before);
          else tile_of(lo + (pos - before));
        } else {
          tile_of(lo + c1 + (pos - (int)n1));
        }
        found = true;
      }
      pos -= c + l;
      total += c + l;
    }
  }
  if (len) *len = total;
  return out;
}

// 16 bytes per lane straight from global memory into LDS (wave-uniform LDS base + lane * 16); out-of-range lanes of the
// buffer descriptor write zeros
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}
// the same with the sc1 cache-policy bit (aux 16): the load is served from the device-coherent level, not from this CU's L1 --
// for operands another workgroup of the SAME launch has just written write-through (the LayerNorm lead role of gemm_h3_kernel)
__device__ __forceinline__ void dma16_to_lds_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 16);
}

}  // namespace anyloc
