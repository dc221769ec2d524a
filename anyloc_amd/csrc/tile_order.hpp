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

// 16 bytes per lane straight from global memory into LDS (wave-uniform LDS base + lane * 16); out-of-range lanes of the
// buffer descriptor write zeros
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

}  // namespace anyloc
