// Node-tile helpers shared by the fused conv kernels (conv_forward.hip: 32 x 32 MFMA tiles, two
// workgroup halves; conv_forward16.hip: 16 x 16 tiles, eight waves on all of a workgroup's tiles).
#pragma once
#include "common.hpp"

namespace {

// The argument block is read where the dispatch packet put it — the kernarg segment (constant
// address space, scalar loads, dynamic indexing of its arrays) — instead of a private copy.
typedef const __attribute__((address_space(4))) lnz_forward_args KArgs;

struct TileDesc {
  int ta;     // molecule in rows [0, split)  (the only one of a single tile)
  int tb;     // molecule in rows [split, 32) or -1
  int split;  // multiple of 8; 32 for a single tile
};

// Slot `slot` of the tile plan (or, without a plan, molecule `slot` as a single tile).
__device__ __forceinline__ TileDesc load_tile_desc(KArgs& a, int slot) {
  TileDesc t;
  if (a.plan) {
    t.ta = a.plan[3 * slot + 0];
    t.tb = a.plan[3 * slot + 1];
    t.split = a.plan[3 * slot + 2];
  } else {
    t.ta = slot < a.B ? slot : -1;
    t.tb = -1;
    t.split = 32;
  }
  t.ta = __builtin_amdgcn_readfirstlane(t.ta);
  t.tb = __builtin_amdgcn_readfirstlane(t.tb);
  t.split = __builtin_amdgcn_readfirstlane(t.split);
  return t;
}

// Node extents (last real node + 1) of the one or two molecules of a tile, wave uniform.
__device__ __forceinline__ void tile_extents(KArgs& a, const TileDesc& t, int lane, int& nA, int& nB) {
  const bool pr = t.tb >= 0;
  int la = 0, lb = 0;
  for (int i = lane; i < a.N; i += 64) {
    la = a.mask[(int64_t)t.ta * a.N + i] ? i + 1 : la;
    if (pr) lb = a.mask[(int64_t)t.tb * a.N + i] ? i + 1 : lb;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    la = max(la, __shfl_xor(la, off, 64));
    lb = max(lb, __shfl_xor(lb, off, 64));
  }
  nA = __builtin_amdgcn_readfirstlane(la);
  nB = __builtin_amdgcn_readfirstlane(lb);
}

// 8-row groups of the tile that hold real nodes (bit g: rows 8g..8g+7): A from row 0, B from `split`.
__device__ __forceinline__ int row_group_mask(int nA, int nB, int split) {
  return ((1 << ((nA + 7) >> 3)) - 1) | (((1 << ((nB + 7) >> 3)) - 1) << (split >> 3));
}

// Molecules that share a tile or a strip meet in the matrix instructions (0 x NaN = NaN): a
// non-finite Ritz entry (a degenerate molecule of AdaLanczosNet's in-model Lanczos layer, whose
// learned Laplacian is 0 / 0 when every node carries the same embedding) is staged as 0, so that
// it stays with its own molecule, as it does in the reference's batched products.
__device__ __forceinline__ float finite_or_zero(float v) {
  return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u ? 0.0f : v;
}

// Ritz tile [node row][slot row] of one node tile, block diagonal: element (jj, rho) belongs to
// the molecule owning BOTH rows, V[mol][local node][local slot] (zero elsewhere / beyond N, K).
__device__ __forceinline__ float ritz_tile_elem(KArgs& a, const TileDesc& t, int jj, int rho) {
  const bool first = jj < t.split, sfirst = rho < t.split;
  const int row = first ? jj : jj - t.split;
  const int k = sfirst ? rho : rho - t.split;
  const int mol = first ? t.ta : t.tb;
  const bool ok = sfirst == first && k < a.K && row < a.N && mol >= 0;
  return ok ? finite_or_zero(a.V[((int64_t)mol * a.N + row) * a.K + k]) : 0.0f;
}

}  // namespace
