"""Matrix-core work the fused forward kernel ISSUES for a given tile plan (measurement helper, pure
numpy: used by bench.py's `roofline` and checked against `rocprofv3 --pmc
SQ_INSTS_VALU_MFMA_MOPS_F32` in tests/test_flop_model.py).

`lanczosnet_forward_kernel<4,*,0,0>` (csrc/conv_forward.hip, diagonal gains, eigen-space long
channels) runs 32-row node tiles; a tile's four wavefronts own 32 output columns each and issue, per
conv layer with input width d_in,

    GEMM1      (n_short + n_long + n_edge) * d_in / 2        X W_c^T, never skipped
    lift-back  4 * popcount(smask)                           V T, 8-slot groups that hold Ritz pairs
    GEMM2      (n_edge - n_ident) * 4 * popcount(g2mask)     M_c Z_c, 8-row groups that hold real
                                                             nodes; identity channels add Z instead
    projection 4 * popcount(g2mask)                          V^T X' for the next layer (not the last)

`v_mfma_f32_32x32x2_f32` instructions, plus the first layer's projection (waves whose 32 columns lie
inside din0) and the head (one wave per tile, dhid / 2 instructions).  g2mask / smask follow
`row_group_mask` (conv_forward.hip:139): molecule A fills groups from row 0, molecule B from the
split row.  One instruction = 2 * 32 * 32 * 2 = 4096 flop = 8 counts of the PMC counter (512 flop
per count).
"""
import numpy as np

FLOP_PER_MFMA = 2 * 32 * 32 * 2          # v_mfma_f32_32x32x2_f32
FLOP_PER_MOPS_COUNT = 512                # SQ_INSTS_VALU_MFMA_MOPS_F32 unit


def _row_groups(n_a, n_b, split):
  ga = (n_a + 7) >> 3
  gb = ((n_b + 7) >> 3) if n_b > 0 else 0
  return ga + gb if split < 32 else ga   # B's groups start at split / 8: disjoint from A's


def tiles_from_plan(plan_entries, extents, ident=None, K=20):
  """plan_entries: int array [cap * 4, 3] (molecule A, molecule B | -1, split row) as written by
  lnz_plan_tiles; extents [B] = last real node + 1 per molecule; ident [B] = identity-channel
  bits of the edge channels (Lp.ident) or None.  Returns one dict per tile in use."""
  out = []
  pe = np.asarray(plan_entries).reshape(-1, 3)
  ext = np.asarray(extents)
  for ta, tb, split in pe:
    if ta < 0:
      continue
    n_a = int(ext[ta])
    n_b = int(ext[tb]) if tb >= 0 else 0
    idb = 0
    if ident is not None:
      idb = int(ident[ta]) & 0xffffffff
      if tb >= 0:
        idb &= int(ident[tb]) & 0xffffffff
    out.append(dict(nA=n_a, nB=n_b, split=int(split) if tb >= 0 else 32, ident=idb,
                    pg=_row_groups(n_a, n_b, int(split) if tb >= 0 else 32),
                    ps=_row_groups(min(n_a, K), min(n_b, K), int(split) if tb >= 0 else 32)))
  return out


def forward_mfma_issued(tiles, cfg, nwv=4):
  """tiles: tiles_from_plan(...) output; cfg: model dict (input_dim, hidden_dim, num_layer,
  short/long diffusion lists, num_bond_type).  Returns a dict with the issued instruction count,
  the count a full 32-row single-molecule tile would issue (no skipping), flops for both and the
  fraction of tile rows that hold real nodes."""
  n_short, n_long = len(cfg['short_diffusion_dist']), len(cfg['long_diffusion_dist'])
  n_edge = cfg['num_bond_type'] + 1
  C = n_short + n_long + n_edge
  din0, dhid, nl = cfg['input_dim'], cfg['hidden_dim'][0], cfg['num_layer']
  assert n_short == 0, 'short-diffusion powers issue p-1 extra GEMM2 chains: not modelled'

  def per_tile(pg, ps, n_ident):
    tot = 0
    for l in range(nl):
      d_in = din0 if l == 0 else dhid
      per_wave = C * (d_in // 2) + (4 * ps if n_long else 0) + (n_edge - n_ident) * 4 * pg
      if n_long and l + 1 < nl:
        per_wave += 4 * pg
      tot += nwv * per_wave
    if n_long:
      tot += min(nwv, (din0 + 31) // 32) * 4 * pg   # first layer's projection
    tot += dhid // 2                                 # head: one wave per tile
    return tot

  issued = 0
  rows_real = 0
  emask = (1 << n_edge) - 1
  for t in tiles:
    issued += per_tile(t['pg'], t['ps'], bin(t['ident'] & emask).count('1'))
    rows_real += t['nA'] + t['nB']
  full = per_tile(4, 4 if cfg['num_eig_vec'] > 24 else (cfg['num_eig_vec'] + 7) // 8, 0) * len(tiles)
  return dict(tiles=len(tiles), mfma_issued=int(issued), mfma_unskipped=int(full),
              flops_issued=int(issued) * FLOP_PER_MFMA, flops_unskipped=int(full) * FLOP_PER_MFMA,
              mops_counts=int(issued) * (FLOP_PER_MFMA // FLOP_PER_MOPS_COUNT),
              useful_row_frac=rows_real / (32.0 * max(1, len(tiles))))
