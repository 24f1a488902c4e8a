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

`lanczosnet_forward16_kernel` (csrc/conv_forward16.hip: the same algebra on
`v_mfma_f32_16x16x4_f32`, 2048 flop = 4 counter units; eight waves of 16 output columns on every tile
of the workgroup) issues per tile, wave and layer

    GEMM1      (n_long + n_edge) * d_in / 2                  (8 instructions per 16-k step)
    lift-back  8 * popcount(slot subtiles)                   16-row subtiles that hold Ritz pairs
    GEMM2      (n_edge - n_ident) * 8 * popcount(row subtiles)
    projection 8 * popcount(row subtiles)                    (not the last layer)

plus the first layer's projection (waves whose 16 columns lie inside din0) and the head (one wave per
tile, dhid / 2 instructions of the 32x32x2 kind).  It takes the launches `forward16_selected` says.

`lanczosnet_strip_kernel` (csrc/conv_strip.hip: the inference forward on the strip plan of
lnz_plan_strips — S subtiles of 16 rows per workgroup, molecules at 4-row granularity) issues per
strip, wave and layer

    GEMM1      (n_long + n_edge) * d_in / 16 * 4 * S
    lift-back  4 * blocks                                    blocks = 3 S - 2: every subtile pair (I, J),
    GEMM2      4 * n_edge * blocks                           |I - J| <= 1 (branch free: a pair no molecule
    projection 4 * blocks      (not the last layer)          touches, or an identity channel's, has zero
                                                             fragments and is multiplied all the same)

plus the first layer's projection (din0 / 16 waves) and the head (S waves, 64 instructions), all of
the 16x16x4 kind.  It takes the launches `strips_selected` says.
"""
import os

import numpy as np

FLOP_PER_MFMA = 2 * 32 * 32 * 2          # v_mfma_f32_32x32x2_f32
FLOP_PER_MFMA16 = 2 * 16 * 16 * 4        # v_mfma_f32_16x16x4_f32
FLOP_PER_MOPS_COUNT = 512                # SQ_INSTS_VALU_MFMA_MOPS_F32 unit


def _row_groups(n_a, n_b, split):
  ga = (n_a + 7) >> 3
  gb = ((n_b + 7) >> 3) if n_b > 0 else 0
  return ga + gb if split < 32 else ga   # B's groups start at split / 8: disjoint from A's


def _subtiles16(n_a, n_b, split):
  """16-row subtiles of a tile that hold a row of A (from row 0) or B (from the split row):
  conv_forward16.hip live16."""
  mask = (1 << ((n_a + 7) >> 3)) - 1
  if split < 32 and n_b > 0:
    mask |= ((1 << ((n_b + 7) >> 3)) - 1) << (split >> 3)
  return (1 if mask & 3 else 0) + (1 if mask & 12 else 0)


def forward16_selected(cfg):
  """Mirror of lnz::forward16_eligible + the LNZ_FORWARD16 switch (csrc/conv_forward.hip): does the
  exact-fp32 inference forward of this LanczosNet model run on the 16 x 16-tile kernel?"""
  if os.environ.get('LNZ_FORWARD16', '1') in ('0',):
    return False
  hid = cfg['hidden_dim']
  return (len(cfg['short_diffusion_dist']) == 0 and all(h == 128 for h in hid) and
          cfg['input_dim'] % 64 == 0 and cfg['input_dim'] <= 128 and
          len(cfg['long_diffusion_dist']) <= 12)


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
    sp = int(split) if tb >= 0 else 32
    out.append(dict(nA=n_a, nB=n_b, split=sp, ident=idb,
                    pg=_row_groups(n_a, n_b, sp), ps=_row_groups(min(n_a, K), min(n_b, K), sp),
                    pg16=_subtiles16(n_a, n_b, sp), ps16=_subtiles16(min(n_a, K), min(n_b, K), sp)))
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


def forward16_mfma_issued(tiles, cfg):
  """The same record as forward_mfma_issued for lanczosnet_forward16_kernel.  `mfma_issued` counts
  v_mfma_f32_16x16x4_f32 instructions (2048 flop) plus the head's 32x32x2 instructions counted
  double (4096 flop): flops_issued = 2048 * mfma_issued."""
  n_long = len(cfg['long_diffusion_dist'])
  n_edge = cfg['num_bond_type'] + 1
  C = n_long + n_edge
  din0, dhid, nl = cfg['input_dim'], cfg['hidden_dim'][0], cfg['num_layer']
  assert len(cfg['short_diffusion_dist']) == 0 and dhid == 128

  def per_tile(pg, ps, n_ident):
    tot = 0
    for l in range(nl):
      d_in = din0 if l == 0 else dhid
      per_wave = C * (d_in // 2) + (8 * ps if n_long else 0) + (n_edge - n_ident) * 8 * pg
      if n_long and l + 1 < nl:
        per_wave += 8 * pg
      tot += 8 * per_wave
    if n_long:
      tot += (din0 // 16) * 8 * pg                   # first layer's projection
    tot += 2 * (dhid // 2)                           # head: one wave per tile, 32x32x2
    return tot

  issued = 0
  rows_real = 0
  emask = (1 << n_edge) - 1
  K = cfg['num_eig_vec']
  for t in tiles:
    pg16 = t['pg16'] if 'pg16' in t else _subtiles16(t['nA'], t['nB'], t['split'])
    ps16 = t['ps16'] if 'ps16' in t else _subtiles16(min(t['nA'], K), min(t['nB'], K), t['split'])
    issued += per_tile(pg16, ps16, bin(t['ident'] & emask).count('1'))
    rows_real += t['nA'] + t['nB']
  full = per_tile(2, 2, 0) * len(tiles)
  return dict(tiles=len(tiles), mfma_issued=int(issued), mfma_unskipped=int(full),
              flops_issued=int(issued) * FLOP_PER_MFMA16, flops_unskipped=int(full) * FLOP_PER_MFMA16,
              mops_counts=int(issued) * (FLOP_PER_MFMA16 // FLOP_PER_MOPS_COUNT),
              useful_row_frac=rows_real / (32.0 * max(1, len(tiles))))


STRIP_INTS = 80   # LNZ_STRIP_INTS


def strips_selected(cfg, B, N):
  """Mirror of lnz::strip_forward_eligible + the LNZ_STRIPS switch for launches that carry a strip
  plan (every batch with N <= 32: ops.strip_plan_wanted)."""
  if os.environ.get('LNZ_STRIPS', '1') == '0' or not forward16_selected(cfg):
    return False
  return N <= 32


def strips_from_plan(strips, ident=None):
  """strips: the int32 array written by lnz_plan_strips ([cap * 80 + 1], last word = strips in use);
  ident [B] = identity-channel bits of the edge channels or None.  Returns one dict per strip:
  subtiles, blocks (number of subtile pairs some molecule touches), per-subtile block counts and
  identity bits, real rows."""
  p = np.asarray(strips)
  cap = (p.size - 1) // STRIP_INTS
  out = []
  for s in range(int(p[cap * STRIP_INTS])):
    e = p[s * STRIP_INTS:(s + 1) * STRIP_INTS]
    nm, sub = int(e[0]), int(e[1])
    owner = -np.ones(16 * sub, int)
    idb = [0xffffffff if ident is not None else 0] * sub
    rows_real = 0
    for i in range(nm):
      b, st, n = (int(x) for x in e[2 + 3 * i:5 + 3 * i])
      rows = 4 if n <= 4 else (n + 3) // 4 * 4
      owner[st:st + rows] = i
      rows_real += n
      if ident is not None:
        for I in range(st // 16, (st + rows - 1) // 16 + 1):
          idb[I] &= int(ident[b]) & 0xffffffff
    blk = []
    for I in range(sub):
      mine = set(owner[16 * I:16 * I + 16]) - {-1}
      blk.append(sum(1 for J in (I - 1, I, I + 1)
                     if 0 <= J < sub and mine & (set(owner[16 * J:16 * J + 16]) - {-1})))
    out.append(dict(sub=sub, blk=blk, ident=idb, rows_real=rows_real, mols=nm))
  return out


def strip_mfma_issued(strips, cfg):
  """The record of forward_mfma_issued for lanczosnet_strip_kernel; `mfma_issued` counts
  v_mfma_f32_16x16x4_f32 instructions (2048 flop)."""
  n_long = len(cfg['long_diffusion_dist'])
  n_edge = cfg['num_bond_type'] + 1
  C = n_long + n_edge
  din0, dhid, nl = cfg['input_dim'], cfg['hidden_dim'][0], cfg['num_layer']
  din0 = (din0 + 63) // 64 * 64
  emask = (1 << n_edge) - 1
  issued = full = rows_real = rows_tile = 0
  useful = 0   # the instructions of blocks some molecule touches, identity channels left out
  for t in strips:
    S, touched = t['sub'], sum(t['blk'])
    blocks = 3 * S - 2
    g2_useful = sum(b * (n_edge - bin(i & emask).count('1')) for b, i in zip(t['blk'], t['ident']))
    for l in range(nl):
      d_in = din0 if l == 0 else dhid
      proj = 1 if (n_long and l + 1 < nl) else 0
      issued += 8 * (C * (d_in // 16) * 4 * S + 4 * blocks * ((1 if n_long else 0) + proj) + 4 * n_edge * blocks)
      useful += 8 * (C * (d_in // 16) * 4 * S + 4 * touched * ((1 if n_long else 0) + proj) + 4 * g2_useful)
    if n_long:
      issued += (din0 // 16) * 4 * blocks
      useful += (din0 // 16) * 4 * touched
    issued += S * 64
    useful += S * 64
    full = issued
    rows_real += t['rows_real']
    rows_tile += 16 * S
  return dict(tiles=len(strips), mfma_issued=int(issued), mfma_unskipped=int(full),
              flops_issued=int(issued) * FLOP_PER_MFMA16, flops_unskipped=int(full) * FLOP_PER_MFMA16,
              mops_counts=int(issued) * (FLOP_PER_MFMA16 // FLOP_PER_MOPS_COUNT),
              useful_row_frac=rows_real / float(max(1, rows_tile)),
              mfma_in_touched_blocks=int(useful),
              subtiles=int(sum(t['sub'] for t in strips)),
              max_subtiles_per_strip=int(max(t['sub'] for t in strips)))


FLOP_PER_MFMA16X32_F16 = 2 * 16 * 16 * 32


def strip_split_mfma_issued(strips, cfg):
  """lanczosnet_strip_kernel<.., HALF> (gemm_mode 1): v_mfma_f32_16x16x32_f16 instructions (16384 flop)
  one launch issues — per layer and wave GEMM1 = channels x four 32-k blocks x three products x S
  subtiles (every layer is 128 wide there), and each block-diagonal product (the edge types' GEMM2,
  the lift, the next layer's projection) = three products per subtile PAIR a row subtile touches.
  The first layer's projection and the head stay on v_mfma_f32_16x16x4_f32 (counted apart)."""
  n_long = len(cfg['long_diffusion_dist'])
  n_edge = cfg['num_bond_type'] + 1
  C = n_long + n_edge
  nl = cfg['num_layer']

  def pairs(S):   # (2 p, 2 p + 1) pairs over the row subtiles' neighbourhoods {I - 1, I, I + 1}
    n = 0
    for I in range(S):
      p0 = max(I - 1, 0) >> 1
      n += 1
      if 2 * (p0 + 1) < S and 2 * (p0 + 1) <= I + 1:
        n += 1
    return n

  f16 = f32 = 0
  for t in strips:
    S = t['sub']
    for l in range(nl):
      prods = n_edge + (1 if n_long else 0) + (1 if (n_long and l + 1 < nl) else 0)
      f16 += 8 * 3 * (C * 4 * S + prods * pairs(S))
    if n_long:
      f32 += 8 * 4 * (3 * S - 2)
    f32 += S * 64
  return dict(mfma_f16_issued=int(f16), mfma_f32_issued=int(f32),
              flops_issued=int(f16) * FLOP_PER_MFMA16X32_F16 + int(f32) * FLOP_PER_MFMA16)
