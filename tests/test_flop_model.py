"""bench.py's `roofline` prices the fused forward at the matrix-core instructions it ISSUES for the
batch's tile plan (lanczosnet_amd/utils/flop_model.py).  This test holds that model against the
hardware counter: profiles/r03_forward_pmc.json is `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32` of
the bench command on an MI355X (tools/pmc_forward_profile.py), profiles/r03_forward_tile_plan.npz
the tile plan (per tile: node counts, split row, identity-channel bits) of the same run."""
import json
import os

import numpy as np

from lanczosnet_amd.utils.flop_model import (forward16_mfma_issued, forward16_selected,
                                             forward_mfma_issued, strip_mfma_issued, strips_from_plan,
                                             strips_selected, tiles_from_plan)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QM8_CFG = dict(num_atom=70, num_bond_type=6, short_diffusion_dist=[],
               long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30], num_eig_vec=20,
               spectral_filter_kind='MLP', input_dim=64, hidden_dim=[128] * 7, output_dim=16,
               num_layer=7)


def test_issued_mfma_model_agrees_with_the_pmc_counter():
  pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r03_forward_pmc.json')))
  plan = np.load(os.path.join(ROOT, 'profiles', 'r03_forward_tile_plan.npz'))
  tiles = [dict(nA=int(a), nB=int(b), split=int(s), ident=int(i), pg=int(pg), ps=int(ps))
           for a, b, s, i, pg, ps in zip(plan['nA'], plan['nB'], plan['split'], plan['ident'],
                                         plan['pg'], plan['ps'])]
  fm = forward_mfma_issued(tiles, QM8_CFG)
  measured = pmc['SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch']
  assert abs(fm['mops_counts'] - measured) <= 0.01 * measured, (fm['mops_counts'], measured)
  assert abs(fm['flops_issued'] - pmc['mfma_flops_per_launch']) <= 0.01 * pmc['mfma_flops_per_launch']
  # the skipped k-groups and identity channels are real: the unskipped count is what r02 priced
  assert fm['mfma_unskipped'] > fm['mfma_issued']
  assert 0.5 < fm['useful_row_frac'] < 1.0


def test_issued_mfma_model_of_the_16x16_tile_kernel_agrees_with_the_pmc_counter():
  """lanczosnet_forward16_kernel (the default inference forward of the QM8 model, round 4):
  profiles/r04_forward16_pmc.json / r04_forward16_tile_plan.npz, same tool and command."""
  pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r04_forward16_pmc.json')))
  assert pmc['kernel'].startswith('lanczosnet_forward16_kernel')
  plan = np.load(os.path.join(ROOT, 'profiles', 'r04_forward16_tile_plan.npz'))
  tiles = [dict(nA=int(a), nB=int(b), split=int(s), ident=int(i))
           for a, b, s, i in zip(plan['nA'], plan['nB'], plan['split'], plan['ident'])]
  fm = forward16_mfma_issued(tiles, QM8_CFG)
  measured = pmc['SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch']
  assert abs(fm['mops_counts'] - measured) <= 0.002 * measured, (fm['mops_counts'], measured)
  assert fm['mfma_unskipped'] > fm['mfma_issued']
  # the recorded subtile counts are the ones the model derives from the extents
  rec = forward16_mfma_issued([dict(t, pg16=int(g), ps16=int(p))
                               for t, g, p in zip(tiles, plan['pg16'], plan['ps16'])], QM8_CFG)
  assert rec['mfma_issued'] == fm['mfma_issued']


def test_issued_mfma_model_of_the_strip_kernel_agrees_with_the_pmc_counter():
  """lanczosnet_strip_kernel (the default inference forward of the bench batch since the strip
  plan): profiles/r04_strip_pmc.json / r04_strip_plan.npz (the int32 strip plan and the identity
  bits of the timed batch), same tool and command."""
  pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r04_strip_pmc.json')))
  assert pmc['kernel'].startswith('lanczosnet_strip_kernel')
  rec = np.load(os.path.join(ROOT, 'profiles', 'r04_strip_plan.npz'))
  strips = strips_from_plan(rec['strips'], rec['ident'])
  fm = strip_mfma_issued(strips, QM8_CFG)
  measured = pmc['SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch']
  assert abs(fm['mops_counts'] - measured) <= 0.002 * measured, (fm['mops_counts'], measured)
  # branch-free block products: every neighbouring subtile pair is multiplied; 94 % of the
  # instructions sit in GEMM1 or in blocks some molecule touches (what bench.py prices)
  assert fm['mfma_unskipped'] == fm['mfma_issued']
  assert 0.9 * fm['mfma_issued'] < fm['mfma_in_touched_blocks'] <= fm['mfma_issued']
  # 1024 QM8-sized molecules: at most five subtiles on a compute unit, one strip per unit
  assert fm['max_subtiles_per_strip'] == 5 and fm['tiles'] <= 256
  assert sum(t['mols'] for t in strips) == 1024
  assert 0.85 < fm['useful_row_frac'] < 1.0


def test_strip_selection_mirrors_the_launcher(monkeypatch):
  monkeypatch.delenv('LNZ_FORWARD16', raising=False)
  monkeypatch.delenv('LNZ_STRIPS', raising=False)
  assert strips_selected(QM8_CFG, 1024, 26) and strips_selected(QM8_CFG, 2048, 32)
  assert strips_selected(QM8_CFG, 16384, 26) and not strips_selected(QM8_CFG, 64, 48)
  assert not strips_selected(dict(QM8_CFG, short_diffusion_dist=[1]), 1024, 26)
  monkeypatch.setenv('LNZ_STRIPS', '0')
  assert not strips_selected(QM8_CFG, 1024, 26)


def test_forward16_selection_mirrors_the_launcher(monkeypatch):
  monkeypatch.delenv('LNZ_FORWARD16', raising=False)
  assert forward16_selected(QM8_CFG)
  assert not forward16_selected(dict(QM8_CFG, short_diffusion_dist=[1, 2]))
  assert not forward16_selected(dict(QM8_CFG, input_dim=32))
  assert not forward16_selected(dict(QM8_CFG, hidden_dim=[64] * 7))
  monkeypatch.setenv('LNZ_FORWARD16', '0')
  assert not forward16_selected(QM8_CFG)


def test_tile_masks_follow_row_group_mask():
  """tiles_from_plan: 8-row groups of A from row 0 and of B from the split row
  (conv_forward.hip:139 row_group_mask), eigen-slot groups capped at K."""
  plan = np.array([[0, -1, 32], [1, 2, 8], [3, 4, 16], [-1, -1, -1]])
  ext = np.array([26, 7, 20, 16, 9])
  t = tiles_from_plan(plan, ext, ident=np.array([0b0111000, 0b1111110, 0b0011110, 0, 0]), K=20)
  assert [x['pg'] for x in t] == [4, 1 + 3, 2 + 2]
  assert [x['ps'] for x in t] == [3, 1 + 3, 2 + 2]
  assert t[0]['ident'] == 0b0111000 and t[1]['ident'] == 0b0011110 and t[2]['ident'] == 0
  fm = forward_mfma_issued(t, QM8_CFG)
  assert fm['tiles'] == 3 and fm['mfma_issued'] < fm['mfma_unskipped']
  assert abs(fm['useful_row_frac'] - (26 + 7 + 20 + 16 + 9) / 96.0) < 1e-12


def test_split_precision_strip_model_counts_subtile_pairs():
  """strip_split_mfma_issued: a row subtile's block products run over the (2p, 2p+1) subtile pairs its
  neighbourhood {I-1, I, I+1} meets — counted independently here; the launch total of the bench batch
  equals the SQ_INSTS_VALU_MFMA_MOPS_F16 / _F32 counters (profiles/r05_split_strip_pmc.txt)."""
  from lanczosnet_amd.utils.flop_model import strip_split_mfma_issued
  cfg = QM8_CFG
  for S in range(1, 7):
    want_pairs = sum(len({J >> 1 for J in (I - 1, I, I + 1) if 0 <= J < S}) for I in range(S))
    got = strip_split_mfma_issued([dict(sub=S)], cfg)
    C, n_edge, nl = 15, 7, 7
    f16 = sum(8 * 3 * (C * 4 * S + (n_edge + 1 + (1 if l + 1 < nl else 0)) * want_pairs) for l in range(nl))
    assert got['mfma_f16_issued'] == f16, S
    assert got['mfma_f32_issued'] == 8 * 4 * (3 * S - 2) + 64 * S, S
  # the bench batch's plan: 243 strips of five subtiles and one of two
  tot = strip_split_mfma_issued([dict(sub=5)] * 243 + [dict(sub=2)], cfg)
  assert tot['mfma_f16_issued'] == 15524592 and tot['mfma_f32_issued'] == 179104
