"""Packed QM8 shard + device-side collate (SURVEY.md §8f rank 1 + 3).

CPU: file format round trip, bond recovery from the reference's pickle fields, and the oracle
restatement of the collate against the reference-generated golden batch.
GPU: lnz_collate_qm8 (+ lnz_lanczos_ritz) against the same golden batch, bit-identical to
lnz_laplacian_l4 on the dense adjacency, and through the model at B = 1024."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from lanczosnet_amd.dataset import (PackedQM8, edges_from_dense, edges_from_laplacians,  # noqa: E402
                                    write_packed)
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402
from tests.conftest import load_golden, rel_err  # noqa: E402


def _molecules(batch):
  out = []
  for b in range(batch['adjs'].shape[0]):
    n = int(batch['n_nodes'][b])
    out.append(dict(node_feat=batch['node_feat'][b, :n], adjs=batch['adjs'][b, :n, :n, :],
                    label=batch['label'][b]))
  return out


def _golden_batch():
  g = load_golden('collate_batch.npz')
  batch = draw_batch(int(g['batch_size']), seed=int(g['seed']), n_min=int(g['n_min']),
                     n_max=int(g['n_max']))
  return g, batch


def test_packed_file_round_trip(tmp_path):
  g, batch = _golden_batch()
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), num_bond_type=6, num_label=16)
  ds = PackedQM8(path)
  assert len(ds) == 24 and ds.num_bond_type == 6 and ds.num_label == 16
  assert ds.max_nodes == int(batch['n_nodes'].max())
  np.testing.assert_array_equal(ds.sizes, batch['n_nodes'])
  for b in range(24):
    n = int(batch['n_nodes'][b])
    atoms, edges, label = ds.molecule(b)
    np.testing.assert_array_equal(atoms, batch['node_feat'][b, :n])
    np.testing.assert_array_equal(label, batch['label'][b])
    np.testing.assert_array_equal(oracle.dense_from_edges(n, edges, 6), batch['adjs'][b, :n, :n])
  # ~160 B per molecule instead of a dense-Laplacian pickle
  assert os.path.getsize(path) < 24 * 400 + 1024


def test_packed_file_rejects_garbage(tmp_path):
  bad = tmp_path / 'bad.lnzq'
  bad.write_bytes(b'not a shard' * 20)
  with pytest.raises(ValueError):
    PackedQM8(str(bad))
  g, batch = _golden_batch()
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), 6, 16)
  raw = open(path, 'rb').read()
  trunc = tmp_path / 'trunc.lnzq'
  trunc.write_bytes(raw[:len(raw) // 2])
  with pytest.raises(ValueError):
    PackedQM8(str(trunc))


def test_bonds_recovered_from_reference_pickle_fields():
  """A reference pickle stores L_multi (L4 per bond type), not the adjacency: its off-diagonal
  support gives the bonds back (dataset/get_qm8_data.py:63-75)."""
  _, batch = _golden_batch()
  for b in range(24):
    n = int(batch['n_nodes'][b])
    adjs = batch['adjs'][b, :n, :n, :]
    L_multi = oracle.laplacian_multi_l4(adjs)[:, :, 1:]
    np.testing.assert_array_equal(np.sort(edges_from_laplacians(L_multi)),
                                  np.sort(edges_from_dense(adjs)))


def test_oracle_packed_collate_matches_reference(tmp_path):
  g, batch = _golden_batch()
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), 6, 16)
  ds = PackedQM8(path)
  out = oracle.collate_packed([ds.molecule(b) for b in range(24)], 6, 20)
  np.testing.assert_array_equal(out['node_feat'], g['node_feat'])
  np.testing.assert_array_equal(out['node_mask'], g['node_mask'])
  np.testing.assert_array_equal(out['label'], g['label'])
  np.testing.assert_allclose(out['L'], g['L'], rtol=0, atol=1e-7)
  np.testing.assert_allclose(out['D'], g['D'], rtol=0, atol=1e-6)
  for p in (1, 5, 30):
    assert rel_err(oracle.spectral_projector(out['D'], out['V'], p),
                   oracle.spectral_projector(g['D'], g['V'], p)) < 1e-5


def test_collate_needs_a_device(tmp_path):
  _, batch = _golden_batch()
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), 6, 16)
  ds = PackedQM8(path)
  with pytest.raises(RuntimeError):
    ds.collate([0, 1], 20)
  with pytest.raises(RuntimeError):
    ds.to('cpu')


# ------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_collate_matches_reference_golden(tmp_path):
  g, batch = _golden_batch()
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), 6, 16)
  ds = PackedQM8(path).to('cuda')
  out = ds.collate(np.arange(24), 20)
  np.testing.assert_array_equal(out['node_feat'].cpu().numpy(), g['node_feat'])
  np.testing.assert_array_equal(out['node_mask'].cpu().numpy(), g['node_mask'])
  np.testing.assert_array_equal(out['label'].cpu().numpy(), g['label'])
  np.testing.assert_array_equal(out['n_nodes'].cpu().numpy(), g['n_nodes'])
  np.testing.assert_allclose(out['L'].cpu().numpy(), g['L'], rtol=0, atol=1e-7)
  D, V = out['D'].cpu().numpy(), out['V'].cpu().numpy()
  np.testing.assert_allclose(D, g['D'], rtol=0, atol=2e-6)
  for p in (1, 5, 30):
    assert rel_err(oracle.spectral_projector(D, V, p),
                   oracle.spectral_projector(g['D'], g['V'], p)) < 1e-5
  # a shuffled, repeated and shorter id list: per-batch N, same molecules
  ids = np.array([5, 5, 0, 17, 3])
  sub = ds.collate(ids, 20)
  N = int(g['n_nodes'][ids].max())
  assert sub['L'].shape == (5, N, N, 7)
  np.testing.assert_allclose(sub['L'].cpu().numpy(), g['L'][ids][:, :N, :N], rtol=0, atol=1e-7)
  np.testing.assert_array_equal(sub['label'].cpu().numpy(), g['label'][ids])
  with pytest.raises(IndexError):
    ds.collate([24], 20)
  with pytest.raises(IndexError):
    ds.collate([-1], 20)


@pytest.mark.gpu
def test_device_collate_is_bit_identical_to_the_dense_path_at_1024(tmp_path):
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config
  batch = draw_batch(1024, seed=0)
  path = str(tmp_path / 'shard.lnzq')
  write_packed(path, _molecules(batch), 6, 16)
  ds = PackedQM8(path).to('cuda')
  out = ds.collate(np.arange(1024), 20)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
  n = t(batch['n_nodes'])
  L = ops.laplacian_l4(t(batch['adjs']), n)
  assert torch.equal(out['L'], L)
  assert torch.equal(out['node_feat'], t(batch['node_feat']))
  assert torch.equal(out['node_mask'], t(batch['node_mask']))
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 20)
  assert torch.equal(out['D'], D) and torch.equal(out['V'], V)
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net = LanczosNet(make_model_config(cfg)).eval()
  net.load_state_dict({k: torch.from_numpy(v)
                       for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
  net = net.cuda()
  with torch.no_grad():
    s1, loss = net(out['node_feat'], out['L'], out['D'], out['V'], label=out['label'],
                   mask=out['node_mask'])
  assert torch.isfinite(s1).all() and torch.isfinite(loss)


@pytest.mark.gpu
def test_device_collate_rejects_oversized_tiles():
  from lanczosnet_amd import ops, _lib
  dev = torch.device('cuda')
  shard = dict(mol_off=torch.tensor([0, 2], device=dev), edge_off=torch.tensor([0, 1], device=dev),
               atoms=torch.tensor([1, 2], dtype=torch.uint8, device=dev),
               edges=torch.tensor([0 | 1 << 8], dtype=torch.int32, device=dev),
               labels=torch.zeros((1, 16), device=dev))
  ids = torch.zeros(1, dtype=torch.int64, device=dev)
  with pytest.raises(_lib.NotSupported):
    ops.collate_qm8(shard, ids, 200, 6, 16)
  out = ops.collate_qm8(shard, ids, 4, 6, 16)
  assert out['n_nodes'].tolist() == [2]
