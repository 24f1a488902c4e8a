"""CPU leg of the reference's graph configuration (config/graph_lanczos_net.yaml; fixture
tests/golden/graph_config.npz = the unmodified reference run end to end, see
tests/golden/make_golden_graph.py):

  * the oracle restatement (L4, eigh + |lambda| sort, pad / cut, LanczosNetGeneral.forward)
    reproduces the reference's collated (D, V), Laplacian and scores at BOTH batch sizes of the yaml;
  * the `GraphData` mirror (lanczosnet_amd/dataset/graph_data.py) collates the reference's pickle
    format to the same tensors; with the reference tree present, bit for bit against the
    reference's own `GraphData.collate_fn`;
  * the algorithm of the workgroup-per-graph Ritz kernel (csrc/lanczos_ritz_wg.hip: full-length
    Lanczos + CGS2 + restart + QL), mirrored in numpy, meets the (D, V) tolerances at n ~ 100.
"""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

import oracle
import runner_harness as H
from algo_mirror import lanczos_ritz_mirror
from graph_fixture import GRAPH_CFG, check_ritz, load_split, pad_batch

K = GRAPH_CFG['num_eig_vec']


def _oracle_preprocess(it):
  """dataset/get_graph_data.py:51-83 through the oracle restatements."""
  adjs = it['adjs'].astype(np.float64)
  simple = adjs.sum(axis=2)
  e, v, L4 = oracle.graph_laplacian_eigs(simple, graph_laplacian_type='L4')
  return dict(node_feat=it['node_feat'], label=it['label'],
              L_multi=np.stack([oracle.laplacian_l4(adjs[:, :, i]) for i in range(adjs.shape[2])],
                               axis=2),
              L_simple_4=L4, L_simple_6=oracle.get_laplacian(simple, 'L6'),
              L_simple_7=oracle.get_laplacian(simple, 'L7'), D_simple=e, V_simple=v)


@pytest.mark.parametrize('split', ['train', 'test'])
def test_oracle_reproduces_reference_graph_pipeline(split):
  items, ref, seed, chk = load_split(split)
  adjs, X, mask, n = pad_batch(items)
  B, N = mask.shape
  pre = [_oracle_preprocess(it) for it in items]
  D, V = oracle.collate_eigs([p['D_simple'] for p in pre], [p['V_simple'] for p in pre], N, K)
  # same LAPACK call on the same matrix: the eigenvalues agree to rounding, the vectors up to sign
  assert np.abs(D - ref['D']).max() < 1e-6
  check_ritz(D, V, ref['D'], ref['V'], n, ref['D_full'], K)
  L = np.zeros((B, N, N, 2), np.float32)
  for b, p in enumerate(pre):
    L[b, :n[b], :n[b], 0] = p['L_simple_4']
    L[b, :n[b], :n[b], 1] = p['L_multi'][:, :, 0]
  if split == 'train':
    assert np.abs(L[..., 0] - ref['L0']).max() < 1e-7
  P = oracle.make_lanczosnet_params(GRAPH_CFG, seed, general=True)
  assert abs(sum(float(np.abs(v.astype(np.float64)).sum()) for _, v in sorted(P.items())) - chk) < 1e-6 * chk
  for dt, tol in ((np.float64, 1e-5), (np.float32, 1e-4)):
    score = oracle.lanczos_net_forward(P, GRAPH_CFG, X, L, ref['D'], ref['V'], mask, dtype=dt,
                                       general=True)
    per = np.abs(score - ref['score']).max(axis=1) / np.abs(ref['score']).max(axis=1)
    assert per.max() < tol, (dt, per.max())
  loss = float(((score.astype(np.float64) - ref['label']) ** 2).mean())
  assert abs(loss - ref['loss']) < 1e-4 * ref['loss']


def _write_pickles(path, items, tag):
  os.makedirs(path, exist_ok=True)
  for i, it in enumerate(items):
    with open(os.path.join(path, 'synthetic_%s_%07d.p' % (tag, i)), 'wb') as f:
      pickle.dump(_oracle_preprocess(it), f)


def test_graphdata_mirror_collates_to_the_reference_tensors(tmp_path):
  from lanczosnet_amd.dataset.graph_data import GraphData
  from lanczosnet_amd.utils.arg_helper import AttrDict
  items, ref, _, _ = load_split('train')
  _write_pickles(str(tmp_path), items, 'train')
  cfg = AttrDict(dict(seed=1234, dataset=dict(data_path=str(tmp_path), num_edge_type=1),
                      model=dict(name='LanczosNetGeneral', num_eig_vec=K)))
  ds = GraphData(cfg, split='train')
  assert len(ds) == len(items) and len(GraphData(cfg, split='dev')) == 0
  data = ds.collate_fn([ds[i] for i in range(len(ds))])
  n = ref['n_nodes']
  assert data['node_feat'].dtype == torch.float32 and data['node_mask'].dtype == torch.uint8
  assert data['L'].shape == (10, n.max(), n.max(), 2) and data['L'].dtype == torch.float32
  np.testing.assert_array_equal(data['node_mask'].numpy().sum(axis=1), n)
  np.testing.assert_array_equal(data['label'].numpy(), ref['label'])
  assert np.abs(data['L'].numpy()[..., 0] - ref['L0']).max() < 1e-7
  np.testing.assert_array_equal(data['L'].numpy()[..., 0], data['L'].numpy()[..., 1])
  assert np.abs(data['D'].numpy() - ref['D']).max() < 1e-6
  check_ritz(data['D'].numpy(), data['V'].numpy(), ref['D'], ref['V'], n, ref['D_full'], K)
  # models without eigen inputs get no D / V (dataset/graph_data.py:20-22,262)
  cfg2 = AttrDict(dict(seed=1, dataset=dict(data_path=str(tmp_path), num_edge_type=1),
                       model=dict(name='GCN')))
  d2 = GraphData(cfg2, split='train').collate_fn([ds[0], ds[1]])
  assert 'D' not in d2 and 'V' not in d2
  # DCNN / ChebyNet pick L_simple_7 / -L_simple_6 (:247-260)
  for name, key, sgn in (('DCNN', 'L_simple_7', 1.0), ('ChebyNet', 'L_simple_6', -1.0)):
    cfg3 = AttrDict(dict(seed=1, dataset=dict(data_path=str(tmp_path), num_edge_type=1),
                         model=dict(name=name)))
    d3 = GraphData(cfg3, split='train').collate_fn([ds[0]])
    np.testing.assert_allclose(d3['L'].numpy()[0, :, :, 0], sgn * ds[0][key].astype(np.float32))


@pytest.mark.skipif(not H.have_reference(), reason='needs the reference tree (build container)')
def test_graphdata_mirror_equals_reference_collate_bitwise(tmp_path):
  """Both collates on the SAME pickles (written in the reference's format): every tensor equal."""
  from lanczosnet_amd.dataset.graph_data import GraphData
  from lanczosnet_amd.utils.arg_helper import AttrDict
  H.import_reference_runner()
  import dataset.graph_data as ref_gd
  items, _, _, _ = load_split('train')
  _write_pickles(str(tmp_path), items[:6], 'train')
  cfg = AttrDict(dict(seed=1234, dataset=dict(data_path=str(tmp_path), num_edge_type=1),
                      model=dict(name='LanczosNetGeneral', num_eig_vec=K)))
  mine = GraphData(cfg, split='train')
  theirs = ref_gd.GraphData(cfg, split='train')
  theirs.train_data_files.sort()
  batch = [mine[i] for i in range(len(mine))]
  with H.numpy_expand_dims_compat():
    want = theirs.collate_fn([theirs[i] for i in range(len(theirs))])
  got = mine.collate_fn(batch)
  assert set(got.keys()) == set(want.keys())
  for k in want:
    assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
    assert torch.equal(got[k], want[k]), k


def test_workgroup_kernel_algorithm_meets_tolerances_at_graph_sizes():
  items, ref, _, _ = load_split('train')
  n = ref['n_nodes']
  pick = [int(np.argmax(n)), int(np.argmin(n))]
  for b in pick:
    L4 = oracle.laplacian_l4(items[b]['adjs'][:, :, 0]).astype(np.float32)
    D, V, _ = lanczos_ritz_mirror(L4, K, solver='ql')
    Vp = np.zeros((1, n.max(), K), np.float32)
    Vp[0, :n[b]] = V
    wd, wp, c = check_ritz(D[None], Vp, ref['D'][b:b + 1], ref['V'][b:b + 1], n[b:b + 1],
                           ref['D_full'][b:b + 1], K)
    assert c == 1 and wd < 1e-6 and wp < 1e-5
