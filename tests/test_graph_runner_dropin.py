"""Drop-in proof for the GRAPH experiment's caller: the reference's `GraphRunner.train()`
(runner/graph_runner.py:36-264, config/graph_lanczos_net.yaml: LanczosNetGeneral, K = 20, batch 10,
Adam 1e-4) with the HIP `LanczosNetGeneral` and the `GraphData` mirror in place of the reference
classes, against the SAME run of the unmodified reference stack on the CPU
(tests/golden/runner_graph.npz, tests/golden/make_golden_graph_runner.py: graphs from the
reference's own generator, every training loss, every validation MSE).

Two legs, as for the QM8 runner (tests/test_runner_dropin.py):
  CPU  `oracle/graph_runner.py` (the restatement that drives the product module on the GPU box) is
       bit-identical to the unmodified reference runner when both drive the reference's CPU model
       and dataset; it also fails in `test()` exactly where the reference does.
  GPU  the restated runner + `lanczosnet_amd.model.LanczosNetGeneral` (graphs of 20..100 nodes:
       streamed HIP kernels for the validation passes, the differentiable device-side restatement
       for `loss.backward()`) under the runner's `nn.DataParallel(...).cuda()`, on pickles written
       in the reference's format by the oracle's preprocessing, reproduce the fixture."""
import os
import pickle

import numpy as np
import pytest
import torch

import oracle
import oracle.graph_runner as restated
import runner_harness as H
from conftest import load_golden
from graph_fixture import unpack_adj


def _splits(g):
  out = {}
  for s in ('train', 'dev'):
    n, off = g[s + '_n_nodes'], g[s + '_adj_off']
    roff = np.cumsum([0] + [int(x) for x in n])
    out[s] = [dict(adjs=unpack_adj(g[s + '_adj_bits'][off[b]:off[b + 1]], int(n[b]))[:, :, None],
                   node_feat=g[s + '_node_feat'][roff[b]:roff[b + 1]],
                   label=g[s + '_label'][b:b + 1]) for b in range(len(n))]
  return out


def _write_pickles(path, splits):
  """dataset/get_graph_data.py:51-92 through the oracle restatements, files numbered in the
  reference's listing order (the mirror sorts its file list)."""
  os.makedirs(path, exist_ok=True)
  for s, items in splits.items():
    for i, it in enumerate(items):
      adjs = it['adjs'].astype(np.float64)
      simple = adjs.sum(axis=2)
      e, v, L4 = oracle.graph_laplacian_eigs(simple, graph_laplacian_type='L4')
      d = dict(node_feat=it['node_feat'], label=it['label'], L_simple_4=L4,
               L_multi=np.stack([oracle.laplacian_l4(adjs[:, :, c]) for c in range(adjs.shape[2])], axis=2),
               L_simple_6=oracle.get_laplacian(simple, 'L6'), L_simple_7=oracle.get_laplacian(simple, 'L7'),
               D_simple=e, V_simple=v)
      with open(os.path.join(path, 'synthetic_%s_%07d.p' % (s, i)), 'wb') as f:
        pickle.dump(d, f)


def test_graph_runner_fixture_is_consistent():
  g = load_golden('runner_graph.npz')
  assert g['train_loss'].shape == (9,) and g['val_loss'].shape == (3,)   # 3 epochs x 3 iterations
  assert float(g['best_val']) == float(g['val_loss'].min())
  assert len(g['train_n_nodes']) == 30 and len(g['dev_n_nodes']) == 10
  assert g['train_n_nodes'].min() >= 20 and g['train_n_nodes'].max() <= 100
  assert str(g['test_raises']) == 'AttributeError'   # self.const_factor, runner/graph_runner.py:348


@pytest.mark.skipif(not H.have_reference(), reason='needs the reference tree (build container)')
def test_graph_runner_restatement_matches_reference_runner(tmp_path):
  H.import_reference_runner()
  import dataset.graph_data as ref_gd
  import model as ref_model
  import runner.graph_runner as ref_gr
  g = load_golden('runner_graph.npz')
  splits = _splits(g)
  small = dict(train=splits['train'][:12], dev=splits['dev'][:4], test=splits['dev'][4:6])
  _write_pickles(str(tmp_path / 'data'), small)
  torch.set_num_threads(4)
  res = {}
  for who in ('reference', 'restated'):
    cfg = H.graph_config(str(tmp_path / 'data'), str(tmp_path / who), use_gpu=False, max_epoch=2)
    cfg.train['batch_size'] = 4
    H.seed_like_run_exp(1234)
    with H.numpy_expand_dims_compat():
      runner = ref_gr.GraphRunner(cfg) if who == 'reference' else restated.GraphRunner(
          cfg, dict(LanczosNetGeneral=ref_model.LanczosNetGeneral, GraphData=ref_gd.GraphData))
      best = runner.train()
      with pytest.raises(AttributeError):
        runner.test()
    stats = pickle.load(open(os.path.join(cfg.save_dir, 'train_stats.p'), 'rb'))
    res[who] = (np.asarray(stats['train_loss']), np.asarray(stats['val_loss']), best)
  a, b = res['reference'], res['restated']
  assert len(a[0]) == 6 and len(a[1]) == 2
  np.testing.assert_array_equal(a[0], b[0])
  np.testing.assert_array_equal(a[1], b[1])
  assert a[2] == b[2]


@pytest.mark.gpu
def test_reference_graph_runner_loop_on_the_hip_module_reproduces_the_reference_run(tmp_path):
  from lanczosnet_amd.dataset.graph_data import GraphData
  from lanczosnet_amd.model import LanczosNetGeneral
  g = load_golden('runner_graph.npz')
  _write_pickles(str(tmp_path / 'data'), _splits(g))
  cfg = H.graph_config(str(tmp_path / 'data'), str(tmp_path / 'exp'), use_gpu=True,
                       max_epoch=int(g['max_epoch']))
  H.seed_like_run_exp(int(g['seed']))
  runner = restated.GraphRunner(cfg, dict(LanczosNetGeneral=LanczosNetGeneral, GraphData=GraphData))
  best = runner.train()
  stats = pickle.load(open(os.path.join(cfg.save_dir, 'train_stats.p'), 'rb'))
  tl, vl = np.asarray(stats['train_loss']), np.asarray(stats['val_loss'])
  rel = np.abs(tl - g['train_loss']) / np.abs(g['train_loss'])
  relv = np.abs(vl - g['val_loss']) / np.abs(g['val_loss'])
  print('graph runner: train-loss rel dev max %.2e (first %.2e, last %.2e); val MSE rel dev %.2e'
        % (rel.max(), rel[0], rel[-1], relv.max()))
  assert tl.shape == g['train_loss'].shape and vl.shape == g['val_loss'].shape
  assert rel[0] < 1e-5           # first iteration: forward parity
  assert rel.max() < 5e-4        # 9 Adam steps: fp32 rounding differences grow through 1/sqrt(v)
  assert relv[0] < 1e-5 and relv.max() < 5e-4   # validation: the streamed HIP kernels
  assert abs(best - float(g['best_val'])) < 5e-4 * float(g['best_val'])
