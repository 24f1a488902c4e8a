#!/usr/bin/env python
"""Reference-side fixture of the GRAPH runner drop-in test: the UNMODIFIED
`runner.graph_runner.GraphRunner.train()` + `dataset.graph_data.GraphData` + `LanczosNetGeneral` on
graphs produced by the reference's own generator (`dataset/get_graph_data.py` gen_data + dump_data),
`config/graph_lanczos_net.yaml` (batch 10, Adam 1e-4, K = 20) with max_epoch = 3 and a validation
every epoch, on the CPU.  Stored: every training loss, every validation MSE, and the graphs in the
order the reference's `glob` listed them.  (`GraphRunner.test()` raises AttributeError in the
reference — `self.const_factor`, runner/graph_runner.py:348 — so there is no test number.)

    python tests/golden/make_golden_graph_runner.py     # needs /root/reference; writes runner_graph.npz
"""
import glob
import os
import pickle
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import runner_harness as H  # noqa: E402
from make_golden_graph import pack_adj  # noqa: E402

N_TRAIN, N_DEV, MAX_EPOCH, SEED = 30, 10, 3, 1234


def main():
  H.import_reference_runner()
  import runner.graph_runner as ref_gr
  torch.set_num_threads(4)
  import networkx as nx
  if not hasattr(nx, 'to_numpy_matrix'):
    nx.to_numpy_matrix = lambda g: np.asmatrix(nx.to_numpy_array(g))
  scratch = tempfile.mkdtemp(prefix='lnz_graph_runner_')
  cwd = os.getcwd()
  out = {}
  try:
    os.makedirs(os.path.join(scratch, 'work'))
    os.makedirs(os.path.join(scratch, 'data'))
    os.chdir(os.path.join(scratch, 'work'))
    import dataset.get_graph_data as gg
    gg.dump_data(gg.gen_data(num_graphs=N_TRAIN, seed=123), 'train')
    gg.dump_data(gg.gen_data(num_graphs=N_DEV, seed=456), 'dev')
    gg.dump_data(gg.gen_data(num_graphs=4, seed=789), 'test')   # only to show how test() fails
    data_path = os.path.abspath(gg.save_dir)
    config = H.graph_config(data_path, os.path.join(scratch, 'exp'), use_gpu=False,
                            max_epoch=MAX_EPOCH)
    for s, k in (('train', N_TRAIN), ('dev', N_DEV)):
      files = glob.glob(os.path.join(data_path, 'synthetic_%s_*.p' % s))   # the reference's order
      assert len(files) == k
      items = [pickle.load(open(f, 'rb')) for f in files]
      adj = []
      for it in items:
        a = (np.asarray(it['L_simple_4']) != 0).astype(np.float32)
        np.fill_diagonal(a, 0)
        adj.append(a)
      out[s + '_n_nodes'] = np.array([a.shape[0] for a in adj], np.int32)
      out[s + '_adj_bits'] = np.concatenate([pack_adj(a) for a in adj])
      out[s + '_adj_off'] = np.cumsum([0] + [len(pack_adj(a)) for a in adj]).astype(np.int64)
      out[s + '_node_feat'] = np.concatenate([np.asarray(it['node_feat']) for it in items])
      out[s + '_label'] = np.concatenate([np.asarray(it['label']) for it in items], axis=0)
    H.seed_like_run_exp(SEED)
    with H.numpy_expand_dims_compat():
      best = ref_gr.GraphRunner(config).train()
    stats = pickle.load(open(os.path.join(config.save_dir, 'train_stats.p'), 'rb'))
    out.update(seed=SEED, max_epoch=MAX_EPOCH, train_loss=np.asarray(stats['train_loss'], np.float64),
               val_loss=np.asarray(stats['val_loss'], np.float64), best_val=float(best),
               torch_version=np.array(torch.__version__))
    try:
      with H.numpy_expand_dims_compat():
        ref_gr.GraphRunner(config).test()
      out['test_raises'] = np.array('')
    except Exception as e:  # noqa: BLE001
      out['test_raises'] = np.array(type(e).__name__)
  finally:
    os.chdir(cwd)
    shutil.rmtree(scratch, ignore_errors=True)
  print('train', out['train_loss'], 'val', out['val_loss'], 'test() raises', out['test_raises'])
  path = os.path.join(HERE, 'runner_graph.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
