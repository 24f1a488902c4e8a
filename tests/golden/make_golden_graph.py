#!/usr/bin/env python
"""Golden fixture for the reference's OWN graph configuration (config/graph_lanczos_net.yaml),
produced by running the UNMODIFIED reference end to end in the build container:

    dataset/get_graph_data.py   gen_data (fast_gnp graphs, n in [20, 100], p = 0.5) + dump_data
                                (get_laplacian 'L4', get_graph_laplacian_eigs = eigh + |lambda| sort)
                                -> the one-pickle-per-graph files
    dataset/graph_data.py       GraphData(config, split).collate_fn           (pad, cut / pad to K)
    model/lanczos_net_general.py LanczosNetGeneral(config).eval()(...)        (score)

at the configuration's two batch sizes: train batch_size 10 (the `train` split, seed 123) and test
batch_size 64 (64 graphs, seed 789).  Stored: the raw graphs (bit-packed adjacency, node features,
labels), the collated (D, V), the full spectra (to recognise top-K cuts through degenerate
clusters), the simple-graph Laplacian channel for the small batch, and the reference scores.

    python tests/golden/make_golden_graph.py        # needs /root/reference; writes graph_config.npz

Patches applied to the ENVIRONMENT only (the reference files are imported as they are):
  * networkx >= 3 dropped `to_numpy_matrix` (get_graph_data.py:41) -> np.asmatrix(to_numpy_array);
  * numpy >= 2 rejects `np.expand_dims(x2d, axis=3)` (graph_data.py:252,262) -> old clamping;
  * get_graph_data.py creates `../data/synthetic/` relative to the cwd at import: the script
    runs inside a scratch directory so that this lands in scratch.
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden import AttrDict, import_reference, params_checksum  # noqa: E402

GRAPH_CFG = dict(  # config/graph_lanczos_net.yaml:9-27
    num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
    num_eig_vec=20, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7, output_dim=2,
    num_layer=7, num_atom=0)


def graph_config(data_path):
  model = dict(name='LanczosNetGeneral', short_diffusion_dist=[],
               long_diffusion_dist=GRAPH_CFG['long_diffusion_dist'], num_eig_vec=20,
               spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7, output_dim=2,
               num_layer=7, loss='MSE', output_func='MLP')
  dataset = dict(loader_name='GraphData', name='synthetic', data_path=data_path, node_emb_dim=10,
                 graph_emb_dim=2, num_edge_type=1)
  return AttrDict(dict(seed=1234, dataset=dataset, model=model))


def pack_adj(adj):
  """Upper triangle (i < j) of a 0/1 adjacency, row-major, bit-packed."""
  n = adj.shape[0]
  iu = np.triu_indices(n, 1)
  return np.packbits(adj[iu].astype(np.uint8))


def unpack_adj(bits, n):
  iu = np.triu_indices(n, 1)
  v = np.unpackbits(bits)[:len(iu[0])]
  a = np.zeros((n, n), np.float32)
  a[iu] = v
  return a + a.T


def main():
  from oracle import make_lanczosnet_params
  ref_model, ref_dh, _ = import_reference()
  torch.set_num_threads(4)
  import networkx as nx
  if not hasattr(nx, 'to_numpy_matrix'):
    nx.to_numpy_matrix = lambda g: np.asmatrix(nx.to_numpy_array(g))
  scratch = tempfile.mkdtemp(prefix='lnz_graph_golden_')
  cwd = os.getcwd()
  out = {}
  try:
    os.makedirs(os.path.join(scratch, 'work'))
    os.makedirs(os.path.join(scratch, 'data'))
    os.chdir(os.path.join(scratch, 'work'))
    import dataset.get_graph_data as gg        # creates ../data/synthetic/ (scratch)
    import dataset.graph_data as ref_gd
    gg.dump_data(gg.gen_data(seed=123), 'train')                   # 10 graphs: one train batch
    gg.dump_data(gg.gen_data(num_graphs=64, seed=789), 'test')     # one test batch of 64
    data_path = os.path.abspath(gg.save_dir)
    config = graph_config(data_path)
    P = make_lanczosnet_params(GRAPH_CFG, seed=4242, general=True)
    net = ref_model.LanczosNetGeneral(config).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    real_expand = np.expand_dims
    for split in ('train', 'test'):
      ds = ref_gd.GraphData(config, split=split)
      getattr(ds, split + '_data_files').sort()
      items = [ds[i] for i in range(len(ds))]
      np.expand_dims = lambda a, axis: real_expand(a, min(axis, np.ndim(a)))
      try:
        data = ds.collate_fn(items)
      finally:
        np.expand_dims = real_expand
      with torch.no_grad():
        score, loss = net(data['node_feat'], data['L'], data['D'], data['V'], label=data['label'],
                          mask=data['node_mask'].bool())
      L = data['L'].numpy()
      assert np.array_equal(L[..., 0], L[..., 1])   # one edge type: L_multi[..., 0] == L_simple_4
      n = np.array([it['node_feat'].shape[0] for it in items], np.int32)
      N = int(n.max())
      adj = []
      for it in items:
        a = (np.asarray(it['L_simple_4']) != 0).astype(np.float32)   # L4 = D^-1/2 (I+A) D^-1/2
        np.fill_diagonal(a, 0)
        adj.append(a)
        assert np.array_equal(unpack_adj(pack_adj(a), a.shape[0]), a)
      t = split + '_'
      out[t + 'n_nodes'] = n
      out[t + 'adj_bits'] = np.concatenate([pack_adj(a) for a in adj])
      out[t + 'adj_off'] = np.cumsum([0] + [len(pack_adj(a)) for a in adj]).astype(np.int64)
      out[t + 'node_feat'] = np.concatenate([np.asarray(it['node_feat']) for it in items])  # f64
      out[t + 'label'] = data['label'].numpy()
      out[t + 'D'] = data['D'].numpy()
      out[t + 'V'] = data['V'].numpy()
      out[t + 'D_full'] = np.stack([np.pad(np.asarray(it['D_simple']), (0, N - len(it['D_simple'])))
                                    for it in items])
      out[t + 'score'] = score.numpy()
      out[t + 'loss'] = np.float64(loss)
      if split == 'train':
        out[t + 'L0'] = L[..., 0]
      print(split, 'B', len(items), 'n', n.min(), n.max(), 'loss %.6f' % float(loss))
    out['param_seed'] = 4242
    out['param_checksum'] = params_checksum(P)
  finally:
    os.chdir(cwd)
    shutil.rmtree(scratch, ignore_errors=True)
  path = os.path.join(HERE, 'graph_config.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
