#!/usr/bin/env python
"""bench.py's `cpu_baseline` is the numpy oracle (`kind: "port"`): the reference tree does not
travel to the GPU box.  This script shows, in the BUILD container where /root/reference exists, that
the port and the UNMODIFIED reference run at the same rate on one host with the same threads:

  reference   per molecule `utils.data_helper.get_graph_laplacian_eigs(adj, 'L4', use_eigen_decomp)`
              (utils/data_helper.py:169-223), `QM8Data.collate_fn` pad/cut, then
              `LanczosNet(config).eval()(...)` (model/lanczos_net.py:125-199) under torch.no_grad()
  port        bench.cpu_baseline: oracle.graph_laplacian_eigs + collate_eigs + lanczos_net_forward

on bench.py's workload (B = 1024 molecules, draw_batch seed 0, parameters seed 1234).
Writes profiles/cpu_port_vs_reference.json.
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
  import make_golden as MG
  import oracle
  import bench
  from lanczosnet_amd.synthetic import draw_batch
  ref_model, ref_dh, ref_qm8 = MG.import_reference()
  threads = os.cpu_count() or 1
  torch.set_num_threads(threads)
  B, reps = 1024, 3
  cfg = dict(bench.QM8_CFG)
  params = oracle.make_lanczosnet_params(cfg, 1234)
  batch = draw_batch(B, seed=0)
  config = MG.make_config(cfg)
  net = ref_model.LanczosNet(config).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
  # L is preprocessing in both (resident before the timed region, as in bench.py)
  mols_L = []
  for b in range(B):
    n = int(batch['n_nodes'][b])
    adjs = batch['adjs'][b, :n, :n, :]
    _, _, L_list = ref_dh.get_multi_graph_laplacian_eigs(adjs, graph_laplacian_type='L4',
                                                         use_eigen_decomp=True, is_sym=True)
    mols_L.append(np.stack(L_list, axis=2))
  t_ref, t_eig = [], []
  score_ref = None
  for _ in range(reps):
    t0 = time.perf_counter()
    mols = []
    for b in range(B):
      n = int(batch['n_nodes'][b])
      adj_simple = np.sum(batch['adjs'][b, :n, :n, :], axis=2)
      D, V, L4 = ref_dh.get_graph_laplacian_eigs(adj_simple, graph_laplacian_type='L4',
                                                 use_eigen_decomp=True, is_sym=True)
      mols.append(dict(L_multi=mols_L[b], L_simple_4=L4, D_simple=D, V_simple=V))
    t1 = time.perf_counter()
    data = MG.reference_collate(ref_qm8, config, mols, batch)
    with torch.no_grad():
      score_ref = net(data['node_feat'], data['L'], data['D'], data['V'],
                      mask=data['node_mask'].bool())
    t_ref.append(time.perf_counter() - t0)
    t_eig.append(t1 - t0)
  v_port, t_port, score_port, amb = bench.cpu_baseline(cfg, params, B, reps)
  dev = np.abs(score_port - score_ref.numpy()).max(axis=1) / np.abs(score_ref.numpy()).max()
  out = {
      'host': {'cpu_count': threads, 'torch_threads': torch.get_num_threads(),
               'model': [ln.split(':')[1].strip() for ln in open('/proc/cpuinfo')
                         if ln.startswith('model name')][:1]},
      'workload': 'bench.py default: B=1024 QM8-schema molecules (draw_batch seed 0), '
                  'config/qm8_lanczos_net.yaml architecture, parameters numpy seed 1234',
      'reference': {'molecules_per_s': B / min(t_ref), 'seconds': t_ref, 'eig_loop_seconds': t_eig,
                    'what': 'unmodified get_graph_laplacian_eigs loop + QM8Data.collate_fn + '
                            'LanczosNet.forward (eval, no_grad), torch CPU'},
      'port': {'molecules_per_s': v_port, 'seconds': t_port,
               'what': 'bench.cpu_baseline (oracle: numpy eigh loop + torch-CPU restatement of the forward)'},
      'port_over_reference': v_port / (B / min(t_ref)),
      'scores_port_vs_reference_max_rel': float(dev[~amb].max()),
  }
  path = os.path.join(ROOT, 'profiles', 'cpu_port_vs_reference.json')
  json.dump(out, open(path, 'w'), indent=1)
  print(json.dumps(out)[:900])


if __name__ == '__main__':
  main()
