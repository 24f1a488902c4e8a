#!/usr/bin/env python
"""Reference-side fixtures for the TRAINING paths outside the fused HIP backward (the device-side
differentiable restatement `_torch_forward`): loss and gradient projections (tests/gradproj.py) of
the unmodified reference classes' `loss.backward()` for

  small     LanczosNet, hidden widths [16, 12] (outside the fused kernel), short [1, 3] + long [2, 5]
            diffusion, K = 6 — the inputs of lanczosnet_small_mlp.npz
  smalldrop the same with config.model.dropout = 0.3 in training mode, `F.dropout` replaced on both
            sides by tests/gradproj.deterministic_dropout (placement / shape / order / call count)
  graph     LanczosNetGeneral on the graph configuration's train batch (N = 98 > 32 nodes:
            beyond the 32-row tile), graph_config.npz inputs

    python tests/golden/make_golden_trainpaths.py    # needs /root/reference; writes train_paths.npz
"""
import ast
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as MG  # noqa: E402
import oracle  # noqa: E402
from gradproj import project_numpy, deterministic_dropout  # noqa: E402


def record(out, tag, net, loss):
  loss.backward()
  named = dict(net.named_parameters())
  names = sorted(named.keys())
  out[tag + '_names'] = np.array(names)
  out[tag + '_loss'] = float(loss)
  out[tag + '_proj'] = np.stack([project_numpy(named[k].grad.double().numpy(), i)
                                 for i, k in enumerate(names)])
  out[tag + '_norm'] = np.array([float(named[k].grad.double().norm()) for k in names])


def main():
  ref_model, _, _ = MG.import_reference()
  torch.set_num_threads(4)
  out = {}
  g = np.load(os.path.join(HERE, 'lanczosnet_small_mlp.npz'))
  cfg = ast.literal_eval(str(g['cfg_json']))
  P = oracle.make_lanczosnet_params(cfg, int(g['param_seed']))
  label = np.random.RandomState(91).randn(g['node_feat'].shape[0], cfg['output_dim']).astype(np.float32)
  out['small_label'] = label
  args = [torch.from_numpy(g[k]) for k in ('node_feat', 'L', 'D', 'V')]
  mask = torch.from_numpy(g['node_mask']).bool()
  for tag, p in (('small', None), ('smalldrop', 0.3)):
    conf = MG.make_config(cfg)
    if p is not None:
      conf['model']['dropout'] = p
    net = ref_model.LanczosNet(conf).train()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    with deterministic_dropout() as dd:
      _, loss = net(*args, label=torch.from_numpy(label), mask=mask)
    if p is not None:
      out['smalldrop_calls'] = np.array([list(s) for s, _ in dd.calls])
      assert len(dd.calls) == cfg['num_layer']
    record(out, tag, net, loss)

  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from graph_fixture import GRAPH_CFG, load_split, pad_batch
  import make_golden_graph as GG
  items, ref, seed, _ = load_split('train')
  adjs, X, gmask, n = pad_batch(items)
  B, N = gmask.shape
  L = np.zeros((B, N, N, 2), np.float32)
  L[..., 0] = ref['L0']
  L[..., 1] = ref['L0']
  Pg = oracle.make_lanczosnet_params(GRAPH_CFG, seed, general=True)
  netg = ref_model.LanczosNetGeneral(GG.graph_config('/nonexistent')).train()
  netg.load_state_dict({k: torch.from_numpy(v) for k, v in Pg.items()})
  _, lossg = netg(torch.from_numpy(X), torch.from_numpy(L), torch.from_numpy(ref['D']),
                  torch.from_numpy(ref['V']), label=torch.from_numpy(ref['label']),
                  mask=torch.from_numpy(gmask).bool())
  assert abs(float(lossg) - ref['loss']) < 1e-6 * ref['loss']
  record(out, 'graph', netg, lossg)
  path = os.path.join(HERE, 'train_paths.npz')
  np.savez_compressed(path, **out)
  print({k: out[k] for k in ('small_loss', 'smalldrop_loss', 'graph_loss')})
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
