#!/usr/bin/env python
"""Reference-side gradient projections (tests/gradproj.py): `loss.backward()` of the UNMODIFIED
reference classes, every parameter tensor's gradient projected on 16 fixed +-1 vectors, plus its
2-norm.  Same batches, parameters and start vectors as the aggregate fixtures:

  LanczosNet      tests/golden/make_golden.py section 6d (first 12 molecules of the collate batch,
                  parameters numpy seed 2024)                          -> lnet_*
  AdaLanczosNet   tests/golden/make_golden_ada.py (the 32 strict-set molecules `grad_idx` of
                  ada_e2e.npz, its q1, parameters seed 31), the reference class in float32 AND in
                  float64 (the exact-arithmetic gradient its fp32 autograd approximates) -> ada_*

    python tests/golden/make_golden_gradproj.py        # needs /root/reference; writes grad_projections.npz
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
from gradproj import project_numpy  # noqa: E402
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402


def proj_all(named):
  names = sorted(named.keys())
  return (np.array(names),
          np.stack([project_numpy(named[k].grad.detach().double().numpy(), i)
                    for i, k in enumerate(names)]),
          np.array([float(named[k].grad.detach().double().norm()) for k in names]))


def main():
  ref_model, ref_dh, ref_qm8 = MG.import_reference()
  torch.set_num_threads(8)
  out = {}
  # ---- LanczosNet (make_golden.py sections 2 + 6d)
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  config = MG.make_config(cfg)
  batch = draw_batch(24, seed=11, n_min=3, n_max=26)
  mols = [MG.reference_preprocess(ref_dh, batch['adjs'][b], int(batch['n_nodes'][b]))
          for b in range(24)]
  data = MG.reference_collate(ref_qm8, config, mols, batch)
  P = oracle.make_lanczosnet_params(cfg, seed=2024)
  net = ref_model.LanczosNet(config).train()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  nb = 12
  _, loss = net(data['node_feat'][:nb], data['L'][:nb], data['D'][:nb], data['V'][:nb],
                label=data['label'][:nb], mask=data['node_mask'][:nb].bool())
  loss.backward()
  out['lnet_names'], out['lnet_proj'], out['lnet_norm'] = proj_all(dict(net.named_parameters()))
  out['lnet_loss'] = float(loss)
  old = np.load(os.path.join(HERE, 'grad_parity.npz'))
  assert abs(float(old['loss']) - float(loss)) < 1e-6 * abs(float(loss)), 'not the grad_parity.npz batch'
  del net

  # ---- AdaLanczosNet (make_golden_ada.py, e2e part)
  g = np.load(os.path.join(HERE, 'ada_e2e.npz'))
  p = np.load(os.path.join(HERE, 'ada_protocol.npz'))
  acfg = ast.literal_eval(str(g['cfg_json']))
  conf = MG.make_config(acfg, name='AdaLanczosNet')
  conf['model']['use_reorthogonalization'] = False
  b = draw_batch(len(p['n_nodes']), seed=int(p['seed']), n_min=int(p['n_min']), n_max=int(p['n_max']))
  nbe = int(g['nb'])
  B, N = b['node_mask'].shape
  L = np.zeros((B, N, N, 7), np.float32)
  for i in range(nbe):
    n = int(b['n_nodes'][i])
    L[i, :n, :n] = oracle.laplacian_multi_l4(b['adjs'][i, :n, :n])
  gi = g['grad_idx']
  q1 = g['q1'][gi][:, :, None]
  Pa = oracle.make_ada_params(acfg, int(g['param_seed']))
  nf = torch.from_numpy(b['node_feat'][:nbe])[gi]
  Lt = torch.from_numpy(L[:nbe])[gi]
  lab = torch.from_numpy(b['label'][:nbe])[gi]
  mask = torch.from_numpy(b['node_mask'][:nbe])[gi].bool()
  real_randn = torch.randn
  for tag, dt in (('ada', torch.float32), ('ada64', torch.float64)):
    net = ref_model.AdaLanczosNet(conf).to(dt).train()
    net.load_state_dict({k: torch.from_numpy(v).to(dt) for k, v in Pa.items()})
    torch.randn = lambda *a, **k: torch.from_numpy(q1.astype(np.float64)).to(dt)
    try:
      _, loss = net(nf, Lt.to(dt), label=lab.to(dt), mask=mask)
    finally:
      torch.randn = real_randn
    loss.backward()
    out[tag + '_names'], out[tag + '_proj'], out[tag + '_norm'] = proj_all(dict(net.named_parameters()))
    out[tag + '_loss'] = float(loss)
    del net
  assert abs(out['ada_loss'] - float(g['loss'])) < 1e-6 * abs(float(g['loss'])), 'not the ada_e2e.npz batch'
  path = os.path.join(HERE, 'grad_projections.npz')
  np.savez_compressed(path, **out)
  w32 = np.abs(out['ada_proj'] - out['ada64_proj']).max(axis=1) / out['ada64_norm']
  print('ada reference fp32-vs-fp64 projections, worst per tensor (rel. to |g|): %.2e' % w32.max())
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
