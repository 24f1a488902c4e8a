#!/usr/bin/env python
"""Reference-side fixtures for the configurations of AdaLanczosNet that its HIP kernels are NOT built
for, but the reference class accepts and runs: each case is the UNMODIFIED
`model/ada_lanczos_net.py` class run in the build container, only its random sources held fixed from
outside (`torch.randn` for the Lanczos start vector; `F.dropout` replaced by the multiplication with
a seeded mask for the training-dropout case — nothing inside the class is patched).

  reorth_off   top-level `use_reorthogonalization` present + `model.use_reorthogonalization: False`
               (the only way to switch it off)            model/ada_lanczos_net.py:35,177
  dropout      train mode, dropout 0.3                                    :347
  big_n        molecules of 28..44 nodes (beyond the 32-node tile)
  width96      hidden_dim [96, 48]
  no_long      long_diffusion_dist []  (no Lanczos layer, no filters)     :308,324
  non_mlp      spectral_filter_kind 'poly': L_s = Q T^p Q^T               :282-284

Stored per case: the scores (and the loss of the training case), the start vector, and for the
cases with a Lanczos layer the molecules on which the reference's fp32 result is within 2e-6 of its
own float64 run (the protocol of SURVEY.md 8c: an fp32 Lanczos recurrence near a breakdown is not a
function of its input alone).  Parameters and inputs are functions of seeds.

    python tests/golden/make_golden_ada_offnominal.py        (needs /root/reference)
"""
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
from make_golden_ada import fixed_randn  # noqa: E402
from ada_offnominal_fixture import CASES, case_inputs, dropout_masks, fixed_dropout  # noqa: E402


def main():
  ref_model, _, _ = MG.import_reference()
  torch.set_num_threads(8)
  out = {}
  for name, spec in CASES.items():
    cfg, conf_extra, b, L, q1 = case_inputs(name)
    conf = MG.make_config(cfg, name='AdaLanczosNet')
    for k, v in conf_extra.get('model', {}).items():
      conf['model'][k] = v
    for k, v in conf_extra.get('top', {}).items():
      conf[k] = v
    net = ref_model.AdaLanczosNet(conf)
    P = oracle.make_ada_params(cfg, seed=spec['param_seed'])
    net.load_state_dict({k: torch.from_numpy(P[k]) for k in net.state_dict().keys()})
    nf, Lt = torch.from_numpy(b['node_feat']), torch.from_numpy(L)
    mask, lab = torch.from_numpy(b['node_mask']).bool(), torch.from_numpy(b['label'])

    def run(model, dt):
      real = torch.randn
      torch.randn = lambda *a, **k: torch.from_numpy(q1.astype(dt))   # (fixed_randn, any dtype)
      try:
        if spec.get('train'):
          model.train()
          masks = dropout_masks(name, b['node_mask'].shape, cfg)
          with fixed_dropout(masks):
            score, loss = model(nf, Lt.to(torch.float64 if dt == np.float64 else torch.float32),
                                label=lab.to(torch.float64 if dt == np.float64 else torch.float32), mask=mask)
          return score.detach().numpy(), float(loss)
        model.eval()
        with torch.no_grad():
          return model(nf, Lt.to(torch.float64 if dt == np.float64 else torch.float32), mask=mask).numpy(), None
      finally:
        torch.randn = real
    score, loss = run(net, np.float32)
    net64 = ref_model.AdaLanczosNet(conf).double()
    net64.load_state_dict({k: torch.from_numpy(P[k]).double() for k in net64.state_dict().keys()})
    s64, loss64 = run(net64, np.float64)
    dev = np.abs(score - s64).max(axis=1) / np.abs(s64).max()
    good = dev <= 2e-6
    print('%-10s B=%d N=%d: reference fp32 within 2e-6 of its float64 run on %d molecules '
          '(median %.1e, max %.1e)%s' % (name, len(score), L.shape[1], good.sum(), np.median(dev), dev.max(),
                                          '' if loss is None else '; loss %.6f (float64 %.6f)' % (loss, loss64)))
    assert good.sum() >= len(score) // 2, name
    out[name + '_score'] = score
    out[name + '_score64'] = s64.astype(np.float64)
    out[name + '_good'] = good
    if loss is not None:
      out[name + '_loss'] = np.float64(loss)
      out[name + '_loss64'] = np.float64(loss64)
  path = os.path.join(HERE, 'ada_offnominal.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
