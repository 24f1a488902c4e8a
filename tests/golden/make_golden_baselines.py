"""Generates tests/golden/baselines.npz by IMPORTING the reference (run in the build container):
the reference GCN (model/gcn.py), DCNN (model/dcnn.py) and ChebyNet (model/cheby_net.py) at their
QM8 config shapes (config/qm8_gcn.yaml, config/qm8_dcnn.yaml, config/qm8_cheby_net.yaml) on the collate_batch fixture, weights from numpy seeds:
scores, loss and per-parameter gradient sums.   python tests/golden/make_golden_baselines.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
from tests.golden.make_golden import import_reference, AttrDict  # noqa: E402
from oracle import make_lanczosnet_params  # noqa: E402

BASE = dict(num_atom=70, num_bond_type=6, input_dim=64, hidden_dim=[128] * 7, output_dim=16,
            num_layer=7, num_eig_vec=1, spectral_filter_kind='None', long_diffusion_dist=[])
DCNN_DIST = [3, 5, 7, 10, 20, 30]  # config/qm8_dcnn.yaml:18
CHEBY_ORDER = 5                     # config/qm8_cheby_net.yaml:20


def main():
  ref_model, _, _ = import_reference()
  torch.set_num_threads(4)
  c = np.load(os.path.join(HERE, 'collate_batch.npz'))
  nf, L = torch.from_numpy(c['node_feat']), torch.from_numpy(c['L'])
  mask, label = torch.from_numpy(c['node_mask']).bool(), torch.from_numpy(c['label'])
  out = {}
  for name, short, seed in (('GCN', [], 31), ('DCNN', DCNN_DIST, 32),
                            ('ChebyNet', list(range(2, CHEBY_ORDER + 1)), 33)):
    cfg = dict(BASE, short_diffusion_dist=short)
    if name == 'ChebyNet':  # polynomial_order + num_bond_type + 1 message blocks (cheby_net.py:31-33)
      cfg['num_bond_type'] = BASE['num_bond_type'] + 1
    P = make_lanczosnet_params(cfg, seed)  # same keys/shapes as the baselines' state_dict
    model = dict(name=name, input_dim=64, hidden_dim=[128] * 7, output_dim=16, num_layer=7,
                 loss='MSE', diffusion_dist=DCNN_DIST, polynomial_order=CHEBY_ORDER)
    conf = AttrDict(dict(seed=1234, dataset=dict(num_atom=70, num_bond_type=6), model=model))
    net = getattr(ref_model, name)(conf).train()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    score, loss = net(nf, L, label=label, mask=mask)
    loss.backward()
    names = [k for k, _ in net.named_parameters()]
    out[name + '_score'] = score.detach().numpy()
    out[name + '_loss'] = float(loss)
    out[name + '_seed'] = seed
    out[name + '_names'] = np.array(names)
    out[name + '_gsum'] = np.array([float(p.grad.double().sum()) for _, p in net.named_parameters()])
    out[name + '_gabs'] = np.array([float(p.grad.double().abs().sum()) for _, p in net.named_parameters()])
  np.savez_compressed(os.path.join(HERE, 'baselines.npz'), **out)
  print('baselines.npz', os.path.getsize(os.path.join(HERE, 'baselines.npz')), 'B')


if __name__ == '__main__':
  main()
