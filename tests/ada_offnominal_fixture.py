"""Seeded inputs of the AdaLanczosNet off-nominal cases, shared by
tests/golden/make_golden_ada_offnominal.py (the unmodified reference class) and
tests/test_gpu_ada_offnominal.py (the HIP module's device-side restatement): configurations,
molecules, start vectors and dropout masks are functions of numpy seeds."""
import numpy as np
import torch

import oracle
from lanczosnet_amd.synthetic import draw_batch

_BASE = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1], long_diffusion_dist=[2, 4], num_eig_vec=8,
             hidden_dim=[128, 128], num_layer=2)

# name -> cfg overrides, config extras (model / top level), batch draw, flags
CASES = {
    'reorth_off': dict(cfg={}, extra=dict(model=dict(use_reorthogonalization=False),
                                          top=dict(use_reorthogonalization=True)),
                       draw=dict(batch_size=24, seed=51, n_min=4, n_max=26), param_seed=61),
    'dropout': dict(cfg={}, extra=dict(model=dict(dropout=0.3)),
                    draw=dict(batch_size=16, seed=52, n_min=4, n_max=26), param_seed=62, train=True),
    'big_n': dict(cfg={}, extra={}, draw=dict(batch_size=12, seed=53, n_min=28, n_max=44), param_seed=63),
    'width96': dict(cfg=dict(hidden_dim=[96, 48]), extra={},
                    draw=dict(batch_size=16, seed=54, n_min=4, n_max=26), param_seed=64),
    'no_long': dict(cfg=dict(long_diffusion_dist=[], short_diffusion_dist=[1, 2]), extra={},
                    draw=dict(batch_size=16, seed=55, n_min=4, n_max=26), param_seed=65),
    'non_mlp': dict(cfg=dict(spectral_filter_kind='poly'), extra={},
                    draw=dict(batch_size=16, seed=56, n_min=4, n_max=26), param_seed=66),
}


def case_inputs(name):
  """-> cfg, config extras, batch (draw_batch dict), L [B,N,N,7] float32 (oracle L4), q1 [B,N,1] float32."""
  spec = CASES[name]
  cfg = dict(_BASE, **spec['cfg'])
  b = draw_batch(**spec['draw'])
  B, N = b['node_mask'].shape
  L = np.zeros((B, N, N, 7), np.float32)
  for i in range(B):
    n = int(b['n_nodes'][i])
    L[i, :n, :n] = oracle.laplacian_multi_l4(b['adjs'][i, :n, :n])
  q1 = np.random.RandomState(1000 + spec['draw']['seed']).randn(B, N, 1).astype(np.float32)
  return cfg, spec['extra'], b, L, q1


def dropout_masks(name, shape_bn, cfg):
  """One mask per conv layer, [B,N,width] of {0, 1 / (1 - p)}: what F.dropout multiplies by."""
  p = CASES[name]['extra']['model']['dropout']
  rs = np.random.RandomState(77)
  return [((rs.rand(shape_bn[0], shape_bn[1], w) >= p) / (1.0 - p)).astype(np.float32)
          for w in cfg['hidden_dim'][:cfg['num_layer']]]


class fixed_dropout(object):
  """torch.nn.functional.dropout(x, p, training=True) -> x * the next seeded mask, for the reference
  class and the restatement alike (both call it through the module attribute)."""

  def __init__(self, masks):
    self.masks, self.i = masks, 0

  def __enter__(self):
    self.real = torch.nn.functional.dropout

    def drop(x, p=0.5, training=True, inplace=False):
      if not training:
        return x
      m = torch.from_numpy(self.masks[self.i]).to(device=x.device, dtype=x.dtype)
      self.i += 1
      return x * m
    torch.nn.functional.dropout = drop
    return self

  def __exit__(self, *exc):
    torch.nn.functional.dropout = self.real
