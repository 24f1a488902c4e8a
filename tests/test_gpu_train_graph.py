"""The training step replayed from a HIP graph (lanczosnet_amd.train.GraphedTrainStep) against the
same steps launched eagerly: same kernels, same parameter trajectory."""
import numpy as np
import pytest
import torch

import oracle
from lanczosnet_amd.synthetic import draw_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _setup(seed):
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net = LanczosNet(make_model_config(cfg)).train()
  net.load_state_dict({k: torch.from_numpy(v)
                       for k, v in oracle.make_lanczosnet_params(cfg, seed).items()})
  net = net.to(DEV)
  batches = []
  for s, nmax in ((1, 26), (2, 19), (3, 26), (4, 19), (5, 26), (6, 26), (7, 19), (8, 26)):
    b = draw_batch(96, seed=s, n_min=6, n_max=nmax)
    n = _t(b['n_nodes'])
    L = ops.laplacian_l4(_t(b['adjs']), n)
    D, V = ops.lanczos_ritz(L[..., 0], n, 20)
    batches.append((_t(b['node_feat']), L, D, V, _t(b['label']), _t(b['node_mask'])))
  return net, batches


@pytest.mark.parametrize('optim', ['sgd', 'adam'])
def test_graphed_train_step_follows_the_eager_trajectory(optim):
  from lanczosnet_amd.train import GraphedTrainStep, make_adam
  net_e, batches = _setup(3)
  net_g, _ = _setup(3)
  # the eager run on the captured step's row set (padded B * N message rows, all B * K eigen rows
  # with the dead ones masked) instead of the live rows only: the two runs then add the same terms
  # in the same order, and Adam — which turns rounding noise on near-zero gradients into +-lr
  # moves — has no noise to amplify
  net_e.train_static_rows = optim == 'adam'
  lr = 1e-2 if optim == 'sgd' else 1e-3
  mk = (lambda p: torch.optim.SGD(p, lr=lr)) if optim == 'sgd' else (lambda p: make_adam(p, lr=lr))
  opt_e, opt_g = mk(net_e.parameters()), mk(net_g.parameters())
  step = GraphedTrainStep(net_g, opt_g, warmup=1)
  le, lg = [], []
  for bt in batches:
    opt_e.zero_grad(set_to_none=True)
    _, loss = net_e(bt[0], bt[1], bt[2], bt[3], label=bt[4], mask=bt[5])
    loss.backward()
    opt_e.step()
    le.append(float(loss.detach()))
    lg.append(float(step(*bt)))
  assert len(step._graphs) == 2          # two padded sizes -> two graphs, replayed 4 and 2 times
  le, lg = np.array(le), np.array(lg)
  assert np.abs(le - lg).max() <= 2e-5 * np.abs(le).max(), (le, lg)
  worst, worst_k = 0.0, None
  for (k, pe), (_, pg) in zip(net_e.named_parameters(), net_g.named_parameters()):
    d = (pe - pg).abs().max().item()
    if d / max(pe.abs().max().item(), 1e-12) > worst:
      worst, worst_k = d / max(pe.abs().max().item(), 1e-12), k
    if optim == 'adam':
      # Adam normalises every element's update to ~lr whatever the gradient's size: elements whose
      # gradient is rounding noise move by +-lr per step in either run — bounded, not comparable
      assert d <= 2.5 * lr * len(batches), k
  print('%s: loss dev %.2e, worst relative parameter deviation after %d steps: %.2e (%s)' %
        (optim, np.abs(le - lg).max() / np.abs(le).max(), len(batches), worst, worst_k))
  if optim == 'sgd':
    assert worst < 2e-5     # SGD is linear in the gradients: the two runs stay together
  # the module stays usable eagerly after replays (plans re-packed from the updated parameters)
  net_e.eval(), net_g.eval()
  with torch.no_grad():
    bt = batches[0]
    se = net_e(bt[0], bt[1], bt[2], bt[3], mask=bt[5])
    sg = net_g(bt[0], bt[1], bt[2], bt[3], mask=bt[5])
  assert (se - sg).abs().max().item() <= (1e-5 if optim == 'sgd' else 1e-3) * se.abs().max().item()


def test_graphed_step_needs_a_capturable_optimizer():
  from lanczosnet_amd.train import GraphedTrainStep
  net, _ = _setup(1)
  with pytest.raises(ValueError):
    GraphedTrainStep(net, torch.optim.Adam(net.parameters(), lr=1e-3))


def test_graphed_ada_train_step_follows_the_eager_trajectory():
  """AdaLanczosNet's training step (HIP forward kernels + _AdaLanczosNetFusedFunction backward +
  the fp64 Lanczos-layer autograd, ~1500 launches) replayed from a HIP graph: the Lanczos start
  vector is drawn by the step object from the CPU generator — one draw of shape (B, N, 1) per step,
  exactly what the eager forward consumes (model/ada_lanczos_net.py:161) — so with the same seed
  the two runs see the same vectors; SGD keeps them together."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import AdaLanczosNet
  from lanczosnet_amd.train import GraphedTrainStep
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1, 2, 3],
             long_diffusion_dist=[5, 7, 10, 20, 30], hidden_dim=[128, 128], num_layer=2)

  def make():
    torch.manual_seed(11)
    return AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet')).train().to(DEV)

  batches = []
  for s in (1, 2, 3, 4, 5):
    b = draw_batch(48, seed=s, n_min=10, n_max=26)
    n = _t(b['n_nodes'])
    L = ops.laplacian_l4(_t(b['adjs']), n)
    batches.append((_t(b['node_feat']), L, None, None, _t(b['label']), _t(b['node_mask'])))
  lr = 1e-3
  net_e, net_g = make(), make()
  opt_e = torch.optim.SGD(net_e.parameters(), lr=lr)
  opt_g = torch.optim.SGD(net_g.parameters(), lr=lr)
  le, lg = [], []
  torch.manual_seed(77)
  for bt in batches:
    opt_e.zero_grad(set_to_none=True)
    _, loss = net_e(bt[0], bt[1], label=bt[4], mask=bt[5])
    loss.backward()
    opt_e.step()
    le.append(float(loss.detach()))
  torch.manual_seed(77)
  step = GraphedTrainStep(net_g, opt_g, warmup=1)
  for bt in batches:
    lg.append(float(step(*bt)))
  assert len(step._graphs) == 1 and net_g._static_q1 is None
  le, lg = np.array(le), np.array(lg)
  assert np.abs(le - lg).max() <= 2e-5 * np.abs(le).max(), (le, lg)
  worst, worst_k = 0.0, None
  for (k, pe), (_, pg) in zip(net_e.named_parameters(), net_g.named_parameters()):
    d = (pe - pg).abs().max().item() / max(pe.abs().max().item(), 1e-12)
    if d >= worst:
      worst, worst_k = d, k
  print('Ada graphed vs eager: loss dev %.2e, worst relative parameter deviation after %d steps: '
        '%.2e (%s)' % (np.abs(le - lg).max() / np.abs(le).max(), len(batches), worst, worst_k))
  assert worst < 2e-5
