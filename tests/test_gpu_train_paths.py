"""Training through the paths OUTSIDE the fused HIP backward — the device-side differentiable
restatement `_LanczosNetBase._torch_forward` (hidden widths other than a uniform 64 / 128, graphs
beyond the 32-row tile, dropout > 0) — against the unmodified reference's autograd
(tests/golden/train_paths.npz, tests/golden/make_golden_trainpaths.py): loss at 1e-5 and 16 fixed
+-1 projections of every parameter tensor's gradient at 1e-5 |g| (tests/gradproj.py).  These are the
configurations the reference runners train that the fused kernels do not cover."""
import ast
import warnings

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden
from gradproj import deterministic_dropout, project_torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _check(net, loss, g, tag):
  assert abs(float(loss.detach()) - float(g[tag + '_loss'])) < 1e-5 * abs(float(g[tag + '_loss'])), tag
  loss.backward()
  gd = dict(net.named_parameters())
  worst = (0.0, None)
  for i, k in enumerate(g[tag + '_names']):
    gr = gd[str(k)].grad
    assert gr is not None, k
    nrm = float(g[tag + '_norm'][i])
    e = float(np.abs(project_torch(gr, i) - g[tag + '_proj'][i]).max() / nrm)
    if e >= worst[0]:
      worst = (e, str(k))
    assert abs(float(gr.double().norm()) - nrm) < 1e-5 * nrm, k
  print('%s: gradient projections vs reference autograd, worst %.2e of |g| (%s)' % (tag, worst[0], worst[1]))
  assert worst[0] < 1e-5, worst


def _small_net(dropout=None):
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config
  g = load_golden('lanczosnet_small_mlp.npz')
  cfg = ast.literal_eval(str(g['cfg_json']))
  conf = make_model_config(cfg)
  if dropout is not None:
    conf.model['dropout'] = dropout
  net = LanczosNet(conf)
  net.load_state_dict({k: torch.from_numpy(v) for k, v in
                       oracle.make_lanczosnet_params(cfg, int(g['param_seed'])).items()})
  return net.to(DEV).train(), g, cfg


def test_training_widths_outside_the_fused_kernel():
  net, g, _ = _small_net()
  tp = load_golden('train_paths.npz')
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    _, loss = net(_t(g['node_feat']), _t(g['L']), _t(g['D']), _t(g['V']), label=_t(tp['small_label']),
                  mask=_t(g['node_mask']))
  _check(net, loss, tp, 'small')


def test_training_with_dropout_places_it_where_the_reference_does():
  net, g, cfg = _small_net(dropout=0.3)
  assert net.dropout == 0.3
  tp = load_golden('train_paths.npz')
  with warnings.catch_warnings(), deterministic_dropout() as dd:
    warnings.simplefilter('ignore')
    _, loss = net(_t(g['node_feat']), _t(g['L']), _t(g['D']), _t(g['V']), label=_t(tp['small_label']),
                  mask=_t(g['node_mask']))
  # one call per conv layer, on the [B, N, width] state (model/lanczos_net.py:182)
  assert [list(s) for s, _ in dd.calls] == tp['smalldrop_calls'].tolist()
  assert all(abs(p - 0.3) < 1e-12 for _, p in dd.calls)
  _check(net, loss, tp, 'smalldrop')
  # eval mode: no dropout call, the deterministic forward
  net.eval()
  with torch.no_grad(), warnings.catch_warnings(), deterministic_dropout() as dd2:
    warnings.simplefilter('ignore')
    s_eval = net(_t(g['node_feat']), _t(g['L']), _t(g['D']), _t(g['V']), mask=_t(g['node_mask']))
  assert dd2.calls == []
  assert float((s_eval.cpu() - torch.from_numpy(g['score'])).abs().max()) < 1e-5 * float(np.abs(g['score']).max())


def test_training_graphs_beyond_the_32_row_tile():
  from graph_fixture import GRAPH_CFG, load_split, pad_batch
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  items, ref, seed, _ = load_split('train')
  _, X, mask, n = pad_batch(items)
  B, N = mask.shape
  assert N > 32
  L = np.zeros((B, N, N, 2), np.float32)
  L[..., 0] = ref['L0']
  L[..., 1] = ref['L0']
  net = LanczosNetGeneral(make_model_config(GRAPH_CFG, general=True))
  net.load_state_dict({k: torch.from_numpy(v) for k, v in
                       oracle.make_lanczosnet_params(GRAPH_CFG, seed, general=True).items()})
  net = net.to(DEV).train()
  _, loss = net(_t(X), _t(L), _t(ref['D']), _t(ref['V']), label=_t(ref['label']), mask=_t(mask))
  _check(net, loss, load_golden('train_paths.npz'), 'graph')


@pytest.mark.parametrize('name', ['GCN', 'ChebyNet'])
def test_baselines_accept_a_host_laplacian(name):
  """runner/qm8_runner.py:301-302 hands the model a HOST `L`; the baselines build D / V / the
  identity channel from their inputs, so those must be moved to the module's device first."""
  from test_baselines import _conf, _params
  from lanczosnet_amd import model as M
  g = load_golden('baselines.npz')
  c = load_golden('collate_batch.npz')
  net = getattr(M, name)(_conf(name))
  net.load_state_dict({k: torch.from_numpy(v) for k, v in _params(name, int(g[name + '_seed'])).items()})
  net = net.to(DEV).eval()
  with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter('ignore')
    s = net(_t(c['node_feat']), torch.from_numpy(c['L']), mask=_t(c['node_mask'])).cpu().numpy()
  ref = g[name + '_score']
  assert np.abs(s - ref).max() <= 1e-5 * np.abs(ref).max()
