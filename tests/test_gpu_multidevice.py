"""One process, several devices — the reference's own multi-GPU mechanism (`nn.DataParallel` with
`gpus: [0, 1, ...]`, runner/qm8_runner.py:62): replicas run concurrently in one thread per device
on dim-0 shards, so nothing in the library may be configured "once per process".  The kernels
with more than 64 KiB of dynamic LDS (the f16x3 forward, the large-graph gemm1 / conv with 2 and 3
operand planes) set their per-device function attribute at every launch; these tests run them on
the SECOND device.  Skipped on a box with one GPU (the driver's GPU tier has one)."""
import numpy as np
import pytest
import torch

import oracle
from lanczosnet_amd.synthetic import draw_batch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs in one process')]


def _net(cfg, P, cls_name='LanczosNet', general=False):
  from lanczosnet_amd import model
  from lanczosnet_amd.utils.arg_helper import make_model_config
  net = getattr(model, cls_name)(make_model_config(cfg, general=general)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return net


def test_f16x3_forward_under_dataparallel_two_devices():
  from lanczosnet_amd import ops
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  P = oracle.make_lanczosnet_params(cfg, 3)
  b = draw_batch(96, seed=8)
  t = lambda x, d: torch.from_numpy(np.ascontiguousarray(x)).to(d)  # noqa: E731
  n = t(b['n_nodes'], 'cuda:0')
  L = ops.laplacian_l4(t(b['adjs'], 'cuda:0'), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, cfg['num_eig_vec'])
  ref = None
  for mode in ('fp32', 'f16x3'):
    net = _net(cfg, P)
    net.gemm_mode = mode
    dp = torch.nn.DataParallel(net, device_ids=[0, 1]).cuda()
    with torch.no_grad():
      score = dp(t(b['node_feat'], 'cuda:0'), L, D, V, mask=t(b['node_mask'], 'cuda:0'))
    assert torch.isfinite(score).all()
    if ref is None:
      ref = score
    else:
      assert float((score - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize('planes', [1, 2, 3])
def test_large_graph_kernels_on_the_second_device(planes):
  from lanczosnet_amd import ops
  from test_gpu_graph import GRAPH_CFG
  cfg = dict(GRAPH_CFG, num_eig_vec=24)
  P = oracle.make_lanczosnet_params(cfg, 5, general=True)
  rs = np.random.RandomState(2)
  B, N = 6, 96
  scores = []
  for dev in ('cuda:0', 'cuda:1'):
    with torch.cuda.device(dev):
      net = _net(cfg, P, 'LanczosNetGeneral', general=True).to(dev)
      net.large_split_planes = planes
      if planes == 1:
        net.gemm_mode = 'bf16'
      rs = np.random.RandomState(2)
      adj = np.triu((rs.rand(B, N, N) < 0.2).astype(np.float32), 1)
      adj = (adj + adj.transpose(0, 2, 1))[..., None]
      n = torch.full((B,), N, dtype=torch.int32, device=dev)
      L = ops.laplacian_l4(torch.from_numpy(adj).to(dev), n)
      D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 24)
      X = torch.from_numpy(rs.randn(B, N, 10).astype(np.float32)).to(dev)
      with torch.no_grad():
        scores.append(net(X, L, D, V, mask=torch.ones((B, N), dtype=torch.uint8, device=dev)).cpu())
  assert torch.equal(scores[0], scores[1])
