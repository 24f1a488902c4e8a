"""Diagnostic: the bench step replayed from a captured HIP graph vs launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
P = oracle.make_lanczosnet_params(cfg, 1)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}); net = net.cuda()
plan = net._plan()
B, K = 1024, 20
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
nf, mk = t(b['node_feat']), t(b['node_mask']).to(torch.uint8).contiguous()


def step():
  Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mk, n, K)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'], rows=rows,
                         zero_fill=False)
  return ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)


def timeit(fn, nrep=100):
  for _ in range(10): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(nrep): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / nrep * 1e3


with torch.no_grad():
  ref = step().clone()
  print('eager   %.4f ms/step' % timeit(step))
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    for _ in range(3): step()
  torch.cuda.current_stream().wait_stream(s)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    out = step()
  g.replay(); torch.cuda.synchronize()
  print('graph   %.4f ms/step   equal=%s' % (timeit(g.replay), bool(torch.equal(out, ref))))
