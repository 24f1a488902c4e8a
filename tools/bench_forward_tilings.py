"""Diagnostic: fused-forward time by tile plan (pairs / singles) and batch size."""
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
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for B in (512, 768, 1024, 2048, 4096):
  b = draw_batch(B, seed=0)
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
  nf, mk = t(b['node_feat']), t(b['node_mask'])
  for tiling in ('auto', 'single'):
    for _ in range(5): ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiling)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiling)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('B=%5d tiling=%-6s forward %.3f ms  %.0f mol/s' % (B, tiling, dt * 1e3, B / dt))
