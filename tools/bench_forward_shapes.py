"""Diagnostic: fused-forward time for hand-made workgroup shapes (tiles per half), singles only."""
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
W = 256
for shape in ((1, 0), (2, 0), (1, 1), (2, 1), (2, 2)):
  B = W * sum(shape)
  b = draw_batch(B, seed=0)
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
  nf, mk = t(b['node_feat']), t(b['node_mask'])
  cap = W
  buf = np.full((12 * cap + 1,), -1, np.int32)
  e = buf[:12 * cap].reshape(cap, 4, 3)
  e[:, :, 2] = 32
  mol = 0
  for w in range(W):
    for h in range(2):
      for m in range(shape[h]):
        e[w, 2 * h + m, 0] = mol; mol += 1
  buf[12 * cap] = W
  tiles = (torch.from_numpy(buf).cuda(), cap)
  for _ in range(5): ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
  print('shape %s: B=%4d forward %.3f ms (%.3f ms per tile-slot)' % (shape, B, dt * 1e3, dt * 1e3 / sum(shape)))
