"""A/B timing of the fused inference forward on the bench batch (QM8 shapes, B = 1024): average
launch time over 300 launches between two events, for whatever library / switches the environment
selects (LANCZOSNET_HIP_LIB, LNZ_FORWARD16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
B = int(os.environ.get('PROBE_B', '1024'))
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
plan = net._plan(); Lp = ops.pack_laplacian_for(plan, L)
G = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
nf, mk = t(b['node_feat']), t(b['node_mask'])
tiling = ops.plan_tiles(mk, True)
run = lambda: ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiling)
for _ in range(20): s = run()
res = []
for rep in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(300): s = run()
  e1.record(); torch.cuda.synchronize()
  res.append(e0.elapsed_time(e1) / 300)
print('%s forward16=%s: %s ms  (score checksum %.9g)' % (
    os.path.basename(os.environ.get('LANCZOSNET_HIP_LIB', 'in-tree')), os.environ.get('LNZ_FORWARD16', '0'),
    ' '.join('%.4f' % r for r in res), float(s.double().sum())))
