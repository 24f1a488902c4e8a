"""Spectral-gains launch time on the bench batch (live rows), 1 MI355X.  usage: bench_gains.py [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1234).items()})
net = net.cuda(); plan = net._plan()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mk = t(b['node_mask'])
Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mk, n, 20)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
def run():
  return ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'], rows=rows, zero_fill=False)
G = run(); torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
for _ in range(reps): run()
e[1].record(); torch.cuda.synchronize()
Gf = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
live = int(rows[1].item())
print(json.dumps({'gains_ms': round(e[0].elapsed_time(e[1]) / reps, 4), 'live_rows': live,
                  'checksum': float(Gf.double().abs().sum())}))
