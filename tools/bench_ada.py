#!/usr/bin/env python
"""Secondary measurement (BASELINE.json configs[3]): AdaLanczosNet forward, batch 1024, 1 x MI355X.
Prints per-stage HIP-event times and molecules/s.  Not the headline bench (bench.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import AdaLanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1024)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--layers', type=int, default=7)
args = ap.parse_args()
cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1, 2, 3],
           long_diffusion_dist=[5, 7, 10, 20, 30], hidden_dim=[128] * args.layers,
           num_layer=args.layers)
torch.manual_seed(1234)
net = AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet')).eval().cuda()
b = draw_batch(args.batch, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
nf, mask = t(b['node_feat']), t(b['node_mask'])
B, N = nf.shape; K, S = 20, 5
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
names = ['learned_laplacian', 'lanczos_layer', 't_powers', 'filter_mlp(hipBLASLt)+sym', 'pack_L', 'fused_forward']
acc = np.zeros(6)
with torch.no_grad():
  plan = net._plan()
  for it in range(args.steps + 2):
    q1 = torch.randn(B, N, 1).cuda()
    ev[0].record()
    Le = ops.ada_graph_laplacian(nf, net.embedding.weight, L[:, :, :, 0]); ev[1].record()
    T, Q = ops.ada_lanczos_layer(Le, mask, q1, K); ev[2].record()
    tcat = ops.ada_t_powers(T, cfg['long_diffusion_dist']).view(B, -1); ev[3].record()
    DDp = torch.empty((args.layers, B, S, K, K), device='cuda')
    for l, seq in enumerate(net.spectral_filter):
      ops.ada_symmetrize_filters(seq(tcat), K, S, out=DDp[l])
    ev[4].record()
    Lp = ops.pack_laplacian(L); ev[5].record()
    score = ops.lanczosnet_forward(plan, nf, Lp, Q, DDp, mask); ev[6].record()
    torch.cuda.synchronize()
    if it >= 2:
      acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]
acc /= args.steps
tot = acc.sum()
print(json.dumps({'workload': 'AdaLanczosNet forward, QM8 batch=%d, %d layers, fp32' % (B, args.layers),
                  'ms_per_step': round(float(tot), 3), 'molecules_per_s': round(B / tot * 1e3, 1),
                  'stage_ms': {k: round(float(v), 4) for k, v in zip(names, acc)},
                  'finite': bool(torch.isfinite(score).all())}))
