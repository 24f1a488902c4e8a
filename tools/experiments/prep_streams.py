"""Batch preparation as ONE fused launch against Ritz kernel + (pack + plan) launch on two streams."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval().cuda(); plan = net._plan()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mk = t(b['node_mask'])
A = L[..., 0]
side = torch.cuda.Stream()
def two_streams():
  cur = torch.cuda.current_stream()
  side.wait_stream(cur)
  with torch.cuda.stream(side):
    Lp, tiles, rows = ops.pack_and_plan(plan, L, mk, 20)
  D, V = ops.lanczos_ritz(A, n, 20)
  cur.wait_stream(side)
  return Lp, tiles, rows, D, V
def timed(f, reps=200):
  for _ in range(5): f()
  torch.cuda.synchronize()
  e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  e[0].record()
  for _ in range(reps): f()
  e[1].record(); torch.cuda.synchronize()
  return round(e[0].elapsed_time(e[1]) / reps, 4)
G = lambda D, rows: ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'], rows=rows, zero_fill=False)
def step(prep):
  Lp, tiles, rows, D, V = prep()
  g = G(D, rows)
  return ops.lanczosnet_forward(plan, t_nf, Lp, V, g, mk, tiling=tiles)
t_nf = t(b['node_feat'])
fused = lambda: ops.prepare_batch(plan, L, mk, n, 20)
print(json.dumps({'fused': timed(fused), 'two_streams': timed(two_streams), 'fused_again': timed(fused),
                  'step_fused': timed(lambda: step(fused)), 'step_two_streams': timed(lambda: step(two_streams)),
                  'step_fused_again': timed(lambda: step(fused))}))
s1 = step(fused); s2 = step(two_streams)
print('scores equal', bool(torch.equal(s1, s2)))
