"""The inference step (prepare + gains + forward) eager against a captured HIP graph replay."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mk = t(b['node_mask'].astype(np.uint8)); nf = t(b['node_feat']); K = 20
res = {}
for mode in ('fp32', 'f16x3'):
  net.gemm_mode = mode
  plan = net._plan()
  def step():
    Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mk, n, K)
    G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'], rows=rows, zero_fill=False,
                           split_pack=Lp if plan['gemm_mode'] == 1 else None)
    if plan['gemm_mode'] == 1:
      G, Lp = G
    return ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
  def timed(f, steps=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
      t0 = time.perf_counter()
      for _ in range(steps): f()
      torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return round(best, 4)
  with torch.no_grad():
    eager = timed(step)
    s_e = step().clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      out = step()
    graphed = timed(g.replay)
    g.replay(); torch.cuda.synchronize()
    res[mode] = {'eager_ms': eager, 'graph_replay_ms': graphed, 'scores_equal': bool(torch.equal(out, s_e))}
print(json.dumps(res))
