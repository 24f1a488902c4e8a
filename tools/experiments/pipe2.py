"""Batch k + 1's preparation (plan + pack launch, Ritz launch) on a SECOND stream under batch k's gains
+ forward, against the one-stream forms.  LANCZOSNET_HIP_LIB selects the Ritz register budget."""
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
net = net.cuda(); plan = net._plan()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mk = t(b['node_mask'].astype(np.uint8))
nf = t(b['node_feat']); A = L[..., 0]; K = 20
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
UNDER_FORWARD = os.environ.get('UNDER_FORWARD', '1') == '1'
gains = lambda D, rows: ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'], rows=rows, zero_fill=False)

def prep_side(after=None):
  if after is not None:
    side.wait_event(after)
  else:
    side.wait_stream(main)
  with torch.cuda.stream(side):
    Lp, tiles, rows = ops.pack_and_plan(plan, L, mk, K)
    D, V = ops.lanczos_ritz(A, n, K)
    ev = torch.cuda.Event(); ev.record(side)
  for x in (Lp, Lp.ident, tiles[0], tiles[0].strips, rows[0], rows[1], D, V):
    x.record_stream(main)
  return (Lp, tiles, rows, D, V), ev

def run_two_streams(steps):
  cur, ev = prep_side(); main.wait_event(ev)
  score = None
  for k in range(steps):
    Lp, tiles, rows, D, V = cur
    G = gains(D, rows)
    if UNDER_FORWARD:   # the preparation becomes eligible together with the forward
      eg = torch.cuda.Event(); eg.record(main)
      score = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
      nxt, ev = prep_side(after=eg)
    else:
      nxt, ev = prep_side()
      score = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
    main.wait_event(ev)
    cur = nxt
  return score

def run_sequential(steps):
  score = None
  for k in range(steps):
    Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mk, n, K)
    score = ops.lanczosnet_forward(plan, nf, Lp, V, gains(D, rows), mk, tiling=tiles)
  return score

def timed(f, steps=100):
  f(10); torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    t0 = time.perf_counter(); s = f(steps); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
  return round(best, 4), s

def ritz_alone():
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  for _ in range(5): ops.lanczos_ritz(A, n, K)
  e0.record()
  for _ in range(50): ops.lanczos_ritz(A, n, K)
  e1.record(); torch.cuda.synchronize()
  return round(e0.elapsed_time(e1) / 50, 4)

with torch.no_grad():
  seq, s1 = timed(run_sequential)
  two, s2 = timed(run_two_streams)
  seq2, _ = timed(run_sequential)
print(json.dumps({'lib': os.environ.get('LANCZOSNET_HIP_LIB', 'default'), 'ritz_alone_ms': ritz_alone(),
                  'sequential_ms': seq, 'two_streams_ms': two, 'sequential_again_ms': seq2,
                  'scores_equal': bool(torch.equal(s1, s2))}))
