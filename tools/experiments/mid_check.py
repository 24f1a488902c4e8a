"""The fused mid-size-graph forward (csrc/conv_mid.hip) against the streamed large-graph kernels on
the reference's graph configuration: scores and launch times."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNetGeneral
from lanczosnet_amd.utils.arg_helper import make_model_config
from graph_fixture import GRAPH_CFG
cfg = dict(GRAPH_CFG)
P = oracle.make_lanczosnet_params(cfg, 3, general=True)
net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}); net = net.cuda()
rs = np.random.RandomState(0)
B, N = int(os.environ.get('MID_B', '64')), 100
ns = rs.randint(20, N + 1, size=B); ns[0] = N
adj = np.zeros((B, N, N, 1), np.float32)
for b in range(B):
  a = np.triu((rs.rand(ns[b], ns[b]) < 0.5).astype(np.float32), 1)
  adj[b, :ns[b], :ns[b], 0] = a + a.T
mask = (np.arange(N)[None, :] < ns[:, None]).astype(np.uint8)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(ns.astype(np.int32)); L = ops.laplacian_l4(t(adj), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
X = t(rs.randn(B, N, 10).astype(np.float32)); mk = t(mask)
def timed(f, reps=30):
  f(); torch.cuda.synchronize()
  e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  e[0].record()
  for _ in range(reps): f()
  e[1].record(); torch.cuda.synchronize()
  return round(e[0].elapsed_time(e[1]) / reps, 4)
with torch.no_grad():
  net.mid_graph_kernel = True
  s_mid = net(X, L, D, V, mask=mk); torch.cuda.synchronize()
  net.mid_graph_kernel = False
  s_large = net(X, L, D, V, mask=mk); torch.cuda.synchronize()
  rel = float(((s_mid - s_large).abs().amax(1) / s_large.abs().amax(1)).max())
  out = {'B': B, 'rel_dev_mid_vs_large': rel, 'finite': bool(torch.isfinite(s_mid).all())}
  net.mid_graph_kernel = True
  out['mid_ms'] = timed(lambda: net(X, L, D, V, mask=mk))
  net.mid_graph_kernel = False
  out['large_ms'] = timed(lambda: net(X, L, D, V, mask=mk))
  # the launch alone
  plan = net._plan_mid(); mid = plan['mid']
  X0 = torch.nn.functional.pad(X, (0, mid['din0p'] - X.shape[2])).contiguous()
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
  Vc = V.float().contiguous()
  out['mid_launch_ms'] = timed(lambda: ops.midgraph_forward(X0, L, Vc, G, mk, mid['W'], mid['bias'], mid['Whead'], mid['bhead'], 7))
  out['gains_ms'] = timed(lambda: ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack']))
  if os.environ.get('MID_PHASES'):
    ops.midgraph_forward(X0, L, Vc, G, mk, mid['W'], mid['bias'], mid['Whead'], mid['bhead'], 7); torch.cuda.synchronize()
    st = ops.midgraph_forward.last_sync[-16:-7].cpu().numpy() * 16 / 1e3
    out['phases_kcycles(prologue,head,Y,long,edge+lift,publish,wait,reload,readout)'] = [round(float(x), 1) for x in st]
print(json.dumps(out))
