"""Random configurations through lnz_midgraph_forward (graphs of 33..128 nodes, one launch) against the
float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import oracle
from graph_fixture import GRAPH_CFG
from lanczosnet_amd import ops, model
from lanczosnet_amd.utils.arg_helper import make_model_config
DEV = 'cuda:0'
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
worst = 0.0
for seed in range(int(os.environ.get('FUZZ_FROM', '0')), int(os.environ.get('FUZZ_TO', '40'))):
  rs = np.random.RandomState(900 + seed)
  N = int(rs.randint(33, 129)); K = int(rs.choice([4, 8, 12, 20, 32])); nl = int(rs.choice([0, 1, 3, 8, 16]))
  din = int(rs.choice([3, 10, 16, 64, 128])); B = int(rs.choice([1, 5, 64, 70, 130])); nlay = int(rs.choice([1, 3, 7]))
  nmin = int(rs.randint(2, N + 1)); p = float(rs.choice([0.02, 0.1, 0.5]))
  dists = sorted(rs.choice(np.arange(1, 31), size=nl, replace=False).tolist())
  cfg = dict(GRAPH_CFG, num_bond_type=1, num_eig_vec=K, long_diffusion_dist=dists, input_dim=din, num_layer=nlay)
  cfg['hidden_dim'] = [128] * nlay
  P = oracle.make_lanczosnet_params(cfg, seed, general=True)
  net = model.LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  ns = rs.randint(nmin, N + 1, size=B); ns[rs.randint(B)] = N
  adj = np.zeros((B, N, N, 1), np.float32)
  for b in range(B):
    n = int(ns[b]); a = np.triu((rs.rand(n, n) < p).astype(np.float32), 1); adj[b, :n, :n, 0] = a + a.T
  mask = (np.arange(N)[None, :] < ns[:, None]).astype(np.uint8)
  nd = t(ns.astype(np.int32)); L = ops.laplacian_l4(t(adj), nd)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], nd, K)
  X = rs.randn(B, N, din).astype(np.float32) * mask[:, :, None]
  if not net._mid_hip_supported(N, K, L.shape[3]):
    print('seed', seed, 'not on the mid kernel', N, K, nl, din); continue
  with torch.no_grad():
    got = net(t(X), L, D, V, mask=t(mask)).cpu().numpy()
  ref = oracle.lanczos_net_forward(P, cfg, X, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(), mask, dtype=np.float64, general=True)
  e = float((np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-30)).max())
  worst = max(worst, e)
  print('seed %2d N=%3d K=%2d nl=%2d din=%3d B=%3d n>=%3d p=%.2f layers=%d: per-graph %.2e%s'
        % (seed, N, K, nl, din, B, nmin, p, nlay, e, '' if (e < 1e-5 and np.isfinite(got).all()) else '   <-- FAIL'))
print('worst', worst)
