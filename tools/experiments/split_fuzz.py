"""Random configurations of the split-precision strip forward (gemm_mode='f16x3') against the float64
oracle: edge types, long scales, K, input width, batch size, node range, holes in the mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops, model
from lanczosnet_amd.utils.arg_helper import make_model_config
from lanczosnet_amd.synthetic import draw_batch

DEV = 'cuda:0'
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
seeds = range(int(os.environ.get('FUZZ_FROM', '0')), int(os.environ.get('FUZZ_TO', '40')))
worst = 0.0
for seed in seeds:
  rs = np.random.RandomState(seed)
  E = int(rs.choice([1, 2, 3, 6]))
  nl = int(rs.choice([0, 1, 3, 8, 12]))
  K = int(rs.choice([4, 8, 12, 20, 32]))
  din = int(rs.choice([16, 30, 64, 100, 128]))
  general = bool(rs.rand() < 0.5)
  B = int(rs.choice([1, 2, 7, 40, 130, 600]))
  nmax = int(rs.choice([3, 9, 17, 26, 32]))
  nmin = int(rs.randint(1, nmax + 1))
  dists = sorted(rs.choice(np.arange(1, 31), size=nl, replace=False).tolist())
  cfg = dict(oracle.DEFAULT_QM8_CFG, num_bond_type=E, long_diffusion_dist=dists, num_eig_vec=K, input_dim=din,
             num_layer=int(rs.choice([1, 2, 4, 7])))
  cfg['hidden_dim'] = [128] * cfg['num_layer']
  P = oracle.make_lanczosnet_params(cfg, seed, general=general)
  cls = model.LanczosNetGeneral if general else model.LanczosNet
  net = cls(make_model_config(cfg, general=general)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  net.gemm_mode = 'f16x3'
  b = draw_batch(B, seed=seed, n_min=nmin, n_max=nmax, num_bond_type=E)
  mask = b['node_mask'].copy()
  if rs.rand() < 0.3:   # holes: the extent stays the last real node + 1
    for i in range(B):
      n = int(b['n_nodes'][i])
      if n > 2:
        mask[i, rs.randint(0, n - 1)] = 0
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, K)
  feat = rs.randn(B, mask.shape[1], din).astype(np.float32) if general else b['node_feat']
  with torch.no_grad():
    got = net(t(feat), L, D, V, mask=t(mask)).cpu().numpy()
  assert net._plan()['gemm_mode'] == 1
  ref = oracle.lanczos_net_forward(P, cfg, feat, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(), mask,
                                   dtype=np.float64, general=general)
  e = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
  worst = max(worst, e)
  flag = '' if (e < 1e-5 and np.isfinite(got).all()) else '   <-- FAIL'
  print('seed %3d E=%d nl=%2d K=%2d din=%3d %s B=%3d n=%d..%d layers=%d: %.2e%s'
        % (seed, E, nl, K, din, 'gen' if general else 'emb', B, nmin, nmax, cfg['num_layer'], e, flag))
print('worst', worst)
