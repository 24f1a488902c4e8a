"""AdaLanczosNet conv launch (dense K x K filters, short-diffusion channels) on the strip plan against
the 32-row tile plan over random batches.  usage: ada_strip_fuzz.py SEED_LO SEED_HI"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
import test_gpu_ada as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
  rs = np.random.RandomState(3000 + seed)
  nl = int(rs.randint(1, 4))
  K = int(rs.choice([4, 8, 12, 20, 24]))
  cfg = dict(oracle.DEFAULT_QM8_CFG,
             short_diffusion_dist=sorted(rs.choice(np.arange(1, 4), size=rs.randint(0, 3), replace=False).tolist()),
             long_diffusion_dist=sorted(rs.choice(np.arange(1, 12), size=rs.randint(1, 5), replace=False).tolist()),
             hidden_dim=[128] * nl, num_layer=nl, num_eig_vec=K)
  P = oracle.make_ada_params(cfg, 5 + seed)
  net = T._ada_model(cfg, P)
  B = int(rs.randint(1, 150))
  b = draw_batch(B, seed=seed, n_min=int(rs.randint(1, 6)), n_max=int(rs.choice([9, 16, 26])))
  N = b['node_mask'].shape[1]
  t = T._t
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  S = len(cfg['long_diffusion_dist'])
  with torch.no_grad():
    plan = net._plan()
    q1 = torch.randn(B, N, 1, device='cuda')
    Le = ops.ada_graph_laplacian(t(b['node_feat']), net.embedding.weight, L[:, :, :, 0])
    Tm, Q = ops.ada_lanczos_layer(Le, t(b['node_mask']), q1, K)
    tcat = ops.ada_t_powers(Tm, cfg['long_diffusion_dist']).view(B, -1)
    DDp = net._ada_dense_filters(plan, tcat)
    Lp = ops.pack_laplacian(L)
    mk = t(b['node_mask'])
    tiles = ops.plan_tiles(mk, True)
    os.environ['LNZ_STRIPS'] = '1'
    s1 = ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, Q, DDp, mk, tiling=tiles)
    os.environ['LNZ_STRIPS'] = '0'
    s0 = ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, Q, DDp, mk, tiling=tiles)
  ok = torch.isfinite(s1).all().item() and (s1 - s0).abs().max().item() <= 3e-6 * (s0.abs().max().item() + 1e-12)
  if not ok:
    bad.append((seed, B, K, cfg['short_diffusion_dist'], cfg['long_diffusion_dist'], float((s1 - s0).abs().max())))
print('seeds %d..%d: %d failures' % (lo, hi - 1, len(bad)))
for x in bad:
  print(x)
