import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
import test_gpu_ada as T
seed = int(sys.argv[1])
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
with torch.no_grad():
  plan = net._plan()
  torch.manual_seed(0)
  q1 = torch.randn(B, N, 1, device='cuda')
  Le = ops.ada_graph_laplacian(t(b['node_feat']), net.embedding.weight, L[:, :, :, 0])
  Tm, Q = ops.ada_lanczos_layer(Le, t(b['node_mask']), q1, K)
  tcat = ops.ada_t_powers(Tm, cfg['long_diffusion_dist']).view(B, -1)
  DDp = net._ada_dense_filters(plan, tcat)
  Lp = ops.pack_laplacian(L)
  mk = t(b['node_mask'])
  tiles = ops.plan_tiles(mk, True)
  out = {}
  for name, env in (('strips', dict(LNZ_STRIPS='1', LNZ_FORWARD16='1')), ('tiles16', dict(LNZ_STRIPS='0', LNZ_FORWARD16='1')),
                    ('tiles32', dict(LNZ_STRIPS='0', LNZ_FORWARD16='0'))):
    os.environ.update(env)
    out[name] = ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, Q, DDp, mk, tiling=tiles).cpu().numpy()
  os.environ.update(dict(LNZ_STRIPS='0', LNZ_FORWARD16='0'))
  out['single32'] = ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, Q, DDp, mk, tiling='single').cpu().numpy()
ref, _ = oracle.ada_lanczos_net_forward(P, cfg, b['node_feat'], L.cpu().numpy(), b['node_mask'], None, dtype=np.float64,
                                        TQ=(Tm.cpu().numpy(), Q.cpu().numpy()))
sp = tiles.strips.cpu().numpy() if getattr(tiles, 'strips', None) is not None else None
print('finite: T', np.isfinite(Tm.cpu().numpy()).all(), 'Q', np.isfinite(Q.cpu().numpy()).all(), 'DDp', np.isfinite(DDp.cpu().numpy()).all())
bad_dd = np.nonzero(~np.isfinite(DDp.cpu().numpy()).reshape(nl, B, -1).all(axis=(0, 2)))[0]
print('molecules with non-finite filters', bad_dd, b['n_nodes'][bad_dd])
print('oracle nan molecules', np.nonzero(~np.isfinite(ref).all(axis=1))[0])
if sp is not None:
  cap = (sp.size - 1) // 80
  for s_ in range(int(sp[cap * 80])):
    e = sp[s_ * 80:(s_ + 1) * 80]
    mols = [(int(e[2 + 3 * i]), int(e[3 + 3 * i]), int(e[4 + 3 * i])) for i in range(int(e[0]))]
    if any(m[0] in (84, 88, 90) for m in mols) or any(m[0] in bad_dd for m in mols): print('strip', s_, 'sub', int(e[1]), mols)
print('cfg', B, K, cfg['short_diffusion_dist'], cfg['long_diffusion_dist'], 'n range', b['n_nodes'].min(), b['n_nodes'].max())
for k, v in out.items():
  print(k, 'nan molecules', np.nonzero(~np.isfinite(v).all(axis=1))[0])
  d = np.nan_to_num(np.abs(v - ref), nan=0.0).max(axis=1)
  print(k, 'max dev vs oracle %.3e' % d.max(), 'molecules off > 1e-4:', np.nonzero(d > 1e-4)[0][:10], b['n_nodes'][np.nonzero(d > 1e-4)[0][:10]])
