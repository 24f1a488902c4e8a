"""One configuration of train_fuzz.py three ways: HIP backward on strips, on 32-row tiles, torch.
(Ran at the commit before the tile form of the message pass was retired; LNZ_STRIPS=0 now refuses it.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops, model
from lanczosnet_amd.utils.arg_helper import make_model_config
from lanczosnet_amd.synthetic import draw_batch
DEV = 'cuda:0'
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
seed = int(sys.argv[1])
rs = np.random.RandomState(500 + seed)
E = int(rs.choice([1, 2, 3, 6])); nl = int(rs.choice([1, 3, 8])); K = int(rs.choice([8, 12, 20]))
din = int(rs.choice([64, 128])); B = int(rs.choice([3, 40, 300, 1024])); nmax = int(rs.choice([9, 26, 32]))
nmin = int(rs.randint(2, nmax + 1)); nlay = int(rs.choice([2, 4, 7]))
dists = sorted(rs.choice(np.arange(1, 31), size=nl, replace=False).tolist())
cfg = dict(oracle.DEFAULT_QM8_CFG, num_bond_type=E, long_diffusion_dist=dists, num_eig_vec=K, input_dim=din, num_layer=nlay)
cfg['hidden_dim'] = [128] * nlay
net = model.LanczosNet(make_model_config(cfg)).train()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, seed).items()})
net = net.to(DEV)
b = draw_batch(B, seed=seed, n_min=nmin, n_max=nmax, num_bond_type=E)
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); D, V = ops.lanczos_ritz(L[..., 0], n, K)
grads = {}
for name, impl, strips in (('strips', 'hip', '1'), ('tiles', 'hip', '0'), ('torch', 'torch', '1')):
  os.environ['LNZ_STRIPS'] = strips
  net.backward_impl = impl
  net.zero_grad(set_to_none=True)
  score, loss = net(t(b['node_feat']), L, D, V, label=t(b['label']), mask=t(b['node_mask']))
  loss.backward()
  grads[name] = {k: p.grad.double().clone() for k, p in net.named_parameters()}
  grads[name]['_score'] = score.detach().double().clone()
def diff(a, c):
  e, who = 0.0, None
  for k in grads[a]:
    ek = float((grads[a][k] - grads[c][k]).abs().max() / grads[c][k].abs().max().clamp_min(1e-30))
    if ek > e: e, who = ek, k
  return '%.2e (%s)' % (e, who)
print('seed', seed, 'strips vs tiles', diff('strips', 'tiles'), '| strips vs torch', diff('strips', 'torch'), '| tiles vs torch', diff('tiles', 'torch'))
for k in ('filter.0.weight', 'embedding.weight', 'filter.%d.weight' % (nlay - 1)):
  g = {m: grads[m][k] for m in ('strips', 'tiles', 'torch')}
  sc = float(g['torch'].abs().max())
  print('  %-18s strips-tiles %.2e  strips-torch %.2e  tiles-torch %.2e  (of max |g|); entries off by > 1e-4: %d %d %d' % (
      k, float((g['strips'] - g['tiles']).abs().max()) / sc, float((g['strips'] - g['torch']).abs().max()) / sc,
      float((g['tiles'] - g['torch']).abs().max()) / sc,
      int(((g['strips'] - g['tiles']).abs() > 1e-4 * sc).sum()), int(((g['strips'] - g['torch']).abs() > 1e-4 * sc).sum()),
      int(((g['tiles'] - g['torch']).abs() > 1e-4 * sc).sum())))
