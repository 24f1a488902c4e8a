"""Random LanczosNet configurations through one training step: every parameter gradient of the HIP
backward (strip kernels, message pass on strips, MLP / embedding gradient kernels) against autograd
through the torch restatement of the same forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops, model
from lanczosnet_amd.utils.arg_helper import make_model_config
from lanczosnet_amd.synthetic import draw_batch
DEV = 'cuda:0'
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
worst = 0.0
for seed in range(int(os.environ.get('FUZZ_FROM', '0')), int(os.environ.get('FUZZ_TO', '24'))):
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
  for impl in ('hip', 'torch'):
    net.backward_impl = impl
    net.zero_grad(set_to_none=True)
    _, loss = net(t(b['node_feat']), L, D, V, label=t(b['label']), mask=t(b['node_mask']))
    loss.backward()
    grads[impl] = {k: p.grad.double().clone() for k, p in net.named_parameters()}
  e, who = 0.0, None
  for k in grads['hip']:
    ek = float((grads['hip'][k] - grads['torch'][k]).abs().max() / grads['torch'][k].abs().max().clamp_min(1e-30))
    if ek > e: e, who = ek, k
  worst = max(worst, e)
  print('seed %2d E=%d nl=%d K=%2d din=%3d B=%4d n=%d..%d layers=%d: worst %.2e (%s)%s'
        % (seed, E, nl, K, din, B, nmin, nmax, nlay, e, who, '' if e < 1e-4 else '   <-- LOOK'))
print('worst', worst)
