"""One-off: HIP backward (input-grad, messages, gain-grad kernels) vs torch autograd on random
architectures / batches.  usage: fuzz_backward.py SEED_LO SEED_HI"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
  rs = np.random.RandomState(1000 + seed)
  nl = int(rs.randint(1, 4))
  cfg = dict(num_atom=17, num_bond_type=int(rs.randint(1, 5)),
             short_diffusion_dist=sorted(rs.choice(np.arange(1, 5), size=rs.randint(0, 3), replace=False).tolist()),
             long_diffusion_dist=sorted(rs.choice(np.arange(1, 12), size=rs.randint(1, 9), replace=False).tolist()),
             num_eig_vec=int(rs.choice([4, 9, 12, 20, 27, 32])), spectral_filter_kind='MLP',
             input_dim=int(rs.choice([32, 64, 96, 128])), hidden_dim=[128] * nl,
             output_dim=int(rs.randint(1, 20)), num_layer=nl)
  B = int(rs.randint(1, 90))
  batch = draw_batch(B, seed=seed, n_min=int(rs.randint(1, 6)), n_max=int(rs.choice([9, 16, 24, 32])),
                     num_atom=17, num_bond_type=cfg['num_bond_type'], num_label=cfg['output_dim'])
  net = T._model(cfg, oracle.make_lanczosnet_params(cfg, 50 + seed)).train()
  assert net._fused_backward_supported(), cfg
  n = T._t(batch['n_nodes'])
  L = ops.laplacian_l4(T._t(batch['adjs']), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, cfg['num_eig_vec'])
  nf, mask, label = T._t(batch['node_feat']), T._t(batch['node_mask']), T._t(batch['label'])
  got = {}
  for impl in ('hip', 'torch'):
    net.backward_impl = impl
    net.zero_grad(set_to_none=True)
    score, loss = net(nf, L, D, V, label=label, mask=mask)
    loss.backward()
    got[impl] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
  for k in got['hip']:
    a, b = got['hip'][k], got['torch'][k]
    if not torch.isfinite(a).all() or (a - b).norm().item() > 1e-3 * b.norm().item() + 1e-9:
      bad.append((seed, k, float((a - b).norm()), float(b.norm())))
      break
print('seeds %d..%d: %d failures' % (lo, hi - 1, len(bad)))
for x in bad[:10]:
  print(x)
