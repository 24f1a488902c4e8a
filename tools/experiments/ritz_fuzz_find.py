import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
found = 0
for seed in range(0, 64):
  kw = {} if seed % 3 else dict(n_min=1, n_max=32, N=32)
  b = draw_batch(1024, seed=seed, **kw)
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V, info = ops.lanczos_ritz(L[..., 0], n, 20, return_info=True)
  Vd = V.double()
  kk = torch.clamp(n, max=20).long()
  eye = torch.diag_embed((torch.arange(20, device='cuda')[None, :] < kk[:, None]).double())
  err = (Vd.transpose(1, 2) @ Vd - eye).abs().amax(dim=(1, 2))
  bad = torch.nonzero(err > 1e-5).flatten().tolist()
  for bb in bad[:3]:
    G = (Vd[bb].T @ Vd[bb]).cpu().numpy()
    nb = int(b['n_nodes'][bb])
    lam = np.linalg.eigvalsh(L[bb, :nb, :nb, 0].double().cpu().numpy())
    print('seed', seed, 'mol', bb, 'n', nb, 'info', int(info[bb]), 'diag', np.round(np.diag(G), 3)[:nb], 'D', np.round(D[bb].cpu().numpy(), 6)[:nb])
    print('   eigh sorted by |.|', np.round(lam[np.argsort(-np.abs(lam), kind='stable')], 6)[:20])
    Go = np.abs(G - np.diag(np.diag(G))); ij = np.unravel_index(np.argmax(Go), Go.shape)
    print('   worst off-diagonal', ij, G[ij], 'D_i, D_j =', float(D[bb, ij[0]]), float(D[bb, ij[1]]), 'resid', float((L[bb, :, :, 0].double() @ Vd[bb] - Vd[bb] * D[bb].double()[None, :]).abs().max()))
    found += 1
  if found >= 4: break
print('found', found)
