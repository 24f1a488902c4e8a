import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
D, V, info = ops.lanczos_ritz(L[..., 0], n, 20, return_info=True)
D = D.cpu().numpy().astype(np.float64); V = V.cpu().numpy().astype(np.float64); info = info.cpu().numpy()
A = L[..., 0].cpu().numpy().astype(np.float64)
bad = 0
for i in range(1024):
  k = min(int(b['n_nodes'][i]), 20)
  G = V[i][:, :k].T @ V[i][:, :k]
  err = np.abs(G - np.eye(k)).max()
  res = np.abs(A[i] @ V[i][:, :k] - V[i][:, :k] * D[i][:k]).max()
  if err > 1e-5 or res > 1e-5:
    bad += 1
    if bad <= 6:
      off = np.abs(G - np.eye(k)); p, q = np.unravel_index(off.argmax(), off.shape)
      ev = np.linalg.eigvalsh(A[i][:b['n_nodes'][i], :b['n_nodes'][i]])
      print('mol', i, 'n', b['n_nodes'][i], 'restarts', info[i], 'gram err %.2e' % err, 'res %.2e' % res, 'pair', p, q, 'D', D[i][p], D[i][q])
      print('   true eigs near:', ev[np.abs(ev - D[i][p]) < 1e-3])
print('bad', bad)
