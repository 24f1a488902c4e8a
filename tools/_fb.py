import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
worst_g = worst_r = 0
for seed in range(8):
  b = draw_batch(1024, seed=seed)
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V, info = ops.lanczos_ritz(L[..., 0], n, 20, return_info=True)
  info = info.cpu().numpy(); fb = info >= 256
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): ops.lanczos_ritz(L[..., 0], n, 20)
  e1.record(); torch.cuda.synchronize()
  Dn = D.cpu().numpy().astype(np.float64); Vn = V.cpu().numpy().astype(np.float64); A = L[..., 0].cpu().numpy().astype(np.float64)
  kk = np.minimum(b['n_nodes'], 20)
  G = np.einsum('bnk,bnl->bkl', Vn, Vn)
  eye = (np.arange(20)[None, :] < kk[:, None])
  gerr = np.abs(G - np.eye(20)[None] * eye[:, :, None]).max()
  res = np.abs(np.einsum('bnm,bmk->bnk', A, Vn) - Vn * Dn[:, None, :]).max()
  worst_g = max(worst_g, gerr); worst_r = max(worst_r, res)
  print('seed', seed, 'QL fallbacks', int(fb.sum()), 'ritz ms %.4f' % (e0.elapsed_time(e1) / 20), 'gram err %.1e res %.1e' % (gerr, res))
print('worst', worst_g, worst_r)
