import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for N in (100, 128, 130, 150, 160, 192):
  for kern in ('auto', 'workgroup_ql'):
    rs = np.random.RandomState(N)
    B = 32
    adj = np.zeros((B, N, N, 1), np.float32)
    for b in range(B):
      a = np.triu((rs.rand(N, N) < 0.2).astype(np.float32), 1); adj[b, :, :, 0] = a + a.T
    n_d = t(np.full(B, N, np.int32)); L = ops.laplacian_l4(t(adj), n_d)
    D, V, info = ops.lanczos_ritz(L[..., 0], n_d, 20, return_info=True, kernel=kern)
    A = L[..., 0].double(); Vd, Dd = V.double(), D.double()
    eye = torch.eye(20, device='cuda', dtype=torch.float64)[None]
    print(N, kern, 'orth %.2e resid %.2e ql %d' % (float((Vd.transpose(1, 2) @ Vd - eye).abs().max()),
          float((A @ Vd - Vd * Dd[:, None, :]).abs().max()), int((info >= 256).sum())))
