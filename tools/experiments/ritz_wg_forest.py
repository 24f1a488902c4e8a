import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for N in [int(x) for x in os.environ.get('FOREST_N', '100,128,136,144,152,160,176,192').split(',')]:
  rs = np.random.RandomState(N + int(os.environ.get('FOREST_SEED', '0')))
  B = int(os.environ.get('FOREST_B', '48'))
  ns = rs.randint(max(33, N - 16), N + 1, size=B)
  adj = np.zeros((B, N, N, 1), np.float32)
  for b in range(B):
    n = int(ns[b]); a = np.zeros((n, n), np.float32); m = rs.randint(3, 9)
    for i in range(1, n):
      a[(i - 1) // m * m if i % m else max(i - m, 0), i] = 1.0
    adj[b, :n, :n, 0] = np.maximum(a, a.T)
  n_d = t(ns.astype(np.int32)); L = ops.laplacian_l4(t(adj), n_d)
  for kern in ('auto', 'workgroup_ql'):
    D, V, info = ops.lanczos_ritz(L[..., 0], n_d, 20, return_info=True, kernel=kern)
    Vd = V.double(); eye = torch.eye(20, device='cuda', dtype=torch.float64)[None]
    per = (Vd.transpose(1, 2) @ Vd - eye).abs().amax(dim=(1, 2)).cpu().numpy()
    bad = np.nonzero(per > 1e-5)[0]
    print(N, kern, 'bad', len(bad), 'of', B, 'ql', int((info >= 256).sum()), 'restarts max', int((info % 256).max()),
          [(int(ns[i]), int(info[i])) for i in bad[:4]])
