"""The workgroup-per-graph Ritz kernel over random graphs of 33..160 nodes (sparse to dense, incl.
highly symmetric ones): V^T V = I, |A V - V D|, D against numpy eigh, QL fallbacks."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
worst = dict(orth=0.0, resid=0.0, dD=0.0, ql=0, graphs=0)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
  rs = np.random.RandomState(seed)
  B, N = 96, int(rs.choice([40, 64, 100, 128, 160]))
  ns = rs.randint(33, N + 1, size=B); ns[0] = N
  adj = np.zeros((B, N, N, 1), np.float32)
  for b in range(B):
    n = int(ns[b]); kind = b % 4
    if kind == 3:   # a forest of equal stars / paths: highly degenerate spectrum
      a = np.zeros((n, n), np.float32); m = rs.randint(3, 9)
      for i in range(1, n):
        a[(i - 1) // m * m if i % m else max(i - m, 0), i] = 1.0
    else:
      a = np.triu((rs.rand(n, n) < [0.03, 0.2, 0.6][kind]).astype(np.float32), 1)
    adj[b, :n, :n, 0] = np.maximum(a, a.T)
  n_d = t(ns.astype(np.int32)); L = ops.laplacian_l4(t(adj), n_d)
  D, V, info = ops.lanczos_ritz(L[..., 0], n_d, 20, return_info=True)
  A = L[..., 0].double(); Vd, Dd = V.double(), D.double()
  eye = torch.eye(20, device='cuda', dtype=torch.float64)[None]
  worst['orth'] = max(worst['orth'], float((Vd.transpose(1, 2) @ Vd - eye).abs().max()))
  worst['resid'] = max(worst['resid'], float((A @ Vd - Vd * Dd[:, None, :]).abs().max()))
  worst['ql'] += int((info >= 256).sum()); worst['graphs'] += B
  per = (Vd.transpose(1, 2) @ Vd - eye).abs().amax(dim=(1, 2)).cpu().numpy()
  bad = np.nonzero(per > 1e-5)[0]
  if len(bad):
    print('seed', seed, 'N', N, 'bad graphs', len(bad), 'kinds', sorted(set(int(x) % 4 for x in bad)), 'n of first', int(ns[bad[0]]), 'info', int(info[bad[0]]), 'err', per[bad[0]], file=sys.stderr)
  for bb in range(0, B, 7):
    nb = int(ns[bb]); lam = torch.linalg.eigvalsh(A[bb, :nb, :nb].cpu())
    want = lam[torch.argsort(-lam.abs(), stable=True)][:20]
    worst['dD'] = max(worst['dD'], float((torch.sort(Dd[bb].cpu()).values - torch.sort(want).values).abs().max()))
print(json.dumps(worst))
