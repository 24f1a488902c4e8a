"""One seed of ritz_wg_fuzz.py in detail: per graph the sorted eigenvalues of every kernel variant
against numpy.linalg.eigh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
for seed in [int(a) for a in sys.argv[1:]]:
  rs = np.random.RandomState(9000 + seed)
  N = int(rs.choice([40, 64, 65, 72, 96, 100, 108]))
  B = 12
  p = float(rs.choice([0.015, 0.03, 0.08, 0.2, 0.5, 0.9]))
  sizes = rs.randint(33 if N > 40 else 20, N + 1, size=B).astype(np.int32)
  sizes[0] = N
  A = np.zeros((B, N, N), np.float32)
  for b, n in enumerate(sizes):
    adj = np.triu((rs.rand(n, n) < p).astype(np.float64), 1)
    A[b, :n, :n] = oracle.laplacian_l4(adj + adj.T)
  Kk = 24 if p >= 0.1 else N
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
  print('seed', seed, 'N', N, 'p', p, 'K', Kk, 'sizes', sizes.tolist())
  out = {}
  for kern in ('auto', 'workgroup_mw', 'workgroup_p1', 'workgroup_ws', 'workgroup_ql'):
    D, V, info = ops.lanczos_ritz(t(A), t(sizes), Kk, return_info=True, kernel=kern)
    out[kern] = (D.cpu().numpy(), V.cpu().numpy(), info.cpu().numpy())
  for b, n in enumerate(sizes):
    e = np.linalg.eigh(A[b, :n, :n].astype(np.float64))[0]
    idx = np.argsort(-np.abs(e), kind='mergesort')
    ref = e[idx][:Kk]
    row = []
    for kern, (D, V, info) in out.items():
      k = min(n, Kk)
      slot = np.abs(D[b, :k] - ref[:k]).max()
      ms = np.abs(np.sort(D[b, :k]) - np.sort(ref[:k])).max()
      Vb = V[b, :n, :k].astype(np.float64)
      orth = np.abs(Vb.T @ Vb - np.eye(k)).max()
      res = np.abs(A[b, :n, :n].astype(np.float64) @ Vb - Vb * D[b, :k][None, :]).max()
      row.append('%s: slot %.1e set %.1e orth %.1e res %.1e info %d' % (kern[-4:], slot, ms, orth, res, info[b]))
    flag = any(float(r.split('slot ')[1].split()[0]) > 1e-6 for r in row)
    if flag:
      print(' graph', b, 'n', n)
      for r in row:
        print('    ', r)
      k = min(n, Kk)
      D = out['auto'][0]
      j = int(np.argmax(np.abs(D[b, :k] - ref[:k])))
      print('     around slot', j, 'auto', D[b, max(0, j - 2):j + 3], 'eigh', ref[max(0, j - 2):j + 3])
