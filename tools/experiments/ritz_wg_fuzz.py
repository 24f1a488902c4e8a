"""Fuzz of the workgroup Ritz kernel (csrc/lanczos_ritz_wg.hip) over graph sizes 20..N and edge
densities from forests to dense: every graph against numpy.linalg.eigh — the kept eigenvalues as a
multiset to 1e-6 (slot by slot the order of +x and -x with equal |x| is decided by the last bit, in
LAPACK as well: ritz_wg_one.py shows such a pair), V^T V = I to 1e-5, |A V - V D| to 1e-6 — and
against the eight-wave Lanczos phase ('workgroup_mw'); the restart branch (sparse graphs) and the
QL fallback are counted.  Graphs whose top-K cut splits a |lambda| cluster are skipped.
  python tools/experiments/ritz_wg_fuzz.py [first seed] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns_ = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad, restarts, ql = [], 0, 0
for seed in range(s0, s0 + ns_):
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
  Kk = 24 if p >= 0.1 else N   # sparse graphs: degenerate clusters everywhere, nothing is cut
  Dl, Vl, full = [], [], np.zeros((B, N))
  for b, n in enumerate(sizes):
    e, v = np.linalg.eigh(A[b, :n, :n].astype(np.float64))
    idx = np.argsort(-np.abs(e), kind='mergesort')
    Dl.append(e[idx]); Vl.append(v[:, idx]); full[b, :n] = e[idx]
  Dr, Vr = oracle.collate_eigs(Dl, Vl, N, Kk)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
  try:
    D, V, info = ops.lanczos_ritz(t(A), t(sizes), Kk, return_info=True)
    Dm, Vm, im = ops.lanczos_ritz(t(A), t(sizes), Kk, return_info=True, kernel='workgroup_mw')
    assert torch.isfinite(D).all() and torch.isfinite(V).all()
    Dn, Vn, Dmn = D.cpu().numpy(), V.cpu().numpy(), Dm.cpu().numpy()
    for b, n in enumerate(sizes):
      if oracle.degenerate_cut(full[b][:n], Kk):
        continue
      k = min(int(n), Kk)
      assert np.abs(np.sort(Dn[b, :k]) - np.sort(Dr[b, :k])).max() < 1e-6, ('D', b, n)
      assert np.abs(np.sort(Dn[b, :k]) - np.sort(Dmn[b, :k])).max() < 1e-6, ('D vs mw', b, n)
      Vb = Vn[b, :n, :k].astype(np.float64)
      assert np.abs(Vb.T @ Vb - np.eye(k)).max() < 1e-5, ('orth', b, n)
      assert np.abs(A[b, :n, :n].astype(np.float64) @ Vb - Vb * Dn[b, :k][None, :]).max() < 1e-6, ('res', b, n)
      assert (Vn[b, n:] == 0).all() and (Vn[b, :, k:] == 0).all()
    restarts += int((info % 256).sum()); ql += int((info >= 256).sum())
  except AssertionError as e:
    bad.append((seed, N, p, str(e)[:80]))
print('seeds %d..%d: %d failures, %d restarts, %d QL fallbacks' % (s0, s0 + ns_ - 1, len(bad), restarts, ql))
for b in bad:
  print(b)
