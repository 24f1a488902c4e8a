"""Adversarial graphs for the one-wavefront Ritz kernel (n <= 32): forests of equal stars / paths,
complete bipartite pieces, rings — massively degenerate spectra."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
N = 32; worst = dict(orth=0.0, resid=0.0, dD=0.0, ql=0, graphs=0, bad=0)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
  rs = np.random.RandomState(seed); B = 1024
  ns = rs.randint(8, N + 1, size=B)
  adj = np.zeros((B, N, N, 1), np.float32)
  for b in range(B):
    n = int(ns[b]); a = np.zeros((n, n), np.float32); kind = b % 4
    if kind == 0:      # caterpillar of equal stars
      m = rs.randint(2, 7)
      for i in range(1, n):
        a[(i - 1) // m * m if i % m else max(i - m, 0), i] = 1.0
    elif kind == 1:    # disjoint equal stars (disconnected)
      m = rs.randint(2, 7)
      for i in range(1, n):
        if i % m: a[i // m * m, i] = 1.0
    elif kind == 2:    # complete bipartite K_{p, n-p}
      p = rs.randint(1, n // 2 + 1); a[:p, p:] = 1.0
    else:              # ring (or two rings)
      for i in range(n): a[i, (i + 1) % n] = 1.0
    adj[b, :n, :n, 0] = np.maximum(a, a.T)
  n_d = t(ns.astype(np.int32)); L = ops.laplacian_l4(t(adj), n_d)
  D, V, info = ops.lanczos_ritz(L[..., 0], n_d, 20, return_info=True)
  A = L[..., 0].double(); Vd, Dd = V.double(), D.double()
  kk = torch.clamp(n_d, max=20).long()
  eye = torch.diag_embed((torch.arange(20, device='cuda')[None, :] < kk[:, None]).double())
  per = (Vd.transpose(1, 2) @ Vd - eye).abs().amax(dim=(1, 2))
  worst['orth'] = max(worst['orth'], float(per.max())); worst['bad'] += int((per > 2e-6).sum())
  worst['resid'] = max(worst['resid'], float((A @ Vd - Vd * Dd[:, None, :]).abs().max()))
  worst['ql'] += int((info >= 256).sum()); worst['graphs'] += B
  for bb in range(0, B, 41):
    nb = int(ns[bb]); lam = torch.linalg.eigvalsh(A[bb, :nb, :nb].cpu())
    want = lam[torch.argsort(-lam.abs(), stable=True)][:min(nb, 20)]
    got = Dd[bb, :min(nb, 20)].cpu()
    worst['dD'] = max(worst['dD'], float((torch.sort(got).values - torch.sort(want).values).abs().max()))
print(json.dumps(worst))
