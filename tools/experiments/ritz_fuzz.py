"""The one-wavefront Ritz kernel over many seeds x 1024 molecules: V^T V = I, |A V - V D|, QL
fallbacks, restarts, and the eigenvalues against numpy eigh on a sample."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
lo, hi = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 48
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
worst = dict(orth=0.0, resid=0.0, dD=0.0, ql=0, seeds=0)
for seed in range(lo, hi):
  kw = {} if seed % 3 else dict(n_min=1, n_max=32, N=32)
  b = draw_batch(1024, seed=seed, **kw)
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V, info = ops.lanczos_ritz(L[..., 0], n, 20, return_info=True)
  A = L[..., 0].double(); Vd, Dd = V.double(), D.double()
  kk = torch.clamp(n, max=20).long()
  eye = torch.diag_embed((torch.arange(20, device='cuda')[None, :] < kk[:, None]).double())
  worst['orth'] = max(worst['orth'], float((Vd.transpose(1, 2) @ Vd - eye).abs().max()))
  worst['resid'] = max(worst['resid'], float((A @ Vd - Vd * Dd[:, None, :]).abs().max()))
  worst['ql'] += int((info >= 256).sum())
  for bb in range(0, 1024, 61):
    nb = int(b['n_nodes'][bb])
    lam = torch.linalg.eigvalsh(A[bb, :nb, :nb].cpu())
    want = lam[torch.argsort(-lam.abs(), stable=True)][:min(nb, 20)]
    got = Dd[bb, :min(nb, 20)].cpu()
    worst['dD'] = max(worst['dD'], float((torch.sort(got).values - torch.sort(want).values).abs().max()))
  worst['seeds'] += 1
print(json.dumps(worst))
