"""One fuzz case of kstep_fuzz.py in detail: per graph, compact vs symmetric stream vs full stream vs the
fp64 restatement (steps taken, D, projector)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
case, N, K, p, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
sizes = [int(x) for x in sys.argv[6].split(',')]
B = len(sizes); dev = 'cuda'
A = torch.zeros((B, N, N), device=dev)
g = torch.Generator(device=dev); g.manual_seed(case)
for b, n in enumerate(sizes):
  if n == 0: continue
  adj = (torch.rand((n, n), generator=g, device=dev) < p).float().triu(1)
  adj = adj + adj.t() + torch.eye(n, device=dev)
  d = adj.sum(1).rsqrt()
  A[b, :n, :n] = d[:, None] * adj * d[None, :]
nn = torch.tensor(sizes, dtype=torch.int32, device=dev)
res = {}
for name, kw in (('sym', dict(compact=False, symmetric=True)), ('full', dict(compact=False, symmetric=False)),
                 ('compact', dict(compact=True, row_cap=cap))):
  D, V, info = ops.lanczos_ritz_kstep(A, nn, K, K, return_info=True, **kw)
  res[name] = (D.cpu().numpy(), V.cpu().numpy().astype(np.float64), info.cpu().numpy())
for b, n in enumerate(sizes):
  if n == 0: continue
  Dr, Vr, (al, be, steps, last) = oracle.lanczos_kstep_fp64(A[b, :n, :n].cpu().numpy(), K, K)
  line = {'b': b, 'n': n, 'oracle_steps': int(steps), 'oracle_betas_min': float(np.min(be[:max(steps - 1, 1)])) if steps > 1 else None}
  for name, (D, V, info) in res.items():
    P = V[b, :n] @ V[b, :n].T
    line[name] = dict(steps=int(info[b]), eD=float(np.abs(D[b] - Dr).max()), eP=float(np.abs(P - Vr @ Vr.T).max()),
                      orth=float(np.abs(V[b].T @ V[b] - np.diag(np.diag(V[b].T @ V[b]))).max()))
  print(json.dumps(line))
