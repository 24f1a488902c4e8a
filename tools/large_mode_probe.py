#!/usr/bin/env python
"""The symmetric large-graph Lanczos launch sits at ~26.3 or ~29.2 ms depending on the process.
Is it where the 4.3 GB of A (or the workspace) landed?  One process, several live copies of the
same A and of the workspace, every combination timed."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lanczosnet_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--copies', type=int, default=5)
args = ap.parse_args()
B, N, M = 256, 2048, 64
g = torch.Generator(device='cuda'); g.manual_seed(0)
A0 = torch.empty((B, N, N), dtype=torch.float32, device='cuda')
for b in range(B):
  adj = (torch.rand((N, N), generator=g, device='cuda') < 0.01).float().triu(1)
  adj = adj + adj.t() + torch.eye(N, device='cuda')
  d = adj.sum(1).rsqrt()
  A0[b] = d[:, None] * adj * d[None, :]
need = ops._abi().lanczos_ritz_large_workspace_bytes(B, N)
As = [A0] + [A0.clone() for _ in range(args.copies - 1)]
Ws = [torch.empty((need,), dtype=torch.uint8, device='cuda') for _ in range(2)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def run(A, ws):
  ops.lanczos_ritz_large(A, M, M, workspace=ws, symmetric=True)
  torch.cuda.synchronize()
  ts = []
  for _ in range(2):
    ev[0].record(); ops.lanczos_ritz_large(A, M, M, workspace=ws, symmetric=True); ev[1].record()
    torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
  return round(min(ts), 2)
out = {'A_ptrs': [hex(a.data_ptr()) for a in As], 'ws_ptrs': [hex(w.data_ptr()) for w in Ws],
       'ms[copy of A][workspace]': [[run(a, w) for w in Ws] for a in As]}
print(json.dumps(out))
