#!/usr/bin/env python
"""SURVEY.md §8(d) 'launch-bound caveat': algorithmic GB/s of the small-graph Lanczos/QL/select
kernel (6,768 B per molecule at N=32, K=20) over a batch sweep."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
base = draw_batch(1024, seed=0, N=32)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n1 = t(base['n_nodes']); L1 = ops.laplacian_l4(t(base['adjs']), n1)[:, :, :, 0].contiguous()
rows = []
for rep in (1, 8, 64, 256):
  B = 1024 * rep
  A = L1.repeat(rep, 1, 1); n = n1.repeat(rep)
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  ops.lanczos_ritz(A, n, 20); torch.cuda.synchronize()
  ts = []
  for _ in range(3):
    ev[0].record(); ops.lanczos_ritz(A, n, 20); ev[1].record(); torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
  ms = min(ts)
  rows.append({'B': B, 'ms': round(ms, 3), 'molecules_per_s': round(B / ms * 1e3),
               'algorithmic_GBps': round(B * 6768 / ms / 1e6, 1)})
  del A
print(json.dumps({'kernel': 'lanczos_ritz32_kernel', 'bytes_per_molecule': 6768, 'sweep': rows}))
