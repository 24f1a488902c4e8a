#!/usr/bin/env python
"""SURVEY.md §8(d) 'launch-bound caveat': the small-graph Lanczos / eigensolve / select kernel
(6,768 algorithmic bytes per molecule at N=32, K=20) over a batch sweep, priced against BOTH
rooflines it could meet: HBM (8 TB/s) and the fp64 vector pipe (78.6 TFLOP/s).  At ~13 useful
flop per byte the kernel is on the fp64 side, and at B = 1024 (one wavefront per SIMD) it is
bound by the LATENCY of its dependent chains, not by either roof.  Useful flops per molecule
(2 per FMA, no lane redundancy counted): Lanczos 2 n^3 (Gram-Schmidt against the basis, twice
where needed ~ once) + 2 n^3 (SpMV), eigenvalue search passes x 4 probes x n^2 x 3 with the
measured ~11 passes, twisted vectors 20 n^2, V = Q S 2 n^3."""
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
nn = base['n_nodes'].astype(np.float64)
flop = float(np.mean(6.0 * nn ** 3 + 11 * 4 * 3 * nn ** 2 + 20 * nn ** 2))
for r in rows:
  r['useful_fp64_TFLOPs'] = round(r['molecules_per_s'] * flop / 1e12, 3)
  r['frac_of_fp64_vector_peak'] = round(r['molecules_per_s'] * flop / 78.6e12, 4)
  r['frac_of_hbm_peak'] = round(r['algorithmic_GBps'] / 8000.0, 4)
print(json.dumps({'kernel': 'lanczos_ritz32_kernel', 'bytes_per_molecule': 6768,
                  'useful_fp64_flop_per_molecule': round(flop), 'flop_per_byte': round(flop / 6768, 1),
                  'peaks': {'hbm_GBps': 8000.0, 'fp64_vector_TFLOPs': 78.6},
                  'bound': 'fp64 vector pipe by arithmetic intensity; latency of the dependent chains at '
                           'B = 1024 (one wavefront per SIMD)', 'sweep': rows}))
