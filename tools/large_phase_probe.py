#!/usr/bin/env python
"""Where the large-graph Lanczos launch spends its time (wave 0 of every workgroup, 100 MHz
wall clock, summed over the workgroups): needs a library built with -DLNZ_LARGE_PROBE
(csrc/lanczos_large.hip) given as LANCZOSNET_HIP_LIB.  Prints microseconds per workgroup."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lanczosnet_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--nodes', type=int, default=2048)
ap.add_argument('--steps', type=int, default=64)
ap.add_argument('--sym', action='store_true')
ap.add_argument('--compact', action='store_true', help='lnz_lanczos_ritz_kstep on the sliced-ELL image')
args = ap.parse_args()
B, N, M = args.batch, args.nodes, args.steps
g = torch.Generator(device='cuda'); g.manual_seed(0)
A = torch.empty((B, N, N), dtype=torch.float32, device='cuda')
for b in range(B):
  adj = (torch.rand((N, N), generator=g, device='cuda') < 0.01).float().triu(1)
  adj = adj + adj.t() + torch.eye(N, device='cuda')
  d = adj.sum(1).rsqrt()
  A[b] = d[:, None] * adj * d[None, :]
lib = C.CDLL(_lib.LIB_PATH)
lib.lnz_debug_large_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
run = (lambda: ops.lanczos_ritz_kstep(A, None, M, M)) if args.compact else \
      (lambda: ops.lanczos_ritz_large(A, M, M, symmetric=args.sym))
run()
torch.cuda.synchronize()
lib.lnz_debug_large_probe(None, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); run(); ev[1].record()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert lib.lnz_debug_large_probe(out, 0) == 0
names = ['normalise + store q', 'SpMV (own jobs)', 'SpMV (wait for the other waves)', 'slot sum',
         'norm before', 'CGS pass 1', 'CGS pass 2', 'norm after', 'tridiagonal QL', 'order + V = Q S']
us = [out[i] / 100.0 / B for i in range(10)]  # 100 MHz ticks -> us, per workgroup
if args.compact:
  names[1:4] = ['-', '-', 'SpMV on the image']
print(json.dumps({'sym': args.sym, 'compact': args.compact, 'launch_ms': round(ev[0].elapsed_time(ev[1]), 3),
                  'sum_ms': round(sum(us) / 1e3, 3),
                  'us_per_workgroup': {n: round(u, 1) for n, u in zip(names, us)}}))
