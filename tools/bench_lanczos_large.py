#!/usr/bin/env python
"""HBM-roofline measurement of the large-graph Lanczos stage (BASELINE.json configs[4] shape:
N = 2048, K = 64, batch 256).  Algorithmic bytes per graph: M * 4 N^2 for A
re-streamed every step + basis traffic 4 * 8 N * M (M + 1) / 2 (fp64, two CGS passes, dots +
update) + 4 N K for V.  Input graphs are drawn with torch on the GPU (generator, not the path)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lanczosnet_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--nodes', type=int, default=2048)
ap.add_argument('--steps', type=int, default=64)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--sym', action='store_true', help='lnz_lanczos_ritz_large_sym (upper chunk blocks only)')
ap.add_argument('--compact', action='store_true',
                help='lnz_lanczos_ritz_kstep with LNZ_KSTEP_COMPACT: A read once into a sliced-ELL image')
ap.add_argument('--density', type=float, default=0.01, help='edge probability of the G(n, p) graphs')
ap.add_argument('--row-cap', type=int, default=64)
ap.add_argument('--row-pad', type=int, default=0,
                help='floats of padding behind every row of A (row stride N + pad): probes the '
                     'sensitivity of the 1 KiB-segment stream to a power-of-two row stride')
args = ap.parse_args()
B, N, M = args.batch, args.nodes, args.steps
g = torch.Generator(device='cuda'); g.manual_seed(0)
A = torch.zeros((B, N, N + args.row_pad), dtype=torch.float32, device='cuda')[:, :, :N]
for b in range(B):  # G(n, p = 0.01) + self loops, symmetric GCN normalisation (L4)
  adj = (torch.rand((N, N), generator=g, device='cuda') < args.density).float().triu(1)
  adj = adj + adj.t() + torch.eye(N, device='cuda')
  d = adj.sum(1).rsqrt()
  A[b] = d[:, None] * adj * d[None, :]
ws = torch.empty((ops._abi().lanczos_ritz_kstep_workspace_bytes(B, N, 3, args.row_cap),), dtype=torch.uint8, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
if args.compact:
  run = lambda: ops.lanczos_ritz_kstep(A, None, M, M, symmetric=True, compact=True, row_cap=args.row_cap,
                                       workspace=ws, return_info=True, return_fallback=True)
else:
  run = lambda: ops.lanczos_ritz_large(A, M, M, workspace=ws, return_info=True, symmetric=args.sym) + (None,)
run()
torch.cuda.synchronize()
ts = []
for _ in range(args.reps):
  ev[0].record(); D, V, info, fb = run(); ev[1].record()
  torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
t = min(ts) * 1e-3
nch = (N + 255) // 256
bytes_A = M * 4 * N * N if not args.sym else M * 4 * 256 * 256 * (nch * (nch + 1) // 2)
bytes_Q = 4 * 8 * N * M * (M + 1) // 2
bytes_V = 4 * N * M
alg = B * (bytes_A + bytes_Q + bytes_V)
if args.compact:
  # what the compacted path moves: A once, the image M times (values + columns of the slab widths),
  # the basis once per Gram-Schmidt pass, V
  nnz = int((A != 0).sum())
  bytes_A = 4 * N * N
  alg_compact = B * (bytes_A + bytes_Q // 4 + bytes_V) + M * nnz * 6
print(json.dumps({'workload': 'lanczos_ritz_%s B=%d N=%d M=K=%d fp32 A (G(n, %g)), fp64 arithmetic' % (
                      'kstep[compact, row_cap %d]' % args.row_cap if args.compact else 'large_sym' if args.sym else 'large', B, N, M, args.density),
                  'dense_fallback_graphs': int(fb.sum()) if fb is not None else None,
                  'compact_path_GB (A once + image x M + basis once per pass + V)': round(alg_compact / 1e9, 2) if args.compact else None,
                  'ms': round(t * 1e3, 3), 'graphs_per_s': round(B / t, 1),
                  'algorithmic_GB': round(alg / 1e9, 2), 'achieved_GBps': round(alg / t / 1e9, 1),
                  'A_only_GBps': round(B * bytes_A / t / 1e9, 1), 'peak_GBps': 8000,
                  'frac_of_hbm_peak': round(alg / t / 8e12, 4), 'steps_taken_min': int(info.min()),
                  'row_pad': args.row_pad, 'all_ms': [round(x, 2) for x in ts]}))
