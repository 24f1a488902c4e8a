#!/usr/bin/env python
"""Workgroup-per-graph Ritz kernel (csrc/lanczos_ritz_wg.hip) on the reference's graph configuration
sizes: G(n, 0.5) graphs with n ~ U{20..100} (dataset/get_graph_data.py:15-49), batch 64 / 256 / 1024,
and fixed-size batches at N = 64, 100, 113, 128, 192 — launch time, graphs/s, algorithmic HBM GB/s
(4 n^2 + 4 K + 4 N K bytes per graph), next to numpy.linalg.eigh + the |lambda| sort on the host."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lanczosnet_amd import ops  # noqa: E402


def laplacians(rs, B, N, n_lo, n_hi, p):
  A = np.zeros((B, N, N), np.float32)
  ns = rs.randint(n_lo, n_hi + 1, size=B).astype(np.int32)
  for b in range(B):
    n = ns[b]
    a = np.triu((rs.rand(n, n) < p).astype(np.float64), 1)
    a = a + a.T + np.eye(n)
    d = 1.0 / np.sqrt(a.sum(axis=1))
    A[b, :n, :n] = d[:, None] * a * d[None, :]
  return A, ns


def time_kernel(A, ns, K, kernel, reps=5):
  ops.lanczos_ritz(A, ns, K, kernel=kernel)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    ops.lanczos_ritz(A, ns, K, kernel=kernel)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def main():
  rs = np.random.RandomState(0)
  K = 20
  out = []
  cases = [('graph config n~U{20..100}', B, 100, 20, 100) for B in (64, 256, 1024)]
  cases += [('fixed n=N', 256, N, N, N) for N in (48, 64, 100, 113, 128, 192)]
  for name, B, N, lo, hi in cases:
    A, ns = laplacians(rs, B, N, lo, hi, 0.5)
    Ad, nd = torch.from_numpy(A).cuda(), torch.from_numpy(ns).cuda()
    row = dict(case=name, B=B, N=N)
    kernels = ['auto'] + (['workgroup'] if N <= 64 else []) + (['workgroup_ws', 'workgroup_mw'] if N <= 108 else [])
    for kern in kernels:
      ms = time_kernel(Ad, nd, K, kern)
      bytes_ = float((4.0 * ns.astype(np.float64) ** 2).sum() + B * (4 * K + 4 * N * K))
      row[kern] = dict(ms=round(ms, 4), graphs_per_s=round(B / ms * 1e3, 1),
                       algorithmic_GBps=round(bytes_ / ms / 1e6, 2))
    t0 = time.perf_counter()
    nb = min(B, 64)
    for b in range(nb):
      n = ns[b]
      e, v = np.linalg.eigh(A[b, :n, :n].astype(np.float64))
      idx = np.argsort(-np.abs(e), kind='mergesort')
      e, v = e[idx[:K]], v[:, idx[:K]]
    row['numpy_eigh_ms_per_graph'] = round((time.perf_counter() - t0) / nb * 1e3, 4)
    out.append(row)
    print(json.dumps(row))
  return out


if __name__ == '__main__':
  main()
