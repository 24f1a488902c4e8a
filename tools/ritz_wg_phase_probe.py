#!/usr/bin/env python
"""Where the workgroup-per-graph Ritz kernel spends its cycles (needs tools/libprobe_ritz_wg.so =
the library with lanczos_ritz_wg.hip built -DLNZ_PROFILE_PHASES):
    LANCZOSNET_HIP_LIB=tools/libprobe_ritz_wg.so python tools/ritz_wg_phase_probe.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lanczosnet_amd import ops  # noqa: E402
from tools.bench_ritz_wg import laplacians  # noqa: E402

rs = np.random.RandomState(0)
for N in (48, 64, 100, 128, 192):
  A, ns = laplacians(rs, 8, N, N, N, 0.5)
  D, V = ops.lanczos_ritz(torch.from_numpy(A).cuda(), torch.from_numpy(ns).cuda(), 20,
                          kernel='workgroup' if N <= 113 else 'auto')
  d = D.cpu().numpy()[:, :8].mean(axis=0)
  print('N=%3d  Lanczos %9.0f cycles  QL %9.0f  order+output %8.0f   (QL %.0f cycles per n^2)'
        % (N, d[0], d[1], d[2], d[1] / (N * N)))
  print('       Lanczos parts: A w + norm %8.0f  dots %8.0f  coefficient sums %8.0f  update %8.0f' % tuple(d[4:8]))
