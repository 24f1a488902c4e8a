#!/usr/bin/env python
"""Where the workgroup-per-graph Ritz kernel spends its cycles (needs tools/libprobe_ritz_wg.so =
the library with lanczos_ritz_wg.hip built -DLNZ_PROFILE_PHASES):
    tools/experiments/build_variant.sh lanczos_ritz_wg.hip probe:"-DLNZ_PROFILE_PHASES"
    LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_wg_probe.so \
        python tools/ritz_wg_phase_probe.py [workgroup workgroup_mw]
('workgroup': the wave-level Lanczos phase of graphs with the basis in LDS; 'workgroup_mw': the
eight-wave one)"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lanczosnet_amd import ops  # noqa: E402
from tools.bench_ritz_wg import laplacians  # noqa: E402

rs = np.random.RandomState(0)
KERNELS = sys.argv[1:] or ['workgroup']
for N, kern in [(N, k) for N in (48, 64, 100, 128, 192) for k in KERNELS]:
  A, ns = laplacians(rs, 8, N, N, N, 0.5)
  if N > 108 and kern != KERNELS[0]:
    continue
  D, V = ops.lanczos_ritz(torch.from_numpy(A).cuda(), torch.from_numpy(ns).cuda(), 20,
                          kernel=kern if N <= 108 else 'auto')
  d = D.cpu().numpy()[:, :13].mean(axis=0)
  print('N=%3d %-13s Lanczos %9.0f cycles  QL %9.0f  order+output %8.0f   (QL %.0f cycles per n^2)'
        % (N, kern, d[0], d[1], d[2], d[1] / (N * N)))
  print('       Lanczos parts: A w + norm %8.0f  dots %8.0f  coefficient sums %8.0f  update %8.0f' % tuple(d[4:8]))
  print('       after the eigenvalues: ordering %7.0f  twisted vectors %8.0f  V = Q S %8.0f  sign scan %8.0f  sign apply %7.0f' % tuple(d[8:13]))
