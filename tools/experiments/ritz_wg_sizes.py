"""Workgroup Ritz kernel: launch time at fixed n = N for the library the environment selects
(LANCZOSNET_HIP_LIB), 'auto' against the eight-wave Lanczos phase ('workgroup_mw')."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
from bench_ritz_wg import laplacians, time_kernel
rs = np.random.RandomState(0)
for N in (40, 40, 56, 64, 72, 88, 100):
  A, ns = laplacians(rs, 256, N, N, N, 0.5)
  Ad, nd = torch.from_numpy(A).cuda(), torch.from_numpy(ns).cuda()
  print(N, {k: round(time_kernel(Ad, nd, 20, k), 4) for k in (os.environ.get('RITZ_WG_KERNELS') or 'workgroup_p1 workgroup_p2 workgroup_p4 workgroup_mw').split()}, flush=True)
