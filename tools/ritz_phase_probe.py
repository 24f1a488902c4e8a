"""Diagnostic: clock64 phase times of the one-wavefront Ritz kernel per molecule size (needs the
library built with -DLNZ_PROFILE_PHASES: tools/experiments/build_variant.sh lanczos_ritz.hip
phases:"-DLNZ_PROFILE_PHASES", then LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_phases.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
for _ in range(3):
  D, V, info = ops.lanczos_ritz(L[..., 0], n, 20, return_info=True)
torch.cuda.synchronize()
D = D.cpu().numpy(); info = info.cpu().numpy()
print('n   count  lanczos   eig   out | eigenvalue  vector (pivots, twist)  V=QS (kcycles, mean)   restarts(mean)')
for nn in sorted(set(D[:, 3].astype(int))):
  m = D[:, 3].astype(int) == nn
  print('%2d %5d %8.1f %8.1f %6.1f | %8.1f %8.1f (%5.1f %5.1f) %8.1f   %.2f' % (nn, m.sum(), D[m, 0].mean() / 1e3, D[m, 1].mean() / 1e3, D[m, 2].mean() / 1e3, D[m, 4].mean() / 1e3, D[m, 5].mean() / 1e3, D[m, 7].mean() / 1e3, D[m, 8].mean() / 1e3, D[m, 6].mean() / 1e3, info[m].mean()))
print('max total kcycles', (D[:, 0] + D[:, 1] + D[:, 2]).max() / 1e3)
tot = D[:, 0] + D[:, 1] + D[:, 2]
order = np.argsort(-tot)[:14]
print('slowest molecules: total | n restarts(+256 = QL sweep) | lanczos eigenvalue vector V=QS out (kcycles)')
for i in order:
  print('%7.1f | %2d %4d | %6.1f %6.1f %6.1f %5.1f %5.1f' % (tot[i] / 1e3, int(D[i, 3]), int(info[i]), D[i, 0] / 1e3, D[i, 4] / 1e3, D[i, 5] / 1e3, D[i, 6] / 1e3, D[i, 2] / 1e3))
print('percentiles of the total (kcycles): 50%% %.1f  90%% %.1f  99%% %.1f  max %.1f' % tuple(np.percentile(tot, [50, 90, 99, 100]) / 1e3))
