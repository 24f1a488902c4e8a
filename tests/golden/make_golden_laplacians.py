#!/usr/bin/env python
"""Golden fixture for EVERY kind of the reference's get_laplacian (utils/data_helper.py:119-166,
'L1' .. 'L7', alpha = 0.5 and 0.3 for 'L6'), produced by calling the UNMODIFIED reference function
in the build container on

  * bond-type adjacencies of six synthetic molecules (lanczosnet_amd.synthetic.draw_batch; 0/1,
    symmetric, four bond types), one of them with an isolated atom (its row sums are zero: the
    reference's inf -> 0 guard) — per channel like the collate layout: channel 0 = the simple
    graph sum_e A_e, channel 1 + e = bond type e;
  * one weighted, NON-symmetric 9 x 9 single-channel matrix (kernel-like positive weights), which
    separates the symmetric from the asymmetric normalisations.

    python tests/golden/make_golden_laplacians.py      # needs /root/reference; writes laplacian_kinds.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

from make_golden import import_reference  # noqa: E402
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402

KINDS = [('L1', 0.5), ('L2', 0.5), ('L3', 0.5), ('L4', 0.5), ('L5', 0.5), ('L6', 0.5), ('L6', 0.3),
         ('L7', 0.5)]


def main():
  _, ref_dh, _ = import_reference()
  b = draw_batch(6, seed=41, n_min=5, n_max=14)
  adjs = b['adjs'].astype(np.float32).copy()          # [B, N, N, E]
  n_nodes = b['n_nodes'].astype(np.int32).copy()
  # isolate atom 2 of molecule 3 (all its bonds removed)
  adjs[3, 2, :, :] = 0.0
  adjs[3, :, 2, :] = 0.0
  B, N, _, E = adjs.shape
  rs = np.random.RandomState(5)
  W = (rs.rand(9, 9) * (rs.rand(9, 9) < 0.6)).astype(np.float32)   # weighted, non-symmetric
  out = dict(adjs=adjs, n_nodes=n_nodes, weighted=W, kinds=np.array([k for k, _ in KINDS]),
             alphas=np.array([a for _, a in KINDS]))
  for idx, (kind, alpha) in enumerate(KINDS):
    L = np.zeros((B, N, N, E + 1), np.float64)
    for m in range(B):
      n = int(n_nodes[m])
      a = adjs[m, :n, :n, :].astype(np.float64)
      L[m, :n, :n, 0] = ref_dh.get_laplacian(a.sum(axis=2), kind, alpha)
      for e in range(E):
        L[m, :n, :n, 1 + e] = ref_dh.get_laplacian(a[:, :, e], kind, alpha)
    out['L_%d' % idx] = L
    out['Lw_%d' % idx] = ref_dh.get_laplacian(W.astype(np.float64), kind, alpha)
  path = os.path.join(HERE, 'laplacian_kinds.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
