"""Oracle (test infrastructure): the reference's (D, V) producer.

Restates `utils/data_helper.py:169-258` (get_graph_laplacian_eigs, the
`use_eigen_decomp=True, is_sym=True` branch both dataset generators take:
dataset/get_qm8_data.py:63-69) and the pad/cut of the collate function
`dataset/qm8.py:264-291`.

Third-party arithmetic: `numpy.linalg.eigh` (LAPACK syevd) exactly as the
reference calls it (utils/data_helper.py:201); the reference pins no numpy
version (requirements.txt:2-3); this container has numpy 2.2.6.
"""
import numpy as np

from .laplacian import get_laplacian


def graph_laplacian_eigs(adj, k=100, graph_laplacian_type='L4'):
  """Returns (eigs[k'], V[n,k'], L[n,n]) sorted by descending |lambda| with a
  stable mergesort on the ascending-lambda eigh output (utils/data_helper.py:197-223)."""
  L = get_laplacian(np.asarray(adj), graph_laplacian_type)
  assert np.allclose(L, L.T, atol=1e-8)  # utils/data_helper.py:195
  eigs, V = np.linalg.eigh(L)
  idx = np.argsort(-np.abs(eigs), kind='mergesort')
  return eigs[idx[:k]], V[:, idx[:k]], L


# A top-K cut (n > K) through a cluster of equal |lambda| keeps an arbitrary vector of the cluster:
# basis dependent in the reference itself (LAPACK's choice), so such molecules are excluded from
# every (D, V) / score comparison and counted.  ONE rule for bench.py and the tests:
# the gap below which two eigenvalues of the fp32 Laplacian the device path is handed
# (dataset/qm8.py:262 casts L to fp32) cannot be ordered — 1e-7, its rounding.  (On the QM8-sized
# synthetic batches the gaps are bimodal: exact symmetries at 1e-16, everything else >= 2e-4.)
CUT_GAP = 1.0e-7


def degenerate_cut(eigs_sorted, K, gap=CUT_GAP):
  """eigs_sorted: one molecule's eigenvalues in the reference order (descending |lambda|).  True
  when the cut behind slot K - 1 separates two eigenvalues whose moduli differ by less than gap."""
  e = np.abs(np.asarray(eigs_sorted, dtype=np.float64))
  return bool(e.shape[0] > K and abs(e[K - 1] - e[K]) < gap)


def collate_eigs(D_list, V_list, N, K):
  """dataset/qm8.py:264-291: pad V rows to N, cut / zero-pad eigen slots to K, cast fp32.

  D_list[b]: [n_b'], V_list[b]: [n_b, n_b'] -> D [B,K] f32, V [B,N,K] f32."""
  B = len(D_list)
  D = np.zeros((B, K), dtype=np.float32)
  V = np.zeros((B, N, K), dtype=np.float32)
  for b in range(B):
    d, v = np.asarray(D_list[b]), np.asarray(V_list[b])
    kk = min(K, d.shape[0])
    D[b, :kk] = d[:kk]
    V[b, :v.shape[0], :kk] = v[:, :kk]
  return D, V


def spectral_projector(D, V, power):
  """V diag(D^power) V^T — the basis-invariant quantity the network consumes
  (model/lanczos_net.py:114-121).  float64."""
  D = np.asarray(D, dtype=np.float64)
  V = np.asarray(V, dtype=np.float64)
  return np.einsum('...ik,...k,...jk->...ij', V, D ** power, V)
