"""Seeded inputs of the large-graph (BASELINE config 5) tests, shared by tests/test_gpu_large.py
and tests/golden/make_golden_config5.py so that both sides build the SAME graphs, features and
parameters from numpy RandomState alone (no large arrays in the fixture)."""
import numpy as np

import oracle


def adjacency(B, N, p, seed):
  """The B symmetric 0/1 adjacency matrices of G(N, p) graphs behind `graphs`, float64 [B,N,N]."""
  rs = np.random.RandomState(seed)
  out = np.zeros((B, N, N), np.float64)
  for b in range(B):
    a = (rs.rand(N, N) < p).astype(np.float64)
    a = np.triu(a, 1)
    out[b] = a + a.T
  return out


def graphs(B, N, p, seed):
  """B dense symmetric L4 Laplacians of G(N, p) graphs, float32 [B,N,N]."""
  adj = adjacency(B, N, p, seed)
  A = np.zeros((B, N, N), np.float32)
  for b in range(B):
    A[b] = oracle.laplacian_l4(adj[b])
  return A


def general_cfg(K, num_layer):
  return dict(num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
              num_eig_vec=K, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * num_layer,
              output_dim=2, num_layer=num_layer, num_atom=0)


def general_inputs(B, N, K, num_layer, seed, p_edge):
  """cfg, parameters (numpy seed 17), node features X [B,N,10], L [B,N,N,2] (simple graph + its one
  edge type, dataset/graph_data.py:222-260), node mask (graph 1 loses its last N/5 nodes)."""
  cfg = general_cfg(K, num_layer)
  A = graphs(B, N, p_edge, seed=seed)
  rs = np.random.RandomState(2)
  X = rs.randn(B, N, 10).astype(np.float32)
  mask = np.ones((B, N), np.uint8)
  if B > 1:
    mask[1, N - N // 5:] = 0
  L = np.stack([A, A], axis=3)
  P = oracle.make_lanczosnet_params(cfg, 17, general=True)
  return cfg, P, X, L, mask


def kstep_ritz(L, K):
  """(D, V) of channel 0 by the fp64 restatement of the K-step Lanczos (oracle/lanczos_kstep.py),
  cast to fp32 like the collate output."""
  B, N = L.shape[0], L.shape[1]
  D = np.zeros((B, K), np.float32)
  V = np.zeros((B, N, K), np.float32)
  for b in range(B):
    d, v, _ = oracle.lanczos_kstep_fp64(L[b, :, :, 0], K, K)
    D[b], V[b] = d, v
  return D, V
