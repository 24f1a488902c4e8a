"""Synthetic QM8-schema molecule batches (host-side input generator, numpy only).

There is no QM8 data in the build or GPU containers, so the bench and
the parity tests draw molecule-like graphs with the schema `dataset/get_qm8_data.py:26-42,
56-96` produces: integer atom ids, 6 bond-type adjacency channels that partition the
edges, 16 regression targets.  Recipe = SURVEY.md §8(d) "Config 2".

This module only draws *inputs* (adjacency, atoms, labels, mask).  Laplacians, Ritz pairs
and the network itself are computed by the HIP path (`lanczosnet_amd.ops`) — or, inside
tests, by the oracle.
"""
import numpy as np

BOND_P = (0.7, 0.15, 0.05, 0.05, 0.03, 0.02)


def draw_molecule(rs, n, num_bond_type=6, max_extra=2):
  """Random spanning tree + U{0..max_extra} ring-closing edges; each edge gets one bond type.

  Returns adjs [n, n, num_bond_type] float32 (symmetric 0/1, channels partition the edges)."""
  adjs = np.zeros((n, n, num_bond_type), dtype=np.float32)
  p = np.asarray(BOND_P[:num_bond_type], dtype=np.float64)
  p = p / p.sum()
  edges = set()
  for v in range(1, n):
    u = int(rs.randint(0, v))
    edges.add((u, v))
  n_extra = int(rs.randint(0, max_extra + 1))
  for _ in range(n_extra):
    if n < 3:
      break
    u, v = sorted(int(x) for x in rs.choice(n, size=2, replace=False))
    edges.add((u, v))
  for (u, v) in sorted(edges):
    t = int(rs.choice(num_bond_type, p=p))
    adjs[u, v, t] = 1.0
    adjs[v, u, t] = 1.0
  return adjs


def draw_batch(batch_size, seed=0, n_min=8, n_max=26, N=None, num_atom=70, num_bond_type=6,
               num_label=16):
  """Padded batch in the collate layout of `dataset/qm8.py:57-100` (minus L/D/V).

  Returns dict of numpy arrays:
    adjs      [B, N, N, E]  float32   bond-type adjacency, zero padded
    node_feat [B, N]        int64     atom ids, padded with 0 (qm8.py:71-77)
    node_mask [B, N]        uint8     1 for real nodes (qm8.py:80-86)
    n_nodes   [B]           int32
    label     [B, P]        float32
  N defaults to the batch max (qm8.py:66)."""
  rs = np.random.RandomState(seed)
  sizes = rs.randint(n_min, n_max + 1, size=batch_size)
  if N is None:
    N = int(sizes.max())
  assert N >= int(sizes.max())
  adjs = np.zeros((batch_size, N, N, num_bond_type), dtype=np.float32)
  node_feat = np.zeros((batch_size, N), dtype=np.int64)
  mask = np.zeros((batch_size, N), dtype=np.uint8)
  for b, n in enumerate(sizes):
    n = int(n)
    adjs[b, :n, :n, :] = draw_molecule(rs, n, num_bond_type)
    node_feat[b, :n] = rs.randint(0, num_atom, size=n)
    mask[b, :n] = 1
  label = rs.randn(batch_size, num_label).astype(np.float32)
  return dict(adjs=adjs, node_feat=node_feat, node_mask=mask,
              n_nodes=sizes.astype(np.int32), label=label)
