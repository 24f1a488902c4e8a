"""ORACLE (test infrastructure only): CPU restatement of the packed-shard collate.

Follows the reference's preprocessing + default collate branch on a molecule given as atom ids and
a bond list: `to_graph` adjacency (dataset/get_qm8_data.py:26-42), `adj_simple = sum_e adjs`
(:62), L4 per bond type and of the simple graph (:63-75 via utils/data_helper.py:92-116,155-156,
261-291), then the zero padding / channel concat / fp32 cast of dataset/qm8.py:57-100,220-262 and
the (D, V) pad/cut of qm8.py:264-291.  Pinned by tests/golden/collate_batch.npz (generated from
the reference's own get_graph_laplacian_eigs + QM8Data.collate_fn).
"""
import numpy as np

from .eigs import collate_eigs, graph_laplacian_eigs
from .laplacian import laplacian_multi_l4


def dense_from_edges(n, edges, num_bond_type):
  """Packed bonds (u | v<<8 | type<<16, each undirected bond once) -> adjs [n,n,E] float32."""
  adjs = np.zeros((n, n, num_bond_type), dtype=np.float32)
  for w in np.asarray(edges, dtype=np.uint32).tolist():
    u, v, t = w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff
    if u < n and v < n and t < num_bond_type:
      adjs[u, v, t] += 1.0
      if u != v:
        adjs[v, u, t] += 1.0
  return adjs


def collate_packed(molecules, num_bond_type, num_eigs):
  """molecules: list of (atoms [n], edges uint32, label [P]).  Returns the reference's batch dict
  (numpy): node_feat [B,N] int64, node_mask [B,N] uint8, label [B,P] f32, L [B,N,N,E+1] f32,
  D [B,K] f32, V [B,N,K] f32, n_nodes [B] int32."""
  B = len(molecules)
  sizes = [len(m[0]) for m in molecules]
  N = max(sizes)  # dataset/qm8.py:66
  P = len(molecules[0][2])
  node_feat = np.zeros((B, N), dtype=np.int64)
  mask = np.zeros((B, N), dtype=np.uint8)
  label = np.zeros((B, P), dtype=np.float32)
  L = np.zeros((B, N, N, num_bond_type + 1), dtype=np.float32)
  Dl, Vl = [], []
  for b, (atoms, edges, lab) in enumerate(molecules):
    n = sizes[b]
    node_feat[b, :n] = np.asarray(atoms)
    mask[b, :n] = 1
    label[b] = np.asarray(lab, dtype=np.float32)
    adjs = dense_from_edges(n, edges, num_bond_type)
    L[b, :n, :n, :] = laplacian_multi_l4(adjs)
    e, V, _ = graph_laplacian_eigs(adjs.sum(axis=2), graph_laplacian_type='L4')
    Dl.append(e)
    Vl.append(V)
  D, V = collate_eigs(Dl, Vl, N, num_eigs)
  return dict(node_feat=node_feat, node_mask=mask, label=label, L=L, D=D, V=V,
              n_nodes=np.asarray(sizes, dtype=np.int32))
