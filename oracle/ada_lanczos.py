"""Oracle (test infrastructure): AdaLanczosNet's in-model Lanczos layer and learned Laplacian.

Restates reference `model/ada_lanczos_net.py:139-247` (`_lanczos_layer`, with the
three quirks of SURVEY.md F6 / §A.3 reproduced on purpose) and `:101-137`
(`_get_graph_laplacian`).  The start vector q1 is an argument (the reference draws
`torch.randn(B,N,1)` on the CPU generator at :161).
"""
import numpy as np

EPS = float(np.finfo(np.float32).eps)  # model/ada_lanczos_net.py:8


def ada_lanczos_layer(A, mask, q1, num_eig_vec, use_reorth=True, dtype=np.float32):
  """A: [B,N,N] sym, mask: [B,N] (0/1) or None, q1: [B,N] raw start vector (before masking).

  Returns T [B,K,K], Q [B,N,K]."""
  A = np.asarray(A, dtype=dtype)
  B, N = A.shape[0], A.shape[1]
  K = num_eig_vec
  T_it = min(N, K)
  eps = dtype(EPS)

  Q = [None] * (T_it + 2)
  alpha = [None] * (T_it + 1)
  beta = [None] * (T_it + 1)
  beta[0] = np.zeros((B, 1, 1), dtype=dtype)
  Q[0] = np.zeros((B, N, 1), dtype=dtype)
  Q[1] = np.asarray(q1, dtype=dtype).reshape(B, N, 1).copy()
  m = None
  if mask is not None:
    m = np.asarray(mask).astype(dtype)[:, :, None]
    Q[1] = Q[1] * m  # :163-165
  Q[1] = Q[1] / np.sqrt((Q[1] * Q[1]).sum(axis=1, keepdims=True))  # :167

  lb = dtype(1.0e-4)
  valid = []
  for ii in range(1, T_it + 1):
    z = A @ Q[ii]  # :173
    alpha[ii] = (Q[ii] * z).sum(axis=1, keepdims=True)  # :174
    z = z - alpha[ii] * Q[ii] - beta[ii - 1] * Q[ii - 1]  # :175
    if use_reorth and ii > 1:  # :177-189 (two passes of sequential Gram-Schmidt)
      for _ in range(2):
        for jj in range(1, ii):
          z = z - (z * Q[jj]).sum(axis=1, keepdims=True) / (
              (Q[jj] * Q[jj]).sum(axis=1, keepdims=True) + eps) * Q[jj]
    beta[ii] = np.sqrt((z * z).sum(axis=1, keepdims=True))  # :191
    ok = (beta[ii] >= lb).astype(dtype)  # :195
    valid.append(ok if ii == 1 else valid[-1] * ok)  # :196-199
    Q[ii + 1] = (z * valid[-1]) / (beta[ii] + eps)  # :202

  alpha = np.concatenate(alpha[1:], axis=1)[:, :, 0]  # B x T
  beta = np.concatenate(beta[1:-1], axis=1)[:, :, 0] if T_it > 1 else np.zeros((B, 0), dtype)
  valid = np.concatenate(valid, axis=1)[:, :, 0]  # B x T
  idx_mask = valid.sum(axis=1).astype(np.int64)  # :209
  if m is not None:
    idx_mask = np.minimum(idx_mask, m[:, :, 0].sum(axis=1).astype(np.int64))  # :210-211
  for b in range(B):
    if idx_mask[b] < valid.shape[1]:
      valid[b, idx_mask[b]:] = 0.0  # :213-215
  alpha = alpha * valid  # :218  (QUIRK 1)
  beta = beta * valid[:, :-1]  # :219

  T = np.zeros((B, T_it, T_it), dtype=dtype)
  for b in range(B):  # :222-226
    T[b] = np.diag(alpha[b]) + np.diag(beta[b], 1) + np.diag(beta[b], -1)
  Qm = np.concatenate(Q[1:-1], axis=2)  # B x N x T   :229
  Q_mask = np.repeat(valid[:, None, :], N, axis=1)  # :230  (QUIRK 2)
  for b in range(B):
    if idx_mask[b] < Q_mask.shape[1]:
      Q_mask[b, idx_mask[b]:, :] = 0.0  # :233-235 (QUIRK 3: zeroes NODE rows)
  Qm = Qm * Q_mask

  if T_it < K:  # :240-245
    Tp = np.zeros((B, K, K), dtype=dtype)
    Tp[:, :T_it, :T_it] = T
    Qp = np.zeros((B, N, K), dtype=dtype)
    Qp[:, :, :T_it] = Qm
    T, Qm = Tp, Qp
  return T, Qm


def ada_graph_laplacian(node_feat, adj_mask, dtype=np.float32):
  """model/ada_lanczos_net.py:101-137.  node_feat [B,N,D], adj_mask [B,N,N] -> L [B,N,N]."""
  X = np.asarray(node_feat, dtype=dtype)
  adj = np.asarray(adj_mask, dtype=dtype)
  B, N, _ = X.shape
  # meshgrid(range(N), range(N)) -> idx_row[k] = k % N, idx_col[k] = k // N   (:115-118)
  diff = X[:, None, :, :] - X[:, :, None, :]  # [B, col, row, D]: X[row] - X[col]
  dist2 = (diff * diff).sum(axis=3).reshape(B, N * N)  # :120-121
  sigma2 = dist2.mean(axis=1, keepdims=True)  # :126
  A = np.exp(-dist2 / sigma2).reshape(B, N, N) * adj  # :128-129
  row_sum = A.sum(axis=2, keepdims=True)
  pad = (row_sum == 0.0).astype(dtype)  # :131-132
  Dm = 1.0 / np.power(row_sum + pad, dtype(0.5))  # :133-134
  return Dm * A * Dm.transpose(0, 2, 1)  # :135
