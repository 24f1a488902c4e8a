"""Oracle (test infrastructure): K-step Lanczos Ritz pairs for LARGE graphs (BASELINE config 5).

The reference's Lanczos branch for large graphs is `scipy.sparse.linalg.eigsh(L, k, which='LM')`
(ARPACK implicitly restarted Lanczos, `utils/data_helper.py:205-208`, third-party, scipy 1.15.3
here).  The HIP kernel `lnz_lanczos_ritz_large` is a plain M-step Lanczos with full
re-orthogonalisation (no implicit restarts) — a different function from ARPACK for unconverged
pairs (SURVEY.md F8, §8d "Config 5").  Its parity is therefore pinned two ways:
  * `lanczos_kstep_fp64` below: fp64 restatement of the SAME M-step algorithm (same start vector,
    classical Gram-Schmidt with a second pass on cancellation, early stop on breakdown) — Ritz values / Ritz-vector subspace compared directly;
  * `eigsh` (the reference's call): the converged leading Ritz pairs must agree with it, and every
    Ritz pair must satisfy the Lanczos residual bound |A v - theta v| = beta_M |e_M^T s|.
"""
import numpy as np


def start_vector(n):
  lanes = np.arange(n, dtype=np.uint64)
  h = ((lanes + np.uint64(1)) * np.uint64(2654435761)) & np.uint64(0xffffffff)
  return 1.0 + ((h >> np.uint64(8)) & np.uint64(0xffff)).astype(np.float64) / 65536.0


def lanczos_kstep_fp64(A, M, K, tol=1e-8, reorth=1e-6):
  """A [n,n] symmetric (float32 values, promoted), M Lanczos steps, top-K Ritz pairs by |theta|.
  Returns D [K], V [n,K], (alpha, beta, steps_done)."""
  A = np.asarray(A, dtype=np.float32).astype(np.float64)
  n = A.shape[0]
  M = min(M, n)
  Q = np.zeros((M, n))
  alpha = np.zeros(M)
  beta = np.zeros(M)
  w = start_vector(n)
  nrm = np.sqrt(w @ w)
  steps = 0
  for j in range(M):
    if j > 0 and nrm <= tol:
      break  # invariant subspace: stop, remaining slots stay zero
    if j > 0:
      beta[j - 1] = nrm
    q = w / nrm
    Q[j] = q
    w = A @ q
    coef = 0.0
    n0 = w @ w
    for p in range(2):
      # second classical Gram-Schmidt pass only when the first cancelled more than 1 - 1e-3 of
      # |w| (the kernel's kReorth rule: the rounding error of one projection relative to what is
      # left is eps |w0| / |w1|)
      if p == 1 and w @ w >= reorth * n0:
        break
      c = Q[:j + 1] @ w
      w = w - Q[:j + 1].T @ c
      coef += c[j]
    alpha[j] = coef
    nrm = np.sqrt(w @ w)
    steps = j + 1
  T = np.diag(alpha[:steps]) + np.diag(beta[:steps - 1], 1) + np.diag(beta[:steps - 1], -1)
  th, S = np.linalg.eigh(T)
  idx = np.argsort(-np.abs(th), kind='mergesort')
  kk = min(K, steps)
  D = np.zeros(K)
  V = np.zeros((n, K))
  D[:kk] = th[idx[:kk]]
  V[:, :kk] = Q[:steps].T @ S[:, idx[:kk]]
  return D, V, (alpha, beta, steps, nrm)
