"""Pure-Python mirrors of the HIP kernels' algorithms (design checks, CPU only, small sizes).

`lanczos_ritz_mirror` follows lanczosnet_amd/csrc/lanczos_ritz.hip statement by statement
(full-length Lanczos + CGS2 + restart, tql2 recurrences on the transposed basis, ordering,
sign convention).  It lets the numerics of the kernel's *algorithm* be checked against the
reference's eigh-based (D, V) on the CPU; the GPU tests then check the kernel itself.
"""
import numpy as np

TOL = 1e-8
EPS = 2.220446049250313e-16


def _cgs2(Qt, w, j):
  coef = 0.0
  for _ in range(2):
    c = Qt[:j + 1] @ w
    w = w - Qt[:j + 1].T @ c
    coef += c[j]
  return w, coef


def lanczos_ritz_mirror(A, K):
  """A: [n, n] float32/64 symmetric.  Returns D [K], V [n, K] (float32), restarts."""
  A = np.asarray(A, dtype=np.float32).astype(np.float64)
  n = A.shape[0]
  kk = min(K, n)
  Qt = np.zeros((n, n))
  dd = np.zeros(n)
  ee = np.zeros(n)
  lanes = np.arange(n)
  h = ((lanes + 1).astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xffffffff)
  q = 1.0 + ((h >> np.uint64(8)) & np.uint64(0xffff)).astype(np.float64) / 65536.0
  q = q / np.sqrt((q * q).sum())
  restarts = 0
  for j in range(n):
    Qt[j] = q
    w = A @ q
    w, alpha = _cgs2(Qt, w, j)
    dd[j] = alpha
    if j == n - 1:
      break
    beta = np.sqrt((w * w).sum())
    if beta > TOL:
      ee[j] = beta
      q = w / beta
    else:
      restarts += 1
      ee[j] = 0.0
      res = 1.0 - (Qt[:j + 1] ** 2).sum(axis=0)
      cand = int(np.argmax(res))
      w = np.zeros(n)
      w[cand] = 1.0
      w, _ = _cgs2(Qt, w, j)
      q = w / np.sqrt((w * w).sum())
  # tql2
  f = 0.0
  tst1 = 0.0
  for l in range(n):
    tst1 = max(tst1, abs(dd[l]) + abs(ee[l]))
    m = l
    while m < n - 1 and abs(ee[m]) > EPS * tst1:
      m += 1
    if m > l:
      it = 0
      while True:
        it += 1
        g = dd[l]
        el = ee[l]
        p = (dd[l + 1] - g) / (2.0 * el)
        r = np.sqrt(p * p + 1.0)
        if p < 0:
          r = -r
        dl = el / (p + r)
        dl1 = el * (p + r)
        hh = g - dl
        dd[l] = dl
        dd[l + 1] = dl1
        dd[l + 2:n] -= hh
        f += hh
        p = dd[m]
        c = c2 = c3 = 1.0
        s = s2 = 0.0
        el1 = ee[l + 1]
        for i in range(m - 1, l - 1, -1):
          c3, c2, s2 = c2, c, s
          ei, di = ee[i], dd[i]
          g = c * ei
          hp = c * p
          r = np.sqrt(p * p + ei * ei)
          rinv = 1.0 / r
          ee[i + 1] = s * r
          s = ei * rinv
          c = p * rinv
          p = c * di - s * g
          dd[i + 1] = hp + s * (c * g + s * di)
          z1 = Qt[i + 1].copy()
          z0 = Qt[i].copy()
          Qt[i + 1] = s * z0 + c * z1
          Qt[i] = c * z0 - s * z1
        p = -s * s2 * c3 * el1 * ee[l] / dl1
        el = s * p
        ee[l] = el
        dd[l] = c * p
        if not (abs(el) > EPS * tst1 and it < 60):
          break
    dd[l] = dd[l] + f
    ee[l] = 0.0
  # ordering
  perm = np.zeros(n, dtype=np.int64)
  for i in range(n):
    di, ai = dd[i], abs(dd[i])
    rank = 0
    for jj in range(n):
      dj, aj = dd[jj], abs(dd[jj])
      if aj > ai or (aj == ai and (dj < di or (dj == di and jj < i))):
        rank += 1
    perm[rank] = i
  D = np.zeros(K, np.float32)
  V = np.zeros((n, K), np.float32)
  for k in range(kk):
    v = Qt[perm[k]]
    sg = 1.0 if v[np.argmax(np.abs(v))] >= 0 else -1.0
    D[k] = dd[perm[k]]
    V[:, k] = (sg * v).astype(np.float32)
  return D, V, restarts


# ---------------------------------------------------------------------------------------
# v_mfma_f32_32x32x2_f32 model (cdna_hip_programming.md §3): lane l supplies A[i=l&31][k=l>>5]
# and B[k=l>>5][j=l&31]; C/D register r of lane l is element [row=(r&3)+8(r>>2)+4(l>>5)][col=l&31]
# ---------------------------------------------------------------------------------------
def cd_row(r, hh):
  return (r & 3) + 8 * (r >> 2) + 4 * hh


def mfma32(a, b, c):
  """a, b: [64] lane values; c: [64, 16] accumulators -> new [64, 16]."""
  A = np.stack([a[:32], a[32:]], axis=1)  # [i, k]
  Bm = np.stack([b[:32], b[32:]], axis=0)  # [k, j]
  Dm = (A.astype(np.float64) @ Bm.astype(np.float64))  # [i, j]
  out = c.copy()
  for lane in range(64):
    for r in range(16):
      out[lane, r] += Dm[cd_row(r, lane >> 5), lane & 31]
  return out


def cd_to_matrix(acc):
  M = np.zeros((32, 32), acc.dtype)
  for lane in range(64):
    for r in range(16):
      M[cd_row(r, lane >> 5), lane & 31] = acc[lane, r]
  return M
