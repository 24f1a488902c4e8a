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
  """Classical Gram-Schmidt against the stored basis; the second pass only where the first one
  removed more than 99 % of the vector's squared length (csrc/lanczos_ritz.hip cgs2_32)."""
  coef = 0.0
  for p in range(2):
    xx = float(w @ w)
    c = Qt[:j + 1] @ w
    w = w - Qt[:j + 1].T @ c
    coef += c[j]
    if p == 0 and float(c @ c) <= 0.99 * xx:
      break
  return w, coef


def _tql2(dd, ee, Qt, n):
  """Implicit-shift QL (EISPACK tql2 recurrences), rotations applied to the rows of Qt."""
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


def _sturm_below(d, e2, a, b, x):
  """Eigenvalues of rows a..b (no outside coupling) below x — the product-form count of the kernel:
  sign changes of p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}."""
  p1, p2, c = 1.0, 0.0, 0
  for i in range(a, b + 1):
    ep = e2[i - 1] if i > a else 0.0
    p = (d[i] - x) * p1 - ep * p2
    if (p < 0) != (p1 < 0):
      c += 1
    p2, p1 = p1, p
    if abs(p1) < 1e-100:
      p1 *= 1e100
      p2 *= 1e100
  return c


def _section_search(d, e2, s, t, j, gsc, may_bail):
  """The (j)-th eigenvalue of block [s, t]: 5-section on Sturm counts.  Returns (lam, lo, hi) — or
  (None, lo, hi) at pass 12 when may_bail (the caller then compares brackets for the cluster test)."""
  lo, hi = -gsc, gsc
  for it in range(30):
    w = (hi - lo) * 0.2
    xs = [lo + w * k for k in (1, 2, 3, 4)]
    cs = [_sturm_below(d, e2, s, t, x) for x in xs]
    if cs[0] > j:
      hi = xs[0]
    elif cs[1] > j:
      lo, hi = xs[0], xs[1]
    elif cs[2] > j:
      lo, hi = xs[1], xs[2]
    elif cs[3] > j:
      lo, hi = xs[2], xs[3]
    else:
      lo = xs[3]
    if may_bail and it == 12:
      return None, lo, hi
    if hi - lo <= 4.0 * EPS * max(abs(lo), abs(hi)) + 1e-300:
      break
  return 0.5 * (lo + hi), lo, hi


def _twisted_vector(d, e, e2, s, t, lam, ws, wt, tiny):
  """dlar1v-style eigenvector of block [s, t]: stationary / progressive qd transforms of T - lam,
  twist at min |gamma| inside the window [ws, wt]."""
  n = len(d)
  Dp = np.zeros(n)
  Dm = np.zeros(n)
  fix = lambda v: (-tiny if v < 0 else tiny) if abs(v) < tiny else v  # noqa: E731
  for i in range(s, t + 1):
    v = d[i] - lam
    if i > s:
      v -= e2[i - 1] / Dp[i - 1]
    Dp[i] = fix(v)
  for i in range(t, s - 1, -1):
    v = d[i] - lam
    if i < t:
      v -= e2[i] / Dm[i + 1]
    Dm[i] = fix(v)
  gam = np.abs(Dp + Dm - (d - lam))
  cand = [i for i in range(s, t + 1) if ws <= i <= wt]
  tw = min(cand, key=lambda i: gam[i])
  z = np.zeros(n)
  z[tw] = 1.0
  for i in range(tw - 1, s - 1, -1):
    z[i] = -(e[i] / Dp[i]) * z[i + 1]
  for i in range(tw + 1, t + 1):
    z[i] = -(e[i - 1] / Dm[i]) * z[i - 1]
  return z / np.sqrt((z * z).sum())


def _tridiag_eig_parallel(dd, ee, Qt, n):
  """Mirror of tridiag_eig_parallel (lanczos_ritz.hip): per-block section search, twisted vectors,
  cluster rescue by twist windows + Gram-Schmidt, V = Q S.  Returns False when the QL sweep has
  to run instead (nothing modified then)."""
  d = dd.copy()
  e = np.zeros(n)
  for i in range(n - 1):
    if abs(ee[i]) > EPS * (abs(dd[i]) + abs(dd[i + 1])):
      e[i] = ee[i]
  e2 = e * e
  gsc = max(abs(d[i]) + (abs(e[i - 1]) if i > 0 else 0.0) + abs(e[i]) for i in range(n)) or 1.0
  blk = []
  for k in range(n):
    s = k
    while s > 0 and e[s - 1] != 0.0:
      s -= 1
    t = k
    while t < n - 1 and e[t] != 0.0:
      t += 1
    blk.append((s, t))
  first = [_section_search(d, e2, blk[k][0], blk[k][1], k - blk[k][0], gsc, True) for k in range(n)]
  bail = [k > 0 and blk[k] == blk[k - 1] and first[k][1] == first[k - 1][1] for k in range(n)]
  win = list(blk)
  member = [False] * n
  if any(bail):
    member = [bail[k] or (k + 1 < n and bail[k + 1]) for k in range(n)]
    cut, wd = 1e-3 * gsc, 1e-5 * gsc
    for k in range(n):
      if not member[k]:
        continue
      s, t = blk[k]
      xl, xh = first[k][1] - wd, first[k][2] + wd
      g = (k - s) - _sturm_below(d, e2, s, t, xl)
      a, found = s, False
      for i in range(s, t + 1):
        if i == t or abs(e[i]) <= cut:
          inside = _sturm_below(d, e2, a, i, xh) - _sturm_below(d, e2, a, i, xl)
          if g < inside:
            if inside != 1:
              return False
            win[k] = (a, i)
            found = True
            break
          g -= inside
          a = i + 1
      if not found:
        return False
  lam = [_section_search(d, e2, blk[k][0], blk[k][1], k - blk[k][0], gsc, False)[0] for k in range(n)]
  tiny = EPS * gsc
  S = np.zeros((n, n))
  for k in range(n):
    S[:, k] = _twisted_vector(d, e, e2, blk[k][0], blk[k][1], lam[k], win[k][0], win[k][1], tiny)
  for k in range(n):  # Gram-Schmidt over cluster lanes, in lane order
    pos = 0
    for dlt in (1, 2, 3):
      if (k - dlt >= 0 and member[k] and member[k - dlt] and blk[k - dlt] == blk[k]
          and abs(lam[k] - lam[k - dlt]) <= 1e-6 * gsc and pos == dlt - 1):
        pos = dlt
    for dlt in range(1, pos + 1):
      S[:, k] -= (S[:, k] @ S[:, k - dlt]) * S[:, k - dlt]
    if pos:
      S[:, k] /= np.sqrt((S[:, k] ** 2).sum())
  Qt[:n] = S.T @ Qt[:n]
  dd[:n] = lam
  return True


def lanczos_ritz_mirror(A, K, solver='parallel'):
  """A: [n, n] float32/64 symmetric.  Returns D [K], V [n, K] (float32), restarts.
  solver: 'parallel' = the N <= 32 kernel's lane-parallel tridiagonal eigensolver (QL sweep as its
  last resort), 'ql' = the implicit-QL sweep of the generic kernel."""
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
  solved = solver == 'parallel' and n <= 32 and _tridiag_eig_parallel(dd, ee, Qt, n)
  if not solved:
    if solver == 'parallel' and n <= 32:
      restarts += 256  # like the kernel's info: the QL sweep ran as the last resort
    _tql2(dd, ee, Qt, n)
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
