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
  return _sturm_poly(d, e2, a, b, x)[0]


def _sturm_poly(d, e2, a, b, x):
  """-> (count, mantissa, exponent): the count above and the value p_len(x) of the block's
  characteristic polynomial from the same recurrence, renormalised every eight rows with the
  binary exponent tracked (csrc/lanczos_ritz.hip sturm_rows)."""
  import math
  p1, p2, c, ex = 1.0, 0.0, 0, 0
  for i in range(a, b + 1):
    ep = e2[i - 1] if i > a else 0.0
    p = (d[i] - x) * p1 - ep * p2
    if (math.copysign(1.0, p) < 0) != (math.copysign(1.0, p1) < 0):
      c += 1
    p2, p1 = p1, p
    if ((i - a) & 7) == 7 or i == b:
      if p1 != 0.0:
        m, k = math.frexp(p1)
        p1, p2, ex = m, math.ldexp(p2, -k), ex + k
  return c, p1, ex


def _search_block(d, e2, s, t, gsc, state=None):
  """All eigenvalues of block [s, t], the lanes in lock step (eigenvalue_search of the kernel):
  pass 0 probes the bracket's ends and thirds, a section pass lo + k w / 5, and once exactly one
  eigenvalue is inside (p changes sign) the secant point x* of the end values with probes at
  x* -+ d1, x* -+ 16 d1, d1 ~ w^2 / (2 gap to the neighbouring lanes' brackets); brackets are
  updated from the COUNTS only.  state=None: passes 0..12, returning (None, state, bail flags)
  when two lanes still share a bracket after pass 12 (a cluster); otherwise passes 13.. ."""
  import math
  L = t - s + 1
  if state is None:
    state = [dict(lo=-gsc, hi=gsc, clo=0, chi=L, flo=None, fhi=None, done=False, sect=True)
             for _ in range(L)]
    if L == 1:
      state[0].update(lo=d[s], hi=d[s], done=True)
    it0, may_bail = 0, True
  else:
    it0, may_bail = 13, False
  tol_of = lambda lo, hi: 4.0 * EPS * max(abs(lo), abs(hi), 0.125 * gsc)  # noqa: E731
  for it in range(it0, 30):
    if all(st['done'] for st in state):
      break
    mids = [0.5 * (st['lo'] + st['hi']) for st in state]
    for j, st in enumerate(state):
      if st['done']:
        continue
      lo, hi = st['lo'], st['hi']
      w = hi - lo
      iso = (it > 0 and not st['sect'] and st['chi'] - st['clo'] == 1 and st['flo'][0] != 0.0
             and (math.copysign(1.0, st['flo'][0]) < 0) != (math.copysign(1.0, st['fhi'][0]) < 0))
      if it == 0:
        xs = [lo, lo + w * (1.0 / 3.0), lo + 2.0 * (w * (1.0 / 3.0)), hi]
      elif iso:
        (ml, el), (mh, eh) = st['flo'], st['fhi']
        fh = math.ldexp(mh, max(-1000, min(1000, eh - el)))
        xstar = lo + w * (ml / (ml - fh))
        g = gsc
        if j > 0:
          g = min(g, abs(xstar - mids[j - 1]))
        if j < L - 1:
          g = min(g, abs(mids[j + 1] - xstar))
        d1 = 0.5 * w * w / max(g, 1e-300)
        d1 = max(min(d1, 0.125 * w), 0.45 * tol_of(lo, hi))
        d2 = min(0.25 * w, 16.0 * d1)
        xs = [min(max(x, lo), hi) for x in (xstar - d2, xstar - d1, xstar + d1, xstar + d2)]
      else:
        xs = [lo + w * 0.2 * k for k in (1, 2, 3, 4)]
      ev = [_sturm_poly(d, e2, s, t, x) for x in xs]
      idx = next((k for k in range(4) if ev[k][0] > j), 4)
      if idx > 0:
        st['lo'], st['clo'], st['flo'] = xs[idx - 1], ev[idx - 1][0], ev[idx - 1][1:]
      if idx < 4:
        st['hi'], st['chi'], st['fhi'] = xs[idx], ev[idx][0], ev[idx][1:]
      nw = st['hi'] - st['lo']
      st['sect'] = nw > 0.25 * w
      st['done'] = nw <= tol_of(st['lo'], st['hi'])
    if may_bail and it == 12:
      bail = [j > 0 and not state[j]['done'] and state[j]['lo'] == state[j - 1]['lo'] for j in range(L)]
      if any(bail):
        return None, state, bail
  return [0.5 * (st['lo'] + st['hi']) for st in state], state, [False] * L


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
  lam = [0.0] * n
  states, bail = {}, [False] * n
  for (s, t) in sorted(set(blk)):
    res, st, bl = _search_block(d, e2, s, t, gsc)
    states[(s, t)] = (res, st)
    for k in range(s, t + 1):
      bail[k] = bool(bl[k - s])
      if res is not None:
        lam[k] = res[k - s]
  win = list(blk)
  member = [False] * n
  if any(bail):
    member = [bail[k] or (k + 1 < n and bail[k + 1]) for k in range(n)]
    # an eigenvalue of the block within 1e-7 |T| of a member belongs to the cluster as well
    # (its rank inside the widened bracket counts when the windows are handed out)
    mid = [0.0] * n
    for (s, t), (res, st) in states.items():
      for k in range(s, t + 1):
        mid[k] = 0.5 * (st[k - s]['lo'] + st[k - s]['hi'])
    for _ in range(3):
      new = list(member)
      for k in range(n):
        s, t = blk[k]
        if (k > s and member[k - 1] and abs(mid[k] - mid[k - 1]) <= 1e-7 * gsc) or \
           (k < t and member[k + 1] and abs(mid[k + 1] - mid[k]) <= 1e-7 * gsc):
          new[k] = True
      member = new
    cut, wd = 1e-3 * gsc, 1e-5 * gsc
    for k in range(n):
      if not member[k]:
        continue
      s, t = blk[k]
      stk = states[(s, t)][1][k - s]
      xl, xh = stk['lo'] - wd, stk['hi'] + wd
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
    for (s, t), (res, st) in states.items():
      if res is None:
        res, _, _ = _search_block(d, e2, s, t, gsc, state=st)
        for k in range(s, t + 1):
          lam[k] = res[k - s]
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
