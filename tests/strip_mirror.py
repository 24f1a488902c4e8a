"""Python restatement of the strip planner (csrc/prep.hpp plan_strips_body: lnz_plan_strips) and the
invariants of a strip plan — shared by the CPU test of the packing rule and the GPU tests that hold
the device plan against it."""
import numpy as np

INTS, SUB = 80, 6


def _place(fill, rows):
  return fill if (fill % 16) + rows <= 32 else (fill + 15) // 16 * 16


def plan_strips_mirror(ext, n_cu, chunk=2048):
  """Batches beyond 2048 molecules: chunk by chunk, consecutive strip ranges."""
  out = []
  for c0 in range(0, len(ext), chunk):
    for mols, sub in _plan_chunk_mirror(ext[c0:c0 + chunk], n_cu):
      out.append(([(b + c0, st, n) for b, st, n in mols], sub))
  return out


def _plan_chunk_mirror(ext, n_cu, bins=1024):
  """First fit decreasing by size class (rows / 4), a class at a time, stable in batch order; strip
  height = the smallest number of subtiles (2..6) for which the strips in use fit the rounds of
  n_cu strips the batch needs at full height."""
  rows4 = np.where(ext <= 4, 4, (ext + 3) // 4 * 4)
  total16 = (int(rows4.sum()) + 15) // 16
  rounds = (total16 + SUB * n_cu - 1) // (SUB * n_cu)
  target = min(rounds * n_cu, bins)
  cap = min(max((total16 + target - 1) // target, 2), SUB)
  while True:
    fill = np.zeros(bins, int)
    mols = [[] for _ in range(bins)]
    for rows in range(32, 0, -4):
      items = [b for b in range(len(ext)) if rows4[b] == rows]
      k = 0
      for bi in range(bins):
        while k < len(items):
          off = _place(fill[bi], rows)
          if off + rows > 16 * cap:
            break
          mols[bi].append((items[k], off, int(ext[items[k]])))
          fill[bi] = off + rows
          k += 1
        if k == len(items):
          break
      assert k == len(items)
    used = max(i + 1 for i in range(bins) if mols[i])
    if used <= target or cap >= SUB:
      return [(mols[i], (fill[i] + 15) // 16) for i in range(used)]
    cap += 1



def check_plan(plan, ext):
  """plan: list of (molecules [(b, first row, extent)], subtiles).  Every molecule exactly once,
  4-aligned rows that do not overlap, at most two subtiles per molecule, rows ascending."""
  seen = np.zeros(len(ext), int)
  for mols, sub in plan:
    assert 1 <= len(mols) <= 24 and 1 <= sub <= SUB
    taken = np.zeros(16 * sub, int)
    for b, st, n in mols:
      rows = 4 if n <= 4 else (n + 3) // 4 * 4
      assert n == ext[b] and st % 4 == 0 and st + rows <= 16 * sub
      assert st // 16 + 1 >= (st + rows - 1) // 16, 'a molecule spans at most two subtiles'
      taken[st:st + rows] += 1
      seen[b] += 1
    assert taken.max() == 1
    assert [m[1] for m in mols] == sorted(m[1] for m in mols), 'rows ascending'
  assert (seen == 1).all()
