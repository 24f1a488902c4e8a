"""The packing rule of the strip plan (lnz_plan_strips), on the CPU: the Python restatement the GPU
tests hold the device planner against (tests/strip_mirror.py) keeps the plan's invariants on random
size mixes, and the bench batch fits five 16-row subtiles per compute unit."""
import numpy as np
import pytest

from strip_mirror import SUB, check_plan, plan_strips_mirror


@pytest.mark.parametrize('B,nmin,nmax,n_cu', [(1024, 8, 26, 256), (700, 1, 32, 256), (2048, 3, 26, 256),
                                              (4500, 8, 26, 256), (5, 1, 9, 256), (300, 20, 32, 64),
                                              (777, 1, 6, 3), (1, 32, 32, 1)])
def test_strip_packing_rule_invariants(B, nmin, nmax, n_cu):
  ext = np.random.RandomState(B + nmax).randint(nmin, nmax + 1, size=B)
  plan = plan_strips_mirror(ext, n_cu)
  check_plan(plan, ext)
  assert plan == plan_strips_mirror(ext, n_cu)
  rows4 = np.where(ext <= 4, 4, (ext + 3) // 4 * 4)
  if B <= 2048:
    # one chunk: the strips are as low as a whole number of rounds over the CUs allows
    total16 = (int(rows4.sum()) + 15) // 16
    rounds = (total16 + SUB * n_cu - 1) // (SUB * n_cu)
    assert max(s for _, s in plan) <= SUB
    if len(plan) <= rounds * n_cu:
      assert max(s for _, s in plan) <= max(2, -(-total16 // (rounds * n_cu)) + 1)


def test_bench_batch_needs_five_subtiles_per_compute_unit():
  from lanczosnet_amd.synthetic import draw_batch
  for seed in range(6):
    n = draw_batch(1024, seed=seed)['n_nodes']
    plan = plan_strips_mirror(n, 256)
    check_plan(plan, n)
    assert len(plan) <= 256 and max(s for _, s in plan) == 5
    # 32-row tiles (8 | 24, 16 | 16 pairs or singles) need six subtiles' worth of rows there
    rows4 = (n + 3) // 4 * 4
    assert rows4.sum() < 0.82 * 744 * 32
