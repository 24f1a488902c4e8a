"""One-off: the tile-plan fuzz test of tests/test_gpu_parity.py over many more seeds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
  try:
    T.test_tile_plan_fuzz_against_unplanned_schedule_and_oracle(seed)
  except AssertionError as e:
    bad.append((seed, str(e)[:300]))
print('seeds %d..%d: %d failures' % (lo, hi - 1, len(bad)))
for b in bad:
  print(b)
