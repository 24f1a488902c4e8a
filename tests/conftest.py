import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
  """`pytest tests` on a machine without an accelerator skips the gpu-marked tests instead of
  failing them.  On a GPU box nothing is skipped: a missing HIP library must fail loudly there."""
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no HIP GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


def load_golden(name):
  return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
  return load_golden


def rel_err(a, b):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_err_rows(a, b):
  """max over rows (molecules) of max|a_row - b_row| / max|b_row|: a molecule whose outputs are
  100x smaller than the batch maximum is held to the same relative bar as the largest one
  (rel_err above normalises by the batch maximum)."""
  a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
  b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
  return float((np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-30)).max())
