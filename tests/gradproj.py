"""Element-level gradient parity without shipping the gradients: every parameter tensor's
gradient is projected on 16 fixed pseudo-random +-1 vectors (numpy RandomState, a function of the
tensor's index in the sorted parameter names and of the projection number).  A projection is a
signed sum over ALL elements with independent signs: an error confined to a few entries, a
permuted channel block or a transposed tile shows up in it at full size, unlike in `sum` /
`abs-sum` / first-entry statistics.  Used by tests/golden/make_golden_gradproj.py (reference side)
and the `-m gpu` gradient tests (HIP side)."""
import numpy as np

NPROJ = 16


def signs(tensor_index, j, numel):
  rs = np.random.RandomState(100003 * (tensor_index + 1) + j)
  return rs.randint(0, 2, size=numel, dtype=np.int8) * np.int8(2) - np.int8(1)


def project_numpy(grad, tensor_index):
  g = np.asarray(grad, dtype=np.float64).reshape(-1)
  return np.array([float(np.dot(g, signs(tensor_index, j, g.size).astype(np.float64)))
                   for j in range(NPROJ)])


def project_torch(grad, tensor_index):
  """grad: torch tensor on any device; the dot products run on its device in float64."""
  import torch
  g = grad.detach().reshape(-1).double()
  out = []
  for j in range(NPROJ):
    s = torch.from_numpy(signs(tensor_index, j, g.numel())).to(g.device)
    out.append(float((g * s.double()).sum()))
  return np.array(out)
