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


class deterministic_dropout(object):
  """Replace `torch.nn.functional.dropout` by a seed-free stand-in while a training forward runs, on
  BOTH sides of a parity test: call number i on a tensor x zeroes the elements whose flat index j
  has hash(j, i) % 10 < 10 p and scales the rest by 1 / (1 - p).  A generator-based mask cannot be
  compared (the reference consumes the CPU generator, the HIP module the device's); this one checks
  what matters for a drop-in: WHERE dropout is applied, on WHICH tensor shape, HOW OFTEN and in
  which order (model/lanczos_net.py:182).  `calls` records (shape, p) per call."""

  def __enter__(self):
    import torch
    self.F = torch.nn.functional
    self.real = self.F.dropout
    self.calls = []

    def fake(x, p=0.5, training=True, inplace=False):
      if not training or p == 0.0:
        return x
      i = len(self.calls)
      self.calls.append((tuple(x.shape), float(p)))
      j = torch.arange(x.numel(), device=x.device, dtype=torch.int64)
      h = ((j * 2654435761 + 40503 * (i + 1)) >> 7) % 10
      keep = (h >= int(round(10 * p))).to(x.dtype).view(x.shape)
      return x * keep / (1.0 - p)
    self.F.dropout = fake
    return self

  def __exit__(self, *exc):
    self.F.dropout = self.real
