"""Oracle (test infrastructure): unsorted_segment_sum.

Restatement of the reference op.  Pin: the reference's CPU source
(`operators/src/segment_reduction.cpp`) is compiled UNMODIFIED where it lies by
`oracle/ref_build.py` (an empty `THC/THC.h` shim satisfies its one dead include) into
`oracle/_ref/`, and `tests/test_segment_sum_reference.py` checks this restatement — and the HIP
operator — against that binary.  The CUDA file cannot be built here; its semantics are restated
from the source:

  * GPU semantics  — `operators/src/cuda/segment_reduction.cu:39-53` (forward,
    atomicAdd scatter) and `:55-69` (backward, gather).  NOTE the reference quirk
    (SURVEY.md §2.1): the output batch stride is `dim1*dim2`, not
    `num_segments*dim2` (:48, :64) — only self-consistent when
    num_segments == dim1 (or B == 1).  Restated verbatim over a flat buffer.
  * CPU semantics  — `operators/src/segment_reduction.cpp:6-30`, which indexes
    `segment_ids_ptr[jj]` (:20), i.e. uses batch-0 ids for every batch.

The product (HIP) op follows the *GPU* semantics with the batch stride FIXED to
`num_segments*dim2` (documented in DESIGN.md); the two agree whenever
num_segments == dim1, which is how the reference's wrapper would have been used.
"""
import numpy as np


def unsorted_segment_sum_forward_gpu_semantics(data, segment_ids, num_segments, fix_stride=True):
  data = np.asarray(data, dtype=np.float32)
  ids = np.asarray(segment_ids, dtype=np.int64)
  B, D1, D2 = data.shape
  out = np.zeros((B * num_segments * D2,), dtype=np.float32)
  bstride = num_segments * D2 if fix_stride else D1 * D2
  for b in range(B):
    for c in range(D1):
      pos = b * bstride + ids[b, c] * D2
      out[pos:pos + D2] += data[b, c]  # segment_reduction.cu:48-50
  return out.reshape(B, num_segments, D2)


def unsorted_segment_sum_backward_gpu_semantics(grad_out, segment_ids, D1, fix_stride=True):
  g = np.asarray(grad_out, dtype=np.float32)
  ids = np.asarray(segment_ids, dtype=np.int64)
  B, S, D2 = g.shape
  flat = g.reshape(-1)
  bstride = S * D2 if fix_stride else D1 * D2
  out = np.zeros((B, D1, D2), dtype=np.float32)
  for b in range(B):
    for c in range(D1):
      pos = b * bstride + ids[b, c] * D2
      out[b, c] = flat[pos:pos + D2]  # segment_reduction.cu:64-66
  return out


def unsorted_segment_sum_forward_cpu_semantics(data, segment_ids, num_segments):
  """operators/src/segment_reduction.cpp:6-30 (batch-0 ids for every batch; stride dim1*dim2)."""
  data = np.asarray(data, dtype=np.float32)
  ids = np.asarray(segment_ids, dtype=np.int64).reshape(-1)
  B, D1, D2 = data.shape
  assert num_segments == D1, "reference CPU loop is only in-bounds when num_segments == dim1"
  out = np.zeros((B, D1, D2), dtype=np.float32)
  for b in range(B):
    for c in range(D1):
      out[b, ids[c]] += data[b, c]  # segment_reduction.cpp:20-25
  return out
