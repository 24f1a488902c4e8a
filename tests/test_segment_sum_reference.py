"""R12 pinned on the REAL reference: `oracle/_ref/segment_reduction_ref*.so` is the reference's CPU
operator compiled from its own unmodified source (`operators/src/segment_reduction.cpp`, built by
`oracle/ref_build.py` in the build container; the binary travels to the GPU box).

* CPU: the oracle's restatement of the CPU semantics equals the built reference bit for bit —
  including its quirk of indexing `segment_ids` with the row index only (batch-0 ids for every
  batch, `segment_reduction.cpp:20,47`).
* GPU: the HIP operator equals the built reference wherever the reference's CPU and GPU semantics
  coincide (all batch entries share their ids and num_segments == dim1 — the only in-bounds use of
  both): forward scatter-add and backward gather.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import ref_build

REF_OP = ref_build.load()
needs_ref = pytest.mark.skipif(REF_OP is None, reason='oracle/_ref not built (no reference tree)')


def _ref_forward(data, ids):
  B, D1, D2 = data.shape
  out = torch.zeros((B, D1, D2), dtype=torch.float32)
  assert REF_OP.unsorted_segment_sum_forward(torch.from_numpy(data), torch.from_numpy(ids),
                                              [B, D1, D2], out) == 1
  return out.numpy()


def _ref_backward(gout, ids):
  B, D1, D2 = gout.shape
  gd = torch.zeros((B, D1, D2), dtype=torch.float32)
  assert REF_OP.unsorted_segment_sum_backward(torch.from_numpy(gout), torch.from_numpy(ids),
                                               [B, D1, D2], gd) == 1
  return gd.numpy()


@needs_ref
def test_oracle_cpu_semantics_equal_the_built_reference():
  rs = np.random.RandomState(0)
  for (B, D1, D2) in ((1, 5, 3), (3, 7, 5), (4, 33, 64), (2, 100, 17)):
    data = rs.randn(B, D1, D2).astype(np.float32)
    ids = rs.randint(0, D1, size=(B, D1)).astype(np.int64)   # rows > 0 are ignored by the reference
    ref = _ref_forward(data, ids)
    np.testing.assert_array_equal(
        ref, oracle.unsorted_segment_sum_forward_cpu_semantics(data, ids[0], D1))
    # shared ids: CPU semantics == GPU semantics (the restatement the HIP kernel is tested against)
    shared = np.repeat(ids[:1], B, axis=0)
    np.testing.assert_array_equal(
        _ref_forward(data, shared), oracle.unsorted_segment_sum_forward_gpu_semantics(data, shared, D1))
    gout = rs.randn(B, D1, D2).astype(np.float32)
    np.testing.assert_array_equal(
        _ref_backward(gout, shared),
        oracle.unsorted_segment_sum_backward_gpu_semantics(gout, shared, D1))


@needs_ref
@pytest.mark.gpu
def test_hip_segment_sum_equals_the_built_reference():
  from lanczosnet_amd import ops
  rs = np.random.RandomState(1)
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')  # noqa: E731
  for (B, D1, D2) in ((3, 7, 5), (4, 33, 64), (2, 100, 128), (5, 40, 260), (600, 24, 256)):
    ids = np.repeat(rs.randint(0, D1, size=(1, D1)), B, axis=0).astype(np.int64)
    data_i = rs.randint(-8, 9, size=(B, D1, D2)).astype(np.float32)
    got = ops.unsorted_segment_sum_forward(dev(data_i), dev(ids), D1).cpu().numpy()
    np.testing.assert_array_equal(got, _ref_forward(data_i, ids))            # integers: exact
    data_f = rs.randn(B, D1, D2).astype(np.float32)
    got = ops.unsorted_segment_sum_forward(dev(data_f), dev(ids), D1).cpu().numpy()
    ref = _ref_forward(data_f, ids)
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())    # float: summation order
    gout = rs.randn(B, D1, D2).astype(np.float32)
    gd = ops.unsorted_segment_sum_backward(dev(gout), dev(ids), D1).cpu().numpy()
    np.testing.assert_array_equal(gd, _ref_backward(gout, ids))              # gather: exact


def test_operator_classes_have_the_reference_surface_and_refuse_cpu_tensors():
  """`operators/functions/unsorted_segment_sum.py:8-44`, `operators/modules/unsorted_segment_sum.py:7-15`:
  same class names, call forms and attribute; no CPU path behind them."""
  from lanczosnet_amd.operators.functions.unsorted_segment_sum import UnsortedSegmentSumFunction
  from lanczosnet_amd.operators.modules.unsorted_segment_sum import UnsortedSegmentSum
  m = UnsortedSegmentSum(7)
  assert m.num_segments == 7 and not list(m.parameters())
  assert issubclass(UnsortedSegmentSumFunction, torch.autograd.Function)
  with pytest.raises(Exception):
    m(torch.zeros(2, 7, 3), torch.zeros(2, 7, dtype=torch.int64))
  with pytest.raises(ValueError):
    UnsortedSegmentSumFunction.apply(torch.zeros(2, 7), torch.zeros(2, 7, dtype=torch.int64), 7)


@needs_ref
@pytest.mark.gpu
def test_operator_function_and_module_autograd_against_the_built_reference():
  """`UnsortedSegmentSumFunction.apply(data, segment_index, num_segments)` (the form
  `model/mpnn.py:9,88` uses) and `UnsortedSegmentSum(num_segments)(data, segment_index)`: forward
  and the gradient autograd returns for `data` against the built reference operator's forward /
  backward on the same inputs (shared ids, num_segments == dim1: the in-bounds use of both), plus
  a finite-difference check of the backward on integer-valued data (exact in fp32)."""
  from lanczosnet_amd.operators.functions.unsorted_segment_sum import UnsortedSegmentSumFunction
  from lanczosnet_amd.operators.modules.unsorted_segment_sum import UnsortedSegmentSum
  rs = np.random.RandomState(4)
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')  # noqa: E731
  for (B, D1, D2) in ((3, 7, 5), (4, 33, 64), (2, 100, 128), (64, 24, 256)):
    ids = np.repeat(rs.randint(0, D1, size=(1, D1)), B, axis=0).astype(np.int64)
    data = rs.randint(-8, 9, size=(B, D1, D2)).astype(np.float32)
    gout = rs.randint(-4, 5, size=(B, D1, D2)).astype(np.float32)
    x = dev(data).requires_grad_(True)
    y = UnsortedSegmentSumFunction.apply(x, dev(ids), D1)
    np.testing.assert_array_equal(y.detach().cpu().numpy(), _ref_forward(data, ids))
    y.backward(dev(gout))
    np.testing.assert_array_equal(x.grad.cpu().numpy(), _ref_backward(gout, ids))
    # module form, with a loss: d/d data of sum(w * y) is the gather of w
    x2 = dev(data).requires_grad_(True)
    (UnsortedSegmentSum(D1)(x2, dev(ids)) * dev(gout)).sum().backward()
    assert torch.equal(x2.grad, x.grad)
    # finite differences on one entry per batch (integers: every sum is exact)
    b, i, j = rs.randint(B), rs.randint(D1), rs.randint(D2)
    bump = data.copy()
    bump[b, i, j] += 1.0
    d = (UnsortedSegmentSumFunction.apply(dev(bump), dev(ids), D1) - y.detach()) * dev(gout)
    assert float(d.sum()) == float(x.grad[b, i, j])
  # per-batch ids and num_segments != dim1 (the GPU semantics): against the oracle's restatement
  B, D1, D2, S = 5, 40, 36, 11
  ids = rs.randint(0, S, size=(B, D1)).astype(np.int64)
  data = rs.randint(-8, 9, size=(B, D1, D2)).astype(np.float32)
  gout = rs.randint(-4, 5, size=(B, S, D2)).astype(np.float32)
  x = dev(data).requires_grad_(True)
  y = UnsortedSegmentSumFunction.apply(x, dev(ids), S)
  np.testing.assert_array_equal(y.detach().cpu().numpy(),
                                oracle.unsorted_segment_sum_forward_gpu_semantics(data, ids, S))
  y.backward(dev(gout))
  np.testing.assert_array_equal(x.grad.cpu().numpy(),
                                oracle.unsorted_segment_sum_backward_gpu_semantics(gout, ids, D1))
