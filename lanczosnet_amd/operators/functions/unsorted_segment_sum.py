"""`UnsortedSegmentSumFunction` — the autograd surface of R12 (reference
`operators/functions/unsorted_segment_sum.py:8-44`; call site `model/mpnn.py:9,88`):
`UnsortedSegmentSumFunction.apply(data, segment_index, num_segments)` with `data [B, D1, D2]`
float32, `segment_index [B, D1]` int64, reduction over dim 1 into `[B, num_segments, D2]`.

Forward = `lnz_unsorted_segment_sum_forward` (LDS-privatised scatter-add, source-row order, the
GPU semantics of `operators/src/cuda/segment_reduction.cu:39-62`: every batch entry uses its own
row of ids); backward = `lnz_unsorted_segment_sum_backward`, the gather
`grad_data[b, i, :] = grad_out[b, ids[b, i], :]` (`segment_reduction.cu:64-95`).  Like the
reference's `.cu` path an id outside `[0, num_segments)` contributes nothing.  Device tensors
only: there is no CPU path in this package (a CPU tensor raises)."""
import torch
from torch.autograd import Function

from ... import ops


class UnsortedSegmentSumFunction(Function):

  @staticmethod
  def forward(ctx, data, segment_index, num_segments):
    if data.dim() != 3 or segment_index.dim() != 2 or tuple(segment_index.shape) != tuple(data.shape[:2]):
      raise ValueError('data [B, D1, D2] and segment_index [B, D1] expected, got %s / %s'
                       % (tuple(data.shape), tuple(segment_index.shape)))
    ctx.save_for_backward(segment_index)
    ctx.dim1 = int(data.shape[1])
    ctx.in_dtype = data.dtype
    return ops.unsorted_segment_sum_forward(data, segment_index, int(num_segments))

  @staticmethod
  def backward(ctx, grad_output):
    segment_index, = ctx.saved_tensors
    grad_data = ops.unsorted_segment_sum_backward(grad_output, segment_index, ctx.dim1)
    return grad_data.to(ctx.in_dtype), None, None
