"""`UnsortedSegmentSum(num_segments)` — module form of R12 (reference
`operators/modules/unsorted_segment_sum.py:7-15`)."""
import torch.nn as nn

from ..functions.unsorted_segment_sum import UnsortedSegmentSumFunction


class UnsortedSegmentSum(nn.Module):

  def __init__(self, num_segments):
    super(UnsortedSegmentSum, self).__init__()
    self.num_segments = num_segments

  def forward(self, data, segment_index):
    return UnsortedSegmentSumFunction.apply(data, segment_index, self.num_segments)
