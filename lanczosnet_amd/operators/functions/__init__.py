from .unsorted_segment_sum import UnsortedSegmentSumFunction  # noqa: F401
