from .unsorted_segment_sum import UnsortedSegmentSum  # noqa: F401
