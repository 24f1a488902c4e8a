"""The reference's custom-operator package surface (`operators/`), backed by the HIP kernels.

`from lanczosnet_amd.operators.functions.unsorted_segment_sum import UnsortedSegmentSumFunction`
replaces `from operators.functions.unsorted_segment_sum import ...` (reference `model/mpnn.py:6`)."""
