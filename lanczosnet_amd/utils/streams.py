"""HIP streams restricted to a set of compute units (hipExtStreamCreateWithCUMask through lnz_stream_create_cu_masked).

A latency-chain kernel that fills a quarter of the chip (the workgroup-per-graph Ritz launch of the
reference's graph configuration: 64 graphs -> 64 of 256 compute units for 0.45 ms) can share the
device with the NEXT stage of the previous batch — but only apart from it: forward waves scheduled
onto the same compute units stretch the Ritz chain from 0.45 to 0.60 ms (measured), and the two
stages then take as long side by side as one after the other.  Two streams on disjoint compute
units: 0.67 -> 0.50 ms per batch (tools/experiments/graph_config_streams.py, bench.py
graph_configuration_mode.stream_of_batches_two_streams)."""
import torch

from .. import _torch_ext


def cu_masked_stream(first_cu, end_cu, device=None):
  """A torch.cuda.ExternalStream whose kernels run on compute units [first_cu, end_cu) only
  (lnz_stream_create_cu_masked)."""
  _torch_ext.load()
  dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
  idx = dev.index if dev.index is not None else torch.cuda.current_device()
  handle = torch.ops.lanczosnet.cu_masked_stream(int(first_cu), int(end_cu), int(idx))
  return torch.cuda.ExternalStream(handle, device=torch.device('cuda', idx))
