"""HIP streams restricted to a set of compute units (hipExtStreamCreateWithCUMask).

A latency-chain kernel that fills a quarter of the chip (the workgroup-per-graph Ritz launch of the
reference's graph configuration: 64 graphs -> 64 of 256 compute units for 0.45 ms) can share the
device with the NEXT stage of the previous batch — but only apart from it: forward waves scheduled
onto the same compute units stretch the Ritz chain from 0.45 to 0.60 ms (measured), and the two
stages then take as long side by side as one after the other.  Two streams on disjoint compute
units: 0.67 -> 0.50 ms per batch (tools/experiments/graph_config_streams.py, bench.py
graph_configuration_mode.stream_of_batches_two_streams)."""
import ctypes

import torch

_hip = None


def cu_masked_stream(first_cu, end_cu, device=None):
  """A torch.cuda.ExternalStream whose kernels run on compute units [first_cu, end_cu) only."""
  global _hip
  if _hip is None:
    _hip = ctypes.CDLL('libamdhip64.so')
  dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
  n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
  assert 0 <= first_cu < end_cu <= n_cu, (first_cu, end_cu, n_cu)
  n_words = (n_cu + 31) // 32
  words = (ctypes.c_uint32 * n_words)(*([0] * n_words))
  for cu in range(first_cu, end_cu):
    words[cu // 32] |= 1 << (cu % 32)
  st = ctypes.c_void_p()
  with torch.cuda.device(dev):
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), n_words, words)
  if rc != 0:
    raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
  return torch.cuda.ExternalStream(st.value, device=dev)
