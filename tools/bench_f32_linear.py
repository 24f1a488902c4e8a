"""lnz_f32_linear (hand-written exact-fp32 Linear) against torch.nn.functional.linear (+ relu) on
the AdaLanczosNet filter-MLP shapes, M = 1024.  One JSON line per shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lanczosnet_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)


def timed(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


for (N, K, relu) in ((4096, 832, True), (4096, 4096, True), (1056, 4096, False)):
  x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5
  b = torch.randn(N, device='cuda')
  ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
  if relu: ref = torch.relu(ref)
  out = ops.f32_linear(x, w, b, relu=relu)
  lib = torch.nn.functional.linear(x, w, b)
  if relu: lib = torch.relu(lib)
  err = float((out.double() - ref).abs().max() / ref.abs().max())
  err_lib = float((lib.double() - ref).abs().max() / ref.abs().max())
  t = timed(lambda: ops.f32_linear(x, w, b, relu=relu))
  o = torch.empty(M, N, device='cuda')
  t_lib = timed(lambda: torch.relu_(torch.nn.functional.linear(x, w, b)) if relu
                else torch.nn.functional.linear(x, w, b))
  fl = 2.0 * M * N * K
  print(json.dumps({'M': M, 'N': N, 'K': K, 'hand_written_ms': round(t, 4),
                    'hand_written_TFLOPs': round(fl / t / 1e9, 1),
                    'frac_of_fp32_mfma_peak': round(fl / t / 1e9 / 157.3, 3),
                    'library_ms': round(t_lib, 4), 'library_TFLOPs': round(fl / t_lib / 1e9, 1),
                    'err_vs_float64': err, 'library_err_vs_float64': err_lib}))
