import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lanczosnet_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((1024, 4096, 4096), (1024, 4096, 2048), (1024, 4096, 1024), (1024, 2048, 4096), (512, 4096, 4096), (128, 128, 4096), (128, 128, 64), (128, 128, 96), (1024, 4096, 4064)):
  x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5
  ref = (x.double() @ w.double().t())
  errs = []
  for rep in range(3):
    out = ops.f32_linear(x, w, None, relu=False)
    e = (out.double() - ref).abs()
    bad = (e > 1e-3).nonzero()
    errs.append((float(e.max() / ref.abs().max()), int(bad.shape[0]), bad[:3].tolist()))
  print(M, N, K, errs)
