"""lnz_large_head (csrc/head_large.hip) against the torch readout on config 5's last state."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanczosnet_amd import ops
B, N, P = 256, 2048, 2
X = torch.randn((B, N, 128), device='cuda'); Wh = torch.randn((P + 1, 128), device='cuda'); bh = torch.randn((P + 1,), device='cuda')
mask = torch.ones((B, N), dtype=torch.uint8, device='cuda')
lin = torch.nn.Linear(128, P).cuda(); att = torch.nn.Sequential(torch.nn.Linear(128, 1), torch.nn.Sigmoid()).cuda()
def t(fn, n=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
def torch_head():
  y = lin(X) * att(X)
  m = (mask != 0).float().unsqueeze(2)
  return (y * m).sum(dim=1) / m.sum(dim=1)
with torch.no_grad():
  print('hip head ms', t(lambda: ops.large_head(X, mask, Wh, bh)), 'torch head ms', t(torch_head))
