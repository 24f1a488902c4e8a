"""Are the seven layers' filter-MLP GEMMs faster as ONE batched library call (batch = 7)?"""
import torch
x = torch.randn(7, 1024, 4096, device='cuda'); w = torch.randn(7, 4096, 4096, device='cuda') / 64
b = torch.randn(7, 1, 4096, device='cuda')
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
def separate():
  return [torch._addmm_activation(b[i, 0], x[i], w[i].t()) for i in range(7)]
def batched():
  return torch.relu_(torch.baddbmm(b, x, w.transpose(1, 2)))
print('7 separate fused linear+relu: %.4f ms' % t(separate))
print('one baddbmm + relu_:          %.4f ms' % t(batched))
print('one bmm (no bias/relu):       %.4f ms' % t(lambda: torch.bmm(x, w.transpose(1, 2))))
