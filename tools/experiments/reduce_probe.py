"""Which of the small torch reductions of the training backward is the 43 us reduce_kernel."""
import torch
def t(fn, reps=50):
  for _ in range(5): fn()
  torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / reps * 1e3, 1)
part = torch.randn(4, 128, 1920, device='cuda')
dy = torch.randn(7, 1024 * 32, 128, device='cuda')
print('part.sum(0) [4,128,1920]           %6.1f us' % t(lambda: part.sum(dim=0)))
print('part[0]+part[1]+part[2]+part[3]     %6.1f us' % t(lambda: (part[0] + part[1]) + (part[2] + part[3])))
print('dy.sum(1) [7,32768,128]             %6.1f us' % t(lambda: dy.sum(dim=1)))
ones = torch.ones(7, 1, 1024 * 32, device='cuda')
print('bmm(ones, dy)                       %6.1f us' % t(lambda: torch.bmm(ones, dy)))
g = torch.randn(1024, 4096, device='cuda')
print('g.sum(0) [1024,4096]                %6.1f us' % t(lambda: g.sum(dim=0)))
