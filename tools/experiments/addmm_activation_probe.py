import torch, time
x=torch.randn(1024,4096,device='cuda'); w=torch.randn(4096,4096,device='cuda')/64; b=torch.randn(4096,device='cuda')
def t(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
a=torch.relu_(torch.nn.functional.linear(x,w,b)); c=torch._addmm_activation(b,x,w.t())
print('equal', torch.equal(a,c), float((a-c).abs().max()))
print('linear+relu_ ms', t(lambda: torch.relu_(torch.nn.functional.linear(x,w,b))))
print('_addmm_activation ms', t(lambda: torch._addmm_activation(b,x,w.t())))
print('linear only ms', t(lambda: torch.nn.functional.linear(x,w,b)))
