import sys, torch
sys.path.insert(0,'/root/repo')
from lanczosnet_amd import ops
B,N,P=1024,26,16
g=torch.Generator(device='cuda'); g.manual_seed(0)
X=torch.relu(torch.randn((B,32,128),generator=g,device='cuda'))
mask=torch.ones((B,N),dtype=torch.uint8,device='cuda')
W=torch.randn((P+1,128),generator=g,device='cuda'); bh=torch.randn((P+1,),generator=g,device='cuda')
gs=torch.randn((B,P),generator=g,device='cuda')
dY=torch.empty((B,32,128),device='cuda')
ro=(torch.arange(B,device='cuda')*N).long()
dYc=torch.empty((B*N,128),device='cuda')
for nwg in (256,512,1024):
  for _ in range(3): ops.head_backward(X,mask,gs,W,bh,N,dY,row_off=ro,dY_compact=dYc,n_wg=nwg)
  e=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
  e[0].record()
  for _ in range(20): ops.head_backward(X,mask,gs,W,bh,N,dY,row_off=ro,dY_compact=dYc,n_wg=nwg)
  e[1].record(); torch.cuda.synchronize()
  print(nwg, e[0].elapsed_time(e[1])/20*1e3,'us')
