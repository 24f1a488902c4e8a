import os, sys, torch, json
sys.path.insert(0, os.getcwd())
from lanczosnet_amd import ops
import ctypes as C
B,N,K,S=256,2048,64,8
X=torch.randn(B,N,128,device='cuda'); V=torch.randn(B,N,K,device='cuda'); G=torch.randn(B,S,K,device='cuda')
Wt=ops.pack_rows_k8(torch.randn(128,S*128,device='cuda')); Y=torch.zeros(B,64,128,device='cuda'); Tt=torch.zeros(1,B,128,64,dtype=torch.bfloat16,device='cuda')
def run():
  ops._abi().large_spectral(X,128,128,V,G,Wt,B,N,K,S,1,Y,Tt)
run(); torch.cuda.synchronize()
e=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
for _ in range(20): run()
e[1].record(); torch.cuda.synchronize()
print(os.environ.get('LNZ_LARGE_PROJECT_WGS'), 'project+spectral ms', e[0].elapsed_time(e[1])/20)
