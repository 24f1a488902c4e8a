"""Wall time of each of 30 eager LanczosNet training steps (a one-off pause in the timed window of
tools/bench_train_step.py showed up as a 3x slower step: the interpreter's cyclic collector; NOGC=1)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
B = 1024
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).train()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); nf, mask, label = t(b['node_feat']), t(b['node_mask']), t(b['label'])
L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
import gc
if os.environ.get('NOGC'):
  gc.collect(); gc.disable()
ts = []
for i in range(30):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  opt.zero_grad(set_to_none=True)
  score, loss = net(nf, L, D, V, label=label, mask=mask)
  loss.backward()
  opt.step()
  torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
print(ts)
