"""One mode of the training step per process, for `rocprofv3 --kernel-trace --stats`:
  python tools/train_step_profile.py eager|eager_fused|graph [steps]     (eager_fused: torch.optim.Adam(fused=True))
prints wall ms per step; the kernel stats of the run give the summed kernel time per step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.train import GraphedTrainStep, make_adam
from lanczosnet_amd.utils.arg_helper import make_model_config

mode = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 1024
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).train()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); nf, mask, label = t(b['node_feat']), t(b['node_mask']), t(b['label'])
L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True) if mode == 'eager_fused' else \
    make_adam(net.parameters(), lr=1e-4)
if mode == 'graph':
  gs = GraphedTrainStep(net, opt, warmup=2)
  step = lambda: gs(nf, L, D, V, label, mask)
else:
  def step():
    opt.zero_grad(set_to_none=True)
    _, loss = net(nf, L, D, V, label=label, mask=mask)
    loss.backward()
    opt.step()
    return loss.detach()
for _ in range(4):
  step()
torch.cuda.synchronize()
import gc
gc.collect()
gc.freeze()
t0 = time.perf_counter()
for _ in range(K):
  loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({'mode': mode, 'steps_timed': K, 'steps_total': K + 4, 'wall_ms_per_step': round(dt * 1e3, 3),
                  'loss': float(loss)}))
