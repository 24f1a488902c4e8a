"""Training-step timing (forward HIP + backward) at the BASELINE config-2 shape.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).train()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=os.environ.get('ADAM_FUSED', '1') == '1')
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); nf, mask, label = t(b['node_feat']), t(b['node_mask']), t(b['label'])
L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)


def step():
  opt.zero_grad(set_to_none=True)
  score, loss = net(nf, L, D, V, label=label, mask=mask)
  loss.backward()
  opt.step()
  return loss


for _ in range(3):
  step()
# (a full cyclic collection is a ~75 ms pause: with ten timed steps one of them inside the window
# reads as a 3x slower step — collected here, frozen, and the window is thirty steps)
import gc
gc.collect()
gc.freeze()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 30
for _ in range(K):
  loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
  with torch.no_grad():
    net(nf, L, D, V, mask=mask)
e1.record(); torch.cuda.synchronize()
# the same step replayed from a HIP graph (lanczosnet_amd.train.GraphedTrainStep)
from lanczosnet_amd.train import GraphedTrainStep, make_adam
loss = float(loss)   # drop the last eager autograd graph (its AccumulateGrad nodes belong to the
opt.zero_grad(set_to_none=True)  # default stream and must not be alive during the capture)
del opt
net.train()
opt_g = make_adam(net.parameters(), lr=1e-4)
gstep = GraphedTrainStep(net, opt_g, warmup=2)
for _ in range(4):
  gstep(nf, L, D, V, label, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
  gl = gstep(nf, L, D, V, label, mask)
torch.cuda.synchronize()
dtg = (time.perf_counter() - t0) / K
print(json.dumps({'workload': 'LanczosNet QM8 train step (fwd + bwd + Adam), B=%d' % B,
                  'train_step_ms': round(dt * 1e3, 3), 'molecules_per_s': round(B / dt, 1),
                  'graphed_train_step_ms': round(dtg * 1e3, 3),
                  'graphed_molecules_per_s': round(B / dtg, 1),
                  'forward_only_ms': round(e0.elapsed_time(e1) / K, 3), 'loss': loss,
                  'graphed_loss': float(gl)}))
