"""AdaLanczosNet training step (HIP forward + HIP conv-stack backward, Adam), B=1024; stage times
of the backward and, for comparison, of the differentiable torch restatement."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import AdaLanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1, 2, 3], long_diffusion_dist=[5, 7, 10, 20, 30])
torch.manual_seed(1234)
net = AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet')).train().cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=os.environ.get('ADAM_FUSED', '1') == '1')
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); nf, mask, label = t(b['node_feat']), t(b['node_mask']), t(b['label'])
L = ops.laplacian_l4(t(b['adjs']), n)
def step():
  opt.zero_grad(set_to_none=True)
  _, loss = net(nf, L, label=label, mask=mask)
  loss.backward(); opt.step(); return loss
for _ in range(4): step()   # (library GEMM selection for new shapes happens in the first calls)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
with torch.no_grad():
  net.eval(); net(nf, L, mask=mask); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): net(nf, L, mask=mask)
  torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 5
# where the backward's time goes: the three stages of the differentiable restatement, each
# forward + backward on its own (events), and the optimizer step
net.train()
q1 = torch.randn(B, nf.shape[1], 1).cuda()
def timed(fn, reps=4):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return round(e0.elapsed_time(e1) / reps, 2)
def stage_spectrum():
  st, tc, Q = net._torch_ada_spectrum(nf, L, mask, q1)
  torch.autograd.backward([st, tc, Q], [torch.ones_like(st), torch.ones_like(tc), torch.ones_like(Q)])
with torch.no_grad():
  st0, tc0, Q0 = net._torch_ada_spectrum(nf, L, mask, q1)
def stage_filters():
  dd = net._torch_ada_filters(tc0.detach().requires_grad_(True))
  torch.autograd.backward(dd, [torch.ones_like(d) for d in dd])
with torch.no_grad():
  DD0 = net._torch_ada_filters(tc0)
def stage_conv():
  sc = net._torch_ada_conv(st0.detach().requires_grad_(True), L, Q0.detach().requires_grad_(True),
                           [d.detach().requires_grad_(True) for d in DD0], mask)
  sc.sum().backward()
stages = {'spectrum_fwd_bwd_ms': timed(stage_spectrum), 'filter_mlps_fwd_bwd_ms': timed(stage_filters),
          'conv_stack_fwd_bwd_ms': timed(stage_conv), 'adam_ms': timed(opt.step)}
# the HIP backward's own stages (events recorded inside _AdaLanczosNetFusedFunction.backward)
os.environ['LNZ_ADA_DEBUG'] = '1'
fe = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
opt.zero_grad(set_to_none=True)
fe[0].record(); _, loss2 = net(nf, L, label=label, mask=mask); fe[1].record()
loss2.backward(); fe[2].record(); torch.cuda.synchronize()
mk = net._dbg['marks']
hip_bwd = {b[0] + '_ms': round(a[1].elapsed_time(b[1]), 2) for a, b in zip(mk[:-1], mk[1:])}
hip_bwd['training_forward_ms'] = round(fe[0].elapsed_time(fe[1]), 2)
hip_bwd['backward_total_ms'] = round(fe[1].elapsed_time(fe[2]), 2)
del net._dbg
os.environ['LNZ_ADA_DEBUG'] = '0'
# the same step replayed from a HIP graph (lanczosnet_amd.train.GraphedTrainStep)
from lanczosnet_amd.train import GraphedTrainStep, make_adam
loss = float(loss); loss2 = float(loss2)
opt.zero_grad(set_to_none=True)
del opt
opt_g = make_adam(net.parameters(), lr=1e-4)
gstep = GraphedTrainStep(net, opt_g, warmup=2)
for _ in range(4): gstep(nf, L, None, None, label, mask)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): gl = gstep(nf, L, None, None, label, mask)
torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / 8
print(json.dumps({'workload': 'AdaLanczosNet train step B=%d' % B, 'graphed_train_step_ms': round(dtg * 1e3, 2),
                  'graphed_loss': float(gl), 'hip_backward_stages': hip_bwd,
                  'restatement_stages': stages, 'train_step_ms': round(dt * 1e3, 2),
                  'forward_ms': round(df * 1e3, 2), 'loss': float(loss)}))
