"""Per-tensor projection error of the Ada gradients against the reference's float64 gradients, for
both backward implementations (diagnostic for tests/test_gpu_ada.py)."""
import ast, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import oracle
from conftest import load_golden
import test_gpu_ada as T
from gradproj import project_torch
g = load_golden('ada_e2e.npz'); gp = load_golden('grad_projections.npz')
cfg = ast.literal_eval(str(g['cfg_json']))
P = oracle.make_ada_params(cfg, int(g['param_seed']))
nf, L, mask, lab = T._e2e_inputs(g)
gi = g['grad_idx']
res = {}
for impl in ('hip', 'torch'):
  net = T._ada_model(cfg, P).train(); net.backward_impl = impl
  with T._fixed_randn(g['q1'][gi]):
    _, loss = net(T._t(nf[gi]), T._t(L[gi]), label=T._t(lab[gi]), mask=T._t(mask[gi]))
  loss.backward()
  gd = dict(net.named_parameters())
  res[impl] = {}
  for i, k in enumerate(gp['ada64_names']):
    pr = project_torch(gd[str(k)].grad, i)
    res[impl][str(k)] = float(np.abs(pr - gp['ada64_proj'][i]).max() / float(gp['ada64_norm'][i]))
for k in res['hip']:
  print('%-32s hip %.2e  torch %.2e' % (k, res['hip'][k], res['torch'][k]))

# stage-level comparison: fused backward's intermediate gradients vs autograd through the restatement
os.environ['LNZ_ADA_DEBUG'] = '1'
net = T._ada_model(cfg, P).train(); net.backward_impl = 'hip'
a = [T._t(x[gi]) for x in (nf, L, lab, mask)]
with T._fixed_randn(g['q1'][gi]):
  _, loss = net(a[0], a[1], label=a[2], mask=a[3])
loss.backward()
d = net._dbg
q1 = torch.from_numpy(np.ascontiguousarray(g['q1'][gi][:, :, None])).cuda()
st, tc, Q = net._torch_ada_spectrum(a[0], a[1], a[3], q1)
st = st.detach().requires_grad_(True); tc = tc.detach().requires_grad_(True); Q = Q.detach().requires_grad_(True)
DDs = net._torch_ada_filters(tc)
for x in DDs: x.retain_grad()
sc = net._torch_ada_conv(st, a[1], Q, DDs, a[3])
ls = net.loss_func(sc, a[2]); ls.backward()
def re(x, y): return float((x.double() - y.double()).abs().max() / y.double().abs().max())
K, S = net.num_eig_vec, net.num_scale_long
print('forward: Q %.2e tcat %.2e DD0 %.2e' % (re(d['Q'], Q), re(d['tcat'], tc), re(d['DDp'][0].permute(0, 2, 3, 1), DDs[0])))
print('dQ %.2e  dtcat %.2e  dstate %.2e' % (re(d['dQ'], Q.grad), re(d['dtcat'], tc.grad), re(d['dx0'], st.grad)))
for t in range(net.num_layer):
  # DDs[t].grad is d/d(symmetrised DD) [B,K,K,S]
  print('layer %d dDD %.2e' % (t, re(d['dDDp'][t].permute(0, 2, 3, 1), DDs[t].grad)))
mk = (a[3] != 0)
e = (d['dQ'].double() - Q.grad.double()).abs()
print('dQ err real rows %.2e, padded rows %.2e (scale %.2e)' % (float(e[mk].max()), float(e[~mk].max()) if (~mk).any() else 0, float(Q.grad.abs().max())))
