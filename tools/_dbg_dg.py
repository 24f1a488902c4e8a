import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).train()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); mask = t(b['node_mask']).to(torch.uint8).contiguous()
L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
plan = net._plan_backward()
Lp = ops.pack_laplacian_for(plan, L)
N, K, Lnum, dh, S = V.shape[1], 20, 7, 128, 8
din0p = plan['din0']
g = torch.Generator(device='cuda').manual_seed(1)
act = torch.zeros((Lnum, B, 32, dh), device='cuda'); dy = torch.zeros_like(act)
x0 = torch.zeros((B, 32, din0p), device='cuda')
rowmask = (torch.arange(32, device='cuda')[None, :] < n[:, None]).float()
act[:] = torch.rand((Lnum, B, 32, dh), device='cuda', generator=g) * rowmask[None, :, :, None]
dy[:] = torch.randn((Lnum, B, 32, dh), device='cuda', generator=g) * rowmask[None, :, :, None]
x0[:] = torch.randn((B, 32, din0p), device='cuda', generator=g) * rowmask[:, :, None]
tiles = ops.plan_tiles(mask, allow_pairs=True)
dG = ops.lanczosnet_gain_grad(plan, Lp, V, None, mask, act, x0, dy, tiles)
# torch reference
n_short = 0; n_chan = 15
Vt = V.transpose(1, 2)
ref = []
for la in range(Lnum):
  d = din0p if la == 0 else dh
  X = x0[:, :N] if la == 0 else act[la - 1][:, :N]
  Y = torch.bmm(Vt, X)                       # [B,K,d]
  Pm = torch.bmm(Vt, dy[la][:, :N])           # [B,K,dh]
  W = net._mix_weight(la).detach().view(dh, n_chan, -1)[:, n_short:n_short + S, :]
  if W.shape[2] != d: W = torch.nn.functional.pad(W, (0, d - W.shape[2]))
  Qs = torch.einsum('bki,osi->bkso', Y, W)    # [B,K,S,dh]
  ref.append((Qs * Pm.unsqueeze(2)).sum(3))   # [B,K,S]
ref = torch.stack(ref)
err = (dG - ref).abs().max().item(); sc = ref.abs().max().item()
print('B=%d max abs err %.3e (scale %.3e)' % (B, err, sc))
bad = ((dG - ref).abs() > 1e-3 * sc).nonzero()
print('bad entries', bad.shape[0], bad[:10].tolist())
for e in bad[:8].tolist():
  print(e, float(dG[tuple(e)]), float(ref[tuple(e)]), 'n=', int(n[e[1]]))
tiles1 = ops.plan_tiles(mask, allow_pairs=False)
dG1 = ops.lanczosnet_gain_grad(plan, Lp, V, None, mask, act, x0, dy, tiles1)
print('singles: max err %.3e' % (dG1 - ref).abs().max().item(), ' repeat equal:', bool(torch.equal(dG, ops.lanczosnet_gain_grad(plan, Lp, V, None, mask, act, x0, dy, tiles))))
buf, cap = tiles
print(buf[:12 * cap].view(cap, 4, 3)[:int(buf[12 * cap])].tolist()[:6])
