#!/usr/bin/env python
"""BASELINE.json configs[4]: LanczosNetGeneral (config/graph_lanczos_net.yaml architecture),
N = 2048 nodes, K = 64, batch 256, 1 x MI355X.  Lanczos (HIP, HBM bound) + gains (HIP) + conv:
the streamed HIP kernels of csrc/conv_large.hip (bf16 operands, and the 3-piece split-precision
parity mode) beside the library-GEMM path they replaced (hipBLASLt via torch.bmm)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNetGeneral
from lanczosnet_amd.utils.arg_helper import make_model_config

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--nodes', type=int, default=2048)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--library', action='store_true', help='also time the hipBLASLt path')
ap.add_argument('--full-stream', action='store_true', help='Lanczos on the whole A (lnz_lanczos_ritz_large) instead of the upper chunk blocks (lnz_lanczos_ritz_large_sym)')
ap.add_argument('--dense-stream', action='store_true', help='the dense streams (A re-read every step) instead of the product entry (ops.lanczos_ritz -> lnz_lanczos_ritz_kstep: A read once into a sliced-ELL image)')
args = ap.parse_args()
B, N, K = args.batch, args.nodes, 64
cfg = dict(num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
           num_eig_vec=K, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7,
           output_dim=2, num_layer=7, num_atom=0)
torch.manual_seed(1234)
net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval().cuda()
g = torch.Generator(device='cuda'); g.manual_seed(0)
L = torch.empty((B, N, N, 2), dtype=torch.float32, device='cuda')
for b in range(B):
  adj = (torch.rand((N, N), generator=g, device='cuda') < 0.01).float().triu(1)
  adj = adj + adj.t() + torch.eye(N, device='cuda')
  d = adj.sum(1).rsqrt()
  A = d[:, None] * adj * d[None, :]
  L[b, :, :, 0] = A
  L[b, :, :, 1] = A
X = torch.randn((B, N, 10), generator=g, device='cuda')
mask = torch.ones((B, N), dtype=torch.uint8, device='cuda')
A0 = L[:, :, :, 0].contiguous()
ws = torch.empty((ops._abi().lanczos_ritz_kstep_workspace_bytes(B, N, 3, ops.kstep_row_cap(N)),), dtype=torch.uint8, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = {}
modes = [('hip_bf16', 1), ('hip_split3', 3), ('hip_f16x2', 2)] + ([('library_fp32', None)] if args.library else [])
SYM = not args.full_stream
if args.dense_stream:
  ritz = lambda: ops.lanczos_ritz_large(A0, K, K, workspace=ws, symmetric=SYM)
else:
  # the product surface's input: channel 0 of the collated channels-last tensor, read in place
  # (dataset/graph_data.py collate_graph_adjacency -> ops.lanczos_ritz(L[:, :, :, 0], ...))
  # — and the same pass leaves the conv's sparse image riding on L (ops.lanczos_ritz_collated)
  import warnings
  warnings.simplefilter('ignore')
  ritz = lambda: ops.lanczos_ritz_collated(L, None, K)
D, V = ritz()
ref = None
for name, planes in modes:
  best = None
  for it in range(args.reps + 1):
    ev[0].record()
    D, V = ritz()
    ev[1].record()
    with torch.no_grad():
      if planes is None:
        score = net._large_graph_forward(X, L, D, V, mask)
      else:
        score = net._large_graph_forward_hip(X, L, D, V, mask, planes=planes)
    ev[2].record()
    torch.cuda.synchronize()
    t = (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
    if it > 0 and (best is None or sum(t) < sum(best)):
      best = t
  if name == 'hip_split3':
    ref = score
  res[name] = {'lanczos_ms': round(best[0], 2), 'forward_ms': round(best[1], 2),
               'graphs_per_s': round(B / (sum(best) * 1e-3), 1), 'finite': bool(torch.isfinite(score).all())}
  if name == 'hip_bf16':
    s1 = score
  if name == 'hip_f16x2':
    s2 = score
if ref is not None:
  res['bf16_vs_split3_rel'] = float((s1 - ref).abs().max() / ref.abs().max())
  res['f16x2_vs_split3_rel'] = float((s2 - ref).abs().max() / ref.abs().max())
# per-stage breakdown of the bf16 mode: as the module runs it in the steady state (equal channels
# folded: one operator packed and streamed, weight blocks summed) and with every channel packed
def stages(classes):
  src = sorted(set(classes))
  rep = [src.index(c) for c in classes]
  with torch.no_grad():
    plan = net._plan_large(1, classes)
    G = ops.spectral_gains(D, net.long_diffusion_dist, net.num_layer, plan['mlp_pack'])
    neq = torch.zeros((1,), dtype=torch.int64, device='cuda')
    ops.large_pack_operators(L, V, 1, chan_src=src, chan_rep=rep, neq=neq)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    Lb, Vb = ops.large_pack_operators(L, V, 1, chan_src=src, chan_rep=rep, neq=neq)
    e[1].record()
    work = ops.large_work_buffers(Lb)
    state = X.contiguous()
    e[1].record()
    for t, lay in enumerate(plan['conv'][(1, tuple(classes))]):
      state = ops.large_conv_layer(state, lay['din'], Lb, Vb, V, lay['Wb'], lay['Wt'], G[t], lay['bias'], work)
    e[2].record()
    torch.cuda.synchronize()
  Nk, Cd = Lb.dims[1], Lb.shape[2]
  layer_bytes = B * (Cd * N * Nk * 2 + N * 64 * 2 + Cd * 128 * Nk * 2 + N * 128 * 4 * 2)
  return {'operators_streamed': Cd, 'pack_ms': round(e[0].elapsed_time(e[1]), 3),
          'pack_GBps': round(B * (N * N * 2 * 4 + Cd * N * Nk * 2) / e[0].elapsed_time(e[1]) / 1e6, 1),
          'layers_ms': round(e[1].elapsed_time(e[2]), 3),
          'layer_stream_GBps': round(7 * layer_bytes / e[1].elapsed_time(e[2]) / 1e6, 1)}
if not args.dense_stream:
  e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  ops.lanczos_ritz_kstep(A0, None, K, K, workspace=ws)
  e[0].record()
  ops.lanczos_ritz_kstep(A0, None, K, K, workspace=ws)
  e[1].record()
  torch.cuda.synchronize()
  res['lanczos_on_a_contiguous_copy_ms'] = round(e[0].elapsed_time(e[1]), 2)
res['bf16_stages_folded'] = stages((0, 0))
res['bf16_stages_every_channel'] = stages((0, 1))
print(json.dumps({'workload': 'LanczosNetGeneral N=%d K=%d batch=%d' % (N, K, B),
                  'lanczos': ('lnz_lanczos_ritz_large' + ('_sym' if SYM else '')) if args.dense_stream else
                             'ops.lanczos_ritz_collated: lnz_lanczos_ritz_kstep_image on L[..., 0] of the collated [B,N,N,2] tensor in place; the pass leaves the conv image of the bf16 mode behind', **res}))
