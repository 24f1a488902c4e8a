"""Diagnostic: per-wave clock64 phase times of the 16 x 16-tile forward (conv_forward16.hip built
with -DLNZ_F16_PHASES: tools/experiments/build_variant.sh conv_forward16.hip phases:"-DLNZ_F16_PHASES",
run with LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_conv_forward16_phases.so LNZ_FORWARD16=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
P = oracle.make_lanczosnet_params(cfg, 1)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}); net = net.cuda()
net.gemm_mode = os.environ.get('PROBE_GEMM', 'fp32')   # 'f16x3': the split-precision strip kernel
B = int(os.environ.get('PROBE_B', '1024'))
b = draw_batch(B, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
plan = net._plan(); Lp = ops.pack_laplacian_for(plan, L)
G = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
orig = torch.zeros
def big(shape, **kw):
  if tuple(shape) == (B, 32, 128):
    buf = orig((B * 32 * 128 + 2048,), **kw); big.buf = buf
    return buf[:B * 32 * 128].view(B, 32, 128)
  return orig(shape, **kw)
torch.zeros = big
for _ in range(3):
  ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, V, G, t(b['node_mask']), return_state=True)
torch.cuda.synchronize(); torch.zeros = orig
rec = big.buf[B * 32 * 128:B * 32 * 128 + 1024].cpu().numpy().reshape(8, 8, 16)
names = (os.environ.get('PROBE_NAMES') or 'layer-head gemm1-long gain-scale lift gemm1-edge gemm2 epilogue prologue total tiles').split()
print('B=%d  [block, wave] kcycles: %s' % (B, ' '.join(names)))
for blk in range(2):
  for w in (0, 3, 7):
    print('   ', blk, w, ' '.join('%8.1f' % (x / 1e3) for x in rec[blk, w][:10]))
# the strip plan of this batch: subtiles per strip (a launch lasts as long as its longest strip)
if os.environ.get('PROBE_STRIPS'):
  st = ops.plan_strips(t(b['node_mask']).to(torch.uint8).contiguous()).cpu().numpy()
  nst = int(st[-1])
  ent = st[:-1].reshape(-1, ops.STRIP_INTS)[:nst]
  print('strips', nst, 'subtiles histogram', np.bincount(ent[:, 1], minlength=7).tolist(),
        'first 8 strips (molecules, subtiles):', ent[:8, :2].tolist())
