"""Diagnostic: per-wave clock64 phase times of the fused forward (needs tools/libprobe_phases.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import oracle
from lanczosnet_amd import ops, _lib
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
P = oracle.make_lanczosnet_params(cfg, 1)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}); net = net.cuda()
net.gemm_mode = os.environ.get('PROBE_GEMM', 'fp32')
for B in (int(os.environ.get('PROBE_B', '1024')),):
  b = draw_batch(B, seed=0)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
  n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  plan = net._plan(); Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], 7, plan['mlp_pack'])
  # state buffer with room for the phase record behind it
  import lanczosnet_amd.ops as o
  orig_empty = torch.zeros
  def big_empty(shape, **kw):
    if tuple(shape) == (B, 32, 128):
      buf = torch.zeros((B * 32 * 128 + 1024,), **kw); big_empty.buf = buf
      return buf[:B * 32 * 128].view(B, 32, 128)
    return orig_empty(shape, **kw)
  torch.zeros = big_empty
  tiling = 'auto'
  if os.environ.get('PROBE_SHAPE'):   # e.g. "2,0": tiles per half, singles, 256 workgroups
    shape = tuple(int(x) for x in os.environ['PROBE_SHAPE'].split(','))
    W = B // sum(shape)
    buf = np.full((12 * W + 1,), -1, np.int32)
    e = buf[:12 * W].reshape(W, 4, 3); e[:, :, 2] = 32
    mol = 0
    for w in range(W):
      for h in range(2):
        for m in range(shape[h]):
          e[w, 2 * h + m, 0] = mol; mol += 1
    buf[12 * W] = W
    tiling = (torch.from_numpy(buf).cuda(), W)
  for _ in range(3):
    o.lanczosnet_forward(plan, t(b['node_feat']), Lp, V, G, t(b['node_mask']), return_state=True, tiling=tiling)
  torch.cuda.synchronize(); torch.zeros = orig_empty
  rec = big_empty.buf[B * 32 * 128:B * 32 * 128 + 512].cpu().numpy().reshape(8, 8, 8)
  print('B=%d  [block, wave] -> gemm1, gemm2+M, epilogue, total, projection, mid barrier (kcycles), tiles' % B)
  for blk in (0, 1):
    for w in range(8):
      print('   ', blk, w, ' '.join('%8.1f' % (x / 1e3) for x in rec[blk, w]))
