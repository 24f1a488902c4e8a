import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda(); plan = net._plan()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mask = t(b['node_mask']).contiguous()
orig = torch.zeros
def big(shape, **kw):
  if kw.get('dtype') == torch.int32 and tuple(shape) == (1026,):
    return orig((1026 + 2 * 4480 + 16,), **kw)
  return orig(shape, **kw)
torch.zeros = big
for _ in range(3):
  out = ops.prepare_batch(plan, L, mask, n, 20, gains=(cfg['long_diffusion_dist'], 7, plan['mlp_pack']))
torch.cuda.synchronize()
sync = out[-1].cpu().numpy().astype(np.int64)
done = sync[1:1025]; tl = sync[1026:1026 + 2 * 4480].reshape(-1, 2)
used = tl[:, 1] != 0
t0 = min(done.min(), tl[used, 0].min())
nn = b['n_nodes']
for lo, hi in ((1, 8), (9, 12), (13, 16), (17, 20), (21, 23), (24, 26)):
  m = (nn >= lo) & (nn <= hi)
  print('ritz done n=%d..%d: mean %.1f us max %.1f us' % (lo, hi, (done[m] - t0).mean() / 100, (done[m] - t0).max() / 100))
st = (tl[used, 0] - t0) / 100.0; en = (tl[used, 1] - t0) / 100.0
print('consumers: n=%d first start %.1f last end %.1f mean dur %.1f us' % (used.sum(), st.min(), en.max(), (en - st).mean()))
for q in (10, 25, 50, 75, 90, 100):
  print(' pct %3d: start %.1f end %.1f' % (q, np.percentile(st, q), np.percentile(en, q)))
