"""The batch-preparation launch against its parts on the bench batch: the fused launch, the Ritz
kernel alone, pack + plan alone (event times over 100 launches each)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval().cuda(); plan = net._plan()
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n); mk = t(b['node_mask'])
def timed(f, reps=100):
  f(); torch.cuda.synchronize()
  e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  e[0].record()
  for _ in range(reps): f()
  e[1].record(); torch.cuda.synchronize()
  return round(e[0].elapsed_time(e[1]) / reps, 4)
A = L[..., 0]
print(json.dumps({'prepare_batch': timed(lambda: ops.prepare_batch(plan, L, mk, n, 20)),
                  'lanczos_ritz': timed(lambda: ops.lanczos_ritz(A, n, 20)),
                  'pack_and_plan': timed(lambda: ops.pack_and_plan(plan, L, mk, 20)),
                  'plan_strips': timed(lambda: ops.plan_strips(mk)),
                  'plan_batch': timed(lambda: ops.plan_batch(mk, True, 20))}))

# the Ritz kernel and pack + plan as two launches on two streams (fork / join by events)
s2 = torch.cuda.Stream()
def two_streams():
  cur = torch.cuda.current_stream()
  s2.wait_stream(cur)
  with torch.cuda.stream(s2):
    ops.pack_and_plan(plan, L, mk, 20)
  ops.lanczos_ritz(A, n, 20)
  cur.wait_stream(s2)
print(json.dumps({'two_streams': timed(two_streams)}))

# plan + Ritz pairs without the pack workgroups (lnz_prepare_batch with Lp = NULL)
from lanczosnet_amd import ops as _o
nn32 = n.to(torch.int32).contiguous()
print(json.dumps({'plan_ritz_no_pack': timed(lambda: _o._ext().plan_ritz(L, mk, nn32, 20, 256, True))}))
