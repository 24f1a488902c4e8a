"""prepare_batch launch time on the bench batch.  usage: bench_prep.py [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
out = ops.prepare_batch(plan, L, mk, n, 20); torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
for _ in range(reps): out = ops.prepare_batch(plan, L, mk, n, 20)
e[1].record(); torch.cuda.synchronize()
D, V = ops.lanczos_ritz(L[..., 0], n, 20)
print(json.dumps({'prepare_batch_ms': round(e[0].elapsed_time(e[1]) / reps, 4),
                  'D_equal': bool(torch.equal(out[3], D)), 'V_equal': bool(torch.equal(out[4], V)),
                  'Lp_equal': bool(torch.equal(out[0], ops.pack_laplacian(L)))}))
