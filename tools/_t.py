import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
b = draw_batch(1024, seed=0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
for _ in range(5): ops.lanczos_ritz(L[..., 0], n, 20)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.lanczos_ritz(L[..., 0], n, 20)
e1.record(); torch.cuda.synchronize()
print('ritz32 standalone ms', e0.elapsed_time(e1) / 50)
