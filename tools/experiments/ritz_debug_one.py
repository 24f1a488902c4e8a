import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lanczosnet_amd import ops
from lanczosnet_amd.synthetic import draw_batch
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
seed, mol = int(sys.argv[1]), int(sys.argv[2])
b = draw_batch(1024, seed=seed, n_min=1, n_max=32, N=32)
n = t(b['n_nodes']); L = ops.laplacian_l4(t(b['adjs']), n)
A = L[..., 0].contiguous()
B, N, K = 1024, 32, 20
D = torch.empty((B, K), device='cuda'); V = torch.empty((B, N, K), device='cuda')
info = torch.zeros((B + 32 * 16,), dtype=torch.int32, device='cuda')
ops._abi().lanczos_ritz(A, A.stride(0), A.stride(1), A.stride(2), n, B, N, K, D, V, info)
torch.cuda.synchronize()
rec = info[B:].view(torch.float32).view(32, 16).cpu().numpy()
nb = int(b['n_nodes'][mol])
print('lane lam(+1e-9 residual) s t member ws wt tw width/1e-12gsc clo chi bail   restarts', int(info[mol]))
for r in range(nb):
  print(r, '%.9f %+.3f' % (rec[r, 0], rec[r, 1]), rec[r, 2:12].astype(int).tolist())
np.save(os.environ.get('DBG_OUT', '/tmp/ritz_dbg.npy'), rec)
Vd = V[mol].double(); G = (Vd.T @ Vd).cpu().numpy()
Go = np.abs(G - np.diag(np.diag(G))); print('worst offdiag', np.unravel_index(np.argmax(Go), Go.shape), Go.max())
