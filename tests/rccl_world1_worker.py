"""Worker of tests/test_gpu_rccl_world1.py: ONE rank, backend `nccl` (= RCCL on ROCm), launched
through torch.distributed.run.  Sends the HIP forward's scores through `AsyncScoreGather` and the
HIP backward's gradients through `all_reduce_gradients` on a real RCCL communicator (a one-rank
group is a legal communicator) and checks that both come back unchanged."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (parameter draw only)
from lanczosnet_amd import dist as lnz_dist, ops  # noqa: E402
from lanczosnet_amd.model import LanczosNet  # noqa: E402
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402
from lanczosnet_amd.utils.arg_helper import make_model_config  # noqa: E402


def main():
  dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
  torch.cuda.set_device(dev)
  dist.init_process_group('nccl', device_id=dev)
  assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  P = oracle.make_lanczosnet_params(cfg, 11)
  net = LanczosNet(make_model_config(cfg)).to(dev)
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  b = draw_batch(64, seed=4)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, cfg['num_eig_vec'])
  net.eval()
  with torch.no_grad():
    score = net(t(b['node_feat']), L, D, V, mask=t(b['node_mask']))
  # 1. the per-step score exchange on RCCL
  g = lnz_dist.AsyncScoreGather(64, cfg['output_dim'], dev, force_collective=True)
  assert g.collective
  tickets = [g.submit(score * (k + 1)) for k in range(3)]
  g.drain()
  assert torch.equal(g.result(tickets[-1]), score * 3)
  assert torch.equal(g.result(tickets[-2]), score * 2)
  # 2. the synchronous gather and the global loss (one-rank short cuts) agree with it
  assert torch.equal(lnz_dist.all_gather_scores(score, 64), score)
  lab = t(b['label'])
  assert abs(float(lnz_dist.global_mse(score, lab)) - float(torch.mean((score - lab) ** 2))) < 1e-6
  # 3. the gradient exchange on RCCL: flat buckets, count-weighted, back into .grad
  net.train()
  _, loss = net(t(b['node_feat']), L, D, V, label=lab, mask=t(b['node_mask']))
  loss.backward()
  params = [p for p in net.parameters() if p.requires_grad]
  before = [p.grad.clone() for p in params]
  lnz_dist.all_reduce_gradients(params, 64, bucket_bytes=1 << 20, force_collective=True)
  torch.cuda.synchronize()
  for p, g0 in zip(params, before):
    assert torch.equal(p.grad, g0)   # x 64 / 64 is exact in fp32
  maps = open('/proc/self/maps').read()
  assert 'librccl' in maps, 'RCCL is not mapped into this process'
  dist.barrier()
  dist.destroy_process_group()
  print('RCCL_WORLD1_OK buckets=%d' % ((sum(p.numel() for p in params) * 4 + (1 << 20) - 1) >> 20))


if __name__ == '__main__':
  main()
