"""Device-side collate timing: packed shard -> padded batch -> scores.
Prints one JSON line.  Run on the GPU box: python tools/bench_collate.py"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
from lanczosnet_amd import ops
from lanczosnet_amd.dataset import PackedQM8, write_packed
from lanczosnet_amd.dataset.qm8 import collate_adjacency
from lanczosnet_amd.model import LanczosNet
from lanczosnet_amd.synthetic import draw_batch
from lanczosnet_amd.utils.arg_helper import make_model_config

B, NMOL = 1024, 21786  # QM8 size
rs = np.random.RandomState(0)
pool = draw_batch(4096, seed=0)
mols = []
for i in range(NMOL):
  b = i % 4096
  n = int(pool['n_nodes'][b])
  mols.append(dict(node_feat=pool['node_feat'][b, :n], adjs=pool['adjs'][b, :n, :n], label=pool['label'][b]))
path = os.path.join(tempfile.mkdtemp(), 'qm8.lnzq')
t0 = time.perf_counter()
write_packed(path, mols, 6, 16)
t_write = time.perf_counter() - t0
ds = PackedQM8(path).to('cuda')
cfg = dict(oracle.DEFAULT_QM8_CFG)
net = LanczosNet(make_model_config(cfg)).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
net = net.cuda()
perm = rs.permutation(NMOL)
nb = NMOL // B


def epoch(with_model):
  for i in range(nb):
    out = ds.collate(perm[i * B:(i + 1) * B], 20)
    if with_model:
      with torch.no_grad():
        net(out['node_feat'], out['L'], out['D'], out['V'], mask=out['node_mask'])
  torch.cuda.synchronize()


epoch(True)
t0 = time.perf_counter(); epoch(False); t_collate = time.perf_counter() - t0
t0 = time.perf_counter(); epoch(True); t_full = time.perf_counter() - t0
# kernel-only time of lnz_collate_qm8
ids = torch.from_numpy(perm[:B]).cuda()
N = int(ds.sizes[perm[:B]].max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
  ops.collate_qm8(ds._dev, ids, N, 6, 16)
e0.record()
for _ in range(20):
  o = ops.collate_qm8(ds._dev, ids, N, 6, 16)
e1.record(); torch.cuda.synchronize()
k_ms = e0.elapsed_time(e1) / 20
out_bytes = o['L'].numel() * 4 + B * N * 9 + B * 68
# host-side padding path for comparison (python loop + 29 MB H2D per batch)
items = mols[:B]
t0 = time.perf_counter(); collate_adjacency(items, 20, 'cuda'); torch.cuda.synchronize()
t_host = time.perf_counter() - t0
print(json.dumps({
    'workload': 'packed QM8-size shard (%d molecules, %d bytes) -> %d batches of %d' % (NMOL, os.path.getsize(path), nb, B),
    'pack_file_s': round(t_write, 2),
    'collate_kernel_ms': round(k_ms, 4), 'collate_kernel_GBps_written': round(out_bytes / k_ms / 1e6, 1),
    'epoch_collate_only_s': round(t_collate, 4), 'epoch_collate_plus_forward_s': round(t_full, 4),
    'molecules_per_s_end_to_end': round(nb * B / t_full, 1),
    'host_padding_collate_one_batch_s': round(t_host, 4)}))
