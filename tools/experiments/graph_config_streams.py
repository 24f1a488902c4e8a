#!/usr/bin/env python
"""The reference's graph configuration (64 graphs, n ~ U{20..100}) as a stream of batches: L4 + Ritz
pairs of batch k+1 beside the forward of batch k on two HIP streams — eager launches, and the two
halves captured in HIP graphs (the eager form is bound by the host's launch rate)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanczosnet_amd import ops
from lanczosnet_amd.model import LanczosNetGeneral
from lanczosnet_amd.utils.arg_helper import make_model_config

dev = torch.device('cuda:0')
B, K = 64, 20
cfg = dict(num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
           num_eig_vec=K, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7,
           output_dim=2, num_layer=7, num_atom=0)
rs = np.random.RandomState(123)
ns = rs.randint(20, 101, size=B).astype(np.int32)
N = int(ns.max())
adjs = np.zeros((B, N, N, 1), np.float32)
for b in range(B):
  a = np.triu((rs.rand(ns[b], ns[b]) < 0.5).astype(np.float32), 1)
  adjs[b, :ns[b], :ns[b], 0] = a + a.T
X = rs.randn(B, N, 10).astype(np.float32)
mask = (np.arange(N)[None, :] < ns[:, None]).astype(np.uint8)
torch.manual_seed(1234)
net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval().to(dev)
t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
ad, nd, Xd, md = t(adjs), t(ns), t(X), t(mask)
res = {}
import gc
gc.collect()
gc.disable()   # (wall-clock windows below: no collection pause inside)
with torch.no_grad():
  def step():
    L = ops.laplacian_l4(ad, nd)
    D, V = ops.lanczos_ritz(L[:, :, :, 0], nd, K)
    return net(Xd, L, D, V, mask=md)
  ref = step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(40):
    step()
  torch.cuda.synchronize()
  res['sequential_eager_ms'] = (time.perf_counter() - t0) / 40 * 1e3
  # whole step as one graph
  g = torch.cuda.CUDAGraph()
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
      out_g = step()
  torch.cuda.current_stream().wait_stream(s)
  g.replay()
  torch.cuda.synchronize()
  res['one_graph_equal'] = bool(torch.equal(out_g, ref))
  t0 = time.perf_counter()
  for _ in range(40):
    g.replay()
  torch.cuda.synchronize()
  res['sequential_graph_ms'] = (time.perf_counter() - t0) / 40 * 1e3
  # two halves, two slots, two streams — optionally on disjoint sets of compute units
  # (hipExtStreamCreateWithCUMask: the Ritz launch is a latency chain per workgroup, and forward
  # waves sharing its compute units stretch it from 0.45 to 0.60 ms)
  from lanczosnet_amd.utils.streams import cu_masked_stream as masked_stream
  split = int(os.environ.get('LNZ_GRAPH_STREAMS_CU_SPLIT', '0'))
  if split:
    s_prep, s_fwd = masked_stream(0, split), masked_stream(split, 256)
  else:
    s_prep, s_fwd = torch.cuda.Stream(), torch.cuda.Stream()
  res['cu_split'] = split
  slots = []
  for i in range(2):
    sl = {}
    gp, gf = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_prep):
      with torch.cuda.graph(gp, stream=s_prep):
        sl['L'] = ops.laplacian_l4(ad, nd)
        sl['D'], sl['V'] = ops.lanczos_ritz(sl['L'][:, :, :, 0], nd, K)
    with torch.cuda.stream(s_fwd):
      with torch.cuda.graph(gf, stream=s_fwd):
        sl['score'] = net(Xd, sl['L'], sl['D'], sl['V'], mask=md)
    sl['gp'], sl['gf'] = gp, gf
    slots.append(sl)
  torch.cuda.synchronize()

  def prep(sl):
    with torch.cuda.stream(s_prep):
      if 'done' in sl:
        s_prep.wait_event(sl['done'])
      sl['gp'].replay()
      sl['ready'] = torch.cuda.Event()
      sl['ready'].record(s_prep)

  def fwd(sl):
    with torch.cuda.stream(s_fwd):
      s_fwd.wait_event(sl['ready'])
      sl['gf'].replay()
      sl['done'] = torch.cuda.Event()
      sl['done'].record(s_fwd)
  nb = 100
  for warm in (True, False):
    prep(slots[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nb):
      prep(slots[(k + 1) & 1])
      fwd(slots[k & 1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nb * 1e3
  res['two_streams_graphs_ms_per_batch'] = dt
  res['two_streams_equal'] = bool(torch.equal(slots[(nb - 1) & 1]['score'], ref))
  # the Ritz launch is one latency chain per graph: 256 graphs take as long as 64 (0.46 ms) — a
  # loader that collates four batches ahead shares ONE Laplacian + Ritz launch among them
  ad4, nd4 = torch.cat([ad] * 4), torch.cat([nd] * 4)
  g4 = torch.cuda.CUDAGraph()
  s4 = torch.cuda.Stream()
  s4.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s4):
    def four():
      L4 = ops.laplacian_l4(ad4, nd4)
      D4, V4 = ops.lanczos_ritz(L4[:, :, :, 0], nd4, K)
      return [net(Xd, L4[B * i:B * (i + 1)], D4[B * i:B * (i + 1)], V4[B * i:B * (i + 1)], mask=md) for i in range(4)]
    four()
    torch.cuda.synchronize()
    with torch.cuda.graph(g4, stream=s4):
      outs = four()
  torch.cuda.current_stream().wait_stream(s4)
  g4.replay()
  torch.cuda.synchronize()
  res['four_batches_one_ritz_launch_equal'] = bool(all(torch.equal(o, ref) for o in outs))
  t0 = time.perf_counter()
  for _ in range(25):
    g4.replay()
  torch.cuda.synchronize()
  res['four_batches_one_ritz_launch_ms_per_batch'] = (time.perf_counter() - t0) / 100 * 1e3
print(json.dumps(res))
