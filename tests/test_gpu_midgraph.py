"""lnz_midgraph_forward (csrc/conv_mid.hip): every conv layer, the head and the gated masked mean of
graphs with 33..128 nodes in ONE launch — a graph on four workgroups (32 output columns each) that
exchange the layer's state through global memory.  The reference's own graph configuration
(config/graph_lanczos_net.yaml against the unmodified reference's run) is tests/test_gpu_graph.py,
which reaches this kernel through the module; here: the oracle (numpy restatement of
model/lanczos_net_general.py:127-201 / model/lanczos_net.py:125-199, float64) on other shapes,
channel folding on / off, more graphs than the chip holds at once, and the hand-off itself —
repeated launches under a concurrent memory stream must give the same bits."""
import numpy as np
import pytest
import torch

import oracle
from graph_fixture import GRAPH_CFG

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _graphs(rs, B, N, nmin, p, edge_types=1):
  ns = rs.randint(nmin, N + 1, size=B)
  ns[rs.randint(B)] = N
  adj = np.zeros((B, N, N, edge_types), np.float32)
  for b in range(B):
    n = int(ns[b])
    a = np.triu((rs.rand(n, n) < p).astype(np.float32), 1)
    kind = rs.randint(edge_types, size=(n, n))
    for e in range(edge_types):
      ae = a * (kind == e)
      adj[b, :n, :n, e] = ae + ae.T
  mask = (np.arange(N)[None, :] < ns[:, None]).astype(np.uint8)
  return ns.astype(np.int32), adj, mask


def _net(cfg, seed, general=True, name='LanczosNetGeneral'):
  from lanczosnet_amd import model
  from lanczosnet_amd.utils.arg_helper import make_model_config
  P = oracle.make_lanczosnet_params(cfg, seed, general=general)
  net = getattr(model, name)(make_model_config(cfg, general=general)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return net.to(DEV), P


def _per_graph(got, ref):
  return float((np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())


@pytest.mark.parametrize('B,N,nmin,edge_types,K,long_dist,din', [
    (64, 100, 20, 1, 20, [1, 2, 3, 5, 7, 10, 20, 30], 10),   # the reference's configuration
    (5, 33, 33, 1, 20, [1, 2, 3, 5, 7, 10, 20, 30], 10),     # just beyond the 32-node tile
    (70, 128, 90, 1, 32, [2, 5], 10),                        # full tile, K = 32, more workgroups than CUs
    (9, 64, 40, 1, 20, [], 16),                              # no long scales at all
    (12, 48, 34, 1, 12, [1, 3, 7], 128),                     # input width 128
])
def test_midgraph_forward_matches_the_oracle_and_the_streamed_kernels(B, N, nmin, edge_types, K, long_dist, din):
  from lanczosnet_amd import ops
  rs = np.random.RandomState(B + N)
  cfg = dict(GRAPH_CFG, num_bond_type=edge_types, num_eig_vec=K, long_diffusion_dist=long_dist, input_dim=din)
  net, P = _net(cfg, 3)
  ns, adj, mask = _graphs(rs, B, N, nmin, 0.3, edge_types)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, K)
  X = rs.randn(B, N, din).astype(np.float32) * mask[:, :, None]
  assert net._mid_hip_supported(N, K, L.shape[3])
  with torch.no_grad():
    s_mid = net(_t(X), L, D, V, mask=_t(mask))
    net.mid_graph_kernel = False
    s_large = net(_t(X), L, D, V, mask=_t(mask))
  ref = oracle.lanczos_net_forward(P, cfg, X, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(), mask,
                                   dtype=np.float64, general=True)
  e_mid, e_large = _per_graph(s_mid.cpu().numpy(), ref), _per_graph(s_large.cpu().numpy(), ref)
  print('midgraph B=%d N=%d K=%d S=%d: per-graph rel err vs float64 oracle %.2e (streamed kernels %.2e)'
        % (B, N, K, len(long_dist), e_mid, e_large))
  assert e_mid < 1e-5


def test_midgraph_channel_folding_is_decided_per_graph():
  """With one edge type the collated L carries the simple graph's operator twice: the kernel
  compares the fragments it holds and runs ONE edge pass with the summed weight blocks.  A batch in
  which some graphs have a second, different channel must give every graph its own answer."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(4)
  cfg = dict(GRAPH_CFG)
  net, P = _net(cfg, 5)
  B, N = 16, 80
  ns, adj, mask = _graphs(rs, B, N, 40, 0.2)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n).clone()
  assert torch.equal(L[..., 0], L[..., 1])
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 20)
  # every other graph: channel 1 becomes a different operator (a sparser graph's Laplacian)
  ns2, adj2, _ = _graphs(np.random.RandomState(5), B, N, 40, 0.05)
  L2 = ops.laplacian_l4(_t(adj2), n)
  L[::2, :, :, 1] = L2[::2, :, :, 0] * _t(mask[::2, :, None] * mask[::2, None, :]).float()
  X = rs.randn(B, N, 10).astype(np.float32) * mask[:, :, None]
  with torch.no_grad():
    got = net(_t(X), L, D, V, mask=_t(mask)).cpu().numpy()
  ref = oracle.lanczos_net_forward(P, cfg, X, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(), mask,
                                   dtype=np.float64, general=True)
  assert _per_graph(got, ref) < 1e-5


def test_midgraph_embedding_model_beyond_the_tile():
  """LanczosNet (atom embedding, several edge types collapse to <= 2 channels only with one bond
  type) on 40..60-node graphs: the embedding rows are the layer-0 state."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(11)
  cfg = dict(oracle.DEFAULT_QM8_CFG, num_bond_type=1, hidden_dim=[128] * 3, num_layer=3)
  net, P = _net(cfg, 2, general=False, name='LanczosNet')
  B, N = 20, 60
  ns, adj, mask = _graphs(rs, B, N, 40, 0.1)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, cfg['num_eig_vec'])
  ids = rs.randint(0, cfg['num_atom'], size=(B, N)).astype(np.int64)
  assert net._mid_hip_supported(N, cfg['num_eig_vec'], L.shape[3])
  with torch.no_grad():
    got = net(_t(ids), L, D, V, mask=_t(mask)).cpu().numpy()
  ref = oracle.lanczos_net_forward(P, cfg, ids, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(), mask,
                                   dtype=np.float64)
  assert _per_graph(got, ref) < 1e-5


def test_midgraph_hand_off_is_stable_under_load():
  """The inter-workgroup hand-off (write-through stores, relaxed counter, sc1 loads; no fences):
  200 launches, alternating two inputs so that every exchange region is rewritten with different
  values each time, while a second stream streams 1 GiB through the memory system — every launch
  must reproduce its input's first result bit for bit (a stale line, a torn slice or a missed
  counter shows up as a different score)."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(21)
  cfg = dict(GRAPH_CFG)
  net, _ = _net(cfg, 7)
  B, N = 64, 100
  ns, adj, mask = _graphs(rs, B, N, 20, 0.5)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 20)
  mk = _t(mask)
  Xs = [_t(rs.randn(B, N, 10).astype(np.float32)) for _ in range(2)]
  with torch.no_grad():
    want = [net(x, L, D, V, mask=mk).clone() for x in Xs]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty((256 << 20,), dtype=torch.float32, device=DEV)
    bad = 0
    for it in range(200):
      if it % 10 == 0:
        with torch.cuda.stream(side):
          big.add_(1.0)          # 2 GiB of traffic per call, concurrently
      got = net(Xs[it & 1], L, D, V, mask=mk)
      bad += int(not torch.equal(got, want[it & 1]))
    torch.cuda.synchronize()
  assert bad == 0, bad


def test_midgraph_batch_of_1024_graphs_goes_out_in_resident_chunks():
  """Forward progress: the four workgroups of a graph wait for each other, so a launch may hold
  only as many blocks as the device keeps resident (lnz_midgraph_forward sizes its launches by the
  occupancy query).  1024 graphs = 4096 workgroups, sixteen times the chip: every graph's score
  must equal the score the same graph gets in a batch of 64 (same kernel, same arithmetic: bitwise)."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(31)
  cfg = dict(GRAPH_CFG, hidden_dim=[128] * 3, num_layer=3)
  net, _ = _net(cfg, 9)
  B, N = 1024, 48
  ns, adj, mask = _graphs(rs, B, N, 33, 0.2)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 20)
  X = _t(rs.randn(B, N, 10).astype(np.float32) * mask[:, :, None])
  mk = _t(mask)
  with torch.no_grad():
    big = net(X, L, D, V, mask=mk)
    torch.cuda.synchronize()
    assert torch.isfinite(big).all()
    for lo in (0, 448, 960):
      part = net(X[lo:lo + 64], L[lo:lo + 64], D[lo:lo + 64], V[lo:lo + 64], mask=mk[lo:lo + 64])
      assert torch.equal(part, big[lo:lo + 64]), lo


def test_midgraph_fenced_exchange_gives_the_same_bits(monkeypatch):
  """The exchange is fence free only when the hardware says the four workgroups of a graph share an
  L2 (HW_REG_XCC_ID, posted in the placement word); otherwise the publisher releases and the consumer
  acquires at agent scope.  LNZ_MID_FENCED=1 takes that path on this box too: same scores, bit for
  bit, and the placement words show what the check saw — four arrivals, four equal XCD ids."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(41)
  cfg = dict(GRAPH_CFG)
  net, _ = _net(cfg, 7)
  B, N = 64, 100
  ns, adj, mask = _graphs(rs, B, N, 20, 0.5)
  n = _t(ns)
  L = ops.laplacian_l4(_t(adj), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, 20)
  X, mk = _t(rs.randn(B, N, 10).astype(np.float32)), _t(mask)
  with torch.no_grad():
    free = net(X, L, D, V, mask=mk).clone()
    place = ops.midgraph_forward.last_sync[B * 7:B * 8].cpu().numpy()
    monkeypatch.setenv('LNZ_MID_FENCED', '1')
    fenced = [net(X, L, D, V, mask=mk).clone() for _ in range(20)]
  assert all(torch.equal(f, free) for f in fenced)
  assert ((place & 0xff) == 4).all()
  ids = np.stack([(place >> (8 + 4 * q)) & 15 for q in range(4)], axis=1)
  assert (ids >= 1).all() and (ids <= 8).all()
  same = (ids == ids[:, :1]).all(axis=1)
  print('graphs whose four workgroups shared an XCD: %d of %d' % (same.sum(), B))
  # (observed on MI355X in SPX mode: all of them — block b runs on XCD b % 8; HIP does not promise
  # it, which is why the kernel checks: a graph outside `same` simply took the fenced exchange)
