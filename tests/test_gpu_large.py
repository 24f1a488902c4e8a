"""GPU parity of the large-graph M-step Lanczos (BASELINE config 5 regime) against its fp64
restatement and against the reference's ARPACK call for the converged leading pairs."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


from conftest import load_golden  # noqa: E402
from large_fixture import general_inputs, graphs as _graphs, kstep_ritz  # noqa: E402


@pytest.mark.parametrize('sym', [False, True])
@pytest.mark.parametrize('N,M,B', [(256, 32, 3), (1000, 48, 2), (2048, 64, 2), (1412, 40, 2)])
def test_large_lanczos_matches_fp64_restatement(N, M, B, sym):
  from lanczosnet_amd import ops
  from scipy.sparse.linalg import eigsh
  A = _graphs(B, N, 8.0 / N, seed=N)
  D, V, info = ops.lanczos_ritz_large(torch.from_numpy(A).to(DEV), M, M, return_info=True,
                                      symmetric=sym)
  D, V = D.cpu().numpy(), V.cpu().numpy().astype(np.float64)
  assert (info.cpu().numpy() == M).all()
  for b in range(B):
    Dr, Vr, (al, be, steps, beta_last) = oracle.lanczos_kstep_fp64(A[b], M, M)
    assert np.abs(D[b] - Dr).max() < 1e-6
    # same Krylov subspace (projector is basis / sign independent)
    Pg, Pr = V[b] @ V[b].T, Vr @ Vr.T
    assert np.abs(Pg - Pr).max() < 1e-5
    assert np.abs(V[b].T @ V[b] - np.eye(M)).max() < 1e-5
    # Lanczos residual identity: |A v_k - theta_k v_k| <= beta_M (bounded by |A| <= 1)
    A64 = A[b].astype(np.float64)
    res = np.linalg.norm(A64 @ V[b] - V[b] * D[b].astype(np.float64), axis=0)
    res_ref = np.linalg.norm(A64 @ Vr - Vr * Dr, axis=0)
    assert np.abs(res - res_ref).max() < 1e-4
    # the reference's Lanczos branch (utils/data_helper.py:205-208): converged pairs agree
    e, _ = eigsh(A64, k=2, which='LM')
    lead = np.sort(np.abs(e))[::-1]
    conv = res[:2] < 1e-6
    assert conv[0] and abs(abs(D[b][0]) - lead[0]) < 1e-6


def test_symmetric_stream_reads_only_the_upper_chunk_blocks():
  """lnz_lanczos_ritz_large_sym never touches the 256 x 256 blocks below the diagonal chunk
  blocks: poisoning them changes nothing; the results agree with the full stream."""
  from lanczosnet_amd import ops
  N, M, B = 1024, 32, 2
  A = _graphs(B, N, 8.0 / N, seed=5)
  D0, V0 = ops.lanczos_ritz_large(torch.from_numpy(A).to(DEV), M, M)
  Ap = A.copy()
  for I in range(N // 256):
    Ap[:, 256 * I:256 * (I + 1), :256 * I] = np.nan
  D1, V1 = ops.lanczos_ritz_large(torch.from_numpy(Ap).to(DEV), M, M, symmetric=True)
  D2, V2 = ops.lanczos_ritz_large(torch.from_numpy(A).to(DEV), M, M, symmetric=True)
  assert torch.equal(D1, D2) and torch.equal(V1, V2)          # deterministic, lower blocks unread
  assert (D1 - D0).abs().max() < 1e-6
  P0 = V0.double() @ V0.double().transpose(1, 2)
  P1 = V1.double() @ V1.double().transpose(1, 2)
  assert (P0 - P1).abs().max() < 1e-5


@pytest.mark.parametrize('sym', [False, True])
def test_large_lanczos_strided_view_equals_contiguous(sym):
  """Row and graph strides come through the ABI (the symmetric kernel addresses A through a buffer
  descriptor with 32-bit offsets): a view into a wider, NaN-padded allocation gives the very same
  Ritz pairs as the contiguous copy — and so does a second call (no atomics anywhere)."""
  from lanczosnet_amd import ops
  N, M, B = 772, 24, 3
  A = torch.from_numpy(_graphs(B, N, 8.0 / N, seed=9)).to(DEV)
  wide = torch.full((B + 1, N + 3, N + 20), float('nan'), dtype=torch.float32, device=DEV)
  view = wide[:B, :N, 4:4 + N]
  view.copy_(A)
  assert not view.is_contiguous() and view.stride(1) == N + 20
  D0, V0 = ops.lanczos_ritz_large(A, M, M, symmetric=sym)
  D1, V1 = ops.lanczos_ritz_large(view, M, M, symmetric=sym)
  D2, V2 = ops.lanczos_ritz_large(view, M, M, symmetric=sym)
  assert torch.isfinite(D1).all() and torch.isfinite(V1).all()
  assert torch.equal(D0, D1) and torch.equal(V0, V1)
  assert torch.equal(D1, D2) and torch.equal(V1, V2)


def test_large_lanczos_config5_full_size_properties():
  """BASELINE configs[4] at its full size (B = 256 graphs of N = 2048 nodes, M = K = 64), where the
  fp64 restatement is too slow to be the checker: what must hold for ANY correct M-step Lanczos —
  orthonormal Ritz vectors, Ritz values inside the spectrum's interval [-1, 1] in the canonical
  order, lambda_max = 1 of the L4 Laplacian found with a vanishing residual, every Ritz pair a
  Rayleigh-Ritz pair (V^T A V = diag(D)), and the symmetric stream equal to the full stream."""
  from lanczosnet_amd import ops
  B, N, M = 256, 2048, 64
  g = torch.Generator(device=DEV)
  g.manual_seed(11)
  A = torch.empty((B, N, N), dtype=torch.float32, device=DEV)
  for b in range(B):
    adj = (torch.rand((N, N), generator=g, device=DEV) < 0.01).float().triu(1)
    adj = adj + adj.t() + torch.eye(N, device=DEV)
    d = adj.sum(1).rsqrt()
    A[b] = d[:, None] * adj * d[None, :]
  D, V, info = ops.lanczos_ritz_large(A, M, M, return_info=True, symmetric=True)
  assert (info == M).all() and torch.isfinite(D).all() and torch.isfinite(V).all()
  Vd = V.double()
  eye = torch.eye(M, dtype=torch.float64, device=DEV)
  assert (Vd.transpose(1, 2) @ Vd - eye).abs().max() < 1e-5           # orthonormal
  assert D.abs().max() <= 1.0 + 1e-6
  assert (D.abs()[:, :-1] >= D.abs()[:, 1:] - 1e-7).all()              # descending |theta|
  assert (D[:, 0] - 1.0).abs().max() < 1e-6                            # lambda_max of L4
  AV = torch.bmm(A, V)                                                 # fp32 is enough here
  res0 = (AV[:, :, 0] - V[:, :, 0] * D[:, None, 0]).norm(dim=1)
  assert res0.max() < 1e-5                                             # the converged leading pair
  H = torch.bmm(V.transpose(1, 2), AV).double()                        # Rayleigh-Ritz: V^T A V = diag(D)
  assert (H - torch.diag_embed(D.double())).abs().max() < 2e-5
  del AV, H, Vd
  D0, V0 = ops.lanczos_ritz_large(A, M, M)                             # full stream: same pairs
  assert (D0 - D).abs().max() < 1e-6
  sgn = torch.sign((V0 * V).sum(dim=1, keepdim=True))
  assert (V0 * sgn - V).abs().max() < 1e-4


def test_large_lanczos_early_stop_on_invariant_subspace():
  from lanczosnet_amd import ops
  # block-diagonal graph whose start vector's Krylov space is tiny: A = I (no edges)
  N, M = 256, 16
  A = np.eye(N, dtype=np.float32)[None]
  D, V, info = ops.lanczos_ritz_large(torch.from_numpy(A).to(DEV), M, M, return_info=True)
  assert int(info[0]) == 1
  D = D.cpu().numpy()[0]
  assert abs(D[0] - 1.0) < 1e-6 and (D[1:] == 0).all()
  assert (V.cpu().numpy()[0][:, 1:] == 0).all()


def _projector_gap(Va, Vb):
  Pa = Va.double() @ Va.double().transpose(1, 2)
  Pb = Vb.double() @ Vb.double().transpose(1, 2)
  return float((Pa - Pb).abs().max())


@pytest.mark.parametrize('N,M,B,p', [(512, 32, 3, 0.02), (2048, 64, 2, 0.01), (1412, 40, 2, 0.006), (300, 24, 4, 0.05)])
def test_kstep_entry_compacted_image_matches_the_dense_streams_and_the_restatement(N, M, B, p):
  """lnz_lanczos_ritz_kstep: the steps on the sliced-ELL image of A (read once) give the Ritz pairs
  of the dense streams (full / symmetric) — skipping exact zeros changes no sum, only the order of
  the fp64 additions — and of the fp64 restatement; no graph fell back; two calls agree bitwise."""
  from lanczosnet_amd import ops
  A = _graphs(B, N, p, seed=N + 1)
  Ad = torch.from_numpy(A).to(DEV)
  Dc, Vc, info, fb = ops.lanczos_ritz_kstep(Ad, None, M, M, return_info=True, return_fallback=True)
  Dc2, Vc2 = ops.lanczos_ritz_kstep(Ad, None, M, M)
  assert torch.equal(Dc, Dc2) and torch.equal(Vc, Vc2)
  assert (fb == 0).all() and (info == M).all()
  for sym in (False, True):
    Dd, Vd = ops.lanczos_ritz_kstep(Ad, None, M, M, symmetric=sym, compact=False)
    Dl, Vl = ops.lanczos_ritz_large(Ad, M, M, symmetric=sym)
    assert torch.equal(Dd, Dl) and torch.equal(Vd, Vl)        # the dense modes ARE the large entries
    assert (Dc - Dd).abs().max() < 1e-6
    assert _projector_gap(Vc, Vd) < 1e-5
  for b in range(B):
    Dr, Vr, _ = oracle.lanczos_kstep_fp64(A[b], M, M)
    assert np.abs(Dc[b].cpu().numpy() - Dr).max() < 1e-6
    Vg = Vc[b].cpu().numpy().astype(np.float64)
    assert np.abs(Vg @ Vg.T - Vr @ Vr.T).max() < 1e-5


def test_kstep_entry_ragged_batch_and_unaligned_width():
  """Real node counts below the padded width (dataset/graph_data.py:222-260), a width that is not a
  multiple of 4: every graph's pairs are those of its own n_b x n_b matrix, V is zero on the padding."""
  from lanczosnet_amd import ops
  N, K = 333, 24
  sizes = [333, 250, 201, 16]
  A = np.zeros((len(sizes), N, N), np.float32)
  for b, n in enumerate(sizes):
    A[b, :n, :n] = _graphs(1, n, 6.0 / n, seed=40 + b)[0]
  nn = torch.tensor(sizes, dtype=torch.int32, device=DEV)
  for compact in (True, False):
    D, V, info = ops.lanczos_ritz_kstep(torch.from_numpy(A).to(DEV), nn, K, K, compact=compact,
                                        return_info=True)
    assert V.shape == (len(sizes), N, K)
    for b, n in enumerate(sizes):
      Dr, Vr, (_, _, kk, _) = oracle.lanczos_kstep_fp64(A[b, :n, :n], K, K)
      assert int(info[b]) == kk and kk <= min(K, n)
      assert np.abs(D[b].cpu().numpy() - Dr).max() < 1e-6 and (D[b, kk:] == 0).all()
      Vg = V[b].cpu().numpy().astype(np.float64)
      assert (Vg[n:] == 0).all() and (Vg[:, kk:] == 0).all()
      assert np.abs(Vg[:n] @ Vg[:n].T - Vr @ Vr.T).max() < 1e-5


def test_kstep_entry_dense_rows_fall_back_to_the_stream_in_the_same_call():
  """A graph with a row beyond the image's capacity is computed by the dense stream (flagged), its
  sparse neighbours in the batch by the image; both equal the all-dense call."""
  from lanczosnet_amd import ops
  N, M = 512, 32
  A = _graphs(3, N, 0.01, seed=77)
  A[1] = _graphs(1, N, 0.3, seed=78)[0]           # ~150 entries per row: over the capacity of 64
  hub = np.zeros((N, N))                           # one hub node: ONE long row (and column)
  hub[0, 1:200] = hub[1:200, 0] = 1.0
  A[2] = oracle.laplacian_l4(hub)
  Ad = torch.from_numpy(A).to(DEV)
  D, V, fb = ops.lanczos_ritz_kstep(Ad, None, M, M, return_fallback=True)
  assert fb.cpu().tolist() == [0, 1, 1]
  Dd, Vd = ops.lanczos_ritz_kstep(Ad, None, M, M, compact=False)
  assert torch.equal(D[1:], Dd[1:]) and torch.equal(V[1:], Vd[1:])     # the same kernel ran them
  assert (D[0] - Dd[0]).abs().max() < 1e-6 and _projector_gap(V[:1], Vd[:1]) < 1e-5
  D8, V8, fb8 = ops.lanczos_ritz_kstep(Ad, None, M, M, row_cap=256, return_fallback=True)
  assert fb8.cpu().tolist() == [0, 0, 0]                                # a roomier image holds all three
  assert (D8 - Dd).abs().max() < 1e-6 and _projector_gap(V8, Vd) < 1e-5


def test_kstep_entry_reads_channel_0_of_the_collated_laplacian_in_place():
  """`L[..., 0]` of the channels-last [B,N,N,2] tensor of the collate (column stride 2) goes to the
  image kernel as it lies — no copy, spied on — and gives the pairs of the contiguous matrix (the
  entries of a row are placed in another order: fp64 sums in another order); a batch with a graph
  beyond the row capacity is copied after the look at the fallback flags and equals the dense call;
  an unaligned width still takes the copy."""
  from lanczosnet_amd import ops
  N, M, B = 512, 32, 3
  A = _graphs(B, N, 0.02, seed=5)
  Ad = torch.from_numpy(A).to(DEV)
  L = torch.stack([Ad, Ad * 0.5], dim=3)                     # channel 1 must not matter
  copies = []
  orig = torch.zeros

  def spy(*a, **kw):
    if len(a) and isinstance(a[0], tuple) and len(a[0]) == 3:
      copies.append(a[0])
    return orig(*a, **kw)
  torch.zeros = spy
  try:
    Dv, Vv, info, fb = ops.lanczos_ritz_kstep(L[..., 0], None, M, M, return_info=True, return_fallback=True)
    assert not copies and (fb == 0).all() and (info == M).all()
    Dc, Vc = ops.lanczos_ritz_kstep(Ad, None, M, M)
    assert not copies
    assert (Dv - Dc).abs().max() < 1e-6 and _projector_gap(Vv, Vc) < 1e-5
    nn = torch.tensor([N, 300, 77], dtype=torch.int32, device=DEV)       # ragged, in place
    Ar = Ad.clone()
    for b in range(B):
      Ar[b, int(nn[b]):, :] = 0
      Ar[b, :, int(nn[b]):] = 0
    Lr = torch.stack([Ar, Ar], dim=3)
    Dr, Vr = ops.lanczos_ritz_kstep(Lr[..., 0], nn, M, M)
    Dr2, Vr2 = ops.lanczos_ritz_kstep(Ar, nn, M, M)
    assert not copies and (Dr - Dr2).abs().max() < 1e-6 and _projector_gap(Vr, Vr2) < 1e-5
    L2 = L.clone()
    L2[1, 7, :200, 0] = 0.01                                   # a row beyond the capacity of 64
    L2[1, :200, 7, 0] = 0.01
    Df, Vf, fbf = ops.lanczos_ritz_kstep(L2[..., 0], None, M, M, return_fallback=True)
    assert fbf.cpu().tolist() == [0, 1, 0]
    Dd, Vd = ops.lanczos_ritz_kstep(L2[..., 0].contiguous(), None, M, M, compact=False)
    assert torch.equal(Df[1], Dd[1]) and torch.equal(Vf[1], Vd[1])
    assert (Df - Dd).abs().max() < 1e-6 and _projector_gap(Vf, Vd) < 1e-5
    del copies[:]
    Lo = torch.stack([Ad[:, :510, :510], Ad[:, :510, :510]], dim=3)      # N = 510: not a multiple of 4
    Do, Vo = ops.lanczos_ritz_kstep(Lo[..., 0], None, M, M)
    assert copies and Vo.shape == (B, 510, M)
    Do2, Vo2 = ops.lanczos_ritz_kstep(Ad[:, :510, :510].contiguous(), None, M, M)
    assert (Do - Do2).abs().max() < 1e-6 and _projector_gap(Vo, Vo2) < 1e-5
  finally:
    torch.zeros = orig


def test_config5_from_raw_adjacency_through_the_dataset_mirror():
  """BASELINE config 5 end to end on the product surface, no hand-called kernel: raw adjacency ->
  `collate_graph_adjacency` (device L4; `ops.lanczos_ritz` routes 2048 nodes to the K-step entry,
  the reference's use_eigen_decomp=False branch, utils/data_helper.py:205-208) ->
  `LanczosNetGeneral.forward`, against the scores the unmodified reference class produced on these
  inputs (tests/golden/config5_full.npz)."""
  import warnings
  from large_fixture import adjacency
  from lanczosnet_amd.dataset.graph_data import collate_graph_adjacency
  g = load_golden('config5_full.npz')
  B, N, K = int(g['B']), int(g['N']), int(g['K'])
  cfg, P, net, X, L, mask = _general_setup(B, N, K, int(g['num_layer']), int(g['seed']),
                                           float(g['p_edge']))
  adj = adjacency(B, N, float(g['p_edge']), int(g['seed']))
  items = [dict(adjs=adj[b][:, :, None].astype(np.float32), node_feat=X[b], label=np.zeros((1, 2)))
           for b in range(B)]
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    from lanczosnet_amd import ops
    ops._WARNED.clear()
    batch = collate_graph_adjacency(items, K, device=DEV)
  assert any('use_eigen_decomp=False' in str(x.message) for x in w)      # the branch is announced
  assert (batch['L'].cpu().numpy() - L).__abs__().max() < 1e-7
  assert np.abs(batch['D'].cpu().numpy() - g['D']).max() < 1e-6
  with torch.no_grad():
    score = net(batch['node_feat'], batch['L'], batch['D'], batch['V'],
                mask=torch.from_numpy(mask).to(DEV))    # (the fixture masks the tail of graph 1)
  ref = g['score']
  err = (np.abs(score.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
  print('config 5 from the raw adjacency vs REFERENCE: %.2e' % err)
  assert err < 1e-4


def _bf16_round(x):
  return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def _pieces_sum(t):
  """[planes, ...] bf16 device tensor -> float64 numpy sum of the pieces."""
  return t.float().double().sum(dim=0).cpu().numpy()


def _untile(x):
  """Fragment-tile order [..., RT, nkb, 4, 64, 8] (include/lanczosnet_hip.h) -> dense
  [..., 32 RT rows, 64 nkb columns]: f = 2 rt + ks, lane = 16 kq + r15 holds row 16 rt + r15,
  k 32 ks + 8 kq + u."""
  lead = x.shape[:-5]
  RT, nkb = x.shape[-5], x.shape[-4]
  x = x.reshape(lead + (RT, nkb, 2, 2, 4, 16, 8))          # rg, kb, rt, ks, kq, r15, u
  nl = len(lead)
  x = np.transpose(x, tuple(range(nl)) + (nl, nl + 2, nl + 5, nl + 1, nl + 3, nl + 4, nl + 6))
  return x.reshape(lead + (RT * 32, nkb * 64))              # rg, rt, r15 | kb, ks, kq, u


@pytest.mark.parametrize('planes', [1, 2, 3])
@pytest.mark.parametrize('B,N,C,K,din,S', [(2, 200, 2, 40, 10, 3), (3, 300, 3, 64, 128, 8),
                                           (1, 64, 1, 7, 33, 1)])
def test_large_conv_stages_match_numpy(planes, B, N, C, K, din, S):
  """lnz_large_pack_operators / gemm1 / spectral / conv one by one against float64 numpy on the
  SAME rounded operands (planes = 1: bf16-rounded inputs make the products exact in fp32, so the
  only difference is fp32 accumulation order; planes = 3: fp32-grade).  Ragged sizes: N not a
  multiple of the 256-row tile nor of the 64-wide k-block, K < 64, input width not a multiple
  of 16, 1..3 channels."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(N + planes)
  L = (rs.randn(B, N, N, C) * (rs.rand(B, N, N, C) < 0.1)).astype(np.float32)
  V = (rs.randn(B, N, K) / np.sqrt(N)).astype(np.float32)
  X = rs.randn(B, N, din).astype(np.float32)
  W = (rs.randn(128, (S + C) * din) / np.sqrt(C * din)).astype(np.float32)
  G = rs.randn(B, S, K).astype(np.float32)
  bias = rs.randn(128).astype(np.float32)
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
  Lb, Vb = ops.large_pack_operators(dev(L), dev(V), planes)
  Nk = Lb.dims[1]
  RT = (N + 31) // 32
  assert Nk % 64 == 0 and Nk >= N and tuple(Lb.shape) == (planes, B, C, RT, Nk // 64, 4, 64, 8)
  # pack: pieces sum to the input (exactly for planes = 3 up to 2^-24, bf16 rounding for 1; the
  # two-plane mode stores fp16 pieces of 1024 x the A operands)
  ascale = ops.LARGE_F16_A_SCALE if planes == 2 else 1.0
  assert Lb.dtype == (torch.float16 if planes == 2 else torch.bfloat16)
  Lsum = _untile(_pieces_sum(Lb)) / ascale
  assert (Lsum[:, :, N:] == 0).all()
  Lsum = Lsum[:, :, :N]
  Lref = L.transpose(0, 3, 1, 2).astype(np.float64)
  if planes == 1:
    np.testing.assert_array_equal(Lsum[..., :N], _bf16_round(L).transpose(0, 3, 1, 2))
  else:
    assert np.abs(Lsum[..., :N] - Lref).max() <= 2.0 ** (-22 if planes == 3 else -20) * np.abs(Lref).max()
  assert (Lsum[..., N:] == 0).all()
  Vsum = _untile(_pieces_sum(Vb)[:, :, None]) / ascale
  assert (Vsum[..., K:] == 0).all() and (Vsum[:, N:] == 0).all()
  Vsum = Vsum[:, :N]
  # weights as the model packs them
  dinp = (din + 15) // 16 * 16
  Wc = np.zeros((128, S + C, dinp), np.float32)
  Wc[:, :, :din] = W.reshape(128, S + C, din)
  Wb = ops.large_weight_fragments(ops.split_bf16_planes(
      dev(Wc[:, S:].transpose(1, 0, 2).reshape(C * 128, dinp)), planes))
  Wt = ops.pack_rows_k8(dev(np.ascontiguousarray(Wc[:, :S].reshape(128, S * dinp))))
  work = ops.large_work_buffers(Lb)
  Zt, Tt, Ybuf = work
  out = ops.large_conv_layer(dev(X), din, Lb, Vb, dev(V), Wb, Wt, dev(G), dev(bias), work)
  assert (Ybuf == 0).all()  # left zeroed for the next layer
  out = out.cpu().numpy().astype(np.float64)
  # ---- stage references in float64 on the operands the kernels saw
  rnd = _bf16_round if planes == 1 else (lambda a: np.asarray(a, np.float64))
  X64, W64 = rnd(X), rnd(Wc)
  Z = np.einsum('bni,oci->bcon', X64[..., :din], W64[:, S:, :din])            # [B,C,128,N]
  Zt_got = _pieces_sum(Zt)
  tol = 2e-6 if planes == 3 else 4e-6 if planes == 2 else 1e-5
  assert (Zt_got[..., N:] == 0).all()
  Zt_ref = rnd(Z) if planes == 1 else Z
  zden = np.abs(Z).max()
  assert np.abs(Zt_got[..., :N] - Zt_ref).max() <= (8e-3 if planes == 1 else tol) * zden
  Y = np.einsum('bnk,bni->bki', V.astype(np.float64), X.astype(np.float64))  # exact fp32 inside
  T = np.einsum('bsk,bki,osi->bko', G.astype(np.float64), Y, Wc[:, :S, :din].astype(np.float64))
  Tt_got = _pieces_sum(Tt).transpose(0, 2, 1)[:, :K]                          # [B,K,128]
  assert np.abs(Tt_got - T).max() <= (8e-3 if planes == 1 else 1e-5) * np.abs(T).max()
  # ---- conv on the operands it was handed (its own Zt / Tt images)
  ref = np.einsum('bcnk,bcok->bno', Lsum[..., :N], Zt_got[..., :N]) + \
      np.einsum('bnk,bok->bno', Vsum, _pieces_sum(Tt)) + bias
  ref = np.maximum(ref, 0)
  assert out.shape == (B, N, 128)
  assert np.abs(out - ref).max() <= (2e-5 if planes == 1 else 5e-6) * np.abs(ref).max()


@pytest.mark.parametrize('planes', [1, 3])
@pytest.mark.parametrize('layout', ['channels_last', 'channel_major', 'pair'])
def test_pack_fold_compares_channels_and_packs_the_distinct_ones(planes, layout):
  """lnz_large_pack_operators_fold: the packed image of the distinct channels is the unfolded
  image of those channels bit for bit; the comparison bits name exactly the channel pairs that
  differ somewhere — also when the only difference is ONE entry in the last ragged row; a zero
  channel stride (expanded view) needs no comparison."""
  from lanczosnet_amd import ops
  B, N, K = 2, 203, 40
  rs = np.random.RandomState(3)
  a = (rs.randn(B, N, N) * (rs.rand(B, N, N) < 0.1)).astype(np.float32)
  b_ = (rs.randn(B, N, N) * (rs.rand(B, N, N) < 0.1)).astype(np.float32)
  V = torch.from_numpy((rs.randn(B, N, K) / np.sqrt(N)).astype(np.float32)).to(DEV)
  if layout == 'pair':                      # the collate layout of one edge type: [B,N,N,2]
    chans, same = [a, a], {(1, 0)}
  else:
    chans, same = [a, b_, a], {(2, 0)}
  Cn = len(chans)
  L = torch.from_numpy(np.stack(chans, axis=3)).to(DEV)
  if layout == 'channel_major':
    L = L.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)   # rows contiguous per channel
    assert L.stride(2) == 1
  bit = lambda c, c2: 1 << (8 * c + c2)  # noqa: E731
  full, _ = ops.large_pack_operators(L, V, planes)
  neq = torch.zeros((1,), dtype=torch.int64, device=DEV)
  full2, _ = ops.large_pack_operators(L, V, planes, neq=neq)
  assert torch.equal(full.view(torch.int16), full2.view(torch.int16))
  want = sum(bit(c, c2) for c in range(Cn) for c2 in range(c) if (c, c2) not in same)
  assert int(neq.item()) == want
  # folded: channel (c) claimed equal to channel 0
  (cf, _), = same
  src = [c for c in range(Cn) if c != cf]
  rep = [src.index(c) if c != cf else 0 for c in range(Cn)]
  neq.zero_()
  fold, Vb = ops.large_pack_operators(L, V, planes, chan_src=src, chan_rep=rep, neq=neq)
  assert fold.shape[2] == Cn - 1 and fold.dims == full.dims
  assert torch.equal(fold.view(torch.int16), full[:, :, src].contiguous().view(torch.int16))
  assert int(neq.item()) & bit(cf, 0) == 0
  # one differing entry in the last row -> the claim fails
  L2 = L.clone() if layout != 'channel_major' else \
      L.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
  L2[B - 1, N - 1, N - 2, cf] += 0.5
  neq.zero_()
  ops.large_pack_operators(L2, V, planes, chan_src=src, chan_rep=rep, neq=neq)
  assert int(neq.item()) & bit(cf, 0)
  # an expanded view: equal by construction, nothing to compare (check = 0), same image
  Le = L[:, :, :, :1].expand(B, N, N, 2)
  assert Le.stride(3) == 0
  neq.zero_()
  ex, _ = ops.large_pack_operators(Le, V, planes, chan_src=[0], chan_rep=[0, 0], chan_check=[1, 0],
                                   neq=neq)
  assert int(neq.item()) == 0
  assert torch.equal(ex.view(torch.int16), full[:, :, :1].contiguous().view(torch.int16))


def test_large_forward_folds_equal_channels_and_survives_a_wrong_guess():
  """The module's fold protocol (model/lanczos_net.py `_large_pack`): batch 1 packs both channels
  and learns they are equal, batch 2 streams ONE operator with the summed weight blocks (same
  scores to fp32 rounding), a batch whose channels differ fails the in-kernel verification and
  is recomputed unfolded — equal to a module that never folds; an expanded view folds at once."""
  from lanczosnet_amd import ops
  B, N, K = 3, 256, 32
  cfg, P, net, X, L, mask = _general_setup(B, N, K, 3, 7, 8.0 / N)
  _, _, plain, _, _, _ = _general_setup(B, N, K, 3, 7, 8.0 / N)
  plain.large_fold = False
  net.large_sparse = plain.large_sparse = False   # (the streamed kernels: the sparse image has its own tests)
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  D, V = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
  seen = []
  # (the eigen-space projection adds its row chunks with fp32 atomics: runs agree to rounding)
  close = lambda x, y: bool((x - y).abs().max() <= 2e-6 * y.abs().max())  # noqa: E731
  orig = ops.large_pack_operators

  def spy(*a, **kw):
    out = orig(*a, **kw)
    seen.append(out[0].shape[2])
    return out
  ops.large_pack_operators = spy
  try:
    with torch.no_grad():
      ref = plain(Xd, Ld, D, V, mask=md)
      assert seen == [2]
      s1 = net(Xd, Ld, D, V, mask=md)            # learns
      s2 = net(Xd, Ld, D, V, mask=md)            # folds
      assert seen[1:] == [2, 1]
      assert close(s1, ref)
      assert close(s2, ref)
      L2 = Ld.clone()
      L2[:, :, :, 1] *= 0.5
      ref2 = plain(Xd, L2, D, V, mask=md)
      del seen[:]
      s3 = net(Xd, L2, D, V, mask=md)            # folded guess fails -> repacked unfolded
      assert seen == [1, 2]
      assert close(s3, ref2)
      s4 = net(Xd, L2, D, V, mask=md)            # guess dropped: unfolded, no verification wait
      assert seen[2:] == [2] and close(s4, ref2)
      s5 = net(Xd, Ld, D, V, mask=md)            # equal again: learns again ...
      s6 = net(Xd, Ld, D, V, mask=md)            # ... and folds
      assert seen[3:] == [2, 1] and close(s5, ref) and close(s6, s2)
      del seen[:]
      s7 = net(Xd, Ld[:, :, :, :1].expand(B, N, N, 2), D, V, mask=md)   # zero channel stride
      assert seen == [1] and close(s7, s2)
      net.gemm_mode = 'bf16'                     # config 5's mode folds the same way
      plain.gemm_mode = 'bf16'
      sb = net(Xd, Ld, D, V, mask=md)
      rb = plain(Xd, Ld, D, V, mask=md)
      assert seen[1:] == [1, 2] and (sb - rb).abs().max() <= 2e-2 * rb.abs().max()
  finally:
    ops.large_pack_operators = orig


def _general_setup(B, N, K, num_layer, seed, p_edge):
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg, P, X, L, mask = general_inputs(B, N, K, num_layer, seed, p_edge)
  net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return cfg, P, net.to(DEV), X, L, mask


def test_large_graph_general_forward_matches_oracle():
  """LanczosNetGeneral beyond the 32-node tile (config 5 regime, reduced size): device pipeline
  (large Lanczos -> gains -> streamed conv kernels) vs the fp64 oracle fed the SAME Ritz pairs:
  the default split-precision mode at north_star's 1e-5, config 5's bf16-operand mode at the
  bf16-appropriate 2e-2, and the library-GEMM path the kernels replaced at 1e-5."""
  from lanczosnet_amd import ops
  B, N, K = 3, 256, 32
  cfg, P, net, X, L, mask = _general_setup(B, N, K, 3, 7, 8.0 / N)
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  D, V = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
  with torch.no_grad():
    net.large_split_planes = 3
    score = net(Xd, Ld, D, V, mask=md)                       # HIP kernels, 3 bf16 planes
    net.large_split_planes = 2
    score2 = net(Xd, Ld, D, V, mask=md)                      # HIP kernels, 2 fp16 planes
    score_lib = net._large_graph_forward(Xd, Ld, D, V, md)   # hipBLASLt path
    net.gemm_mode = 'bf16'
    score_bf16 = net(Xd, Ld, D, V, mask=md)
    net.gemm_mode = 'fp32'
  ref = oracle.lanczos_net_forward(P, cfg, X, L, D.cpu().numpy(), V.cpu().numpy(), mask,
                                   dtype=np.float64, general=True)
  e = np.abs(score.cpu().numpy() - ref).max() / np.abs(ref).max()
  el = np.abs(score_lib.cpu().numpy() - ref).max() / np.abs(ref).max()
  eb = np.abs(score_bf16.cpu().numpy() - ref).max() / np.abs(ref).max()
  print('large-graph forward rel err: split-precision kernels %.2e, library path %.2e, bf16 '
        'operands %.2e' % (e, el, eb))
  e2 = np.abs(score2.cpu().numpy() - ref).max() / np.abs(ref).max()
  print('two fp16 planes %.2e' % e2)
  assert e < 1e-5
  assert e2 < 1e-5
  assert el < 1e-5
  assert eb < 2e-2  # bf16 operands: 8-bit mantissa (config 5's mode, opt-in)


def test_large_graph_forward_config5_shape_matches_library_path():
  """BASELINE config 5's shape (N = 2048, K = 64, 7 layers; B = 2 here): the streamed kernels in
  the split-precision mode against the fp32 library-GEMM path on the same Ritz pairs at 1e-5 (the
  fp64 oracle's explicit N x N filters are too slow at this size), bf16 mode at 2e-2; padded
  nodes (mask) and finite outputs."""
  from lanczosnet_amd import ops
  B, N, K = 2, 2048, 64
  cfg, P, net, X, L, mask = _general_setup(B, N, K, 7, 11, 0.01)
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  D, V = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
  with torch.no_grad():
    net.large_split_planes = 3
    s3 = net(Xd, Ld, D, V, mask=md)
    net.large_split_planes = 2
    s2 = net(Xd, Ld, D, V, mask=md)
    sl = net._large_graph_forward(Xd, Ld, D, V, md)
    net.gemm_mode = 'bf16'
    s1 = net(Xd, Ld, D, V, mask=md)
  assert torch.isfinite(s3).all() and torch.isfinite(s1).all()
  den = sl.abs().max().item()
  e3 = (s3 - sl).abs().max().item() / den
  e1 = (s1 - sl).abs().max().item() / den
  print('N=2048: split-precision vs library %.2e, bf16 vs library %.2e' % (e3, e1))
  e2 = (s2 - sl).abs().max().item() / den
  print('two fp16 planes vs library %.2e' % e2)
  assert e3 < 1e-5
  assert e2 < 1e-5 and torch.isfinite(s2).all()
  assert e1 < 2e-2


def test_config5_full_size_matches_the_reference_golden():
  """BASELINE config 5 at its full size (N = 2048, K = 64, 7 x 128, E+1 = 2; B = 2) against the
  REFERENCE: tests/golden/config5_full.npz holds the scores of the unmodified LanczosNetGeneral
  (CPU) on these seeded inputs with the Ritz pairs of the fp64 K-step restatement
  (tests/golden/make_golden_config5.py).  The same (D, V) — recomputed here by the oracle and
  checked against the fixture — go to the HIP forward: split-precision modes at north_star's 1e-5
  per graph, the bf16-operand mode (config 5's) at 2e-2; then the device's own Ritz pairs
  (lnz_lanczos_ritz_large), whose unconverged trailing pairs differ from the fp64 run's by rounding
  amplified through the Lanczos recurrence — the score moves by < 1e-4."""
  from lanczosnet_amd import ops
  g = load_golden('config5_full.npz')
  B, N, K = int(g['B']), int(g['N']), int(g['K'])
  cfg, P, net, X, L, mask = _general_setup(B, N, K, int(g['num_layer']), int(g['seed']),
                                           float(g['p_edge']))
  D, V = kstep_ritz(L, K)
  assert np.abs(D - g['D']).max() < 1e-6
  assert np.abs(np.abs(V.astype(np.float64)).sum(axis=1) - g['V_abs_colsum']).max() < 1e-3
  ref = g['score']
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  Dd, Vd = torch.from_numpy(D).to(DEV), torch.from_numpy(V).to(DEV)

  def per_graph(s):
    return (np.abs(s.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
  with torch.no_grad():
    net.large_split_planes = 3
    e3 = per_graph(net(Xd, Ld, Dd, Vd, mask=md))
    net.large_split_planes = 2
    e2 = per_graph(net(Xd, Ld, Dd, Vd, mask=md))
    el = per_graph(net._large_graph_forward(Xd, Ld, Dd, Vd, md))
    net.gemm_mode = 'bf16'
    e1 = per_graph(net(Xd, Ld, Dd, Vd, mask=md))
    net.gemm_mode = 'fp32'
    net.large_split_planes = 3
    Dk, Vk = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
    ed = per_graph(net(Xd, Ld, Dk, Vk, mask=md))
  print('config 5 full size vs REFERENCE: 3 bf16 planes %.2e, 2 fp16 planes %.2e, library path '
        '%.2e, bf16 operands %.2e; device Ritz pairs %.2e' % (e3, e2, el, e1, ed))
  assert e3 < 1e-5 and e2 < 1e-5 and el < 1e-5
  assert e1 < 2e-2
  assert ed < 1e-4


def test_qm8_schema_molecules_beyond_the_32_node_tile():
  """LanczosNet (atom embedding, 7 bond-type channels, full QM8 widths) on molecules of up to 44
  atoms: beyond the fused 32-row kernel the module takes the streamed kernels (general channel
  count, ragged N, K = 48 < 64) — against the fp64 oracle on the reference pipeline's (D, V)."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  K = 48   # >= n: no top-K cut, so no molecule's (D, V) is basis dependent
  cfg = dict(oracle.DEFAULT_QM8_CFG, num_eig_vec=K)
  b = draw_batch(12, seed=9, n_min=20, n_max=44)
  B, N = b['node_mask'].shape
  assert N > 32
  P = oracle.make_lanczosnet_params(cfg, 4)
  net = LanczosNet(make_model_config(cfg)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], n, K)
  with torch.no_grad():
    score = net(t(b['node_feat']), L, D, V, mask=t(b['node_mask'])).cpu().numpy()
  Lo = np.zeros((B, N, N, 7), np.float32)
  Dl, Vl = [], []
  for i in range(B):
    nb = int(b['n_nodes'][i])
    Lo[i, :nb, :nb] = oracle.laplacian_multi_l4(b['adjs'][i, :nb, :nb])
    e, v, _ = oracle.graph_laplacian_eigs(b['adjs'][i, :nb, :nb].sum(axis=2), graph_laplacian_type='L4')
    Dl.append(e)
    Vl.append(v)
  Do, Vo = oracle.collate_eigs(Dl, Vl, N, K)
  ref = oracle.lanczos_net_forward(P, cfg, b['node_feat'], Lo, Do, Vo, b['node_mask'], dtype=np.float64)
  per = np.abs(score - ref).max(axis=1) / np.abs(ref).max()
  assert per.max() < 1e-5, per


def test_kstep_entry_stride_2_view_without_room_behind_its_last_row_is_copied():
  """The in-place pair form reads float4s; a stride-2 view whose last row ends at the end of its
  storage (every other column of a [B,N,2N-1] tensor) has no element behind its last entry: it takes
  the copy, and gives the pairs of the contiguous matrix."""
  from lanczosnet_amd import ops
  N, M, B = 256, 16, 2
  A = torch.from_numpy(_graphs(B, N, 0.03, seed=9)).to(DEV)
  wide = torch.zeros((B, N, 2 * N - 1), device=DEV)
  wide[:, :, ::2] = A
  view = wide[:, :, ::2]
  assert view.stride(2) == 2 and view.stride(1) % 4 != 0      # (odd row pitch: not aligned either)
  Dv, Vv = ops.lanczos_ritz_kstep(view, None, M, M)
  Dc, Vc = ops.lanczos_ritz_kstep(A, None, M, M)
  assert (Dv - Dc).abs().max() < 1e-6 and _projector_gap(Vv, Vc) < 1e-5
  wide2 = torch.zeros((B, N, 2 * N), device=DEV)               # aligned rows, room behind: in place
  wide2[:, :, ::2] = A
  Dw, Vw = ops.lanczos_ritz_kstep(wide2[:, :, ::2], None, M, M)
  assert (Dw - Dc).abs().max() < 1e-6 and _projector_gap(Vw, Vc) < 1e-5
  flat = torch.zeros((B * N * 2 * N + 0,), device=DEV)        # exactly [B,N,N,2] minus nothing: the collate case
  L = flat.view(B, N, N, 2)
  L[..., 0] = A
  Dl, Vl = ops.lanczos_ritz_kstep(L[..., 0], None, M, M)
  assert (Dl - Dc).abs().max() < 1e-6 and _projector_gap(Vl, Vc) < 1e-5
