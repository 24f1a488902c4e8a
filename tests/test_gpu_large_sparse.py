"""Large-graph conv on the nonzeros of the Laplacian (csrc/conv_sparse.hip; BASELINE config 5's
bf16 mode): the image against numpy, the layer against the streamed layer (same products, other
summation order) and an fp64 layer, the module's fallback protocol."""
import numpy as np
import pytest
import torch

from large_fixture import general_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bf16(x):
  return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _sparse_L(B, N, C, p, seed, layout):
  rs = np.random.RandomState(seed)
  A = (rs.rand(B, N, N) < p) * rs.randn(B, N, N)
  A = (A + A.transpose(0, 2, 1)).astype(np.float32)
  A[0, 0, :] = 0.0                                   # an empty row
  if layout == 'channels_last':
    L = torch.from_numpy(np.stack([A] * C, axis=3)).to(DEV)
  elif layout == 'channel_major':                    # [B,C,N,N] storage, channels-last view
    L = torch.from_numpy(np.stack([A] * C, axis=1)).to(DEV).permute(0, 2, 3, 1)
  elif layout == 'expanded':
    L = torch.from_numpy(A).to(DEV).unsqueeze(3).expand(B, N, N, C)
  else:                                              # a row-padded parent: strides not those of a contiguous tensor
    big = torch.zeros((B, N, N + 4, C), device=DEV)
    big[:, :, :N] = torch.from_numpy(np.stack([A] * C, axis=3)).to(DEV)
    L = big[:, :, :N]
  return A, L


@pytest.mark.parametrize('B,N,C,p,layout', [
    (3, 256, 2, 0.03, 'channels_last'), (2, 301, 2, 0.02, 'channels_last'), (2, 200, 1, 0.05, 'channels_last'),
    (2, 264, 3, 0.03, 'channels_last'), (2, 256, 2, 0.03, 'channel_major'), (2, 256, 2, 0.03, 'expanded'), (2, 301, 2, 0.02, 'expanded'),
    (2, 256, 2, 0.03, 'padded_parent'), (1, 2048, 2, 0.01, 'channels_last')])
def test_sparse_image_holds_exactly_the_nonzeros(B, N, C, p, layout):
  """entries = bf16(value) << 16 | column for every nonzero of channel 0, each row's set complete
  and duplicate free, counts exact, zero padding to a multiple of 8, flags clear — for the
  channels-last pair fast path, odd N, one and three channels, channel-major / expanded /
  non-contiguous views."""
  from lanczosnet_amd import ops
  A, L = _sparse_L(B, N, C, p, 5 + N, layout)
  img = ops.large_sparse_image(L, 64)
  assert int(img.flags.item()) == 0
  cnt = img.counts.cpu().numpy()
  ent = img.entries.cpu().numpy().view(np.uint32)
  assert np.array_equal(cnt, (A != 0).sum(2))
  assert cnt[0, 0] == 0
  Ab = _bf16(A)
  for b in range(B):
    for r in range(N):
      c = int(cnt[b, r])
      e = ent[b, r, :(c + 7) // 8 * 8]
      assert np.all(e[c:] == 0)
      cols = (e[:c] & 0xffff).astype(np.int64)
      vals = (e[:c] & np.uint32(0xffff0000)).view(np.float32)
      assert np.array_equal(np.sort(cols), np.nonzero(A[b, r])[0])
      assert np.array_equal(vals, Ab[b, r, cols])


def test_sparse_image_flags_differing_channels_dense_rows_and_keeps_a_nan():
  from lanczosnet_amd import ops
  A, L = _sparse_L(2, 256, 2, 0.03, 9, 'channels_last')
  L2 = L.clone()
  L2[1, 200, 3, 1] += 0.25                           # channel 1 differs in ONE entry
  assert int(ops.large_sparse_image(L2, 64).flags.item()) == 1
  L3 = torch.stack([L[..., 0], L[..., 0] * 2.0, L[..., 0]], dim=3)          # generic path, middle channel
  assert int(ops.large_sparse_image(L3, 64).flags.item()) == 1
  L4 = L.clone()
  L4[0, 7, :100, :] = 1.0                            # a row of 100 nonzeros
  assert int(ops.large_sparse_image(L4, 64).flags.item()) == 2
  img = ops.large_sparse_image(L4, 128)
  assert int(img.flags.item()) == 0 and int(img.counts[0, 7]) >= 100
  L5 = L.clone()
  L5[1, 5, 9, :] = float('nan')
  img = ops.large_sparse_image(L5, 64)               # NaN != NaN: reported as a differing channel ...
  assert int(img.flags.item()) == 1
  L6 = L5[..., :1].expand(-1, -1, -1, 2)             # ... and kept as an entry when there is nothing to compare
  img = ops.large_sparse_image(L6, 64)
  e = img.entries[1, 5, :int(img.counts[1, 5])].cpu().numpy().view(np.uint32)
  assert int(img.flags.item()) == 0
  assert np.isnan((e & np.uint32(0xffff0000)).view(np.float32)[(e & 0xffff) == 9]).all()
  with pytest.raises(ops.LnzError):
    ops.large_sparse_image(L, 20)                    # row_cap: a multiple of 8, at least 32


@pytest.mark.parametrize('B,N,K,din,S,p', [(2, 200, 40, 10, 3, 0.04), (3, 300, 64, 128, 8, 0.02),
                                           (9, 130, 20, 16, 2, 0.05)])
def test_sparse_layer_matches_the_streamed_layer_and_fp64(B, N, K, din, S, p):
  """Same bf16-rounded operands, fp32 accumulation: against the streamed layer only the order of the
  sums differs (1e-5); against an fp64 layer on the bf16-rounded operator / features 2e-2 (the
  spectral block's bf16 T and V pieces)."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(N)
  A, L = _sparse_L(B, N, 2, p, 3 + N, 'channels_last')
  V = (rs.randn(B, N, K) / np.sqrt(N)).astype(np.float32)
  G = rs.rand(B, S, K).astype(np.float32)
  X = rs.randn(B, N, din).astype(np.float32)
  W = (rs.randn(128, S + 2, din) / np.sqrt(din * (S + 2))).astype(np.float32)
  bias = (rs.randn(128) * 0.1).astype(np.float32)
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
  dinp = (din + 15) // 16 * 16
  Wc = np.pad(W, ((0, 0), (0, 0), (0, dinp - din)))
  Wn = Wc[:, S] + Wc[:, S + 1]
  Wf = ops.large_weight_fragments(ops.split_bf16_planes(dev(Wn).reshape(128, dinp), 1).reshape(1, 128, dinp))
  Wt = ops.pack_rows_k8(dev(Wc[:, :S].reshape(128, S * dinp)))
  Xd, Vd, Gd, bd = dev(X), dev(V), dev(G), dev(bias)
  Lb, Vb = ops.large_pack_operators(L, Vd, 1, chan_src=[0], chan_rep=[0, 0])
  dense = ops.large_conv_layer(Xd, din, Lb, Vb, Vd, Wf, Wt, Gd, bd, ops.large_work_buffers(Lb))
  img = ops.large_sparse_image(L, 64)
  assert int(img.flags.item()) == 0
  Vb2 = ops.large_pack_vectors(Vd, 1)
  assert torch.equal(Vb2.view(torch.int16), Vb.view(torch.int16))
  for relu in (True, False):
    # (the eigen-space projection adds its row chunks with fp32 atomics: two runs of it may round an
    # entry of T to different bf16 values — the sparse layer is handed the streamed layer's T)
    dwork = ops.large_work_buffers(Lb)
    dense = ops.large_conv_layer(Xd, din, Lb, Vb, Vd, Wf, Wt, Gd, bd, dwork, relu=relu)
    swork = ops.large_sparse_work_buffers(B, N, DEV)
    swork[1].copy_(dwork[1])
    sparse = ops.large_sparse_conv_layer(Xd, din, img, Vb2, Vd, Wf, None, None, bd, swork, relu=relu)
    den = dense.abs().max().item()
    assert (sparse - dense).abs().max().item() <= 1e-5 * den
  Z = _bf16(np.einsum('bnd,od->bno', _bf16(X).astype(np.float64), _bf16(Wn[:, :din]).astype(np.float64)))
  ref = np.einsum('bnm,bmo->bno', _bf16(A).astype(np.float64), Z.astype(np.float64)) + bias
  Y = np.einsum('bnk,bnd->bkd', V.astype(np.float64), X.astype(np.float64))
  T = sum(G[:, s, :, None] * np.einsum('bkd,od->bko', Y, W[:, s].astype(np.float64)) for s in range(S))
  ref = np.maximum(ref + np.einsum('bnk,bko->bno', V.astype(np.float64), T), 0.0)
  wfull = ops.large_sparse_work_buffers(B, N, DEV)
  sparse = ops.large_sparse_conv_layer(Xd, din, img, Vb2, Vd, Wf, Wt, Gd, bd, wfull)
  assert np.abs(sparse.cpu().numpy() - ref).max() <= 2e-2 * np.abs(ref).max()
  # the projection launch also wrote Z (one pass over X): lnz_large_gemm1_rows's bits
  Zsep = torch.empty_like(wfull[0])
  ops._abi().large_gemm1_rows(Xd, din, din, Wf, B, N, Zsep)
  assert torch.equal(wfull[0].view(torch.int16), Zsep.view(torch.int16))
  # without long scales: Tt stays zero, the lift adds the bias alone
  # (against the launch's OWN Z — a product near a bf16 rounding tie may round the other way in the
  # fp64 restatement — only the fp32 accumulation order is left: 1e-5)
  work0 = ops.large_sparse_work_buffers(B, N, DEV)
  sparse0 = ops.large_sparse_conv_layer(Xd, din, img, Vb2, Vd, Wf, None, None, bd, work0)
  Zdev = work0[0].float().cpu().numpy().astype(np.float64)
  assert np.abs(Zdev - Z).max() <= 8e-3 * np.abs(Z).max()
  ref0 = np.maximum(np.einsum('bnm,bmo->bno', _bf16(A).astype(np.float64), Zdev) + bias, 0.0)
  assert np.abs(sparse0.cpu().numpy() - ref0).max() <= 1e-5 * np.abs(ref0).max()


def test_module_takes_the_sparse_image_in_bf16_mode_and_falls_back_when_it_must():
  """LanczosNetGeneral, gemm_mode = 'bf16' (config 5's): the sparse image serves the batch (no
  operator pack at all) with the streamed kernels' scores; channels that differ, or a graph too
  dense for the row capacity, raise the image's flag — the batch is recomputed on the streamed
  kernels (equal to a module with the sparse path off) and the next calls do not try again until
  the back-off has run out; the split-precision modes take the image too, with the node-space
  term in exact fp32 (1e-5 against their streamed kernels)."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  B, N, K = 3, 256, 32
  cfg, P, X, L, mask = general_inputs(B, N, K, 3, 7, 8.0 / N)

  def make():
    net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    net = net.to(DEV)
    net.gemm_mode = 'bf16'
    return net
  net, plain = make(), make()
  plain.large_sparse = False
  net.large_sparse_backoff = 2
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  D, V = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
  packs, images = [], []
  orig_pack, orig_img = ops.large_pack_operators, ops.large_sparse_image

  def spy_pack(*a, **kw):
    packs.append(1)
    return orig_pack(*a, **kw)

  def spy_img(*a, **kw):
    images.append(1)
    return orig_img(*a, **kw)
  ops.large_pack_operators, ops.large_sparse_image = spy_pack, spy_img
  close = lambda x, y, t: bool((x - y).abs().max() <= t * y.abs().max())  # noqa: E731
  try:
    with torch.no_grad():
      plain(Xd, Ld, D, V, mask=md)                             # (learns that the channels are equal)
      ref = plain(Xd, Ld, D, V, mask=md)                       # one operator, summed weight blocks
      n_plain = len(packs)
      assert n_plain >= 2 and not images
      s1 = net(Xd, Ld, D, V, mask=md)
      assert len(packs) == n_plain and len(images) == 1        # no operator pack
      # same products in another order; the bf16 rounding of T can flip between two runs of the
      # projection's fp32 atomics (the layer test pins the gather itself at 1e-5 on a shared T)
      assert close(s1, ref, 2e-3)
      st = net._large_sparse_state[Ld.device.index]
      assert st['last_flags'] == 0
      L2 = Ld.clone()
      L2[:, :, :, 1] *= 0.5                                    # the channels differ
      ref2 = plain(Xd, L2, D, V, mask=md)
      n_plain = len(packs)
      s2 = net(Xd, L2, D, V, mask=md)
      assert st['last_flags'] == 1 and st['skip'] == 2 and len(packs) > n_plain
      assert close(s2, ref2, 2e-3)
      n_img = len(images)
      net(Xd, Ld, D, V, mask=md)
      net(Xd, Ld, D, V, mask=md)                               # back-off: the streamed kernels, no image
      assert len(images) == n_img and st['skip'] == 0
      s3 = net(Xd, Ld, D, V, mask=md)                          # tries again
      assert len(images) == n_img + 1 and st['last_flags'] == 0 and close(s3, s1, 2e-3)
      L3 = Ld.clone()
      L3[0, 10, :60, :] = 0.01                                 # a row beyond the capacity (32 at N = 256)
      ref3 = plain(Xd, L3, D, V, mask=md)
      s4 = net(Xd, L3, D, V, mask=md)
      # (this module's streamed path has not learned the fold yet: two operators, two bf16 roundings
      # of X W_c^T, against `plain`'s one — the bf16 mode's own tolerance)
      assert st['last_flags'] == 2 and close(s4, ref3, 2e-2)
      assert st['skip'] == 2
      st['skip'] = 0
      net(Xd, L3, D, V, mask=md)                               # again: the pause doubles
      assert st['last_flags'] == 2 and st['skip'] == 4
      st['skip'] = 0
      n_img, n_pack = len(images), len(packs)
      net.gemm_mode = plain.gemm_mode = 'fp32'                 # split-precision modes: the exact-fp32 gather
      s5 = net(Xd, Ld, D, V, mask=md)
      assert len(images) == n_img + 1 and len(packs) == n_pack and st['last_flags'] == 0
      assert ops.last_kernel() == 'sparse_conv_f32_kernel'
      assert close(s5, plain(Xd, Ld, D, V, mask=md), 1e-5)     # against the streamed three-piece kernels
      for planes in (2, 3):
        net.large_split_planes = plain.large_split_planes = planes
        assert close(net(Xd, Ld, D, V, mask=md), plain(Xd, Ld, D, V, mask=md), 1e-5)
  finally:
    ops.large_pack_operators, ops.large_sparse_image = orig_pack, orig_img


def test_config5_shape_sparse_and_streamed_scores_agree():
  """N = 2048, K = 64, 7 layers (B = 2): the bf16 mode through the sparse image against the same
  mode on the streamed kernels (2e-3: summation order and bf16 roundings of T that flip between
  runs, seven layers) and the fp32 library path (2e-2)."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  B, N, K = 2, 2048, 64
  cfg, P, X, L, mask = general_inputs(B, N, K, 7, 11, 0.01)
  net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  Ld = torch.from_numpy(L).to(DEV)
  Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
  D, V = ops.lanczos_ritz_large(Ld[:, :, :, 0].contiguous(), K, K)
  with torch.no_grad():
    sl = net._large_graph_forward(Xd, Ld, D, V, md)
    net.gemm_mode = 'bf16'
    ss = net(Xd, Ld, D, V, mask=md)
    assert net._large_sparse_state[Ld.device.index]['last_flags'] == 0
    assert ops.last_kernel() == 'sparse_conv_kernel'
    net.large_sparse = False
    sd = net(Xd, Ld, D, V, mask=md)
  den = sl.abs().max().item()
  print('N=2048 bf16 mode: sparse vs streamed %.2e, sparse vs library %.2e' %
        ((ss - sd).abs().max().item() / den, (ss - sl).abs().max().item() / den))
  assert torch.isfinite(ss).all()
  assert (ss - sd).abs().max().item() <= 2e-3 * den
  assert (ss - sl).abs().max().item() <= 2e-2 * den


def test_kstep_pass_leaves_the_conv_image_behind():
  """lnz_lanczos_ritz_kstep_image: the compaction pass of the K-step entry writes the conv's image
  too — bit for bit lnz_large_sparse_image's (same entry order), for the channels-last pair read
  in place (channel difference and row overflow flagged alike) and for one operator in contiguous
  rows; the Ritz pairs are those of the plain entry."""
  from lanczosnet_amd import ops
  B, N, K = 3, 512, 32
  A, L = _sparse_L(B, N, 2, 0.02, 21, 'channels_last')
  for b in range(B):                                         # (a Laplacian-like diagonal: no empty Krylov start)
    L[b, range(N), range(N), :] += 1.0
  D0, V0 = ops.lanczos_ritz_kstep(L[..., 0], None, K, K)
  D1, V1, img = ops.lanczos_ritz_kstep(L[..., 0], None, K, K, conv_image=64)
  assert torch.equal(D0, D1) and torch.equal(V0, V1)
  ref = ops.large_sparse_image(L, 64)
  assert int(img.flags.item()) == 0 and int(ref.flags.item()) == 0
  assert torch.equal(img.counts, ref.counts)
  keep = (torch.arange(64, device=DEV)[None, None, :] < ((ref.counts + 7) // 8 * 8)[:, :, None])
  assert torch.equal(img.entries[keep], ref.entries[keep])
  refv = ops.large_sparse_image(L, 64, values=True)         # the exact form's unrounded values ride along
  assert torch.equal(img.values[keep], refv.values[keep]) and torch.equal(refv.entries[keep], ref.entries[keep])
  live = torch.arange(64, device=DEV)[None, None, :] < ref.counts[:, :, None]
  cols = (refv.entries[live].long() & 0xffff)
  bi, ri, _ = torch.nonzero(live, as_tuple=True)
  assert torch.equal(refv.values[live], L[bi, ri, cols, 0])
  L2 = L.clone()
  L2[1, 9, 11, 1] += 0.5
  _, _, img2 = ops.lanczos_ritz_kstep(L2[..., 0], None, K, K, conv_image=64)
  assert int(img2.flags.item()) == 1
  _, _, img3 = ops.lanczos_ritz_kstep(L[..., 0], None, K, K, conv_image=32)
  assert int(img3.flags.item()) == int(ops.large_sparse_image(L, 32).flags.item())
  Ac = L[..., 0].contiguous()                                # one operator, contiguous rows
  D4, V4, img4 = ops.lanczos_ritz_kstep(Ac, None, K, K, conv_image=64)
  ref4 = ops.large_sparse_image(Ac.unsqueeze(3), 64)
  assert int(img4.flags.item()) == 0 and torch.equal(img4.counts, ref4.counts)
  assert torch.equal(img4.entries[keep], ref4.entries[keep])
  assert (D4 - D0).abs().max() < 1e-6
  Lo = L[:, :510, :510]                                      # not read in place: no image
  assert ops.lanczos_ritz_kstep(Lo[..., 0], None, K, K, conv_image=64)[2] is None


def test_collate_leaves_the_image_on_L_and_the_module_uses_it():
  """`collate_graph_adjacency` on graphs beyond 192 nodes: ONE pass over L gives the Ritz pairs and
  leaves the conv's image riding on the tensor; the module (bf16 mode) finds it — no image launch of
  its own — and scores as with its own image; an in-place write to L afterwards invalidates it."""
  import warnings
  from large_fixture import adjacency
  from lanczosnet_amd import ops
  from lanczosnet_amd.dataset.graph_data import collate_graph_adjacency
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  B, N, K = 3, 256, 32
  cfg, P, X, L, mask = general_inputs(B, N, K, 3, 7, 8.0 / N)
  adj = adjacency(B, N, 8.0 / N, 7)
  items = [dict(adjs=adj[b][:, :, None].astype(np.float32), node_feat=X[b], label=np.zeros((1, 2)))
           for b in range(B)]
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    batch = collate_graph_adjacency(items, K, device=DEV)
  Ld = batch['L']
  img = ops.attached_sparse_image(Ld)
  assert img is not None and int(img.flags.item()) == 0 and img.cap == ops.large_sparse_row_cap(N)
  own = ops.large_sparse_image(Ld)
  assert torch.equal(img.counts, own.counts)
  net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  net.gemm_mode = 'bf16'
  md = torch.from_numpy(mask).to(DEV)
  calls = []
  orig = ops.large_sparse_image

  def spy(*a, **kw):
    calls.append(1)
    return orig(*a, **kw)
  ops.large_sparse_image = spy
  try:
    with torch.no_grad():
      s1 = net(batch['node_feat'], Ld, batch['D'], batch['V'], mask=md)
      st = net._large_sparse_state[Ld.device.index]
      assert not calls and st['image_from'] == 'collate' and st['last_flags'] == 0
      net.gemm_mode = 'fp32'                                  # the fp32 modes find it too (its fp32 values)
      s1f = net(batch['node_feat'], Ld, batch['D'], batch['V'], mask=md)
      assert not calls and st['image_from'] == 'collate' and ops.last_kernel() == 'sparse_conv_f32_kernel'
      assert (s1f - s1).abs().max() <= 2e-2 * s1f.abs().max()
      net.gemm_mode = 'bf16'
      s2 = net(batch['node_feat'], Ld.clone(), batch['D'], batch['V'], mask=md)    # a copy does not carry it
      assert len(calls) == 1 and st['image_from'] == 'forward'
      assert (s1 - s2).abs().max() <= 2e-3 * s2.abs().max()
      Ld[0, 0, 0, :] += 0.0                                   # an in-place write: the version moves on
      assert ops.attached_sparse_image(Ld) is None
      net(batch['node_feat'], Ld, batch['D'], batch['V'], mask=md)
      assert len(calls) == 2 and st['image_from'] == 'forward'
  finally:
    ops.large_sparse_image = orig


@pytest.mark.parametrize('B,N,P', [(3, 300, 2), (2, 2048, 2), (5, 77, 5), (2, 130, 16), (1, 33, 1)])
def test_large_head_matches_the_torch_readout(B, N, P):
  """lnz_large_head = (W_h x + b_h) * sigmoid(w_g x + b_g), mean over the real nodes (ragged masks,
  holes in the mask), against the fp64 torch form at 1e-6; deterministic."""
  from lanczosnet_amd import ops
  g = torch.Generator(device=DEV).manual_seed(N + P)
  X = torch.randn((B, N, 128), device=DEV, generator=g)
  Wh = torch.randn((P + 1, 128), device=DEV, generator=g) / 11.0
  bh = torch.randn((P + 1,), device=DEV, generator=g)
  mask = (torch.rand((B, N), device=DEV, generator=g) < 0.8).to(torch.uint8)
  mask[0, N // 2:] = 0
  mask[0, 0] = 1
  got = ops.large_head(X, mask, Wh, bh)
  assert torch.equal(got, ops.large_head(X, mask, Wh, bh))
  Xd, Wd, bd = X.double(), Wh.double(), bh.double()
  y = (Xd @ Wd[:P].T + bd[:P]) * torch.sigmoid(Xd @ Wd[P:].T + bd[P:])
  m = (mask != 0).double().unsqueeze(2)
  ref = (y * m).sum(1) / m.sum(1)
  assert (got.double() - ref).abs().max() <= 1e-6 * ref.abs().max()
