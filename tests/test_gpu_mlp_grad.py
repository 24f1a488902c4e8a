"""lnz_spectral_mlp_grad (csrc/spectral_gains_grad.hip): the parameter gradients of the spectral-filter
MLPs of all conv layers from dG in one launch, against float64 autograd through the same
nn.Sequential chain (the reference's own route: model/lanczos_net.py:95-123,146-149), on row lists
with ragged tails, with and without the live-row list, for other S and layer counts; and the
training step that uses it against the one that keeps autograd + library GEMMs."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _mlps(rs, L, S):
  return [[(torch.from_numpy((rs.randn(o, i) / np.sqrt(i)).astype(np.float32)).to(DEV),
            torch.from_numpy((0.1 * rs.randn(o)).astype(np.float32)).to(DEV))
           for (o, i) in ((128, S), (128, 128), (128, 128), (S, 128))] for _ in range(L)]


def _autograd64(D, dist, layers, dG, idx):
  pows = torch.stack([torch.pow(D.double(), p) for p in dist], dim=2).view(-1, len(dist))
  out = []
  for l, lins in enumerate(layers):
    ps = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for (w, b) in lins]
    h = pows[idx]
    for i, (w, b) in enumerate(ps):
      h = h @ w.t() + b
      if i < 3:
        h = torch.relu(h)
    g = torch.autograd.grad(h, [t for wb in ps for t in wb], dG[l].double()[idx])
    out.append(g)
  return out   # [layer][W0, b0, W2, b2, W4, b4, W6, b6]


def _off_the_kink(D, dist, layers, dG, idx, margin=1e-5):
  """Zero the incoming gradient of every (layer, row) in which some hidden unit's pre-activation lies
  within `margin` of the ReLU kink.  Such a unit takes either side in fp32 — in the kernel and in
  float32 autograd alike — which moves the gradients upstream of it by the row's share (1e-3 at these
  row counts; found by tools/experiments/train_kernels_fuzz.py): float64 is an oracle only off the
  kink.  With a zero incoming gradient the row still runs through the kernel but contributes nothing
  on either side.  Returns the fraction of (layer, row) pairs zeroed."""
  pows = torch.stack([torch.pow(D.double(), p) for p in dist], dim=2).view(-1, len(dist))[idx]
  zeroed = 0
  for l, lins in enumerate(layers):
    h = pows
    near = torch.zeros(h.shape[0], dtype=torch.bool, device=h.device)
    for (w, b) in lins[:3]:
      z = h @ w.double().t() + b.double()
      near |= (z.abs() < margin).any(dim=1)
      h = torch.relu(z)
    dG[l, idx[near]] = 0.0
    zeroed += int(near.sum())
  return zeroed / float(len(layers) * max(1, idx.numel()))


@pytest.mark.parametrize('B,K,S,L,live', [(1024, 20, 7, 7, True), (1024, 20, 7, 7, False), (37, 20, 7, 2, True),
                                          (3, 8, 1, 1, False), (200, 12, 8, 16, True), (64, 20, 3, 5, True)])
def test_mlp_grad_matches_float64_autograd(B, K, S, L, live):
  from lanczosnet_amd import ops
  rs = np.random.RandomState(B + S)
  dist = [1, 2, 3, 5, 7, 10, 20, 30][:S]
  D = torch.from_numpy((rs.rand(B, K) * 1.9 - 0.95).astype(np.float32)).to(DEV)   # |lambda(L4)| <= 1
  layers = _mlps(rs, L, S)
  dG = torch.from_numpy(rs.randn(L, B * K, S).astype(np.float32)).to(DEV)
  rows = None
  idx = torch.arange(B * K, device=DEV)
  if live:   # a ragged subset in ascending order, as lnz_plan_batch lists the live eigen slots
    n = rs.randint(1, K + 1, size=B)
    keep = np.concatenate([b * K + np.arange(n[b]) for b in range(B)]).astype(np.int32)
    buf = np.full(B * K, -7, np.int32)
    buf[:len(keep)] = keep
    rows = (torch.from_numpy(buf).to(DEV), torch.tensor([len(keep)], dtype=torch.int32, device=DEV))
    idx = torch.from_numpy(keep.astype(np.int64)).to(DEV)
  frac = _off_the_kink(D, dist, layers, dG, idx)
  assert frac < 0.05, frac
  got = ops.spectral_mlp_grad(D, dist, layers, dG, rows=rows, rows_max=int(idx.numel()))
  want = _autograd64(D, dist, layers, dG, idx)
  worst = 0.0
  for l in range(L):
    for li in range(4):
      for which in range(2):
        g = got[li][which][l].double()
        w = want[l][2 * li + which]
        assert g.shape == w.shape
        e = float((g - w).abs().max() / w.abs().max().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 2e-5, (l, li, which, e)
  print('spectral MLP gradients vs float64 autograd: worst %.2e of max |g| (B=%d K=%d S=%d L=%d rows=%d; %.2f %% of the '
        '(layer, row) pairs on the ReLU kink: incoming gradient zeroed)' % (worst, B, K, S, L, int(idx.numel()), 100 * frac))
  # deterministic: partials are added in a fixed order
  again = ops.spectral_mlp_grad(D, dist, layers, dG, rows=rows, rows_max=int(idx.numel()))
  for a, b in zip(got, again):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_training_step_with_the_mlp_grad_kernel_matches_the_autograd_route():
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net = LanczosNet(make_model_config(cfg)).train()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 9).items()})
  net = net.to(DEV)
  b = draw_batch(300, seed=2, n_min=2)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  grads = {}
  for impl in ('hip', 'torch'):
    net.mlp_grad_impl = impl
    net.zero_grad(set_to_none=True)
    _, loss = net(t(b['node_feat']), L, D, V, label=t(b['label']), mask=t(b['node_mask']))
    loss.backward()
    grads[impl] = {k: p.grad.clone() for k, p in net.named_parameters()}
  for k in grads['hip']:
    a, w = grads['hip'][k].double(), grads['torch'][k].double()
    assert float((a - w).abs().max()) <= 2e-5 * float(w.abs().max()) + 1e-12, k
    if 'spectral_filter' not in k:
      assert torch.equal(grads['hip'][k], grads['torch'][k]), k   # (everything else is the same code)


@pytest.mark.parametrize('B,N,width,atoms,chunks', [(1024, 26, 64, 70, 32), (3, 5, 16, 4, 7), (200, 32, 128, 9, 1),
                                                   (77, 19, 32, 70, 5)])
def test_embedding_grad_matches_index_add_in_float64(B, N, width, atoms, chunks):
  """lnz_embedding_grad against index_add in float64 (out-of-range ids clamp as in the forward), on a
  strided dX_0 block ([B, 32, width] rows, N of them used); repeats are bit-identical."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(B)
  ids = rs.randint(-1, atoms + 1, size=(B, N)).astype(np.int64)
  dx = torch.from_numpy(rs.randn(B, 32, width).astype(np.float32)).to(DEV)
  got = ops.embedding_grad(torch.from_numpy(ids).to(DEV), dx, width, atoms, chunks=chunks)
  want = torch.zeros((atoms, width), dtype=torch.float64, device=DEV)
  want.index_add_(0, torch.from_numpy(np.clip(ids, 0, atoms - 1)).to(DEV).reshape(-1), dx[:, :N].double().reshape(-1, width))
  assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
  assert torch.equal(got, ops.embedding_grad(torch.from_numpy(ids).to(DEV), dx, width, atoms, chunks=chunks))
