"""lnz_head_backward (csrc/head_grad.hip): the readout head's backward — model/lanczos_net.py:185-194
under loss.backward() — in one launch, against torch autograd in float64 on the same stored state:
dY of the last conv layer (through its ReLU, zero on padding and masked rows, and its compact copy),
the head's weight / bias gradients and the column sums of dY.  Then the training step's gradients
with the kernel against the same step with the autograd head (LANCZOSNET_HEAD_GRAD=torch)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('B,N,P,n_wg', [(1024, 26, 16, 256), (37, 32, 31, 256), (5, 9, 1, 3), (300, 20, 2, 64)])
def test_head_backward_matches_float64_autograd(B, N, P, n_wg):
  from lanczosnet_amd import ops
  g = torch.Generator(device=DEV)
  g.manual_seed(B + N + P)
  X = torch.relu(torch.randn((B, 32, 128), generator=g, device=DEV))     # post-ReLU state, ~half zeros
  n = torch.randint(1, N + 1, (B,), generator=g, device=DEV)
  mask = (torch.arange(N, device=DEV)[None, :] < n[:, None])
  mask[0, 0] = True
  if N > 4:
    mask[1, 1] = False                                                     # a hole: masked row below the extent
    mask[1, :1] = True
  X = X * (torch.arange(32, device=DEV)[None, :, None] < n[:, None, None])
  W = torch.randn((P + 1, 128), generator=g, device=DEV) * 0.1
  bh = torch.randn((P + 1,), generator=g, device=DEV) * 0.1
  gs = torch.randn((B, P), generator=g, device=DEV)
  mask_u8 = mask.to(torch.uint8).contiguous()
  extent = (mask_u8.long() * torch.arange(1, N + 1, device=DEV)).amax(dim=1)
  row_end = torch.cumsum(extent, 0)
  row_off = (row_end - extent).contiguous()
  R = int(row_end[-1])
  dY = torch.full((B, 32, 128), float('nan'), device=DEV)
  dYc = torch.full((R, 128), float('nan'), device=DEV)
  dW, db, dbl = ops.head_backward(X, mask_u8, gs, W, bh, N, dY, row_off=row_off, dY_compact=dYc, n_wg=n_wg)
  dW2, db2, dbl2 = ops.head_backward(X, mask_u8, gs, W, bh, N, dY.clone(), n_wg=n_wg)
  assert torch.equal(dW, dW2) and torch.equal(db, db2) and torch.equal(dbl, dbl2)      # deterministic
  # float64 autograd on the same state
  X64 = X[:, :N].double().requires_grad_(True)
  W64, b64 = W.double().requires_grad_(True), bh.double().requires_grad_(True)
  Z = torch.nn.functional.linear(X64, W64, b64)
  y = Z[..., :P] * torch.sigmoid(Z[..., P:])
  mk = mask.double().unsqueeze(2)
  score = (y * mk).sum(dim=1) / mk.sum(dim=1)
  gX, gW, gb = torch.autograd.grad(score, [X64, W64, b64], gs.double())
  ref = torch.zeros((B, 32, 128), dtype=torch.float64, device=DEV)
  ref[:, :N] = gX * (X[:, :N] > 0)
  rel = lambda a, b_: float((a.double() - b_).abs().max() / b_.abs().max())  # noqa: E731
  assert torch.isfinite(dY).all() and torch.isfinite(dYc).all()
  assert rel(dY, ref) < 2e-6
  assert rel(dW, gW) < 2e-6 and rel(db, gb) < 2e-6
  assert rel(dbl, ref.sum(dim=(0, 1))) < 2e-6
  for b in (0, 1, B - 1):
    lo, e = int(row_off[b]), int(extent[b])
    assert torch.equal(dYc[lo:lo + e], dY[b, :e])                          # the compact copy, bit for bit
  assert (dY[:, N:] == 0).all()


def test_training_gradients_with_the_head_kernel_equal_the_autograd_head():
  """One LanczosNet training step at the bench shape: every parameter gradient with lnz_head_backward
  against the same step with the torch-autograd head (same HIP kernels elsewhere)."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  b = draw_batch(256, seed=4)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  grads = {}
  for impl in ('hip', 'torch'):
    net = LanczosNet(make_model_config(cfg)).train()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_lanczosnet_params(cfg, 1).items()})
    net = net.to(DEV)
    net.head_grad_impl = impl
    _, loss = net(t(b['node_feat']), L, D, V, label=t(b['label']), mask=t(b['node_mask']))
    loss.backward()
    grads[impl] = {k: p.grad.double().clone() for k, p in net.named_parameters()}
  worst = max(float((grads['hip'][k] - grads['torch'][k]).abs().max() / grads['torch'][k].abs().max())
              for k in grads['torch'])
  print('worst relative gradient deviation, head kernel vs autograd head: %.2e' % worst)
  assert worst < 5e-6


@pytest.mark.parametrize('B,N', [(1024, 26), (3000, 32), (1, 5), (70, 100)])
def test_node_extents_one_launch(B, N):
  """lnz_node_extents: extents (last real node + 1, holes in the mask included), their exclusive prefix
  sums and the total — the compact row numbering the training kernels share."""
  from lanczosnet_amd import ops
  g = torch.Generator(device=DEV)
  g.manual_seed(B)
  mask = (torch.rand((B, N), generator=g, device=DEV) < 0.6).to(torch.uint8)
  mask[0] = 0
  ext, off, tot = ops.node_extents(mask)
  want = (mask.long() * torch.arange(1, N + 1, device=DEV)).amax(dim=1)
  assert torch.equal(ext, want)
  assert torch.equal(off, torch.cumsum(want, 0) - want) and int(tot) == int(want.sum())
