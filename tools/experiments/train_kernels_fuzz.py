"""Random shapes through lnz_spectral_mlp_grad and lnz_embedding_grad against float64 references."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from lanczosnet_amd import ops
from test_gpu_mlp_grad import _mlps, _autograd64, _off_the_kink
DEV = 'cuda:0'
worst = [0.0, 0.0]
for seed in range(int(os.environ.get('FUZZ_FROM', '0')), int(os.environ.get('FUZZ_TO', '60'))):
  rs = np.random.RandomState(1000 + seed)
  B, K, S, L = int(rs.randint(1, 700)), int(rs.choice([1, 4, 12, 20, 32])), int(rs.randint(1, 9)), int(rs.randint(1, 17))
  dist = sorted(rs.choice(np.arange(1, 31), size=S, replace=False).tolist())
  D = torch.from_numpy((rs.rand(B, K) * 1.9 - 0.95).astype(np.float32)).to(DEV)
  layers = _mlps(rs, L, S)
  dG = torch.from_numpy(rs.randn(L, B * K, S).astype(np.float32)).to(DEV)
  rows, idx = None, torch.arange(B * K, device=DEV)
  if rs.rand() < 0.6:
    n = rs.randint(0, K + 1, size=B)
    if n.sum() == 0:
      n[0] = 1
    keep = np.concatenate([b * K + np.arange(n[b]) for b in range(B)]).astype(np.int32)
    buf = np.full(B * K, -7, np.int32); buf[:len(keep)] = keep
    rows = (torch.from_numpy(buf).to(DEV), torch.tensor([len(keep)], dtype=torch.int32, device=DEV))
    idx = torch.from_numpy(keep.astype(np.int64)).to(DEV)
  if os.environ.get('OFF_KINK', '1') == '1':
    _off_the_kink(D, dist, layers, dG, idx)
  got = ops.spectral_mlp_grad(D, dist, layers, dG, rows=rows, rows_max=int(idx.numel()))
  want = _autograd64(D, dist, layers, dG, idx)
  e = max(float((got[li][w][l].double() - want[l][2 * li + w]).abs().max() / want[l][2 * li + w].abs().max().clamp_min(1e-30))
          for l in range(L) for li in range(4) for w in range(2))
  worst[0] = max(worst[0], e)
  if e > 2e-5:
    errs = {(l, li, w): float((got[li][w][l].double() - want[l][2 * li + w]).abs().max() / want[l][2 * li + w].abs().max().clamp_min(1e-30))
            for l in range(L) for li in range(4) for w in range(2)}
    bad = sorted([k for k, v in errs.items() if v > 2e-5])
    # is it the ReLU kink?  smallest |pre-activation| over the bad layers' hidden units in float64, and
    # the same gradients by float32 autograd
    pows = torch.stack([torch.pow(D.double(), p) for p in dist], dim=2).view(-1, S)[idx]
    for l in sorted({k[0] for k in bad}):
      h, zmin = pows, []
      for i, (w, b_) in enumerate(layers[l][:3]):
        z = h @ w.double().t() + b_.double()
        zmin.append(float(z.abs().min()))
        h = torch.relu(z)
      ps = [(w.clone().requires_grad_(True), b_.clone().requires_grad_(True)) for (w, b_) in layers[l]]
      h32 = pows.float()
      for i, (w, b_) in enumerate(ps):
        h32 = h32 @ w.t() + b_
        if i < 3: h32 = torch.relu(h32)
      g32 = torch.autograd.grad(h32, [t_ for wb in ps for t_ in wb], dG[l][idx])
      e32 = max(float((got[li][w_][l] - g32[2 * li + w_]).abs().max() / g32[2 * li + w_].abs().max().clamp_min(1e-30))
                for li in range(4) for w_ in range(2))
      print('   layer %d: min |z| of the three hidden layers (float64) %s ; kernel vs float32 autograd %.2e' % (l, ['%.1e' % z for z in zmin], e32))
    print('MLP FAIL seed', seed, 'B K S L', B, K, S, L, 'rows', int(idx.numel()), 'live' if rows is not None else 'all', 'dist', dist,
          'worst %.2e' % e, 'bad (layer, linear, w|b):', bad[:12], len(bad))
  # embedding gradient
  N, width, atoms, chunks = int(rs.randint(1, 33)), int(rs.choice([16, 32, 64, 128])), int(rs.randint(1, 90)), int(rs.randint(1, 70))
  ids = rs.randint(-1, atoms + 1, size=(B, N)).astype(np.int64)
  dx = torch.from_numpy(rs.randn(B, 32, width).astype(np.float32)).to(DEV)
  g = ops.embedding_grad(torch.from_numpy(ids).to(DEV), dx, width, atoms, chunks=chunks)
  w64 = torch.zeros((atoms, width), dtype=torch.float64, device=DEV)
  w64.index_add_(0, torch.from_numpy(np.clip(ids, 0, atoms - 1)).to(DEV).reshape(-1), dx[:, :N].double().reshape(-1, width))
  e2 = float((g.double() - w64).abs().max() / w64.abs().max().clamp_min(1e-30))
  worst[1] = max(worst[1], e2)
  if e2 > 5e-6: print('EMB FAIL seed', seed, B, N, width, atoms, chunks, e2)
print('worst mlp %.2e  embedding %.2e' % tuple(worst))
