#!/usr/bin/env python
"""Config-5 conv layer on the sparse image (csrc/conv_sparse.hip) against the streamed layer
(csrc/conv_large.hip) and an fp32 torch layer: parity on a small batch, per-launch times on the
full one.

    python tools/experiments/bench_sparse_conv.py [--batch 256] [--nodes 2048] [--density 0.01]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanczosnet_amd import ops  # noqa: E402


def make_batch(B, N, p, K, S, din, dev, seed=0):
  g = torch.Generator(device=dev).manual_seed(seed)
  A = (torch.rand((B, N, N), device=dev, generator=g) < p).float()
  A = torch.triu(A, 1)
  A = A + A.transpose(1, 2)
  deg = A.sum(2).clamp_min(1.0)
  Lm = A * deg.rsqrt().unsqueeze(2) * deg.rsqrt().unsqueeze(1)
  Lm = Lm + torch.eye(N, device=dev).unsqueeze(0) * 0.5
  L = torch.stack([Lm, Lm], dim=3).contiguous()
  V = torch.randn((B, N, K), device=dev, generator=g) / N ** 0.5
  G = torch.rand((B, S, K), device=dev, generator=g)
  X = torch.randn((B, N, din), device=dev, generator=g)
  W = torch.randn((128, S + 2, din), device=dev, generator=g) / (din * (S + 2)) ** 0.5
  bias = torch.randn((128,), device=dev, generator=g) * 0.1
  return L, V, G, X, W, bias


def weights(W, S, planes=1):
  dout, _, din = W.shape
  dinp = (din + 15) // 16 * 16
  Wc = torch.nn.functional.pad(W, (0, dinp - din))
  Wn = (Wc[:, S] + Wc[:, S + 1]).unsqueeze(1)     # the two equal channels folded
  Wf = ops.large_weight_fragments(ops.split_bf16_planes(Wn.permute(1, 0, 2).reshape(dout, dinp), planes))
  Wt = ops.pack_rows_k8(Wc[:, :S].reshape(dout, S * dinp).contiguous())
  return Wf, Wt


def torch_layer(L, V, G, X, W, bias, S):
  Z = torch.einsum('bnd,ocd->bcno', X, W)
  out = bias.view(1, 1, -1) + torch.bmm(L[..., 0], Z[:, S] + Z[:, S + 1])
  Y = torch.bmm(V.transpose(1, 2), X)                               # [B,K,din]
  T = sum(G[:, s].unsqueeze(2) * torch.einsum('bkd,od->bko', Y, W[:, s]) for s in range(S))
  return torch.relu(out + torch.bmm(V, T))


def timed(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=256)
  ap.add_argument('--nodes', type=int, default=2048)
  ap.add_argument('--density', type=float, default=0.01)
  ap.add_argument('--row-cap', type=int, default=0)
  ap.add_argument('--out', default='')
  a = ap.parse_args()
  dev = torch.device('cuda:0')
  K, S, din, N = 64, 3, 128, a.nodes
  res = {}
  # ---- parity, small batch
  L, V, G, X, W, bias = make_batch(4, N, a.density, K, S, din, dev)
  Wf, Wt = weights(W, S)
  ref = torch_layer(L, V, G, X, W, bias, S)
  Lb, Vb = ops.large_pack_operators(L, V, 1, chan_src=[0], chan_rep=[0, 0])
  dense = ops.large_conv_layer(X, din, Lb, Vb, V, Wf, Wt, G, bias, ops.large_work_buffers(Lb))
  img = ops.large_sparse_image(L, a.row_cap or None)
  res['flags'] = int(img.flags.item())
  cnt = img.counts.long()
  nnz = (L[..., 0] != 0).sum(2)
  res['counts_equal'] = bool(torch.equal(cnt, nnz))
  res['max_row'] = int(cnt.max().item())
  res['mean_row'] = float(cnt.float().mean().item())
  Vb2 = ops.large_pack_vectors(V, 1)
  res['vb_equal'] = bool(torch.equal(Vb2.view(torch.int16), Vb.view(torch.int16)))
  sparse = ops.large_sparse_conv_layer(X, din, img, Vb2, V, Wf, Wt, G, bias,
                                       ops.large_sparse_work_buffers(4, N, dev))
  sc = float(ref.abs().max().item())
  res['err_dense_vs_fp32'] = float((dense - ref).abs().max().item()) / sc
  res['err_sparse_vs_fp32'] = float((sparse - ref).abs().max().item()) / sc
  res['err_sparse_vs_dense'] = float((sparse - dense).abs().max().item()) / sc
  # strided view (expanded single channel) and a differing channel
  Lx = L[..., :1].expand(-1, -1, -1, 2)
  img2 = ops.large_sparse_image(Lx)
  res['expanded_flags'] = int(img2.flags.item())
  res['expanded_counts_equal'] = bool(torch.equal(img2.counts, img.counts))
  L2 = L.clone()
  L2[1, 5, 7, 1] += 1.0
  res['differing_flags'] = int(ops.large_sparse_image(L2).flags.item())
  res['overflow_flags'] = int(ops.large_sparse_image(L, 32).flags.item())
  del L, V, G, X, Lb, Vb, dense, sparse, ref, img, img2, L2, Lx
  torch.cuda.empty_cache()
  # ---- times, full batch
  B = a.batch
  L, V, G, X, W, bias = make_batch(B, N, a.density, K, S, din, dev, seed=1)
  Wf, Wt = weights(W, S)
  t = {}
  t['pack_operators_fold'] = timed(lambda: ops.large_pack_operators(L, V, 1, chan_src=[0], chan_rep=[0, 0]), 5)
  Lb, Vb = ops.large_pack_operators(L, V, 1, chan_src=[0], chan_rep=[0, 0])
  work = ops.large_work_buffers(Lb)
  out = torch.empty((B, N, 128), device=dev)
  t['dense_layer'] = timed(lambda: ops.large_conv_layer(X, din, Lb, Vb, V, Wf, Wt, G, bias, work, out=out))
  t['dense_gemm1'] = timed(lambda: ops.large_gemm1(X, din, Lb, Wf, work[0]))
  t['spectral'] = timed(lambda: ops.large_spectral(X, din, Lb, V, G, Wt, work[2], work[1]))
  t['dense_conv'] = timed(lambda: ops.large_conv(Lb, Vb, work[0], work[1], bias, out=out))
  del Lb, work
  torch.cuda.empty_cache()
  t['sparse_image'] = timed(lambda: ops.large_sparse_image(L, a.row_cap or None), 5)
  Lx = L[..., :1].expand(-1, -1, -1, 2)
  t['sparse_image_expanded_view'] = timed(lambda: ops.large_sparse_image(Lx, a.row_cap or None), 5)
  img = ops.large_sparse_image(L, a.row_cap or None)
  assert int(img.flags.item()) == 0
  t['pack_vectors'] = timed(lambda: ops.large_pack_vectors(V, 1))
  swork = ops.large_sparse_work_buffers(B, N, dev)
  abi = ops._abi()
  t['sparse_layer'] = timed(lambda: ops.large_sparse_conv_layer(X, din, img, Vb, V, Wf, Wt, G, bias, swork, out=out))
  t['sparse_gemm1_rows'] = timed(lambda: abi.large_gemm1_rows(X, din, din, Wf, B, N, swork[0]))
  t['lift'] = timed(lambda: abi.large_conv(None, Vb, None, swork[1], bias, B, N, 0, 1, 0, out))
  t['sparse_conv'] = timed(lambda: abi.large_sparse_conv(img.entries, img.counts, img.cap, swork[0], B, N, 1, out))
  res['ms'] = {k: round(v, 4) for k, v in t.items()}
  res['config'] = dict(B=B, N=N, density=a.density, row_cap=img.cap,
                       mean_row=float(img.counts.float().mean().item()))
  print(json.dumps(res))
  if a.out:
    with open(a.out, 'w') as f:
      f.write(json.dumps(res) + '\n')


if __name__ == '__main__':
  main()
