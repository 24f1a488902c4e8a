#!/usr/bin/env python
"""Random shapes through the sparse large-graph path (csrc/conv_sparse.hip and the K-step entry's
image output): node counts that are odd / not multiples of 4 / 64, densities from empty to beyond
the row capacity, every L layout, ragged graphs.  Checks per case: the image against torch (counts,
entry sets, bf16 values, flags), lnz_lanczos_ritz_kstep_image's image against lnz_large_sparse_image's
bit for bit, the sparse layer against the streamed layer on a shared T (1e-5).

    python tools/experiments/sparse_fuzz.py [--cases 200] [--seed 0]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanczosnet_amd import ops  # noqa: E402

DEV = 'cuda:0'


def bf16(x):
  return x.to(torch.bfloat16).to(torch.float32)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--cases', type=int, default=200)
  ap.add_argument('--seed', type=int, default=0)
  a = ap.parse_args()
  rs = np.random.RandomState(a.seed)
  g = torch.Generator(device=DEV).manual_seed(a.seed)
  stats = dict(cases=0, image_checked=0, kstep_image_checked=0, layers_checked=0, flagged=0, worst_layer_rel=0.0,
               failures=[])
  for case in range(a.cases):
    B = int(rs.randint(1, 6))
    N = int(rs.choice([33, 64, 100, 130, 200, 256, 301, 512, 777, 1024]))
    C = int(rs.choice([1, 2, 2, 2, 3]))
    p = float(rs.choice([0.0, 0.005, 0.02, 0.05, 0.2]))
    layout = str(rs.choice(['channels_last', 'channel_major', 'expanded', 'padded_parent']))
    cap = int(rs.choice([32, 40, 64, 128]))
    A = (torch.rand((B, N, N), device=DEV, generator=g) < p).float() * torch.randn((B, N, N), device=DEV, generator=g)
    A = A + A.transpose(1, 2)
    if rs.rand() < 0.5:                                   # ragged: zero padding beyond n_b
      for b in range(B):
        n = int(rs.randint(1, N + 1))
        A[b, n:, :] = 0
        A[b, :, n:] = 0
    if layout == 'channels_last':
      L = torch.stack([A] * C, dim=3).contiguous()
    elif layout == 'channel_major':
      L = torch.stack([A] * C, dim=1).contiguous().permute(0, 2, 3, 1)
    elif layout == 'expanded':
      L = A.unsqueeze(3).expand(B, N, N, C)
    else:
      big = torch.zeros((B, N, N + 4, C), device=DEV)
      big[:, :, :N] = torch.stack([A] * C, dim=3)
      L = big[:, :, :N]
    differ = bool(C > 1 and layout != 'expanded' and rs.rand() < 0.15)
    if differ:
      L = L.clone() if layout == 'padded_parent' else L
      L[int(rs.randint(B)), int(rs.randint(N)), int(rs.randint(N)), C - 1] += 1.0
    tag = dict(case=case, B=B, N=N, C=C, p=p, layout=layout, cap=cap, differ=differ)
    try:
      img = ops.large_sparse_image(L, cap)
      nnz = (A != 0).sum(2)
      want = (1 if differ else 0) | (2 if int(nnz.max()) > cap else 0)
      flags = int(img.flags.item())
      assert flags == want, ('flags', flags, want)
      stats['cases'] += 1
      if flags:
        stats['flagged'] += 1
      if not (flags & 2):
        assert torch.equal(img.counts.long(), nnz), 'counts'
        ent = img.entries.view(torch.int32).long() & 0xffffffff
        k = torch.arange(cap, device=DEV)[None, None, :]
        live = k < nnz[:, :, None]
        cols = ent & 0xffff
        vals = ((ent >> 16) << 16).to(torch.int32).view(torch.float32) if False else None
        # scatter the entries back into a dense matrix of bf16 values
        dense = torch.zeros((B, N, N), device=DEV)
        hit = torch.zeros((B, N, N), device=DEV)
        bi, ri, ki = torch.nonzero(live, as_tuple=True)
        cc = cols[bi, ri, ki]
        vv = (ent[bi, ri, ki] & 0xffff0000).to(torch.int64)
        vv = torch.where(vv >= 2 ** 31, vv - 2 ** 32, vv).to(torch.int32).view(torch.float32)
        dense[bi, ri, cc] = vv
        hit[bi, ri, cc] += 1
        assert float(hit.max()) <= 1.0, 'duplicate column'
        assert torch.equal(dense, bf16(A)), 'values'
        pad = (k >= nnz[:, :, None]) & (k < ((nnz + 7) // 8 * 8)[:, :, None])
        assert bool((ent[pad] == 0).all()), 'padding'
        stats['image_checked'] += 1
      # the K-step entry's image (layouts it reads in place)
      A0 = L[:, :, :, 0]
      pair = C == 2 and L.stride(3) == 1 and L.stride(2) == 2
      single = L.stride(2) == 1 and (C == 1 or L.stride(3) == 0)
      if (pair or single) and N % 4 == 0 and A0.stride(1) % 4 == 0 and A0.stride(0) % 4 == 0 and \
          A0.data_ptr() % 16 == 0 and N >= 64:
        M = 16
        out = ops.lanczos_ritz_kstep(A0, None, M, M, conv_image=cap)
        img2 = out[2]
        if img2 is not None:
          f2 = int(img2.flags.item())
          assert f2 == (want if pair else (want & 2)), ('kstep flags', f2, want)
          if not (want & 2):
            assert torch.equal(img2.counts, img.counts), 'kstep counts'
            keep = torch.arange(cap, device=DEV)[None, None, :] < ((img.counts + 7) // 8 * 8)[:, :, None]
            assert torch.equal(img2.entries[keep], img.entries[keep]), 'kstep entries'
          stats['kstep_image_checked'] += 1
      # the layer (one operator class, no flag)
      if flags == 0 and C <= 2:
        K, S, din = int(rs.choice([8, 20, 64])), int(rs.choice([1, 3, 8])), int(rs.choice([10, 64, 128]))
        V = torch.randn((B, N, K), device=DEV, generator=g) / N ** 0.5
        G = torch.rand((B, S, K), device=DEV, generator=g)
        X = torch.randn((B, N, din), device=DEV, generator=g)
        W = torch.randn((128, S + C, din), device=DEV, generator=g) / (din * (S + C)) ** 0.5
        bias = torch.randn((128,), device=DEV, generator=g) * 0.1
        dinp = (din + 15) // 16 * 16
        Wc = torch.nn.functional.pad(W, (0, dinp - din))
        Wn = Wc[:, S:].sum(1)
        Wf = ops.large_weight_fragments(ops.split_bf16_planes(Wn.reshape(128, dinp), 1).reshape(1, 128, dinp))
        Wt = ops.pack_rows_k8(Wc[:, :S].reshape(128, S * dinp).contiguous())
        Lb, Vb = ops.large_pack_operators(L, V, 1, chan_src=[0], chan_rep=[0] * C)
        dwork = ops.large_work_buffers(Lb)
        dense_out = ops.large_conv_layer(X, din, Lb, Vb, V, Wf, Wt, G, bias, dwork)
        swork = ops.large_sparse_work_buffers(B, N, DEV)
        swork[1].copy_(dwork[1])
        Vb2 = ops.large_pack_vectors(V, 1)
        sparse_out = ops.large_sparse_conv_layer(X, din, img, Vb2, V, Wf, None, None, bias, swork)
        den = float(dense_out.abs().max())
        rel = float((sparse_out - dense_out).abs().max()) / max(den, 1e-30)
        stats['worst_layer_rel'] = max(stats['worst_layer_rel'], rel)
        assert rel <= 1e-5, ('layer', rel)
        stats['layers_checked'] += 1
    except Exception as e:   # noqa: BLE001
      stats['failures'].append(dict(tag, error=repr(e)[:200]))
  print(json.dumps(stats))


if __name__ == '__main__':
  main()
