#!/usr/bin/env python
"""bench.py — molecules/s of the LanczosNet forward hot path on MI355X.

One "step" = one pass of the hot path over one batch of B synthetic QM8-schema molecules that
is already resident in HBM (the channels-last Laplacian L, atom ids, mask):

    batch preparation, ONE launch                         (lnz_prepare_batch)      -> Lp, D, V
      = tile plan + live eigen slots, pack L into MFMA fragment order, Lanczos
        tridiagonalisation + tridiagonal eigensolve (Sturm / twisted factorisation) + select
    spectral filter gains, all 7 layers                   (lnz_spectral_gains_rows)
    fused 7-layer spectral conv + mix + head + readout    (lnz_lanczosnet_forward) -> score
    [N > 1 ranks only] RCCL all-gather of the per-shard scores

i.e. three launches per step.  The JSON line is self-verifying: `parity_rel_err` is the score of
the timed seed-0 batch against the CPU oracle's score of the same batch (the `cpu_baseline` leg
computes it anyway) and the run FAILS if it exceeds 1e-5.

Workload = BASELINE.json configs[1]: QM8 LanczosNet, batch 1024 per GPU, N <= 32 dense L,
K = 20, fp32, config/qm8_lanczos_net.yaml architecture.  Molecules are independent, so ranks
shard the batch with no data-path collective (weak scaling, 1024 molecules per GPU).

Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the roofline arithmetic.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lanczosnet_amd import ops  # noqa: E402
from lanczosnet_amd.model import LanczosNet  # noqa: E402
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402
from lanczosnet_amd.utils.arg_helper import make_model_config  # noqa: E402

QM8_CFG = dict(num_atom=70, num_bond_type=6, short_diffusion_dist=[],
               long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30], num_eig_vec=20,
               spectral_filter_kind='MLP', input_dim=64, hidden_dim=[128] * 7, output_dim=16,
               num_layer=7)

# Algorithmic FLOPs per 32-row node tile (= per molecule in the reference's padding) of the fused
# forward kernel in the REFERENCE's association
# (SURVEY.md §8d, N = 32 tile, K = 20, S = 8, E+1 = 7, 64 -> 128 x 7 -> 16):
#   filter build S*2N^2K per layer, long S*2N^2 d, edge (E+1)*2N^2 d, mix 2N(15d)128, head 2N*128*17
FWD_FLOP_PER_MOL = (7 * 8 * 2 * 32 * 32 * 20
                    + 8 * 2 * 1024 * (64 + 6 * 128)
                    + 7 * 2 * 1024 * (64 + 6 * 128)
                    + 2 * 32 * 15 * 128 * (64 + 6 * 128)
                    + 2 * 32 * 128 * 17)  # = 130,228,224
# What the kernel executes since the long-scale channels moved to eigen space
# (V [sum_s diag(g_s) (V^T X) W_s^T]): no filter build, no per-channel L_s Z; one projection
# 2N^2 d_in and one lift 2N^2 d_out per layer instead.
FWD_FLOP_EXECUTED = (FWD_FLOP_PER_MOL - 7 * 8 * 2 * 32 * 32 * 20 - 8 * 2 * 1024 * (64 + 6 * 128)
                     + 2 * 1024 * (64 + 6 * 128) + 7 * 2 * 1024 * 128)  # = 117,841,920
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def cpu_baseline(cfg, params, batch_size, reps):
  """The oracle on the host cores, bounded sample: per-molecule numpy eigh + |lambda| sort (the
  reference's (D, V) producer) and the torch-CPU restatement of LanczosNet.forward — the
  reference's own operator sequence on the same library, hence its rate
  (profiles/cpu_port_vs_reference.json: port / reference on one host)."""
  import oracle
  batch = draw_batch(batch_size, seed=0)
  B, N = batch['node_mask'].shape
  L = np.zeros((B, N, N, 7), np.float32)
  for b in range(B):
    nb = int(batch['n_nodes'][b])
    L[b, :nb, :nb] = oracle.laplacian_multi_l4(batch['adjs'][b, :nb, :nb])
  times = []
  K = cfg['num_eig_vec']
  # a top-K cut through a degenerate |lambda| cluster (n > K) keeps an arbitrary vector of the
  # cluster: basis dependent in the reference itself (LAPACK's choice), excluded from the parity
  # figure as SURVEY.md 8(c) prescribes — and counted.  The rule (oracle.degenerate_cut, gap
  # oracle.CUT_GAP = 1e-7 = the rounding of the fp32 Laplacian the device path is handed) is the
  # one every parity test uses.
  ambiguous = np.zeros(B, bool)
  t_start = time.perf_counter()
  for _ in range(reps):
    if times and time.perf_counter() - t_start > 30.0:
      break  # bounded sample: about 30 s of CPU work at most
    t0 = time.perf_counter()
    Dl, Vl = [], []
    for b in range(B):  # (D, V) producer: utils/data_helper.py:169-223 per molecule, fp64 L4 -> eigh
      nb = int(batch['n_nodes'][b])
      e, V, _ = oracle.graph_laplacian_eigs(batch['adjs'][b, :nb, :nb].sum(axis=2),
                                            graph_laplacian_type='L4')
      Dl.append(e)
      Vl.append(V)
      ambiguous[b] = oracle.degenerate_cut(e, K)
    D, V = oracle.collate_eigs(Dl, Vl, N, cfg['num_eig_vec'])
    score = oracle.lanczos_net_forward_torch(params, cfg, batch['node_feat'], L, D, V,
                                             batch['node_mask'])
    times.append(time.perf_counter() - t0)
  return batch_size / min(times), times, score, ambiguous


def forward_batch_sweep(net, plan, L, node_feat, mask_u8, n_nodes, cfg, sizes, reps=5):
  """Fused-forward launch time at growing batch (the bench batch repeated r times, so the size
  mix — and with it the tile plan's pairing rate — is the same): separates tile quantisation
  (B=1024 is 2.9 tiles per CU, rounded up to 3) from in-loop stalls.  Forward launch only,
  `reps` launches back to back (sustained matrix-core clock: the B=1024 entry is slower than the
  timed loop's `avg_launch_ms`, where the short preparation/gains launches sit between forwards)."""
  out = []
  B0 = L.shape[0]
  K = cfg['num_eig_vec']
  with torch.no_grad():
    for Bs in sizes:
      r = max(1, Bs // B0)
      Lr, nf, mk, nn_ = (x.repeat(*([r] + [1] * (x.dim() - 1))).contiguous()
                         for x in (L, node_feat, mask_u8, n_nodes))
      Lp, tiles, rows, D, V = ops.prepare_batch(plan, Lr, mk, nn_, K)
      G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'],
                             rows=rows, zero_fill=not ops.pairing_supported(plan))
      buf, cap = tiles
      n_tiles = int((buf[:12 * cap].view(cap, 4, 3)[:, :, 0] >= 0).sum().item())
      n_wg = int(buf[12 * cap].item())
      ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
      e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
      e[0].record()
      for _ in range(reps):
        ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
      e[1].record()
      torch.cuda.synchronize()
      ms = e[0].elapsed_time(e[1]) / reps
      tf = FWD_FLOP_EXECUTED * n_tiles / (ms * 1e-3) / 1e12
      kern = 'tiles'
      from lanczosnet_amd.utils.flop_model import strips_selected, strips_from_plan, strip_mfma_issued
      if getattr(buf, 'strips', None) is not None and strips_selected(cfg, B0 * r, Lr.shape[1]):
        # this size runs on the strip plan: flops from the instructions that plan issues
        fm = strip_mfma_issued(strips_from_plan(buf.strips.cpu().numpy(),
                                                Lp.ident.cpu().numpy() if hasattr(Lp, 'ident') else None), cfg)
        tf = fm['mfma_in_touched_blocks'] * 2048 / (ms * 1e-3) / 1e12   # (priced like `roofline`)
        kern, n_tiles, n_wg = 'strips (%d subtiles)' % fm['subtiles'], fm['tiles'], fm['tiles']
      out.append({'batch': B0 * r, 'plan': kern, 'tiles': n_tiles, 'workgroups': n_wg, 'forward_ms': round(ms, 4),
                  'executed_tflops': round(tf, 2), 'frac': round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                  'molecules_per_s_forward_only': round(B0 * r / ms * 1e3, 1)})
      del Lr, Lp, V, G
  return out


def bytes_survey_full(N, M):
  # SURVEY.md 8(d), large-N regime: A re-streamed every step + basis traffic + V
  return M * 4 * N * N + M * M * N * 4 + N * M * 4


def lanczos_large_leg(dev, B=256, N=2048, M=64, reps=3):
  """BASELINE configs[4] Lanczos stage, the HBM-bound regime north_star's ">= 40 % of HBM roofline
  on the Lanczos SpMV" is about: lnz_lanczos_ritz_large on B dense graphs of N nodes, M = K steps.
  Algorithmic bytes per graph (large-N regime): A re-streamed every step M*4N^2, +
  basis traffic M^2*N*4 + N*M*4 (SURVEY's fp32 accounting; this kernel keeps an fp64 basis and
  moves more)."""
  g = torch.Generator(device=dev)
  g.manual_seed(0)
  A = torch.empty((B, N, N), dtype=torch.float32, device=dev)
  eye = torch.eye(N, device=dev)
  for b in range(B):  # G(n, p = 0.01) + self loops, symmetric normalisation (L4)
    adj = (torch.rand((N, N), generator=g, device=dev) < 0.01).float().triu(1)
    adj = adj + adj.t() + eye
    d = adj.sum(1).rsqrt()
    A[b] = d[:, None] * adj * d[None, :]
  ws = torch.empty((ops._abi().lanczos_ritz_large_workspace_bytes(B, N),), dtype=torch.uint8,
                   device=dev)
  ops.lanczos_ritz_large(A, M, M, workspace=ws)
  torch.cuda.synchronize()
  ts = []
  for _ in range(reps):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    D, V, info = ops.lanczos_ritz_large(A, M, M, workspace=ws, return_info=True)
    e[1].record()
    torch.cuda.synchronize()
    ts.append(e[0].elapsed_time(e[1]))
  t = float(np.mean(ts)) * 1e-3
  # the same Ritz pairs from the upper chunk blocks only (lnz_lanczos_ritz_large_sym)
  ops.lanczos_ritz_large(A, M, M, workspace=ws, symmetric=True)
  torch.cuda.synchronize()
  ts_sym = []
  for _ in range(reps):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    Ds, Vs = ops.lanczos_ritz_large(A, M, M, workspace=ws, symmetric=True)
    e[1].record()
    torch.cuda.synchronize()
    ts_sym.append(e[0].elapsed_time(e[1]))
  tsym = float(np.mean(ts_sym)) * 1e-3
  nch = (N + 255) // 256
  bytes_A_sym = M * 4 * 256 * 256 * (nch * (nch + 1) // 2)
  sym = {'kernel': 'lanczos_ritz_large_kernel<true> (lnz_lanczos_ritz_large_sym: 256 x 256 chunk '
                   'blocks (I, J >= I) only; deterministic)',
         'ms': round(tsym * 1e3, 3), 'graphs_per_s': round(B / tsym, 1),
         'algorithmic_bytes_per_graph': bytes_A_sym + M * M * N * 4 + N * M * 4,
         'achieved': round(B * (bytes_A_sym + M * M * N * 4 + N * M * 4) / tsym / 1e9, 1),
         'peak': 8000.0, 'unit': 'GB/s',
         'frac': round(B * (bytes_A_sym + M * M * N * 4 + N * M * 4) / tsym / 8e12, 4),
         'survey_accounting': {
             'note': 'SURVEY.md 8(d) counts the full dense matrix even if the kernel exploits '
                     'symmetry: this kernel does (it streams %d of %d chunk blocks), so the figure '
                     'below is an effective rate, not HBM traffic' % (nch * (nch + 1) // 2, nch * nch),
             'algorithmic_bytes_per_graph': M * 4 * N * N + M * M * N * 4 + N * M * 4,
             'achieved': round(B * (M * 4 * N * N + M * M * N * 4 + N * M * 4) / tsym / 1e9, 1),
             'frac': round(B * (M * 4 * N * N + M * M * N * 4 + N * M * 4) / tsym / 8e12, 4)},
         'max_abs_dev_D_vs_full_stream': float((Ds - D).abs().max()),
         'reps_ms': [round(x, 3) for x in ts_sym]}
  del Ds, Vs
  # the PRODUCT entry (what collate_graph_adjacency / get_graph_laplacian_eigs_batched call for a
  # graph of this size): ops.lanczos_ritz routes N > 192 to lnz_lanczos_ritz_kstep, which reads A
  # once into a sliced-ELL image and runs the M steps on that image
  nn_full = torch.full((B,), N, dtype=torch.int32, device=dev)
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    ops.lanczos_ritz(A, nn_full, M)
    torch.cuda.synchronize()
    ts_k = []
    for _ in range(reps):
      e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
      e[0].record()
      Dk, Vk = ops.lanczos_ritz(A, nn_full, M)
      e[1].record()
      torch.cuda.synchronize()
      ts_k.append(e[0].elapsed_time(e[1]))
  tk = float(np.mean(ts_k)) * 1e-3
  _, _, fb = ops.lanczos_ritz_kstep(A, nn_full, M, M, return_fallback=True)
  nnz = int((A != 0).sum())
  moved = B * (4 * N * N + 8 * N * M * (M + 1) // 2 + 4 * N * M) + M * nnz * 6
  Pk = torch.bmm(Vk[:4].double(), Vk[:4].double().transpose(1, 2))
  Pf = torch.bmm(V[:4].double(), V[:4].double().transpose(1, 2))
  kstep = {'entry': 'ops.lanczos_ritz(A, n_nodes, K) -> lnz_lanczos_ritz_kstep(LNZ_KSTEP_SYMMETRIC | '
                    'LNZ_KSTEP_COMPACT): ell_compact_kernel (A read ONCE, nonzeros of every 64-row slab '
                    'gathered into a sliced-ELL image) + lanczos_ritz_large_kernel<2> (the M steps on '
                    'the image) + the dense stream for graphs with a row beyond the image capacity',
           'ms': round(tk * 1e3, 3), 'graphs_per_s': round(B / tk, 1),
           'dense_fallback_graphs': int(fb.sum()), 'nonzeros_per_graph': nnz // B,
           'bytes_moved_per_launch': moved,
           'bytes_moved_note': 'A once (4 N^2) + the image every step (6 bytes x nonzeros, without the '
                               'slab padding) + the fp64 basis once per Gram-Schmidt pass + V; the dense '
                               'streams move %.1f (full) / %.1f (symmetric) GB for the same pairs'
                               % (B * bytes_survey_full(N, M) / 1e9, B * (bytes_A_sym + M * M * N * 4 + N * M * 4) / 1e9),
           'moved_GBps': round(moved / tk / 1e9, 1),
           'survey_accounting': {
               'note': 'SURVEY.md 8(d) prices the dense matrix re-streamed every step; this path does '
                       'not stream it (99 % of it is zeros) — an effective rate, far above the HBM '
                       'peak, that says how much of the priced traffic the design avoids, not a '
                       'bandwidth',
               'achieved': round(B * bytes_survey_full(N, M) / tk / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
               'frac': round(B * bytes_survey_full(N, M) / tk / 8e12, 3)},
           'max_abs_dev_D_vs_full_stream': float((Dk - D).abs().max()),
           'max_projector_dev_vs_full_stream(4 graphs)': float((Pk - Pf).abs().max()),
           'speedup_vs_symmetric_stream': round(tsym / tk, 2), 'reps_ms': [round(x, 3) for x in ts_k]}
  del Pk, Pf
  D, V = Dk, Vk     # the conv leg below runs on the product entry's pairs
  # Ritz residual of the leading pair (lambda_max = 1 of L4): a correctness witness in the line
  r0 = torch.linalg.norm(torch.bmm(A[:4], V[:4, :, :1]) - V[:4, :, :1] * D[:4, None, :1], dim=1).max()
  bytes_A = M * 4 * N * N
  bytes_survey = bytes_A + M * M * N * 4 + N * M * 4
  del ws
  keep = (A, D, V)
  return keep, {'workload': 'lnz_lanczos_ritz_large: B=%d dense graphs, N=%d, M=K=%d Lanczos steps, fp32 A, '
                      'fp64 arithmetic, G(n,0.01) + I, L4 normalised' % (B, N, M),
          'kernel': 'lanczos_ritz_large_kernel<false>', 'ms': round(t * 1e3, 3),
          'graphs_per_s': round(B / t, 1), 'bound': 'hbm',
          'algorithmic_bytes_per_graph': bytes_survey,
          'achieved': round(B * bytes_survey / t / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
          'frac': round(B * bytes_survey / t / 8e12, 4),
          'A_stream_only_GBps': round(B * bytes_A / t / 1e9, 1),
          'frac_A_stream_only': round(B * bytes_A / t / 8e12, 4),
          'leading_pair_residual': float(r0), 'min_steps_taken': int(info.min()),
          'reps_ms': [round(x, 3) for x in ts], 'symmetric_stream': sym,
          'product_entry_compacted': kstep}


def graph_config_leg(dev, B=64, reps=5):
  """The reference's own graph configuration (config/graph_lanczos_net.yaml with
  dataset/get_graph_data.py:15-49): G(n, 0.5) graphs, n ~ U{20..100}, K = 20, one edge type,
  LanczosNetGeneral 10 -> 7 x 128 -> 2, test batch size 64.  Device pipeline from the raw adjacency:
  L4 (lnz_laplacian_l4) -> full-length Lanczos + parallel tridiagonal eigensolver, one workgroup
  per graph (lnz_lanczos_ritz, csrc/lanczos_ritz_wg.hip) -> forward (streamed kernels, split
  precision).  Next to it the host's numpy.linalg.eigh + |lambda| sort on the same graphs (the
  reference's (D, V) producer, utils/data_helper.py:197-223), 16 graphs."""
  from lanczosnet_amd.model import LanczosNetGeneral
  K = 20
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
  acc = np.zeros(3)
  with torch.no_grad():
    for it in range(reps + 2):
      ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
      ev[0].record()
      L = ops.laplacian_l4(ad, nd)
      ev[1].record()
      D, V, info = ops.lanczos_ritz(L[:, :, :, 0], nd, K, return_info=True)
      ev[2].record()
      score = net(Xd, L, D, V, mask=md)
      ev[3].record()
      torch.cuda.synchronize()
      if it >= 2:
        acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
  acc /= reps
  # the forward's one launch (lnz_midgraph_forward, csrc/conv_mid.hip) on its own, and the streamed
  # large-graph kernels it replaced for this size class
  fwd = {}
  with torch.no_grad():
    def timed(fn, n=20):
      fn()
      torch.cuda.synchronize()
      e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
      e[0].record()
      for _ in range(n):
        fn()
      e[1].record()
      torch.cuda.synchronize()
      return e[0].elapsed_time(e[1]) / n
    if net._mid_hip_supported(N, K, L.shape[3]):
      plan = net._plan_mid()
      mid = plan['mid']
      X0 = torch.nn.functional.pad(Xd, (0, mid['din0p'] - Xd.shape[2])).contiguous()
      G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
      Vc = V.float().contiguous()
      launch_ms = timed(lambda: ops.midgraph_forward(X0, L, Vc, G, md, mid['W'], mid['bias'],
                                                     mid['Whead'], mid['bhead'], cfg['num_layer']))
      # v_mfma_f32_16x16x4_f32 instructions the kernel issues (2048 flop each): per workgroup and
      # layer, R = ceil(N / 16) node subtiles, nk = din / 16 k-blocks, S long scales on eight waves,
      # ONE edge pass (the two channels of L are equal and folded in the kernel)
      R, S = (N + 15) // 16, len(cfg['long_diffusion_dist'])
      def layer_mfma(din):
        nk = din // 16
        y = min(8, nk) * 2 * R * 4
        long_ = max(S, 0) * 4 * nk * 4
        return y + long_ + R * 2 * nk * 4 + R * 2 * R * 4 + R * 2 * 8
      mfma = 4 * B * (layer_mfma(mid['din0p']) + (cfg['num_layer'] - 1) * layer_mfma(128))
      fl = 2048.0 * mfma
      net.mid_graph_kernel = False
      streamed_ms = timed(lambda: net(Xd, L, D, V, mask=md), n=10)
      net.mid_graph_kernel = True
      fwd = {'kernel': 'midgraph_forward_kernel<2> (csrc/conv_mid.hip): every conv layer, the head and the '
                       'readout in ONE launch, a graph on four workgroups (32 output columns each), exact fp32 '
                       'on v_mfma_f32_16x16x4_f32; equal operator channels folded in the kernel',
             'launch_ms': round(launch_ms, 4),
             'roofline': {'bound': 'mfma', 'flops_per_launch_issued': fl,
                          'achieved': round(fl / launch_ms / 1e9, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                          'unit': 'TFLOP/s', 'frac': round(fl / launch_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                          'note': 'issued matrix instructions x 2048 flop / launch time; about 40 % of the launch '
                                  'is the per-layer exchange between the four workgroups of a graph (DESIGN.md 4.5c)'},
             'streamed_large_graph_kernels_ms': round(streamed_ms, 4)}
  # The stream mode: the Ritz launch keeps 64 of 256 compute units busy for 0.45 ms and the forward
  # needs its result — but the NEXT batch's does not.  Two streams: Laplacian + Ritz pairs of batch
  # k + 1 beside the forward of batch k (each half a captured HIP graph, double-buffered outputs,
  # events both ways), on DISJOINT compute units (hipExtStreamCreateWithCUMask: sharing them
  # stretches the Ritz chain to 0.60 ms and the overlap buys nothing).  A secondary line: per-batch
  # time of a stream of batches, not the latency of one.
  pipelined = None
  try:
    from lanczosnet_amd.utils.streams import cu_masked_stream
    with torch.no_grad():
      cu_split = 64
      s_prep, s_fwd = cu_masked_stream(0, cu_split, dev), cu_masked_stream(cu_split, 256, dev)
      torch.cuda.synchronize()
      slots = []
      for i in range(2):
        sl = {'gp': torch.cuda.CUDAGraph(), 'gf': torch.cuda.CUDAGraph()}
        with torch.cuda.stream(s_prep):
          with torch.cuda.graph(sl['gp'], stream=s_prep):
            sl['L'] = ops.laplacian_l4(ad, nd)
            sl['D'], sl['V'] = ops.lanczos_ritz(sl['L'][:, :, :, 0], nd, K)
        with torch.cuda.stream(s_fwd):
          with torch.cuda.graph(sl['gf'], stream=s_fwd):
            sl['score'] = net(Xd, sl['L'], sl['D'], sl['V'], mask=md)
        slots.append(sl)
      torch.cuda.synchronize()

      def prep(sl):
        with torch.cuda.stream(s_prep):
          if 'done' in sl:
            s_prep.wait_event(sl['done'])
          sl['gp'].replay()
          sl['ready'] = torch.cuda.Event()
          sl['ready'].record(s_prep)

      def fwd_(sl):
        with torch.cuda.stream(s_fwd):
          s_fwd.wait_event(sl['ready'])
          sl['gf'].replay()
          sl['done'] = torch.cuda.Event()
          sl['done'].record(s_fwd)
      nb = 100
      import gc
      gc.collect()
      gc.disable()   # (wall-clock windows: no ~75 ms collection pause inside, see main())
      for warm in (True, False):
        prep(slots[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nb):
          prep(slots[(k + 1) & 1])
          fwd_(slots[k & 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb * 1e3
      pipelined = {'ms_per_batch': round(dt, 4), 'graphs_per_s': round(B / dt * 1e3, 1), 'batches': nb,
                   'how': 'two HIP streams on disjoint compute units (%d for L4 + Ritz pairs of batch k+1, %d for '
                          'the forward of batch k), each half a captured HIP graph; wall clock over %d batches'
                          % (cu_split, 256 - cu_split, nb),
                   'scores_equal_sequential': bool(torch.equal(slots[(nb - 1) & 1]['score'], score))}
      del slots
      # the Ritz launch is one latency chain per graph — 256 graphs take as long as 64 — so a loader
      # that collates four batches ahead shares ONE Laplacian + Ritz launch among them
      ad4, nd4 = torch.cat([ad] * 4), torch.cat([nd] * 4)
      g4, s4 = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
      s4.wait_stream(torch.cuda.current_stream())

      def four():
        L4 = ops.laplacian_l4(ad4, nd4)
        D4, V4 = ops.lanczos_ritz(L4[:, :, :, 0], nd4, K)
        return [net(Xd, L4[B * i:B * (i + 1)], D4[B * i:B * (i + 1)], V4[B * i:B * (i + 1)], mask=md)
                for i in range(4)]
      with torch.cuda.stream(s4):
        four()
        torch.cuda.synchronize()
        with torch.cuda.graph(g4, stream=s4):
          outs4 = four()
      torch.cuda.current_stream().wait_stream(s4)
      g4.replay()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(25):
        g4.replay()
      torch.cuda.synchronize()
      pipelined['ritz_launch_shared_by_four_batches'] = {
          'ms_per_batch': round((time.perf_counter() - t0) / 100 * 1e3, 4),
          'scores_equal_sequential': bool(all(torch.equal(o, score) for o in outs4)),
          'how': 'one stream, one captured graph: L4 + Ritz pairs of 4 x %d graphs in one launch each '
                 '(a latency chain per graph: 0.46 ms for 64 or 256 graphs), then the four forwards' % B}
      del g4, outs4
  except Exception as e:   # (a secondary line: the leg's numbers above do not depend on it)
    pipelined = {'error': repr(e)[:200]}
  finally:
    import gc
    gc.enable()
  t0 = time.perf_counter()
  Ln = L[:16, :, :, 0].cpu().numpy().astype(np.float64)
  worst = 0.0
  Dn = D[:16].cpu().numpy()
  t0 = time.perf_counter()
  for b in range(16):
    e, v = np.linalg.eigh(Ln[b, :ns[b], :ns[b]])
    idx = np.argsort(-np.abs(e), kind='mergesort')
    worst = max(worst, float(np.abs(e[idx[:K]] - Dn[b]).max()))
  eigh_ms = (time.perf_counter() - t0) / 16 * 1e3
  bytes_ = float((4.0 * ns.astype(np.float64) ** 2).sum() + B * (4 * K + 4 * N * K))
  return {'workload': 'config/graph_lanczos_net.yaml: B=%d graphs G(n,0.5), n~U{20..100} (N=%d), K=20, '
                      'E+1=2, LanczosNetGeneral 10->7x128->2, exact fp32 (one fused launch for the forward)'
                      % (B, N),
          'stage_ms': {'laplacian_l4': round(acc[0], 4), 'lanczos_ritz(workgroup per graph)':
                       round(acc[1], 4), 'forward': round(acc[2], 4)},
          'ms_per_batch': round(float(acc.sum()), 4),
          'graphs_per_s': round(B / float(acc.sum()) * 1e3, 1),
          'ritz': {'kernel': 'lanczos_ritz_wg_kernel<false>', 'graphs_per_s': round(B / acc[1] * 1e3, 1),
                   'algorithmic_GBps': round(bytes_ / acc[1] / 1e6, 3),
                   'bound': 'latency (one workgroup per graph, %d of 256 CUs busy; serial Lanczos '
                            'recurrence of n steps on four waves, then the section search on all eight)' % B,
                   'qL_fallbacks': int((info >= 256).sum().item()),
                   'max_abs_dD_vs_numpy_eigh_16_graphs': worst,
                   'host_numpy_eigh_ms_per_graph': round(eigh_ms, 4)},
          'forward': fwd,
          'stream_of_batches_two_streams': pipelined,
          'finite': bool(torch.isfinite(score).all())}


def large_graph_leg(dev, A, D, V, reps=3):
  """BASELINE configs[4] conv stage: LanczosNetGeneral (config/graph_lanczos_net.yaml widths:
  input 10, 7 x 128, output 2, E+1 = 2 channels, S = 8 long scales, K = 64) on B dense graphs of
  N = 2048 nodes through the streamed kernels of csrc/conv_large.hip, bf16 operands / fp32
  accumulate; the Ritz pairs are the ones the product entry (ops.lanczos_ritz -> lnz_lanczos_ritz_kstep)
  just produced.
  With ONE edge type (the yaml's num_edge_type: 1) the two channels of the collated L are the same
  operator (reference dataset/graph_data.py:225-262): the pack kernel finds that out while it
  converts (first batch) and from then on packs and streams it once, with the two weight blocks
  summed (model/lanczos_net.py `_large_pack`) — `forward_ms` is that steady state on the
  MATERIALISED [B,N,N,2] tensor of the reference's collate; `forward_expanded_view_ms` the same
  batch handed over as a zero-channel-stride view (what a device-side collate can emit: equality
  is then structural, the pack reads one channel); `forward_unfolded_ms` with folding off.
  [r06] `forward_ms` is the module's default in this mode: the node-space term on the NONZEROS of L
  (csrc/conv_sparse.hip: L read once into a row-by-row image, per layer a gather of Z's rows) —
  `sparse_path` prices its launches; `forward_streamed_ms` is the streamed form (the in-call
  fallback for dense graphs or differing channels), whose numbers follow.
  The dominant kernel of the streamed form (lnz_large_conv) is HBM bound on the packed-operator stream: algorithmic
  bytes per launch = B * (Cd N Nk + N 64 + Cd 128 Nk + 128 64) * 2 (bf16 Lb, Vb, Zt, Tt read once)
  + B N 128 * 4 (X' written), Cd = distinct operators streamed (1); SURVEY 8(d)'s accounting
  counts every channel of L (C = 2) whether or not the kernel has to read it: reported beside it,
  as the symmetric Lanczos leg does."""
  from lanczosnet_amd.model import LanczosNetGeneral
  B, N, _ = A.shape
  K = V.shape[2]
  cfg = dict(num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
             num_eig_vec=K, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7,
             output_dim=2, num_layer=7, num_atom=0)
  torch.manual_seed(1234)
  net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval().to(dev)
  L = torch.stack([A, A], dim=3)          # channels-last collate layout [B,N,N,E+1]
  Lx = A.unsqueeze(3).expand(B, N, N, 2)  # the same batch as a zero-channel-stride view
  g = torch.Generator(device=dev)
  g.manual_seed(1)
  X = torch.randn((B, N, 10), generator=g, device=dev)
  mask = torch.ones((B, N), dtype=torch.uint8, device=dev)
  out = {}

  def timed(fn, n=reps):
    fn()
    ts = []
    for _ in range(n):
      e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
      e[0].record()
      r = fn()
      e[1].record()
      torch.cuda.synchronize()
      ts.append(e[0].elapsed_time(e[1]))
    return float(np.mean(ts)), r
  with torch.no_grad():
    net.large_sparse = False
    net._large_graph_forward_hip(X, L, D, V, mask, planes=1)   # first batch: learns the fold
    for name, planes, Lin, fold, sparse in (('bf16', 1, L, True, False), ('bf16_view', 1, Lx, True, False),
                                            ('split3', 3, L, True, False), ('bf16_unfolded', 1, L, False, False),
                                            ('split3_sparse', 3, L, True, True),
                                            ('sparse', 1, L, True, True), ('sparse_view', 1, Lx, True, True)):
      net.large_fold, net.large_sparse = fold, sparse
      out[name] = timed(lambda: net._large_graph_forward_hip(X, Lin, D, V, mask, planes=planes))
    sparse_flags = net._large_sparse_state[dev.index if dev.index is not None else 0]['last_flags']
    sparse_kernel = ops.last_kernel()
    # the product surface: ONE pass over the collated L for the Ritz pairs and the conv image
    # (dataset/graph_data.py collate_graph_adjacency -> ops.lanczos_ritz_collated), then the forward
    import warnings
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      collated_ms, (Dc, Vc) = timed(lambda: ops.lanczos_ritz_collated(L, None, K))
    fwd_img_ms, s_img = timed(lambda: net._large_graph_forward_hip(X, L, Dc, Vc, mask, planes=1))
    image_from = net._large_sparse_state[dev.index if dev.index is not None else 0]['image_from']
    del L._lnz_sparse_image
    net.large_fold = True
    classes = net._large_fold_classes(L)[0]
    folded = len(set(classes)) == 1
    # the stages of the steady state: pack (both input forms), then the dominant kernel alone on
    # the packed operators (the last layer's buffers)
    src = sorted(set(classes))
    rep = [src.index(c) for c in classes]
    neq = torch.zeros((1,), dtype=torch.int64, device=dev)
    pack_ms, (Lb, Vb) = timed(lambda: ops.large_pack_operators(L, V, 1, chan_src=src, chan_rep=rep, neq=neq))
    pack_view_ms, _ = timed(lambda: ops.large_pack_operators(Lx, V, 1, chan_src=[0], chan_rep=[0, 0],
                                                             chan_check=[1, 0]))
    plan = net._plan_large(1, classes)
    work = ops.large_work_buffers(Lb)
    Zt, Tt, _ = work
    Gs = ops.spectral_gains(D, net.long_diffusion_dist, net.num_layer, plan['mlp_pack'])
    lay = plan['conv'][(1, tuple(classes))][1]
    state = torch.randn((B, N, 128), generator=g, device=dev)
    ops.large_gemm1(state, 128, Lb, lay['Wb'], Zt)
    ops.large_spectral(state, 128, Lb, V, Gs[1], lay['Wt'], work[2], Tt)
    buf = torch.empty((B, N, 128), dtype=torch.float32, device=dev)
    ops.large_conv(Lb, Vb, Zt, Tt, lay['bias'], out=buf)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(7):
      ops.large_conv(Lb, Vb, Zt, Tt, lay['bias'], out=buf)
    e[1].record()
    torch.cuda.synchronize()
    conv_ms = e[0].elapsed_time(e[1]) / 7
    # the sparse path's launches on the same layer
    img_ms, img = timed(lambda: ops.large_sparse_image(L))
    img_view_ms, _ = timed(lambda: ops.large_sparse_image(Lx))
    swork = ops.large_sparse_work_buffers(B, N, dev)
    abi = ops._abi()
    folded_plan = net._plan_large(1, (0, 0))['conv'][(1, (0, 0))][1]

    def per_launch(fn, n=7):
      fn()
      e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
      e[0].record()
      for _ in range(n):
        fn()
      e[1].record()
      torch.cuda.synchronize()
      return e[0].elapsed_time(e[1]) / n
    sp_gemm1 = per_launch(lambda: abi.large_gemm1_rows(state, 128, 128, folded_plan['Wb'], B, N, swork[0]))
    sp_spec = per_launch(lambda: abi.large_spectral(state, 128, 128, V, Gs[1], folded_plan['Wt'], B, N, K, 8, 1,
                                                    swork[2], swork[1]))
    sp_fused = per_launch(lambda: abi.large_spectral_gemm1_rows(state, 128, 128, V, Gs[1], folded_plan['Wt'],
                                                                folded_plan['Wb'], B, N, K, 8, swork[2], swork[1],
                                                                swork[0]))
    sp_lift = per_launch(lambda: abi.large_conv(None, Vb, None, swork[1], lay['bias'], B, N, 0, 1, 0, buf))
    sp_conv = per_launch(lambda: abi.large_sparse_conv(img.entries, img.counts, img.cap, swork[0], B, N, 1, buf))
    cnt = img.counts.float()
    slots = float(((img.counts + 7) // 8 * 8).sum().item())
  Nk = Lb.dims[1]
  Cd = Lb.shape[2]
  alg_of = lambda c: B * (c * N * Nk + N * 64 + c * 128 * Nk + 128 * 64) * 2 + B * N * 128 * 4  # noqa: E731
  alg, alg_full = alg_of(Cd), alg_of(2)
  pack_bytes = B * (2 * N * N * 4 + Cd * N * Nk * 2)
  dev_rel = float((out['bf16'][1] - out['split3'][1]).abs().max() / out['split3'][1].abs().max())
  fold_rel = float((out['bf16'][1] - out['bf16_unfolded'][1]).abs().max() / out['bf16_unfolded'][1].abs().max())
  res = {'workload': 'LanczosNetGeneral conv stack on the graphs of lanczos_large_mode: B=%d, N=%d, '
                     'K=%d, E+1=2, S=8, 10 -> 7 x 128 -> 2, bf16 operands / fp32 accumulate '
                     '(sparse image of L + 7 x [gemm1, eigen-space block, lift, gather] + head; streamed form: pack + 7 x '
                     '[gemm1, eigen-space block, streamed conv] + head)' % (B, N, K),
         'forward_ms': round(out['sparse'][0], 3),
         'graphs_per_s_forward': round(B / out['sparse'][0] * 1e3, 1),
         'forward_path': 'sparse image of L (csrc/conv_sparse.hip; the module\'s default in this mode, '
                         'streamed kernels as the in-call fallback); last dominant kernel: %s' % sparse_kernel,
         'forward_expanded_view_ms': round(out['sparse_view'][0], 3),
         'sparse_vs_streamed_rel': float((out['sparse'][1] - out['bf16'][1]).abs().max() / out['bf16'][1].abs().max()),
         'sparse_path': {
             'image_flags': int(sparse_flags), 'row_cap': int(img.cap),
             'entries_per_row_mean': round(float(cnt.mean().item()), 2), 'entries_per_row_max': int(cnt.max().item()),
             'image_ms': {'materialised_[B,N,N,2]': round(img_ms, 3),
                          'GBps': round(B * 2 * N * N * 4 / img_ms / 1e6, 1),
                          'zero_channel_stride_view': round(img_view_ms, 3),
                          'view_GBps': round(B * N * N * 4 / img_view_ms / 1e6, 1)},
             'layer_ms': {'eigen_space_block+gemm1_rows_in_one_pass_over_X': round(sp_fused, 4),
                          'gemm1_rows_alone': round(sp_gemm1, 4), 'eigen_space_block_alone': round(sp_spec, 4),
                          'lift_launch': round(sp_lift, 4), 'sparse_conv': round(sp_conv, 4)},
             'gather': {'kernel': 'sparse_conv_kernel', 'bound': 'L2 -> L1 path (256 B per nonzero)',
                        'bytes_per_launch': int(slots * 256),
                        'GBps': round(slots * 256 / sp_conv / 1e6, 1)}},
         'product_surface': {
             'what': 'ops.lanczos_ritz_collated(L) [lnz_lanczos_ritz_kstep_image: L[..., 0] of the collated '
                     '[B,N,N,2] tensor read in place, ONCE, for the Ritz pairs and the conv image] + forward',
             'ritz_pairs_and_image_ms': round(collated_ms, 3), 'forward_ms': round(fwd_img_ms, 3),
             'image_from': image_from, 'end_to_end_ms': round(collated_ms + fwd_img_ms, 3),
             'graphs_per_s': round(B / (collated_ms + fwd_img_ms) * 1e3, 1),
             'vs_forward_with_own_image_rel': float((s_img - out['sparse'][1]).abs().max() / out['sparse'][1].abs().max())},
         'forward_streamed_ms': round(out['bf16'][0], 3),
         'forward_streamed_expanded_view_ms': round(out['bf16_view'][0], 3),
         'forward_unfolded_ms': round(out['bf16_unfolded'][0], 3),
         'split_precision_forward_ms': round(out['split3_sparse'][0], 3),
         'split_precision_path': 'the module\'s default in the fp32 modes: sparse image with fp32 values, node-space '
                                 'term in EXACT fp32 (lnz_f32_linear + lnz_large_sparse_conv_f32), lift from 3 bf16 pieces',
         'split_precision_streamed_forward_ms': round(out['split3'][0], 3),
         'split_precision_sparse_vs_streamed_rel': float((out['split3_sparse'][1] - out['split3'][1]).abs().max() /
                                                         out['split3'][1].abs().max()),
         'bf16_vs_split_precision_rel': dev_rel,
         'channel_fold': {'classes': list(classes), 'operators_streamed': Cd, 'of_channels': 2,
                          'folded_vs_unfolded_rel': fold_rel,
                          'how': 'pack kernel compares the channels it converts (bits read back '
                                 'through pinned memory at the next batch, no pipeline sync); the '
                                 'claim is verified in-kernel in every later batch'},
         'pack_ms': {'materialised_[B,N,N,2]': round(pack_ms, 3),
                     'GBps': round(pack_bytes / pack_ms / 1e6, 1),
                     'zero_channel_stride_view': round(pack_view_ms, 3),
                     'view_GBps': round(B * (N * N * 4 + N * Nk * 2) / pack_view_ms / 1e6, 1)},
         'roofline': {'kernel': 'large_conv_kernel<1, 8>', 'bound': 'hbm',
                      'algorithmic_bytes_per_launch': alg, 'avg_launch_ms': round(conv_ms, 4),
                      'achieved': round(alg / conv_ms / 1e6, 1), 'peak': 8000.0, 'unit': 'GB/s',
                      'frac': round(alg / conv_ms / 1e6 / 8000.0, 4),
                      'survey_accounting': {
                          'bytes_per_launch_all_channels': alg_full,
                          'equivalent_GBps': round(alg_full / conv_ms / 1e6, 1),
                          'note': 'SURVEY 8(d) counts every channel of the dense L; the kernel '
                                  'streams each DISTINCT operator once (frac above prices the '
                                  'bytes it actually moves)'}}}
  if not folded:
    res['channel_fold']['note'] = 'fold not taken'
  del net, L, Lb, Vb, work, img, swork
  torch.cuda.empty_cache()
  return res


def ada_leg(dev, L, node_feat, mask_u8, reps=5):
  """BASELINE configs[3]: AdaLanczosNet forward (config/qm8_ada_lanczos_net.yaml architecture:
  short [1,2,3], long [5,7,10,20,30], K=20, 7 x 128) on the bench batch.  Per-stage HIP-event
  times; the filter MLPs (2000-4096-4096-4096-2000 per layer, M = batch) run on the hand-written
  exact-fp32 Linear (lnz_f32_linear; the default since r04) and are priced against the fp32 MFMA
  peak; the vendor-library mode is timed beside it."""
  from lanczosnet_amd.model import AdaLanczosNet
  cfg = dict(QM8_CFG, short_diffusion_dist=[1, 2, 3], long_diffusion_dist=[5, 7, 10, 20, 30])
  torch.manual_seed(1234)
  net = AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet')).eval().to(dev)
  B, N = node_feat.shape
  K, S, nl = cfg['num_eig_vec'], 5, cfg['num_layer']
  names = ['learned_laplacian', 'lanczos_layer', 't_powers', 'filter_mlp_gemms+symmetrize',
           'pack_L', 'fused_forward']
  acc = np.zeros(6)
  tot = []
  with torch.no_grad():
    plan = net._plan()
    for it in range(reps + 2):
      q1 = torch.randn(B, N, 1).to(dev)
      ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
      ev[0].record()
      Le = ops.ada_graph_laplacian(node_feat, net.embedding.weight, L[:, :, :, 0])
      ev[1].record()
      T, Q = ops.ada_lanczos_layer(Le, mask_u8, q1, K)
      ev[2].record()
      tcat = ops.ada_t_powers(T, cfg['long_diffusion_dist']).view(B, -1)
      ev[3].record()
      DDp = net._ada_dense_filters(plan, tcat)
      ev[4].record()
      Lp = ops.pack_laplacian(L)
      ev[5].record()
      score = ops.lanczosnet_forward(plan, node_feat, Lp, Q, DDp, mask_u8)
      conv_kernel_ran = ops.last_kernel()
      ev[6].record()
      torch.cuda.synchronize()
      if it >= 2:
        acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]
        tot.append(ev[0].elapsed_time(ev[6]))
  acc /= reps
  ms = float(np.mean(tot))
  # the other filter-GEMM modes on the same stages and inputs: the opt-in split-precision chain
  # (net.filter_gemm_mode = 'f16x3') and the vendor library (hipBLASLt with its bias + ReLU
  # epilogue, 'fp32': the default until r04, now the comparison)
  default_mode = net.filter_gemm_mode
  split = None

  def run_mode(mode):
    net.filter_gemm_mode = mode
    tt, ff = [], []
    for it in range(reps + 2):
      ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
      ev[0].record()
      Le = ops.ada_graph_laplacian(node_feat, net.embedding.weight, L[:, :, :, 0])
      T, Q = ops.ada_lanczos_layer(Le, mask_u8, q1, K)
      tcat = ops.ada_t_powers(T, cfg['long_diffusion_dist']).view(B, -1)
      ev[1].record()
      DDp = net._ada_dense_filters(plan, tcat)
      ev[2].record()
      Lp = ops.pack_laplacian(L)
      score = ops.lanczosnet_forward(plan, node_feat, Lp, Q, DDp, mask_u8)
      ev[3].record()
      torch.cuda.synchronize()
      if it >= 2:
        tt.append(ev[0].elapsed_time(ev[3]))
        ff.append(ev[1].elapsed_time(ev[2]))
    net.filter_gemm_mode = default_mode
    return tt, ff, DDp, score
  with torch.no_grad():
    score32 = score
    DD32 = DDp
    tl, fl, DDl, scorel = run_mode('fp32')
    library = {'mode': "filter_gemm_mode='fp32': the filter MLPs' Linears in hipBLASLt "
                       '(torch._addmm_activation: bias + ReLU in the GEMM epilogue)',
               'ms_per_step': round(float(np.mean(tl)), 4), 'filter_mlp_ms': round(float(np.mean(fl)), 4),
               'filters_bit_identical_to_default': bool(torch.equal(DDl, DD32)),
               'max_rel_dev_scores_vs_default': float((scorel - score32).abs().max() / score32.abs().max())}
    del DDl, scorel
    t16, f16, DDp, score = run_mode('f16x3')
    split = {'mode': "filter_gemm_mode='f16x3': each operand of the filter MLPs' Linears as two fp16 "
                     'pieces, x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 by the hand-written '
                     'lnz_f16x3_linear chain (csrc/f16x3_linear.hip: v_mfma_f32_32x32x16_f16, the four '
                     'pieces of a k-slice staged once through LDS, bias + ReLU + split fused into the '
                     'epilogue, split-K for the last Linear); everything else as in the default mode '
                     '(opt-in, parity-tested at the same 1e-5 bar)',
             'ms_per_step': round(float(np.mean(t16)), 4),
             'value': round(B / float(np.mean(t16)) * 1e3, 1), 'unit': 'molecules/s',
             'filter_mlp_ms': round(float(np.mean(f16)), 4),
             'max_rel_dev_filters_vs_fp32_mode': float((DDp - DD32).abs().max() / DD32.abs().max()),
             'max_rel_dev_scores_vs_fp32_mode': float((score - score32).abs().max() /
                                                      score32.abs().max())}
    split_score = score
    score = score32
  # self-verification of config 4 on a bounded sample of the timed batch: the scores of both filter
  # GEMM modes against the fp64 oracle (exact arithmetic of the reference's formulas, same q1).
  # An fp32 Lanczos recurrence is a noisy function (the reference's own fp32 run sits 1e-6 .. 2e-4
  # from exact arithmetic), so molecules are classified by the oracle's RAW betas:
  # "well separated" = every beta at least 10x away from the 1e-4 breakdown threshold;
  # within 10x the breakdown decision itself is a coin flip of rounding — counted, not judged.
  parity = None
  try:
    import oracle
    ns_ = min(B, 128)
    Pn = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    ref64, _, betas = oracle.ada_lanczos_net_forward(
        Pn, cfg, node_feat[:ns_].cpu().numpy(), L[:ns_].cpu().numpy(), mask_u8[:ns_].cpu().numpy(),
        q1[:ns_, :, 0].cpu().numpy(), dtype=np.float64, return_raw_betas=True)
    with np.errstate(divide='ignore'):
      sep = np.maximum(betas / 1e-4, 1e-4 / np.maximum(betas, 1e-300)).min(axis=1)
    well = sep >= 10
    parity = {'against': 'oracle.ada_lanczos_net_forward in float64 (exact arithmetic of '
                         'model/ada_lanczos_net.py:289-368, same start vector) on the first %d '
                         'molecules of the timed batch; per molecule max |score - ref| / max |ref_b|' % ns_,
              'well_separated_n': int(well.sum()), 'near_threshold_n': int((~well).sum())}
    for tag, sc in (('fp32', score32), ('f16x3', split_score)):
      e = np.abs(sc[:ns_].cpu().numpy().astype(np.float64) - ref64).max(axis=1) / \
          np.abs(ref64).max(axis=1)
      parity[tag] = {'well_separated_within_1e-5': int((e[well] <= 1e-5).sum()),
                     'well_separated_median': float(np.median(e[well])) if well.any() else None,
                     'well_separated_worst': float(e[well].max()) if well.any() else None,
                     'near_threshold_worst': float(e[~well].max()) if (~well).any() else None}
    parity['note'] = ('the unmodified reference run in fp32 is itself up to ~2e-4 from exact arithmetic '
                      'on well-separated molecules (tests/golden/ada_e2e.npz): the asserted protocol '
                      '— 1e-5 where the reference is within 2e-6 of fp64, else 3x the reference\'s own '
                      'noise — lives in tests/test_gpu_ada.py; these are the raw distances from '
                      'exact arithmetic')
    del Pn
  except Exception as e:  # noqa: BLE001
    parity = {'error': repr(e)[:300]}
  fp = net._ada_filter_plan(plan)  # folded first / last Linear (symmetry + band of T^p)
  n_in, n_out = fp['W1'][0].shape[1], fp['W4'][0].shape[0]
  mlp_flops = nl * 2 * B * (n_in * 4096 + 2 * 4096 * 4096 + 4096 * n_out)
  mlp_flops_reference = nl * 2 * B * (2000 * 4096 + 2 * 4096 * 4096 + 4096 * 2000)
  mlp_tf = mlp_flops / (acc[3] * 1e-3) / 1e12
  finite = bool(torch.isfinite(score).all())
  # training step (loss.backward() + Adam) on the same batch: HIP forward kernels, the HIP
  # conv-stack backward with dense filters (_AdaLanczosNetFusedFunction), library GEMMs for the
  # filter MLPs' gradients, autograd through the fp64 Lanczos-layer graph
  train = None
  try:
    net.train()
    label = torch.zeros((B, cfg['output_dim']), dtype=torch.float32, device=dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-5, fused=True)   # (one multi-tensor kernel per step)

    def train_step():
      opt.zero_grad(set_to_none=True)
      _, loss = net(node_feat, L, label=label, mask=mask_u8)
      loss.backward()
      opt.step()
      return loss
    for _ in range(3):
      train_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
      loss = train_step()
    e1.record()
    torch.cuda.synchronize()
    tms = e0.elapsed_time(e1) / 4
    train = {'ms_per_step': round(tms, 3), 'value': round(B / tms * 1e3, 1), 'unit': 'molecules/s',
             'backward': 'hip' if net._fused_backward_supported() else 'torch restatement',
             'what': 'forward + loss.backward() + torch.optim.Adam(fused=True).step(), 351 M parameters',
             'loss_finite': bool(torch.isfinite(loss))}
    del opt, loss
    net.zero_grad(set_to_none=True)
  except Exception as e:  # noqa: BLE001
    train = {'error': repr(e)[:300]}
  del net, plan
  torch.cuda.empty_cache()
  return {'workload': 'AdaLanczosNet forward, QM8 batch=%d, N<=32, K=20, short [1,2,3], long '
                      '[5,7,10,20,30], 7 x 128, fp32, 351 M parameters' % B,
          'ms_per_step': round(ms, 4), 'value': round(B / ms * 1e3, 1), 'unit': 'molecules/s',
          'stage_ms': {k: round(float(v), 4) for k, v in zip(names, acc)},
          'filter_gemm_mode': default_mode,
          'filter_mlp_gemm': {'kernel': 'lnz_f32_linear (csrc/f32_linear.hip: v_mfma_f32_16x16x4_f32, exact fp32, '
                                        'bias + ReLU in the epilogue, stream-K for the last Linear) — no vendor '
                                        'GEMM in the step' if default_mode == 'fp32_hip' else
                                        'hipBLASLt via torch.nn.functional.linear', 'flops': mlp_flops,
                              'achieved': round(mlp_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': round(mlp_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                              'flops_reference_shapes': mlp_flops_reference,
                              'shapes': '%d-4096-4096-4096-%d (reference: 2000-...-2000; the '
                                        'symmetric, banded T^p and the symmetrised output are '
                                        'folded into the first / last weights)' % (n_in, n_out),
                              'note': 'achieved prices the flops executed; stage time includes '
                                      'the input gather and the 7 output scatters'},
          'conv_kernel': conv_kernel_ran + ' (lnz_last_kernel(): what the launcher selected for this '
                         'run): dense K x K filters in eigen space (Q [sum_s DD_s (Q^T X W_s^T)])',
          'library_filter_gemm_mode': library, 'split_precision_mode': split, 'train_step': train,
          'parity': parity, 'finite': finite}


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher around it: re-run this command line as N ranks
  under torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a free port) and hand
  back its exit status.  The children see RANK / WORLD_SIZE and take the ordinary path."""
  import socket
  import subprocess
  rc = 1
  for attempt in range(3):  # a port taken between the probe and the rendezvous: try another
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               LNZ_BENCH_SELF_SPAWNED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)]
    p = subprocess.run(cmd + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(p.stderr)
    rc = p.returncode
    if rc == 0 or 'ddress already in use' not in p.stderr:
      break
  return rc


def shard_parity(cfg, params, batch, score, n_check):
  """This rank's own shard against the oracle on a bounded sample (the first n_check molecules of
  the shard it timed): the same figure as `parity_rel_err`, normalised by the sample's largest
  reference score, degenerate top-K cuts excluded by the same rule.  Every rank runs it, so that an
  N-rank line verifies N shards and not only seed 0."""
  import oracle
  n_check = min(n_check, batch['node_mask'].shape[0])
  N = batch['node_mask'].shape[1]
  K = cfg['num_eig_vec']
  L = np.zeros((n_check, N, N, 7), np.float32)
  Dl, Vl = [], []
  ambiguous = np.zeros(n_check, bool)
  for b in range(n_check):
    nb = int(batch['n_nodes'][b])
    L[b, :nb, :nb] = oracle.laplacian_multi_l4(batch['adjs'][b, :nb, :nb])
    e, V, _ = oracle.graph_laplacian_eigs(batch['adjs'][b, :nb, :nb].sum(axis=2),
                                          graph_laplacian_type='L4')
    Dl.append(e)
    Vl.append(V)
    ambiguous[b] = oracle.degenerate_cut(e, K)
  D, V = oracle.collate_eigs(Dl, Vl, N, K)
  ref = oracle.lanczos_net_forward_torch(params, cfg, batch['node_feat'][:n_check], L, D, V,
                                         batch['node_mask'][:n_check]).astype(np.float64)
  got = score[:n_check].cpu().numpy().astype(np.float64)
  keep = ~ambiguous
  dev = np.abs(got - ref).max(axis=1) / np.abs(ref).max()
  return float(dev[keep].max()) if keep.any() else 0.0, int(keep.sum())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--batch', type=int, default=1024, help='molecules per GPU')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--pipeline', action='store_true',
                  help='software pipeline over the stream of batches: one launch prepares batch k+1 '
                       'and computes the spectral gains of batch k')
  ap.add_argument('--cpu-reps', type=int, default=5)
  ap.add_argument('--shard-parity', type=int, default=64,
                  help='N > 1: molecules of its own shard every rank checks against the oracle')
  ap.add_argument('--no-secondary', action='store_true',
                  help='skip the secondary legs (large-graph Lanczos, AdaLanczosNet)')
  ap.add_argument('--sweep', action='store_true',
                  help='add config.forward_batch_sweep (fused forward at B = 1024 / 4096 / 16384); '
                       'opt-in because its launches are the SAME kernel as the roofline one and '
                       'would mix into a rocprofv3 --stats average of the default command '
                       '(profiles/r02_c_bench.json holds a run with it)')
  ap.add_argument('--gemm', default='fp32', choices=['fp32', 'f16x3'],
                  help="fp32 = exact fp32 MFMA (headline); f16x3 = opt-in split-precision GEMM1")
  ap.add_argument('--dist', default='auto', choices=['auto', 'nccl'],
                  help="auto: a process group only for --gpus > 1; nccl: initialise the RCCL group "
                       "even at world size 1 (launched through torch.distributed.run) and send every "
                       "step's scores through the real all-gather — how the exchange is exercised on "
                       "a box with one GPU")
  ap.add_argument('--zero-params', action='store_true',
                  help='diagnostic only (power/DVFS probe): all-zero weights; never reported')
  args = ap.parse_args()

  one_dev = os.environ.get('LNZ_BENCH_ONE_DEVICE', '0') == '1'
  launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
  if args.gpus < 1:
    raise SystemExit('bench.py: --gpus must be >= 1')
  # the N ranks need N devices (runner/qm8_runner.py:62 hands nn.DataParallel its `gpus` list the
  # same way): fewer visible is an error, not a silently smaller job
  n_dev = torch.cuda.device_count()
  if not one_dev and n_dev < args.gpus:
    raise SystemExit('bench.py: --gpus %d but only %d HIP device(s) visible; nothing measured '
                     '(LNZ_BENCH_ONE_DEVICE=1 runs the N-rank path on one device over gloo as a '
                     'functional test)' % (args.gpus, n_dev))
  if not launched and args.gpus > 1:
    # plain `python bench.py --gpus N`: become the launcher — N ranks, one per GPU, under
    # torch.distributed.run on the loopback rendezvous; rank 0's JSON line is this process's stdout
    raise SystemExit(spawn_ranks(args.gpus))
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d inside a %d-rank launch (WORLD_SIZE): the two must agree'
                     % (args.gpus, world))
  dist = None
  # LNZ_BENCH_ONE_DEVICE=1 (functional test of the N > 1 path on a box with ONE GPU): every rank
  # uses cuda:0 and the exchange runs on gloo — RCCL refuses two ranks on one device.  Not a
  # measurement mode.
  if one_dev:
    local_rank = 0
  if world > 1:
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if one_dev:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
  force_rccl = False
  if world == 1 and args.dist == 'nccl':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(local_rank)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
    force_rccl = True
  dev = torch.device('cuda', local_rank)
  torch.cuda.set_device(dev)

  import oracle  # parameters only (numpy RandomState draw); not on the measured path
  cfg = dict(QM8_CFG)
  params = oracle.make_lanczosnet_params(cfg, 1234)
  if args.zero_params:
    params = {k: np.zeros_like(v) for k, v in params.items()}
  net = LanczosNet(make_model_config(cfg)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
  net = net.to(dev)
  net.gemm_mode = args.gemm

  B = args.batch
  batch = draw_batch(B, seed=rank)  # config 3: seeds 0..7 per shard
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
  n_nodes = t(batch['n_nodes'])
  node_feat, mask = t(batch['node_feat']), t(batch['node_mask'])
  L = ops.laplacian_l4(t(batch['adjs']), n_nodes)  # dataset preprocessing, resident before timing
  A = L[:, :, :, 0]
  K = cfg['num_eig_vec']
  plan = net._plan()
  # per-step score all-gather (the path's one exchange), issued asynchronously so that
  # the next batch's kernels do not queue behind a latency-bound 64 KiB collective
  from lanczosnet_amd.dist import AsyncScoreGather
  gather = AsyncScoreGather(B, cfg['output_dim'], dev, force_collective=force_rccl) if dist else None

  ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(6)] for k in range(args.steps)}
  mask_u8 = mask.to(torch.uint8).contiguous()

  # LNZ_BENCH_SPLIT_PACK=1: the Laplacian pack of a step on a second stream under its
  # spectral-gains launch instead of inside the preparation launch (measured in r05: 0.802 against
  # 0.787 ms per step — the gains launch slows down by what the preparation saves)
  pack_stream = torch.cuda.Stream(device=dev) if os.environ.get('LNZ_BENCH_SPLIT_PACK', '0') == '1' else None

  def step(events=None, all_stages=False):
    # the timed loop brackets only the dominant kernel with HIP events (the roofline's live launch
    # duration); the per-stage breakdown comes from a separate, untimed pass (all_stages) so that
    # the event markers between the short launches do not sit in the measured region
    if events and all_stages:
      events[0].record()
    Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mask_u8, n_nodes, K, pack_stream=pack_stream)
    if events and all_stages:
      events[1].record()
    G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'],
                           rows=rows, zero_fill=not ops.pairing_supported(plan),
                           split_pack=Lp if plan.get('gemm_mode', 0) == 1 else None)
    if plan.get('gemm_mode', 0) == 1:
      G, Lp = G   # (split-precision mode: the pack's float16 form is written under the gains launch)
    if events and all_stages:
      events[2].record()
    if events:
      events[4].record()
    score = ops.lanczosnet_forward(plan, node_feat, Lp, V, G, mask_u8, tiling=tiles)
    if events:
      events[5].record()
    if gather:
      gather.submit(score)
    return score

  gains_cfg = (cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])

  def run_pipelined(n_steps, events=None):
    # batch k+1 is prepared while the gains of batch k are computed (one launch), then batch k is
    # forwarded: every batch still gets its own preparation, gains and forward
    Lp, tiles, rows, D, V = ops.prepare_batch(plan, L, mask_u8, n_nodes, K)      # prologue: batch 0
    score = None
    for k in range(n_steps):
      ev_k = events[k] if events else None
      if ev_k:
        ev_k[0].record()
      nLp, ntiles, nrows, nD, nV, G = ops.prepare_batch_prev_gains(
          plan, L, mask_u8, n_nodes, K, prev=(D, rows), gains=gains_cfg)
      if ev_k:
        ev_k[1].record()
        ev_k[4].record()
      score = ops.lanczosnet_forward(plan, node_feat, Lp, V, G, mask_u8, tiling=tiles)
      if ev_k:
        ev_k[5].record()
      if gather:
        gather.submit(score)
      Lp, tiles, rows, D, V = nLp, ntiles, nrows, nD, nV
    return score

  # The interpreter's cyclic collector is kept out of the timed region: a full collection over the
  # process's objects is a ~75 ms pause (seen as ONE 3x slower step in tools/bench_train_step.py when
  # an unrelated change moved its schedule) — a hundred steps of this loop.  Collected now, off until
  # the clock stops.
  import gc
  gc.collect()
  gc.disable()
  with torch.no_grad():
    if args.pipeline:
      score = run_pipelined(args.warmup)
    else:
      for _ in range(args.warmup):
        score = step()
    torch.cuda.synchronize()
    if dist:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.pipeline:
      score = run_pipelined(args.steps, ev)
    else:
      for i in range(args.steps):
        score = step(ev[i])
    if gather:
      gather.drain()  # every step's gather has landed before the clock stops
    torch.cuda.synchronize()
    if dist:
      dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
  gc.enable()
  per_rank_ms = None
  if dist:
    # every rank's own clock (a SCALE run explains itself: which rank was the slow one), then the
    # maximum over the ranks is the job's time
    tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    parts = [torch.zeros_like(tt) for _ in range(world)]
    dist.all_gather(parts, tt)
    per_rank_ms = [round(1e3 * float(p_.item()) / args.steps, 4) for p_ in parts]
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
  assert torch.isfinite(score).all()
  timed_score = score.clone()
  kernel_ran = ops.last_kernel()   # what the launcher selected for the timed forwards
  shard_check = None
  if dist and not args.zero_params:
    # every rank verifies ITS shard (seed = rank) on a bounded sample, after the clock has stopped
    # (N ranks share the host: a few intra-op threads each, not one pool of all cores per rank)
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
    err_r, n_r = shard_parity(cfg, params, batch, timed_score, args.shard_parity)
    mine = {'rank': rank, 'seed': rank, 'device': '%s:%d' % (torch.cuda.get_device_name(dev), dev.index),
            'parity_rel_err': err_r, 'molecules_checked': n_r}
    shard_check = [None] * world
    dist.all_gather_object(shard_check, mine)
    # the ranks the backend really formed: a sum of ones over the group
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    ranks_formed = int(ones.item())
  # the last step's gathered scores carry this rank's shard unchanged
  gather_ok = bool(torch.equal(gather.result(gather.issued - 1)[rank * B:(rank + 1) * B],
                               timed_score)) if gather else None

  # secondary measurement (N = 1 only, never `value`): the same sequential step once the device has
  # been under this load for a while.  The first ~25 steps after start-up run at ramping matrix clocks
  # (forward launch 0.605 against 0.588 ms): a short run (the driver's --steps 20 --warmup 5) reads
  # about 3 % slower than a long one for that reason alone, on the same box.
  sustained = None
  if world == 1 and not args.zero_params and not args.pipeline:
    with torch.no_grad():
      for _ in range(60):
        step()
      torch.cuda.synchronize()
      gc.collect()
      gc.disable()
      t1 = time.perf_counter()
      for _ in range(50):
        step()
      torch.cuda.synchronize()
      gc.enable()
      sustained = {'ms_per_step': round(1e3 * (time.perf_counter() - t1) / 50, 4),
                   'note': '50 steps timed behind %d + 60 untimed ones (the timed region of `value` sits behind %d)'
                           % (args.warmup + args.steps, args.warmup)}

  # secondary measurement (N = 1 only, never `value`): software pipeline over the stream of batches
  pipe = None
  if world == 1 and args.gemm == 'fp32' and not args.zero_params and not args.pipeline:
    with torch.no_grad():
      run_pipelined(args.warmup)
      torch.cuda.synchronize()
      gc.collect()
      gc.disable()
      t1 = time.perf_counter()
      sp = run_pipelined(args.steps)
      torch.cuda.synchronize()
      elp = time.perf_counter() - t1
      gc.enable()
    pipe = {'mode': 'one launch prepares batch k+1 (plan + Lanczos/eigensolve + pack) and computes the '
                    'spectral gains of batch k; then forward of batch k (lnz_prepare_batch_prev_gains)',
            'value': round(B * args.steps / elp, 1), 'unit': 'molecules/s',
            'ms_per_step': round(1e3 * elp / args.steps, 4),
            'scores_equal_sequential': bool(torch.equal(sp, score))}

  # secondary measurement (N = 1 only, never `value`): the opt-in split-precision GEMM mode
  split = None
  if world == 1 and args.gemm == 'fp32' and not args.zero_params:
    net.gemm_mode = 'f16x3'
    plan_fp32, plan = plan, net._plan()
    ref_score = score.clone()
    with torch.no_grad():
      for _ in range(args.warmup):
        s2 = step()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      gc.collect()
      gc.disable()
      t1 = time.perf_counter()
      for i in range(args.steps):
        s2 = step()
      torch.cuda.synchronize()
      el2 = time.perf_counter() - t1
      gc.enable()
      # the stages of this mode's step (the gains launch carries the pack's conversion along)
      evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(min(args.steps, 20))]
      for e_ in evs:
        step(e_, all_stages=True)
      torch.cuda.synchronize()
      split_stage_ms = {'prepare_batch(plan+lanczos_ritz+pack)': round(float(np.mean([e_[0].elapsed_time(e_[1]) for e_ in evs])), 4),
                        'spectral_gains(+pack conversion)': round(float(np.mean([e_[1].elapsed_time(e_[2]) for e_ in evs])), 4),
                        'lanczosnet_forward': round(float(np.mean([e_[4].elapsed_time(e_[5]) for e_ in evs])), 4)}
      Lp = ops.pack_laplacian_for(plan, L)
      D, V = ops.lanczos_ritz(A, n_nodes, K)
      G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
      tiles = ops.plan_tiles(mask_u8, allow_pairs=ops.pairing_supported(plan))
      N_FWD = 50
      for i in range(5):
        ops.lanczosnet_forward(plan, node_feat, Lp, V, G, mask_u8, tiling=tiles)
      e0.record()
      for i in range(N_FWD):
        ops.lanczosnet_forward(plan, node_feat, Lp, V, G, mask_u8, tiling=tiles)
      e1.record()
      torch.cuda.synchronize()
    fwd2_ms = e0.elapsed_time(e1) / N_FWD
    dev_rel = float((s2 - ref_score).abs().max() / ref_score.abs().max())
    split_score = s2.clone()
    # matrix-pipe work of the split-precision launch on the strip plan (utils/flop_model.py):
    # v_mfma_f32_16x16x32_f16 instructions of GEMM1 and of the block-diagonal products, three per
    # fp32 product; the first projection and the head stay on the fp32 shape
    from lanczosnet_amd.utils.flop_model import strips_from_plan, strip_split_mfma_issued
    assert plan.get('gemm_mode', 0) == 1
    fm2 = strip_split_mfma_issued(strips_from_plan(tiles[0].strips.cpu().numpy(), None), cfg)
    f16_flop = float(fm2['flops_issued'])
    mode_txt = ('f16x3 on the strip plan (lanczosnet_strip_kernel<.., HALF>, gemm_mode 1): X W^T as x_hi w_hi '
                '+ x_lo w_hi + x_hi w_lo on v_mfma_f32_16x16x32_f16, fp32 accumulate, node state in LDS as '
                'fp16 hi | lo pieces; the Laplacian products, the lift and the projection in the same '
                'split (Ritz blocks split once per launch, the Laplacian pack converted in place by '
                'workgroups that ride along with the gains launch); gains, biases, activations, head exact '
                'fp32; the rest of the step is the exact path\'s (same plan, same Ritz pairs) — opt-in, '
                'parity-tested at the same 1e-5 bar')
    roof_note = ('issued f16 matrix flops (16384 x the v_mfma_f32_16x16x32_f16 the strip plan issues: %d per '
                 'launch, + %d v_mfma_f32_16x16x4_f32): three split products per fp32 product, so the '
                 'fp32-equivalent rate is a third of `achieved`' % (fm2['mfma_f16_issued'], fm2['mfma_f32_issued']))
    roof_kernel = 'lanczosnet_strip_kernel<0,0,false,true>'
    split = {'mode': mode_txt,
             'value': round(B * args.steps / el2, 1), 'unit': 'molecules/s',
             'ms_per_step': round(1e3 * el2 / args.steps, 4),
             'stage_ms': split_stage_ms,
             'forward_ms': round(fwd2_ms, 4),
             'forward_ms_note': 'mean of %d back-to-back launches behind 5 warm ones (HIP events)' % N_FWD,
             'max_rel_dev_vs_fp32_path': dev_rel,
             'roofline': {'kernel': roof_kernel, 'bound': 'mfma',
                          'flops_per_launch_issued': f16_flop,
                          'achieved': round(f16_flop / fwd2_ms / 1e9, 1),
                          'peak': 2500.0, 'unit': 'TFLOP/s (dense f16)',
                          'frac': round(f16_flop / fwd2_ms / 1e9 / 2500.0, 4),
                          'note': roof_note}}
    net.gemm_mode = 'fp32'
    plan = plan_fp32

  # forward launch duration: HIP events of the timed region; the other stages: an untimed pass
  stage_ms = {}
  if not args.pipeline:
    ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(min(args.steps, 20))]
    with torch.no_grad():
      for e in ev2:
        step(e, all_stages=True)
    torch.cuda.synchronize()
    stage_ms['prepare_batch(plan+lanczos_ritz+pack)'] = float(np.mean([e[0].elapsed_time(e[1]) for e in ev2]))
    stage_ms['spectral_gains'] = float(np.mean([e[1].elapsed_time(e[2]) for e in ev2]))
  else:
    stage_ms['prepare_batch(k+1)+spectral_gains(k)'] = float(
        np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)]))
  stage_ms['lanczosnet_forward'] = float(
      np.mean([ev[i][4].elapsed_time(ev[i][5]) for i in range(args.steps)]))

  # secondary measurements (N = 1 only, never `value`)
  sweep = large = large_conv = ada = graphcfg = None
  if world == 1 and args.gemm == 'fp32' and not args.zero_params and args.sweep:
    sweep = forward_batch_sweep(net, plan, L, node_feat, mask_u8, n_nodes, cfg, (1024, 4096, 16384)
                                if B == 1024 else (B,))
  if world == 1 and args.gemm == 'fp32' and not args.zero_params and not args.no_secondary:
    # a failing secondary leg must not take the headline line with it: it is reported as an error
    def _leg(fn, *a_):
      try:
        return fn(*a_)
      except Exception as e:  # noqa: BLE001
        sys.stderr.write('bench: secondary leg %s failed: %r\n' % (fn.__name__, e))
        torch.cuda.synchronize()
        return {'error': repr(e)[:300]}
    r_ = _leg(lanczos_large_leg, dev)
    if isinstance(r_, tuple):
      (Ag, Dg, Vg), large = r_
      large_conv = _leg(large_graph_leg, dev, Ag, Dg, Vg)
      del Ag, Dg, Vg
    else:
      large = r_
    torch.cuda.empty_cache()
    ada = _leg(ada_leg, dev, L, node_feat, mask_u8)
    graphcfg = _leg(graph_config_leg, dev)

  if rank == 0:
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    fwd_s = stage_ms['lanczosnet_forward'] * 1e-3
    # matrix-core work the kernel ISSUES for this batch's tile plan: the kernel runs 32-row node
    # tiles (small molecules share one) and skips the GEMM2 / lift-back / projection k-groups of
    # padded rows and the GEMM2 of identity bond-type channels — counted instruction by
    # instruction from the plan (lanczosnet_amd/utils/flop_model.py; agrees with rocprofv3's
    # SQ_INSTS_VALU_MFMA_MOPS_F32 of this kernel, tests/test_flop_model.py)
    from lanczosnet_amd.utils.flop_model import (tiles_from_plan, forward_mfma_issued,
                                                 forward16_mfma_issued, forward16_selected,
                                                 strips_selected, strips_from_plan, strip_mfma_issued)
    with torch.no_grad():
      Lp_m, (buf, cap), _, _, _ = ops.prepare_batch(plan, L, mask_u8, n_nodes, K)
    torch.cuda.synchronize()
    mk = mask_u8.cpu().numpy()
    extents = np.where(mk.any(axis=1), mk.shape[1] - np.argmax(mk[:, ::-1] != 0, axis=1), 0)
    tile_list = tiles_from_plan(buf[:12 * cap].cpu().numpy(), extents,
                                Lp_m.ident.cpu().numpy() if hasattr(Lp_m, 'ident') else None, K)
    # which of the exact-fp32 forward kernels took the launches (csrc/conv_forward.hip launch_conv)
    f16 = args.gemm == 'fp32' and forward16_selected(cfg)
    strip = f16 and getattr(buf, 'strips', None) is not None and strips_selected(cfg, B, L.shape[1])
    if strip:
      ident_np = Lp_m.ident.cpu().numpy() if hasattr(Lp_m, 'ident') else None
      tile_list = strips_from_plan(buf.strips.cpu().numpy(), ident_np)
      fm = strip_mfma_issued(tile_list, cfg)
    else:
      fm = forward16_mfma_issued(tile_list, cfg) if f16 else forward_mfma_issued(tile_list, cfg)
    roof_kernel = ('lanczosnet_strip_kernel' if strip else 'lanczosnet_forward16_kernel' if f16
                   else 'lanczosnet_forward_kernel<4,10,0,0>')
    # the flop model above follows the launcher's rule; the name in the line is the launcher's own
    assert kernel_ran.startswith(roof_kernel.split('<')[0]), (kernel_ran, roof_kernel)
    roof_kernel = kernel_ran
    roof_insn = ('2048 flop x the v_mfma_f32_16x16x4_f32' if f16 else
                 '4096 flop x the v_mfma_f32_32x32x2_f32')
    n_tiles = fm['tiles']
    # strips: the block products are branch free — the kernel also multiplies the neighbouring
    # subtile pairs no molecule touches and the identity channels' zero fragments (6 % of its
    # instructions); `achieved` prices only the instructions of GEMM1 and of touched blocks
    flops_exec = fm['mfma_in_touched_blocks'] * 2048 if strip else fm['flops_issued']
    achieved = flops_exec / fwd_s / 1e12
    if os.environ.get('LNZ_BENCH_DUMP_PLAN'):
      if strip:
        np.savez_compressed(os.environ['LNZ_BENCH_DUMP_PLAN'], strips=buf.strips.cpu().numpy(),
                            ident=ident_np if ident_np is not None else np.zeros(0, np.int32))
      else:
        np.savez_compressed(os.environ['LNZ_BENCH_DUMP_PLAN'],
                            **{k_: np.array([t_[k_] for t_ in tile_list]) for k_ in tile_list[0]})
    # HBM bytes per launch of the roofline kernel: PMC counters cannot be read from inside this
    # process, so `traffic` cites the committed counter run of the SAME command and workload
    # (tools/pmc_forward_profile.py -> profiles/), never a number measured in this run
    traffic, traffic_source = None, None
    for name in (('r06_strip_pmc.json', 'r05_strip_pmc.json', 'r04_strip_pmc.json') if strip else ('r04_forward16_pmc.json',) if f16 else
                 ('r03_forward_pmc.json', 'pmc_forward_hbm_bytes.json')):
      prof = os.path.join(ROOT, 'profiles', name)
      if os.path.exists(prof) and B == 1024:
        try:
          traffic = json.load(open(prof)).get('hbm_bytes_per_launch')
          traffic_source = ('profiles/%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of this kernel '
                            'on this workload, separate counters-only passes (a committed profile, not '
                            'measured in this process), FETCH_SIZE x 2 per MI355X_MICROARCH.md' % name)
          break
        except Exception:
          traffic = None
    out = {
        'metric': 'molecules/sec LanczosNet forward, QM8 batch=1024',
        'value': round(value, 1), 'unit': 'molecules/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.gemm == 'fp32' else 'f16x3 (split fp16 products, f32 accumulate)',
        'data': 'synthetic' if not one_dev else 'synthetic (LNZ_BENCH_ONE_DEVICE functional test: all ranks on one GPU, gloo)',
        'config': {'workload': 'QM8 LanczosNet batch=%d/GPU, N<=32 dense L (tile N=%d), K=20, '
                               'fp32, 7x128 layers, 1xMI355X per rank; step = [pack L + batch plan + '
                               'Lanczos + tridiagonal eigensolver (Ritz pairs)] (one launch) + spectral gains + fused forward' % (B, L.shape[1]),
                   'global_batch': world * B, 'parallelism': 'dp%d (batch shards, async score all-gather per step)'
                   % world, 'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()}},
        # frac = flops the kernel ISSUES / launch time / peak: the long-scale
        # channels run in eigen space, 9.5 % fewer flops than the reference's association
        'roofline': {'kernel': roof_kernel, 'bound': 'mfma',
                     'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                     'traffic': traffic, 'traffic_source': traffic_source,
                     'flops_per_launch_executed': flops_exec,
                     'mfma_instructions_per_launch': fm['mfma_issued'],
                     'flops_per_launch_issued': fm['flops_issued'],
                     'flops_per_launch_without_skips': fm['flops_unskipped'],
                     'tiles_per_launch': n_tiles,
                     'useful_row_frac': round(fm['useful_row_frac'], 4),
                     'useful_frac': round(achieved / PEAK_FP32_MFMA_TFLOPS * fm['useful_row_frac'], 4),
                     'avg_launch_ms': round(stage_ms['lanczosnet_forward'], 4),
                     'reference_association_tflops': round(
                         FWD_FLOP_PER_MOL * (fm['subtiles'] / 2.0 if strip else n_tiles) / fwd_s / 1e12, 2),
                     'note': ('achieved = frac * peak = flops_per_launch_executed / avg_launch_ms. '
                              'flops_per_launch_executed = %s instructions the kernel issues for THIS '
                              'batch\'s strip plan (lnz_plan_strips: %d molecules at 4-row granularity '
                              'in %d strips = %d subtiles of 16 rows, at most %d per strip; the block '
                              'products of subtile pairs no molecule touches and of identity bond-type '
                              'channels are NOT counted, although the branch-free kernel issues them '
                              'on zero fragments: flops_per_launch_issued = mfma_instructions_per_launch '
                              'x 2048 is what the PMC counter SQ_INSTS_VALU_MFMA_MOPS_F32 shows; '
                              'utils/flop_model.py, profiles/). useful_row_frac = real node rows / strip rows; '
                              'useful_frac = frac x useful_row_frac. reference_association_tflops '
                              'prices every 32 rows at SURVEY 8(d)\'s %d flop (filter build + L_s Z per '
                              'long channel, which the kernel replaces by one projection and one lift '
                              'per layer)' % (roof_insn, B, n_tiles, fm['subtiles'],
                                              fm['max_subtiles_per_strip'], FWD_FLOP_PER_MOL))
                             if strip else
                             ('achieved = frac * peak = flops_per_launch_executed / avg_launch_ms. '
                              'flops_per_launch_executed = %s '
                              'instructions the kernel issues for THIS batch\'s tile plan (k-groups of '
                              'padded rows / empty eigen slots and identity bond-type channels are '
                              'skipped; utils/flop_model.py, checked against the PMC counter '
                              'SQ_INSTS_VALU_MFMA_MOPS_F32 in profiles/); without the skips the same '
                              '%d tiles would issue flops_per_launch_without_skips. useful_row_frac = '
                              'real node rows / tile rows (%d molecules ride in %d 32-row tiles, '
                              'lnz_plan_tiles); useful_frac = frac x useful_row_frac. '
                              'reference_association_tflops prices the tiles at SURVEY 8(d)\'s %d flop '
                              '(filter build + L_s Z per long channel, which the kernel replaces by one '
                              'projection and one lift per layer)'
                              % (roof_insn, n_tiles, B, n_tiles, FWD_FLOP_PER_MOL))},
    }
    if dist:
      out['config']['exchange'] = {
          'backend': dist.get_backend(), 'world': world,
          'ranks_formed': ranks_formed if shard_check is not None else dist.get_world_size(),
          'launcher': ('bench.py --gpus %d re-ran itself under torch.distributed.run' % world
                       if os.environ.get('LNZ_BENCH_SELF_SPAWNED') == '1' else 'torch.distributed.run (external)'),
          'devices_visible_per_rank': n_dev,
          'collective': 'all_gather_into_tensor of the [%d,%d] f32 shard scores, every step, async '
                        '(AsyncScoreGather)' % (B, cfg['output_dim']),
          'gathered_equals_local': gather_ok,
          'ms_per_step_per_rank': per_rank_ms,
          'note': 'ms_per_step (top level) = max over the ranks; every rank times its own %d steps '
                  'between the two barriers' % args.steps}
    if shard_check is not None:
      out['config']['exchange']['shards'] = shard_check
      out['config']['exchange']['shards_note'] = (
          'every rank checks the scores of the shard it timed (draw_batch seed = rank) against the '
          'oracle on its first %d molecules; bar 1e-5, the run fails above it' % args.shard_parity)
      out['parity_rel_err'] = max(s_['parity_rel_err'] for s_ in shard_check)
    if split is not None:
      out['config']['split_precision_mode'] = split
    if sustained is not None:
      out['config']['sustained_clock_mode'] = sustained
    if pipe is not None:
      out['config']['pipelined_stream_mode'] = pipe
    if sweep is not None:
      out['config']['forward_batch_sweep'] = sweep
    if large is not None:
      out['config']['lanczos_large_mode'] = large
    if large_conv is not None:
      out['config']['large_graph_conv_mode'] = large_conv
    if ada is not None:
      out['config']['ada_mode'] = ada
    if graphcfg is not None:
      out['config']['graph_config_mode'] = graphcfg
    if world == 1 and not args.no_cpu_baseline:
      # the reference's CPU path on the host cores, at most 32 intra-op threads: the batched GEMMs
      # of a 32-node tile stop scaling long before that, and a 256-thread pool on them does not
      # finish in minutes (measured: the r03 run with every core of the 256-core box timed out)
      ncpu = os.cpu_count() or 1
      torch.set_num_threads(min(ncpu, 32))
      v, times, ref_score, ambiguous = cpu_baseline(cfg, params, B, args.cpu_reps)
      ratio_note = ''
      try:
        pr = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_port_vs_reference.json')))
        ratio_note = ('; on the build container\'s %d cores the same port runs at %.2fx the rate of the '
                      'UNMODIFIED reference (get_graph_laplacian_eigs loop + collate_fn + '
                      'LanczosNet.forward: %.0f molecules/s) with identical scores '
                      '(profiles/cpu_port_vs_reference.json, tools/cpu_port_vs_reference.py)'
                      % (pr['host']['cpu_count'], pr['port_over_reference'],
                         pr['reference']['molecules_per_s']))
      except Exception:
        pass
      out['cpu_baseline'] = {'value': round(v, 1), 'unit': 'molecules/s',
                             'cores': torch.get_num_threads(), 'kind': 'port',
                             'sample': 'oracle port of the reference CPU path: single-thread LAPACK '
                                       'eigh + |lambda| sort per molecule (numpy), then the '
                                       'reference\'s LanczosNet.forward operator sequence restated on '
                                       'torch CPU tensors (%d intra-op threads; host has %d cores), '
                                       'B=%d, best of %d runs (%.2f s each, %.1f s in all)%s' %
                                       (torch.get_num_threads(), os.cpu_count() or 1, B, len(times),
                                        min(times), sum(times), ratio_note)}
      # self-verification: the scores of the timed batch (rank 0's seed-0 batch, same parameters)
      # against the oracle's scores of the same batch, all B molecules
      if not args.zero_params:
        got = timed_score.cpu().numpy().astype(np.float64)
        ref = ref_score.astype(np.float64)
        dev_mol = np.abs(got - ref).max(axis=1) / np.abs(ref).max()
        keep = ~ambiguous
        own = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
        out['parity_rel_err'] = float(dev_mol[keep].max())
        out['parity_rel_err_per_molecule'] = float(own[keep].max())
        out['parity'] = {'against': 'oracle (numpy port of the reference pipeline, fp32) on the timed '
                                    'batch: %d of %d molecules x %d outputs'
                                    % (int(keep.sum()), ref.shape[0], ref.shape[1]),
                         'metric': 'parity_rel_err = max |score - ref| / max |ref| over the batch; '
                                   'parity_rel_err_per_molecule = max over molecules of '
                                   'max |score_b - ref_b| / max |ref_b| (each molecule on its own scale)',
                         'bar': 1e-5,
                         'excluded': int(ambiguous.sum()),
                         'excluded_why': 'n > K and the top-K cut splits a degenerate |lambda| '
                                         'cluster (oracle.degenerate_cut: gap < 1e-7, the fp32 rounding of L): the reference keeps a LAPACK-chosen '
                                         'vector of the cluster (SURVEY.md 8c)',
                         'excluded_max_rel_dev': float(dev_mol[ambiguous].max()) if ambiguous.any()
                         else None}
        if split is not None:
          # the split-precision mode's OWN parity on the same molecules (it is opt-in and never the
          # `value`; the bar is the same)
          got2 = split_score.cpu().numpy().astype(np.float64)
          d2 = np.abs(got2 - ref).max(axis=1)
          out['config']['split_precision_mode']['parity'] = {
              'molecules': int(keep.sum()), 'bar': 1e-5,
              'parity_rel_err': float((d2 / np.abs(ref).max())[keep].max()),
              'parity_rel_err_per_molecule': float((d2 / np.abs(ref).max(axis=1))[keep].max())}
    print(json.dumps(out))
    if max(out.get('parity_rel_err', 0.0), out.get('parity_rel_err_per_molecule', 0.0)) > 1e-5:
      raise SystemExit('bench.py: parity_rel_err %.3e / per molecule %.3e exceeds 1e-5'
                       % (out['parity_rel_err'], out.get('parity_rel_err_per_molecule', 0.0)))
    if dist and shard_check is not None and ranks_formed != world:
      raise SystemExit('bench.py: the backend formed %d ranks, %d asked' % (ranks_formed, world))
  if dist:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
