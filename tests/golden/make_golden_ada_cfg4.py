#!/usr/bin/env python
"""Reference-side fixture for BASELINE configs[3] at its FULL depth: the UNMODIFIED reference
`AdaLanczosNet` built from the reference's own yaml (`config/qm8_ada_lanczos_net.yaml`: 7 conv
layers of width 128, short [1,2,3], long [5,7,10,20,30], K = 20, seven 2000-4096-4096-4096-2000
filter MLPs = 350 M parameters) on the first 128 molecules of the bench batch
(`draw_batch(1024, seed=0)`, the batch `bench.py` times), parameters from
`oracle.make_ada_params(cfg, 41)` (numpy seed: nothing large in the fixture), start vectors
`RandomState(83).randn`.

  ada_cfg4.npz   scores of the reference (fp32 torch CPU), the RAW beta of every Lanczos step (`torch.norm` is wrapped while the unmodified forward runs), and the
                 reference's own distance from exact arithmetic per molecule (the SAME reference
                 class run in float64) — the two numbers the parity protocol of
                 tests/test_gpu_ada.py classifies molecules by.

    python tests/golden/make_golden_ada_cfg4.py        (needs /root/reference; ~2 min, ~6 GB RAM)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import make_golden as MG  # noqa: E402
import make_golden_ada as MA  # noqa: E402
import oracle  # noqa: E402
from lanczosnet_amd.synthetic import draw_batch  # noqa: E402

NB, BATCH, PARAM_SEED, Q1_SEED = 128, 1024, 41, 83


def main():
  ref_model, _, _ = MG.import_reference()
  torch.set_num_threads(8)
  y = yaml.safe_load(open(os.path.join(MG.REF, 'config', 'qm8_ada_lanczos_net.yaml')))
  ym, yd = y['model'], y['dataset']
  cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=list(ym['short_diffusion_dist']),
             long_diffusion_dist=list(ym['long_diffusion_dist']), hidden_dim=list(ym['hidden_dim']),
             num_layer=int(ym['num_layer']), num_eig_vec=int(ym['num_eig_vec']),
             output_dim=int(ym['output_dim']), num_atom=int(yd['num_atom']),
             num_bond_type=int(yd['num_bond_type']))
  assert cfg['num_layer'] == 7 and cfg['hidden_dim'] == [128] * 7
  conf = MG.make_config(cfg, name='AdaLanczosNet')
  conf['model']['use_reorthogonalization'] = ym['use_reorthogonalization']  # (ignored: SURVEY F7)
  b = draw_batch(BATCH, seed=0)
  N = b['node_mask'].shape[1]
  L = np.zeros((NB, N, N, 7), np.float32)
  for i in range(NB):
    n = int(b['n_nodes'][i])
    L[i, :n, :n] = oracle.laplacian_multi_l4(b['adjs'][i, :n, :n])
  nf = torch.from_numpy(b['node_feat'][:NB])
  mask = torch.from_numpy(b['node_mask'][:NB])
  q1 = np.random.RandomState(Q1_SEED).randn(NB, N, 1).astype(np.float32)
  P = oracle.make_ada_params(cfg, PARAM_SEED)
  net = ref_model.AdaLanczosNet(conf).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  with torch.no_grad(), MA.fixed_randn(q1), MA.capture_norms() as cap:
    score = net(nf, torch.from_numpy(L), mask=mask.bool()).numpy()
  # ONE Lanczos run per forward (model/ada_lanczos_net.py:316-318: the learned Laplacian and its
  # tridiagonalisation come from the embedded input; every layer has its own filter MLP on the same
  # T): 1 + T norms (start vector, then beta_1..beta_T)
  T = min(N, cfg['num_eig_vec'])
  assert len(cap.out) == T + 1, len(cap.out)
  betas = cap.betas()                                                    # [NB, T]
  del net
  # exact arithmetic: the same reference class in float64
  net64 = ref_model.AdaLanczosNet(conf).double().eval()
  net64.load_state_dict({k: torch.from_numpy(v).double() for k, v in P.items()})
  real = torch.randn
  torch.randn = lambda *a, **k: torch.from_numpy(q1.astype(np.float64))
  try:
    with torch.no_grad():
      s64 = net64(nf, torch.from_numpy(L).double(), mask=mask.bool()).numpy()
  finally:
    torch.randn = real
  del net64
  # and the numpy oracle the GPU tests use as "exact" must be that function
  so, _ = oracle.ada_lanczos_net_forward(P, cfg, b['node_feat'][:NB], L, b['node_mask'][:NB],
                                         q1[:, :, 0], dtype=np.float64)
  dev_oracle = np.abs(so - s64).max() / np.abs(s64).max()
  print('numpy fp64 oracle vs the reference class in float64: %.2e' % dev_oracle)
  assert dev_oracle < 1e-9
  e_ref = np.abs(score - s64).max(axis=1) / np.abs(s64).max()
  sep = MA.separation(betas)
  strict = (sep >= 10) & (e_ref <= 2e-6)
  print('cfg4: %d molecules, 7 layers; separation >= 10: %d; reference within 2e-6 '
        'of float64: %d; both: %d; reference-vs-float64 median %.2e max %.2e'
        % (NB, (sep >= 10).sum(), (e_ref <= 2e-6).sum(), strict.sum(), np.median(e_ref), e_ref.max()))
  out = os.path.join(HERE, 'ada_cfg4.npz')
  np.savez_compressed(out, cfg_json=np.array(repr(cfg)), param_seed=PARAM_SEED, q1_seed=Q1_SEED,
                      nb=NB, batch=BATCH, batch_seed=0, q1=q1[:, :, 0], score=score, score64=s64,
                      betas_raw=betas.astype(np.float32), n_nodes=b['n_nodes'][:NB])
  print('  %-24s %8d B' % ('ada_cfg4.npz', os.path.getsize(out)))


if __name__ == '__main__':
  main()
