#!/usr/bin/env python
"""BASELINE config 5 at its FULL size pinned on the reference: the unmodified
`model/lanczos_net_general.py` LanczosNetGeneral (7 x 128, E+1 = 2, K = 64) on CPU at B = 2,
N = 2048 (G(n, p = 0.01) graphs), fed the Ritz pairs of `oracle.lanczos_kstep_fp64` (the fp64
restatement of the K-step Lanczos: full-length eigh is not what the large-graph path computes, and
the reference's eigsh is a different function for unconverged pairs).

The inputs are functions of numpy seeds (tests/large_fixture.py), so the fixture holds only the
reference's scores, D and a checksum of V:  tests/golden/config5_full.npz.

    python tests/golden/make_golden_config5.py       # needs /root/reference, ~1 min
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden import AttrDict, import_reference, params_checksum  # noqa: E402
from large_fixture import general_inputs, kstep_ritz  # noqa: E402

B, N, K, LAYERS, SEED, P_EDGE = 2, 2048, 64, 7, 11, 0.01


def main():
  ref_model, _, _ = import_reference()
  torch.set_num_threads(os.cpu_count() or 4)
  cfg, P, X, L, mask = general_inputs(B, N, K, LAYERS, SEED, P_EDGE)
  D, V = kstep_ritz(L, K)
  model = dict(name='LanczosNetGeneral', short_diffusion_dist=[],
               long_diffusion_dist=cfg['long_diffusion_dist'], num_eig_vec=K,
               spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * LAYERS, output_dim=2,
               num_layer=LAYERS, loss='MSE')
  config = AttrDict(dict(seed=1234, model=model,
                         dataset=dict(node_emb_dim=10, graph_emb_dim=2, num_edge_type=1)))
  net = ref_model.LanczosNetGeneral(config).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  with torch.no_grad():
    score = net(torch.from_numpy(X), torch.from_numpy(L), torch.from_numpy(D), torch.from_numpy(V),
                mask=torch.from_numpy(mask).bool())
  path = os.path.join(HERE, 'config5_full.npz')
  np.savez_compressed(path, B=B, N=N, K=K, num_layer=LAYERS, seed=SEED, p_edge=P_EDGE,
                      score=score.numpy(), D=D, V_abs_colsum=np.abs(V.astype(np.float64)).sum(axis=1),
                      param_checksum=params_checksum(P))
  print('score', score.numpy())
  print('wrote', path, os.path.getsize(path), 'B')


if __name__ == '__main__':
  main()
