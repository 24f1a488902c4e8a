#!/usr/bin/env python
"""Reference-side fixture of the runner drop-in test: run the UNMODIFIED reference stack —
`runner.qm8_runner.QM8Runner` + `dataset.qm8.QM8Data` + `model.LanczosNet`, preprocessing by
`utils.data_helper.get_*_laplacian_eigs` — on a seeded QM8-schema surrogate set with
`config/qm8_lanczos_net.yaml` (BASELINE configs[0]: B = 64, K = 20, Adam 1e-4) on the CPU, and
store what it produced: the loss of every training iteration, the validation MAE of every epoch,
the test MAE of the best snapshot, and the molecules in the order the reference's `glob` listed
them (packed: atoms, bonds, labels — ~160 B per molecule).

Build container only (needs /root/reference):   python tests/golden/make_golden_runner.py
Writes tests/golden/runner_qm8.npz.  The GPU box replays the same run through the HIP module
(tests/test_runner_dropin.py) and compares.
"""
import glob
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import runner_harness as H  # noqa: E402

SIZES = dict(train=640, dev=128, test=128)
MAX_EPOCH, SEED, DATA_SEED = 3, 1234, 77


def main():
  ref_runner, ref_model, ref_qm8, ref_dh = H.import_reference_runner()
  torch.set_num_threads(8)
  mols = H.draw_surrogate(sum(SIZES.values()), DATA_SEED)
  splits, o = {}, 0
  for s, k in SIZES.items():
    splits[s] = mols[o:o + k]
    o += k
  meta = H.standardise(splits)
  with tempfile.TemporaryDirectory() as tmp:
    data_dir, save_dir = os.path.join(tmp, 'data'), os.path.join(tmp, 'exp')
    H.write_reference_pickles(data_dir, splits, meta, H.reference_preprocess_fn(ref_dh))
    config = H.qm8_config(data_dir, save_dir, use_gpu=False, max_epoch=MAX_EPOCH)
    # the order in which the reference's dataset lists the files (dataset/qm8.py:29-34: glob order)
    order = {}
    for s in SIZES:
      files = glob.glob(os.path.join(config.dataset.data_path, 'QM8_preprocess_%s_*.p' % s))
      order[s] = [int(os.path.basename(f).split('_')[-1][:-2]) for f in files]
      assert sorted(order[s]) == list(range(SIZES[s]))
    H.seed_like_run_exp(SEED)
    with H.numpy_expand_dims_compat():
      res = H.run_runner(ref_runner.QM8Runner(config))
  print('reference run: %d iterations, train loss %.5f -> %.5f, val MAE %s, test MAE %.6f' %
        (len(res['train_loss']), res['train_loss'][0], res['train_loss'][-1],
         np.array2string(res['val_loss'], precision=6), res['test_mae']))

  out = dict(seed=SEED, data_seed=DATA_SEED, max_epoch=MAX_EPOCH, mean=meta['mean'], std=meta['std'],
             train_loss=res['train_loss'], val_loss=res['val_loss'], best_val=res['best_val'],
             test_mae=res['test_mae'], torch_version=np.array(torch.__version__))
  for s in SIZES:
    ms = [splits[s][i] for i in order[s]]  # reference listing order
    out[s + '_atoms'] = np.concatenate([m['node_feat'] for m in ms]).astype(np.uint8)
    out[s + '_mol_off'] = np.cumsum([0] + [len(m['node_feat']) for m in ms]).astype(np.int64)
    from lanczosnet_amd.dataset import edges_from_dense
    ed = [edges_from_dense(m['adjs']) for m in ms]
    out[s + '_edges'] = np.concatenate(ed).astype(np.uint32)
    out[s + '_edge_off'] = np.cumsum([0] + [len(e) for e in ed]).astype(np.int64)
    out[s + '_label'] = np.concatenate([m['label'] for m in ms], axis=0)  # standardised, float64
  np.savez_compressed(os.path.join(HERE, 'runner_qm8.npz'), **out)
  print('wrote runner_qm8.npz (%d bytes)' % os.path.getsize(os.path.join(HERE, 'runner_qm8.npz')))


if __name__ == '__main__':
  main()
