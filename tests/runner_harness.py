"""Shared pieces of the runner drop-in tests (tests/test_runner_dropin.py) and of the script that
generates their reference-side fixture (tests/golden/make_golden_runner.py).

* a seeded QM8-schema surrogate set (random tree + <= 2 ring closures,
  6 bond types, atom ids < 70) with LEARNABLE labels and per-target mean/std meta data;
* writers for the reference's on-disk formats (`dataset/get_qm8_data.py:56-96,120-121`: one
  pickle per molecule + QM8_meta.p) parameterised by the preprocessing function, so the SAME
  writer runs with the reference's `get_*_laplacian_eigs` (build container) or with the oracle's
  restatement (GPU box, where /root/reference does not exist);
* the config of `config/qm8_lanczos_net.yaml` (BASELINE configs[0]: B = 64, K = 20) as an
  attribute dict, restated here because the yaml lives in the reference tree;
* import of the unmodified reference with the stubs SURVEY.md F13 lists.
"""
import os
import pickle
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
REF = os.environ.get('LANCZOS_REFERENCE', '/root/reference')

from lanczosnet_amd.synthetic import draw_molecule  # noqa: E402
from lanczosnet_amd.utils.arg_helper import AttrDict  # noqa: E402

K_EIG, NUM_BOND, NUM_ATOM, NUM_LABEL = 20, 6, 70, 16


# ---------------------------------------------------------------------------------- surrogate
def draw_surrogate(num, seed, n_min=4, n_max=26):
  """`num` molecules: dict(adjs [n,n,6] float32, node_feat [n] int64, label [1,16] float64 RAW).
  Molecules whose top-K cut would split a degenerate |lambda| cluster (n > K, gap < 1e-7) are
  re-drawn: their (D, V) are basis dependent in the reference itself."""
  import oracle
  rs = np.random.RandomState(seed)
  w_atom = rs.randn(NUM_ATOM, NUM_LABEL) * 0.3
  w_bond = rs.randn(NUM_BOND, NUM_LABEL) * 0.2
  w_n = rs.randn(NUM_LABEL) * 0.05
  mols = []
  while len(mols) < num:
    n = int(rs.randint(n_min, n_max + 1))
    adjs = draw_molecule(rs, n, NUM_BOND)
    atoms = rs.randint(0, NUM_ATOM, size=n).astype(np.int64)
    if n > K_EIG:
      e, _, _ = oracle.graph_laplacian_eigs(adjs.sum(axis=2), graph_laplacian_type='L4')
      if abs(abs(e[K_EIG - 1]) - abs(e[K_EIG])) < 1e-7:
        continue
    # a smooth function of composition, bonds and size + noise: something a GNN can fit
    # (scaled so that the per-target std is ~0.02, the magnitude of QM8's 16 targets: the MAE
    # gate of +-0.05e-3 is quoted for MAEs around 1e-2, README.md:44)
    lab = 0.07 * (w_atom[atoms].mean(axis=0) + (adjs.sum(axis=(0, 1)) / (2.0 * n)) @ w_bond
                  + w_n * n + 0.05 * rs.randn(NUM_LABEL))
    mols.append(dict(adjs=adjs, node_feat=atoms, label=lab.reshape(1, -1)))
  return mols


def standardise(splits):
  """Per-target mean/std over the train split; labels become (raw - mean) / std like
  deepchem's normalisation transformer feeding `dataset/get_qm8_data.py:104-121`."""
  raw = np.concatenate([m['label'] for m in splits['train']], axis=0)
  mean, std = raw.mean(axis=0), raw.std(axis=0)
  for mols in splits.values():
    for m in mols:
      m['label'] = ((m['label'] - mean) / std).astype(np.float64)
  return dict(mean=mean, std=std)


# -------------------------------------------------------------------------- preprocessing fns
def oracle_preprocess(adjs):
  """The oracle's restatement of dataset/get_qm8_data.py:59-81 for one molecule."""
  import oracle
  Lm = oracle.laplacian_multi_l4(adjs)
  D, V, L4 = oracle.graph_laplacian_eigs(adjs.sum(axis=2), graph_laplacian_type='L4')
  return dict(L_multi=Lm[:, :, 1:], L_simple_4=L4, D_simple=D, V_simple=V)


def reference_preprocess_fn(ref_dh):
  def fn(adjs):
    n = adjs.shape[0]
    _, _, L_list = ref_dh.get_multi_graph_laplacian_eigs(
        adjs, graph_laplacian_type='L4', use_eigen_decomp=True, is_sym=True)
    D, V, L4 = ref_dh.get_graph_laplacian_eigs(
        adjs.sum(axis=2), graph_laplacian_type='L4', use_eigen_decomp=True, is_sym=True)
    return dict(L_multi=np.stack(L_list, axis=2), L_simple_4=L4,
                D_simple=D if D is not None else np.ones(n),
                V_simple=V if V is not None else np.eye(n))
  return fn


def write_reference_pickles(data_dir, splits, meta, preprocess):
  """One `QM8_preprocess_{split}_{i:07d}.p` per molecule with the key set of
  dataset/get_qm8_data.py:56-96 that the default collate branch reads, + QM8_meta.p."""
  os.makedirs(os.path.join(data_dir, 'preprocess'), exist_ok=True)
  for split, mols in splits.items():
    for i, m in enumerate(mols):
      d = dict(node_feat=m['node_feat'], label=m['label'], **preprocess(m['adjs']))
      with open(os.path.join(data_dir, 'preprocess', 'QM8_preprocess_%s_%07d.p' % (split, i)),
                'wb') as f:
        pickle.dump(d, f)
  with open(os.path.join(data_dir, 'QM8_meta.p'), 'wb') as f:
    pickle.dump(meta, f)


def write_packed_shards(data_dir, splits):
  from lanczosnet_amd.dataset import write_packed
  os.makedirs(os.path.join(data_dir, 'preprocess'), exist_ok=True)
  for split, mols in splits.items():
    write_packed(os.path.join(data_dir, 'preprocess', 'QM8_packed_%s.bin' % split),
                 [dict(node_feat=m['node_feat'], adjs=m['adjs'], label=m['label'].reshape(-1))
                  for m in mols], NUM_BOND, NUM_LABEL)


# --------------------------------------------------------------------------------------- config
def qm8_config(data_dir, save_dir, use_gpu, max_epoch, batch_size=64, loader='QM8Data', seed=1234):
  """config/qm8_lanczos_net.yaml (BASELINE configs[0]) with the paths, the epoch count and
  num_workers = 0 overridden (a device-side collate cannot run in worker processes; the sample
  order does not depend on the worker count)."""
  os.makedirs(save_dir, exist_ok=True)
  return AttrDict(dict(
      exp_name='qm8_lanczos_net', exp_dir=save_dir, save_dir=save_dir, runner='QM8Runner',
      use_gpu=use_gpu, gpus=[0], seed=seed,
      dataset=dict(loader_name=loader, name='chemistry',
                   data_path=os.path.join(data_dir, 'preprocess'),
                   meta_data_path=os.path.join(data_dir, 'QM8_meta.p'), num_atom=NUM_ATOM,
                   num_bond_type=NUM_BOND),
      model=dict(name='LanczosNet', short_diffusion_dist=[],
                 long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30], num_eig_vec=K_EIG,
                 spectral_filter_kind='MLP', input_dim=64, hidden_dim=[128] * 7, output_dim=16,
                 num_layer=7, loss='MSE', output_func='MLP'),
      train=dict(optimizer='Adam', lr_decay=0.1, lr_decay_steps=[10000], num_workers=0,
                 max_epoch=max_epoch, batch_size=batch_size, display_iter=100,
                 snapshot_epoch=10000, valid_epoch=1, lr=1.0e-4, wd=0.0, momentum=0.9,
                 shuffle=True, is_resume=False, resume_model='None'),
      test=dict(batch_size=batch_size, num_workers=0,
                test_model=os.path.join(save_dir, 'model_snapshot_best.pth'))))


def graph_config(data_path, save_dir, use_gpu, max_epoch, loader='GraphData', seed=1234):
  """config/graph_lanczos_net.yaml with the paths and the epoch count overridden and a validation
  every epoch (yaml: valid_epoch 100)."""
  os.makedirs(save_dir, exist_ok=True)
  return AttrDict(dict(
      exp_name='graph_lanczos_net', exp_dir=save_dir, save_dir=save_dir, runner='GraphRunner',
      use_gpu=use_gpu, gpus=[0], seed=seed,
      dataset=dict(loader_name=loader, name='synthetic', data_path=data_path, node_emb_dim=10,
                   graph_emb_dim=2, num_edge_type=1),
      model=dict(name='LanczosNetGeneral', short_diffusion_dist=[],
                 long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30], num_eig_vec=20,
                 spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7, output_dim=2,
                 num_layer=7, loss='MSE', output_func='MLP'),
      train=dict(optimizer='Adam', lr_decay=0.1, lr_decay_steps=[10000], num_workers=0,
                 max_epoch=max_epoch, batch_size=10, display_iter=10, snapshot_epoch=10000,
                 valid_epoch=1, lr=1.0e-4, wd=0.0, momentum=0.9, shuffle=True, is_resume=False,
                 resume_model='None'),
      test=dict(batch_size=64, num_workers=0,
                test_model=os.path.join(save_dir, 'model_snapshot_best.pth'))))


def seed_like_run_exp(seed):
  """run_exp.py:18-20."""
  import torch
  np.random.seed(seed)
  torch.manual_seed(seed)
  if torch.cuda.is_available():
    torch.cuda.manual_seed_all(seed)


def run_runner(runner):
  """train() then test() of a QM8Runner-like object -> dict of the numbers the runner produces."""
  best = runner.train()
  with open(os.path.join(runner.config.save_dir, 'train_stats.p'), 'rb') as f:
    stats = pickle.load(f)
  test_mae = runner.test()
  return dict(train_loss=np.asarray(stats['train_loss'], np.float64),
              val_loss=np.asarray(stats['val_loss'], np.float64), best_val=float(best),
              test_mae=float(test_mae))


# ------------------------------------------------------------------------------------ reference
def have_reference():
  return os.path.isdir(os.path.join(REF, 'runner'))


def import_reference_runner():
  """The UNMODIFIED reference runner + model + dataset + data helper, read-only: bytecode writing
  off, stubs for the two absent third-party modules the runner imports at module level
  (`tensorboardX`) and for the un-buildable `operators._ext` (F11)."""
  sys.dont_write_bytecode = True
  for name in ('operators._ext', 'operators._ext.segment_reduction'):
    sys.modules.setdefault(name, types.ModuleType(name))
  sys.modules['operators._ext'].segment_reduction = sys.modules['operators._ext.segment_reduction']
  if 'tensorboardX' not in sys.modules:
    tb = types.ModuleType('tensorboardX')

    class SummaryWriter(object):
      def __init__(self, *a, **k):
        pass

      def add_scalar(self, *a, **k):
        pass

      def close(self):
        pass
    tb.SummaryWriter = SummaryWriter
    sys.modules['tensorboardX'] = tb
  if REF not in sys.path:
    sys.path.insert(1, REF)
  import model as ref_model  # noqa
  import dataset.qm8 as ref_qm8  # noqa
  import runner.qm8_runner as ref_runner  # noqa
  import utils.data_helper as ref_dh  # noqa
  return ref_runner, ref_model, ref_qm8, ref_dh


class numpy_expand_dims_compat(object):
  """dataset/qm8.py:254-259 calls np.expand_dims(x2d, axis=3); numpy < 1.18 clamped an
  out-of-range axis (-> [n,n,1]), numpy 2 raises.  Emulate the old numpy around reference calls,
  leaving the reference untouched."""

  def __enter__(self):
    self.real = np.expand_dims
    real = self.real
    np.expand_dims = lambda a, axis: real(a, min(axis, np.ndim(a)))

  def __exit__(self, *exc):
    np.expand_dims = self.real
