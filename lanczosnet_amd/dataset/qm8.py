"""Collate for QM8-schema molecules (reference `dataset/qm8.py:57-100,220-291`).

Two entry points:

* `collate_preprocessed(items, num_eigs)` — items are the reference's per-molecule pickle dicts
  (`dataset/get_qm8_data.py:56-96`: node_feat, L_multi, L_simple_4, D_simple, V_simple, label);
  host-side padding exactly like the reference's default branch, for drop-in DataLoaders.
* `collate_adjacency(items, num_eigs, device)` — items carry only the RAW graph
  (`adjs [n,n,E]`, `node_feat [n]`, `label [1,P]`); padding happens on the host, but the
  Laplacians (`lnz_laplacian_l4`, replacing get_qm8_data.py:62-75) and the Ritz pairs
  (`lnz_lanczos_ritz`, replacing utils/data_helper.py:197-223 + the pad/cut of qm8.py:264-291) are
  computed ON THE DEVICE — no offline eigendecomposition pickles.

`QM8Data(config, split)` is the reference's dataset class surface (`dataset/qm8.py:11-55`:
constructor, `__getitem__`, `__len__`, `collate_fn`) over the same per-molecule pickles, for the
runner's `eval(config.dataset.loader_name)(config, split=...)` (runner/qm8_runner.py:40-56).

Both return the reference's dict keys: node_feat [B,N] int64, node_mask [B,N] uint8,
label [B,P] float32, L [B,N,N,E+1] float32, D [B,K], V [B,N,K].
"""
import numpy as np
import torch


def _pad_common(items):
    sizes = [int(np.asarray(it['node_feat']).shape[0]) for it in items]
    B, N = len(items), max(sizes)
    node_feat = np.zeros((B, N), dtype=np.int64)
    mask = np.zeros((B, N), dtype=np.uint8)
    for b, (it, n) in enumerate(zip(items, sizes)):
        node_feat[b, :n] = np.asarray(it['node_feat'])
        mask[b, :n] = 1
    label = np.concatenate([np.asarray(it['label'], dtype=np.float32).reshape(1, -1)
                            for it in items], axis=0)
    return sizes, B, N, node_feat, mask, label


def collate_preprocessed(items, num_eigs):
    """Host-side restatement of the reference default branch (dataset/qm8.py:220-291)."""
    sizes, B, N, node_feat, mask, label = _pad_common(items)
    E = np.asarray(items[0]['L_multi']).shape[2]
    L = np.zeros((B, N, N, E + 1), dtype=np.float32)
    D = np.zeros((B, num_eigs), dtype=np.float32)
    V = np.zeros((B, N, num_eigs), dtype=np.float32)
    for b, (it, n) in enumerate(zip(items, sizes)):
        L[b, :n, :n, 0] = it['L_simple_4']
        L[b, :n, :n, 1:] = it['L_multi']
        d, v = np.asarray(it['D_simple']), np.asarray(it['V_simple'])
        kk = min(num_eigs, d.shape[0])
        D[b, :kk] = d[:kk]
        V[b, :n, :kk] = v[:, :kk]
    return dict(node_feat=torch.from_numpy(node_feat), node_mask=torch.from_numpy(mask),
                label=torch.from_numpy(label), L=torch.from_numpy(L), D=torch.from_numpy(D),
                V=torch.from_numpy(V))


def collate_adjacency(items, num_eigs, device='cuda'):
    """Raw graphs in, device-resident batch out (L4 and Ritz pairs by the HIP kernels)."""
    from .. import ops
    sizes, B, N, node_feat, mask, label = _pad_common(items)
    E = np.asarray(items[0]['adjs']).shape[2]
    adjs = np.zeros((B, N, N, E), dtype=np.float32)
    for b, (it, n) in enumerate(zip(items, sizes)):
        adjs[b, :n, :n, :] = it['adjs']
    dev = torch.device(device)
    n_nodes = torch.tensor(sizes, dtype=torch.int32, device=dev)
    L = ops.laplacian_l4(torch.from_numpy(adjs).to(dev), n_nodes)
    D, V = ops.lanczos_ritz(L[:, :, :, 0], n_nodes, num_eigs)
    return dict(node_feat=torch.from_numpy(node_feat).to(dev),
                node_mask=torch.from_numpy(mask).to(dev), label=torch.from_numpy(label).to(dev),
                L=L, D=D, V=V, n_nodes=n_nodes)


class QM8Data(object):
    """Drop-in for reference `dataset/qm8.py:11-293` (LanczosNet / default branch): globs
    `QM8_preprocess_{split}_*.p` under `config.dataset.data_path`, one pickle per molecule
    (`dataset/get_qm8_data.py:86-96`), `collate_fn` pads a list of them to the batch maximum.
    The file lists are SORTED (the reference keeps `glob.glob`'s directory order, :29-34, which
    is file-system dependent)."""

    def __init__(self, config, split='train'):
        import glob
        import os
        assert split in ('train', 'dev', 'test'), 'no such split'
        self.split, self.config = split, config
        self.data_path = config.dataset.data_path
        self.num_edgetype = config.dataset.num_bond_type
        self.model_name = config.model.name
        self.use_eigs = hasattr(config.model, 'num_eig_vec')
        self.num_eigs = config.model.num_eig_vec if self.use_eigs else 0
        self.files = sorted(glob.glob(os.path.join(self.data_path,
                                                   'QM8_preprocess_%s_*.p' % split)))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        import pickle
        with open(self.files[index], 'rb') as f:
            return pickle.load(f)

    def collate_fn(self, batch):
        assert isinstance(batch, list)
        if self.model_name in ('GPNN', 'GraphSAGE', 'DCNN', 'ChebyNet'):
            raise NotImplementedError('QM8Data mirrors the default collate branch (LanczosNet, '
                                      'AdaLanczosNet, GCN); got %s' % self.model_name)
        out = collate_preprocessed(batch, self.num_eigs if self.use_eigs else 1)
        if not self.use_eigs:
            out.pop('D')
            out.pop('V')
        return out
