"""Packed QM8 shard + device-side collate.

The reference stores one pickle per molecule with dense float64 Laplacians and an offline
eigendecomposition (`dataset/get_qm8_data.py:56-96`), opens 21k of them per epoch
(`dataset/qm8.py:41-47`) and pads them in Python (`dataset/qm8.py:57-100,220-291`).  A packed shard
keeps only what defines a molecule — atom ids, the bond list, the labels — in one memory-mappable
file (~160 B per molecule); it is uploaded once and `PackedQM8.collate(ids, K)` builds a padded,
device-resident batch with two launches: `lnz_collate_qm8` (bond scatter + all L4 Laplacians) and
`lnz_lanczos_ritz` (the (D, V) the reference reads from `D_simple`/`V_simple`).

File layout (little endian, every array 64-byte aligned, in this order):
  header  64 B : magic b'LNZQM8\\0\\0', u32 version (1), u32 num_bond_type, u32 num_label,
                 u32 max_nodes, u64 num_mol, u64 total_atoms, u64 total_bonds
  mol_off  [num_mol+1] int64      atom offsets
  edge_off [num_mol+1] int64      bond offsets
  labels   [num_mol, num_label] float32
  edges    [total_bonds] uint32   u | v << 8 | type << 16, each undirected bond once (u <= v)
  atoms    [total_atoms] uint8    atom ids (reference `node_feat`, < 256)
"""
import struct

import numpy as np

MAGIC = b'LNZQM8\0\0'
VERSION = 1
_HDR = struct.Struct('<8sIIIIQQQ')


def _align(x, a=64):
    return (x + a - 1) // a * a


def edges_from_dense(adjs):
    """`adjs [n,n,E]` (symmetric, entries = bond multiplicity as in `to_graph`,
    dataset/get_qm8_data.py:26-42) -> packed uint32 bond list (multiplicity m = m entries)."""
    adjs = np.asarray(adjs)
    n, _, E = adjs.shape
    assert n <= 255 and E <= 255
    out = []
    iu, ju = np.triu_indices(n)
    for e in range(E):
        a = adjs[:, :, e]
        cnt = np.rint(a[iu, ju]).astype(np.int64)
        assert np.array_equal(cnt, a[iu, ju]) and (cnt >= 0).all(), 'bond multiplicities only'
        for u, v, c in zip(iu[cnt > 0], ju[cnt > 0], cnt[cnt > 0]):
            out.extend([int(u) | int(v) << 8 | e << 16] * int(c))
    return np.asarray(out, dtype=np.uint32)


def edges_from_laplacians(L_multi):
    """Recover the bond list from a reference pickle: `L_multi [n,n,E]` holds L4 of I + A_e per
    bond type (dataset/get_qm8_data.py:63-64,75), whose off-diagonal support is A_e's."""
    L_multi = np.asarray(L_multi)
    n, _, E = L_multi.shape
    adjs = np.zeros((n, n, E), dtype=np.float64)
    for e in range(E):
        a = (L_multi[:, :, e] != 0).astype(np.float64)
        np.fill_diagonal(a, 0.0)
        adjs[:, :, e] = np.maximum(a, a.T)
    return edges_from_dense(adjs)


def write_packed(path, molecules, num_bond_type, num_label):
    """molecules: iterable of dicts with `node_feat [n]` ints < 256, `label [num_label]`, and one of
    `edges` (packed uint32), `adjs [n,n,E]` or `L_multi [n,n,E]` (a reference pickle dict)."""
    atoms, edges, labels, mol_off, edge_off = [], [], [], [0], [0]
    max_nodes = 0
    for m in molecules:
        nf = np.asarray(m['node_feat']).astype(np.int64).reshape(-1)
        assert nf.size <= 255 and (nf >= 0).all() and (nf < 256).all()
        if 'edges' in m:
            ed = np.asarray(m['edges'], dtype=np.uint32)
        elif 'adjs' in m:
            ed = edges_from_dense(m['adjs'])
        else:
            ed = edges_from_laplacians(m['L_multi'])
        lab = np.asarray(m['label'], dtype=np.float32).reshape(-1)
        assert lab.size == num_label
        atoms.append(nf.astype(np.uint8))
        edges.append(ed)
        labels.append(lab)
        mol_off.append(mol_off[-1] + nf.size)
        edge_off.append(edge_off[-1] + ed.size)
        max_nodes = max(max_nodes, nf.size)
    num_mol = len(labels)
    hdr = _HDR.pack(MAGIC, VERSION, num_bond_type, num_label, max_nodes, num_mol, mol_off[-1],
                    edge_off[-1])
    parts = [np.asarray(mol_off, dtype='<i8'), np.asarray(edge_off, dtype='<i8'),
             (np.stack(labels) if num_mol else np.zeros((0, num_label))).astype('<f4'),
             (np.concatenate(edges) if num_mol else np.zeros(0)).astype('<u4'),
             (np.concatenate(atoms) if num_mol else np.zeros(0)).astype(np.uint8)]
    with open(path, 'wb') as f:
        f.write(hdr.ljust(64, b'\0'))
        for arr in parts:
            f.write(arr.tobytes())
            f.write(b'\0' * (_align(f.tell()) - f.tell()))


def convert_reference_pickles(files, out_path, num_bond_type=None):
    """Pack the reference's `QM8_preprocess_*.p` files (dataset/get_qm8_data.py:86-96; dicts with
    node_feat, L_multi, label, ...) into one shard.  The pickled Laplacians and eigenpairs are
    dropped: both are recomputed on the device at collate time."""
    import pickle
    mols, E, P = [], num_bond_type, None
    for f in files:
        with open(f, 'rb') as fh:
            d = pickle.load(fh)
        lab = np.asarray(d['label'], dtype=np.float32).reshape(-1)
        E = E if E is not None else int(np.asarray(d['L_multi']).shape[2])
        P = P if P is not None else lab.size
        mols.append(dict(node_feat=d['node_feat'], L_multi=d['L_multi'], label=lab))
    write_packed(out_path, mols, E, P)
    return len(mols)


class PackedQM8:
    """Memory-mapped packed shard; `.to(device)` uploads it once, `.collate()` builds batches."""

    def __init__(self, path):
        self.path = path
        raw = np.memmap(path, dtype=np.uint8, mode='r')
        magic, ver, E, P, max_nodes, num_mol, n_atoms, n_bonds = _HDR.unpack(bytes(raw[:_HDR.size]))
        if magic != MAGIC or ver != VERSION:
            raise ValueError('%s is not a packed QM8 shard (version %d)' % (path, VERSION))
        self.num_bond_type, self.num_label, self.max_nodes = E, P, max_nodes
        self.num_mol = num_mol
        off = 64

        def take(dtype, count):
            nonlocal off
            nbytes = np.dtype(dtype).itemsize * count
            if off + nbytes > raw.size:
                raise ValueError('%s is truncated' % path)
            arr = raw[off:off + nbytes].view(dtype)
            off = _align(off + nbytes)
            return arr
        self.mol_off = take('<i8', num_mol + 1)
        self.edge_off = take('<i8', num_mol + 1)
        self.labels = take('<f4', num_mol * P).reshape(num_mol, P)
        self.edges = take('<u4', n_bonds)
        self.atoms = take(np.uint8, n_atoms)
        if int(self.mol_off[-1]) != n_atoms or int(self.edge_off[-1]) != n_bonds:
            raise ValueError('%s: offsets do not match the header' % path)
        self.sizes = np.diff(self.mol_off).astype(np.int64)
        self._dev = None

    def __len__(self):
        return self.num_mol

    def molecule(self, i):
        """Host view of molecule i: (node_feat [n] uint8, edges uint32, label [P])."""
        a0, a1 = int(self.mol_off[i]), int(self.mol_off[i + 1])
        e0, e1 = int(self.edge_off[i]), int(self.edge_off[i + 1])
        return self.atoms[a0:a1], self.edges[e0:e1], self.labels[i]

    def to(self, device='cuda'):
        import torch
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('PackedQM8.to: the collate kernel needs a HIP device, got %s' % dev)
        up = lambda a: torch.from_numpy(np.array(a, copy=True)).to(dev)  # noqa: E731
        # keep at least one element so data_ptr() is valid for bond-free shards
        edges = self.edges if self.edges.size else np.zeros(1, dtype='<u4')
        self._dev = dict(mol_off=up(self.mol_off.astype(np.int64)),
                         edge_off=up(self.edge_off.astype(np.int64)),
                         labels=up(self.labels.astype(np.float32)),
                         edges=up(edges.view(np.int32)),
                         atoms=up(self.atoms if self.atoms.size else np.zeros(1, np.uint8)))
        self.device = dev
        return self

    def collate(self, ids, num_eigs):
        """Padded device batch with the reference's keys (dataset/qm8.py:262,289-291):
        node_feat [B,N] int64, node_mask [B,N] uint8, label [B,P], L [B,N,N,E+1], D [B,K],
        V [B,N,K]; plus n_nodes [B] int32."""
        import torch
        from .. import ops
        if self._dev is None:
            raise RuntimeError('PackedQM8.collate: call .to(device) first')
        ids_np = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids_np.size == 0:
            raise ValueError('empty batch')
        if ids_np.min() < 0 or ids_np.max() >= self.num_mol:
            raise IndexError('molecule id out of range')
        N = int(self.sizes[ids_np].max())  # dataset/qm8.py:66
        ids_t = torch.from_numpy(ids_np).to(self.device)
        out = ops.collate_qm8(self._dev, ids_t, N, self.num_bond_type, self.num_label)
        D, V = ops.lanczos_ritz(out['L'][:, :, :, 0], out['n_nodes'], num_eigs)
        out['D'], out['V'] = D, V
        return out



class PackedQM8Data(object):
    """Dataset-class surface of the runner (`eval(loader_name)(config, split=...)`,
    runner/qm8_runner.py:40-56) over packed shards: `config.dataset.data_path/QM8_packed_{split}.bin`.
    `__getitem__` returns a molecule id, `collate_fn` is the device-side collate, so a DataLoader
    (num_workers = 0: the collate launches kernels) yields device-resident batches with the
    reference's keys — Laplacians and Ritz pairs computed on the GPU, nothing pickled."""

    def __init__(self, config, split='train', device='cuda'):
        import os
        assert split in ('train', 'dev', 'test'), 'no such split'
        self.split, self.config = split, config
        self.num_eigs = config.model.num_eig_vec if hasattr(config.model, 'num_eig_vec') else 1
        self.shard = PackedQM8(os.path.join(config.dataset.data_path,
                                            'QM8_packed_%s.bin' % split)).to(device)

    def __len__(self):
        return len(self.shard)

    def __getitem__(self, index):
        return int(index)

    def collate_fn(self, batch):
        return self.shard.collate(batch, self.num_eigs)
