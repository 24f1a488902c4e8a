"""CPU-only tests: the C-ABI library loads and exports every symbol the header declares, rejects
bad arguments without touching a GPU, and the nn.Module mirror keeps the reference's drop-in
surface (state_dict keys, parameter count, init RNG order) and refuses to compute on the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _header_symbols():
  txt = open(os.path.join(ROOT, 'include', 'lanczosnet_hip.h')).read()
  txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
  return sorted(set(re.findall(r'\b(lnz_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
  from lanczosnet_amd import _lib
  lib = _lib.load()
  syms = _header_symbols()
  assert len(syms) >= 15
  for name in syms:
    assert hasattr(lib, name), name
    assert name in _lib.SIGNATURES, 'binding lacks a typed signature for ' + name
  assert set(_lib.SIGNATURES) == set(syms)
  assert lib.lnz_abi_version() == _lib.ABI_VERSION
  assert lib.lnz_forward_args_size() == C.sizeof(_lib.ForwardArgs)


def test_argument_validation_without_gpu():
  from lanczosnet_amd import _lib
  lib = _lib.load()
  null = C.c_void_p(0)
  one = C.c_void_p(16)  # never dereferenced: validation fails before any launch
  assert lib.lnz_pack_rows_k8(null, 4, 4, 4, null, null) == _lib.LNZ_EINVAL
  assert lib.lnz_pack_laplacian(one, 1, 1, 1, 1, 2, 40, 7, one, null) == _lib.LNZ_ENOTSUP
  assert b'32-node tile' in lib.lnz_last_error() or b'exceeds' in lib.lnz_last_error()
  # one workgroup owns a graph of up to 192 nodes; beyond that only the streamed K-step kernels apply
  assert lib.lnz_lanczos_ritz(one, 1, 1, 1, one, 4, 193, 20, one, one, null, null) == _lib.LNZ_ENOTSUP
  assert b'lnz_lanczos_ritz_large' in lib.lnz_last_error()
  assert lib.lnz_lanczos_ritz_ws(one, 1, 1, 1, one, 4, 193, 20, one, one, null, null, 0, 0, null) == \
      _lib.LNZ_ENOTSUP
  # the fp64 basis sits in LDS next to A while both fit (every graph of the reference generator,
  # n <= 100, does); above that boundary N (N|1) 8 bytes per graph of workspace
  fit = max(N for N in range(33, 193) if lib.lnz_lanczos_ritz_workspace_bytes(64, N) == 0)
  assert 100 <= fit < 128
  for N in range(33, fit + 1):
    assert lib.lnz_lanczos_ritz_workspace_bytes(64, N) == 0
  for N in (fit + 1, 128, 150):
    assert lib.lnz_lanczos_ritz_workspace_bytes(64, N) == 64 * N * (N | 1) * 8
  assert lib.lnz_lanczos_ritz_workspace_bytes(3, 192) == 3 * 192 * 193 * 8
  assert lib.lnz_lanczos_ritz_workspace_bytes(3, 193) == 0 and lib.lnz_lanczos_ritz_workspace_bytes(3, 32) == 0
  assert lib.lnz_lanczos_ritz(null, 1, 1, 1, one, 4, 10, 20, one, one, null, null) == _lib.LNZ_EINVAL
  assert lib.lnz_spectral_gains(one, 4, 20, (C.c_int32 * 20)(), 20, 7, 0, one, one, null) == _lib.LNZ_ENOTSUP
  a = _lib.ForwardArgs()
  assert lib.lnz_lanczosnet_forward(C.byref(a), null) == _lib.LNZ_EINVAL
  a.B, a.N, a.K, a.num_layer, a.dhid, a.din0, a.dout = 4, 33, 20, 7, 128, 64, 16
  assert lib.lnz_lanczosnet_forward(C.byref(a), null) == _lib.LNZ_ENOTSUP
  a.N, a.dhid = 26, 96
  assert lib.lnz_lanczosnet_forward(C.byref(a), null) == _lib.LNZ_ENOTSUP
  with pytest.raises(_lib.NotSupported):
    _lib.check(_lib.LNZ_ENOTSUP)
  assert lib.lnz_packed_rows_k8_size(128, 1920) == 128 * 1920
  assert lib.lnz_packed_rows_k8_size(17, 20) == 32 * 24


def test_torch_extension_registers_the_ops_and_refuses_cpu_tensors():
  """liblanczosnet_torch.so (csrc/torch_ext.cpp): the forward step's ops live in the dispatcher as
  torch.ops.lanczosnet.*, implemented for the HIP ("CUDA") key only — a CPU tensor finds no kernel."""
  from lanczosnet_amd import _lib, _torch_ext
  _torch_ext.load()
  ns = torch.ops.lanczosnet
  assert ns.abi_version() == _lib.ABI_VERSION
  for name in ('laplacian_l4', 'lanczos_ritz', 'prepare_batch', 'spectral_gains', 'forward',
               'unsorted_segment_sum_forward', 'unsorted_segment_sum_backward'):
    assert hasattr(ns, name), name
  with pytest.raises((NotImplementedError, RuntimeError)):
    ns.laplacian_l4(torch.zeros(1, 4, 4, 1), torch.zeros(1, dtype=torch.int32))
  with pytest.raises((NotImplementedError, RuntimeError)):
    ns.unsorted_segment_sum_forward(torch.zeros(1, 4, 2), torch.zeros(1, 4, dtype=torch.int64), 3)
  # the front end (lanczosnet_amd.ops) refuses before it gets there, with the module's message
  from lanczosnet_amd import ops
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ops.laplacian_l4(torch.zeros(1, 4, 4, 1), torch.zeros(1, dtype=torch.int32))


def _qm8_config():
  from lanczosnet_amd.utils.arg_helper import make_model_config
  import oracle
  return make_model_config(dict(oracle.DEFAULT_QM8_CFG))


def test_module_state_dict_and_init_rng_parity_with_reference():
  """Same keys / shapes / parameter count as the reference class and — under the same
  torch.manual_seed — the same initial weights (creation + init order preserved).
  Fixture: tests/golden/init_parity.npz from the unmodified reference."""
  from lanczosnet_amd.model import LanczosNet
  g = load_golden('init_parity.npz')
  torch.manual_seed(1234)
  net = LanczosNet(_qm8_config())
  sd = net.state_dict()
  assert sorted(sd.keys()) == list(g['keys'])
  assert sum(int(v.numel()) for v in sd.values()) == int(g['num_params']) == 1851465
  for k, shp, s, f in zip(g['keys'], g['shapes'], g['sums'], g['first']):
    assert repr(tuple(sd[k].shape)) == shp, k
    if str(g['torch_version']) == torch.__version__:
      assert float(sd[k].double().sum()) == float(s), k
      assert float(sd[k].reshape(-1)[0]) == float(f), k


def test_module_refuses_cpu_and_training_backward():
  from lanczosnet_amd.model import LanczosNet
  net = LanczosNet(_qm8_config()).eval()
  B, N, K = 2, 5, 20
  args = (torch.zeros(B, N, dtype=torch.long), torch.zeros(B, N, N, 7), torch.zeros(B, K),
          torch.zeros(B, N, K))
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    net(*args, mask=torch.ones(B, N, dtype=torch.uint8))
  with pytest.raises(ValueError):
    net(*args, mask=None)


def test_config_surface():
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.utils.arg_helper import AttrDict, make_model_config
  from lanczosnet_amd.utils.data_helper import check_dist
  import oracle
  cfg = AttrDict(dict(a=dict(b=1)))
  assert cfg.a.b == 1 and not hasattr(cfg.a, 'dropout')  # hasattr probing (lanczos_net.py:24)
  assert check_dist([1, 2, 'inf']) == [1, 2, 'inf']
  with pytest.raises(ValueError):
    check_dist([1.5])
  bad = make_model_config(dict(oracle.DEFAULT_QM8_CFG), loss='Hinge')
  with pytest.raises(ValueError, match='Non-supported loss'):
    LanczosNet(bad)
  c = make_model_config(dict(oracle.DEFAULT_QM8_CFG))
  c.model.dropout = 0.25
  assert LanczosNet(c).dropout == 0.25


def test_yaml_config_roundtrip(tmp_path):
  from lanczosnet_amd.utils.arg_helper import get_config
  y = tmp_path / 'c.yaml'
  y.write_text('exp_dir: %s\nseed: 1\nmodel:\n  name: LanczosNet\n  hidden_dim: [128, 128]\n'
               'dataset:\n  name: chemistry\n' % tmp_path)
  cfg = get_config(str(y))
  assert cfg.model.hidden_dim == [128, 128] and os.path.isdir(cfg.save_dir)
  assert os.path.exists(os.path.join(cfg.save_dir, 'config.yaml'))


def test_synthetic_batch_schema():
  from lanczosnet_amd.synthetic import draw_batch
  b = draw_batch(64, seed=0)
  assert b['adjs'].shape[1] == b['adjs'].shape[2] == b['n_nodes'].max()
  assert b['n_nodes'].min() >= 8 and b['n_nodes'].max() <= 26
  a = b['adjs']
  np.testing.assert_array_equal(a, a.transpose(0, 2, 1, 3))
  assert a.sum(axis=3).max() == 1.0  # bond channels partition the edges
  for i in range(64):
    n = int(b['n_nodes'][i])
    assert a[i, n:].sum() == 0 and a[i, :, n:].sum() == 0
    assert b['node_mask'][i].sum() == n
    assert a[i].sum() / 2 >= n - 1  # spanning tree => connected


def test_collate_preprocessed_matches_reference_collate():
  """Host collate of reference-format items == the reference's collate_fn output (fixture)."""
  import oracle
  from lanczosnet_amd.dataset import collate_preprocessed
  from lanczosnet_amd.synthetic import draw_batch
  g = load_golden('collate_batch.npz')
  b = draw_batch(int(g['batch_size']), seed=int(g['seed']), n_min=int(g['n_min']),
                 n_max=int(g['n_max']))
  items = []
  for i in range(len(b['n_nodes'])):
    n = int(b['n_nodes'][i])
    adjs = b['adjs'][i, :n, :n]
    Lm = oracle.laplacian_multi_l4(adjs)
    e, V, _ = oracle.graph_laplacian_eigs(adjs.sum(axis=2), graph_laplacian_type='L4')
    items.append(dict(node_feat=b['node_feat'][i, :n], label=b['label'][i:i + 1],
                      L_multi=Lm[:, :, 1:], L_simple_4=Lm[:, :, 0], D_simple=e, V_simple=V))
  out = collate_preprocessed(items, 20)
  np.testing.assert_array_equal(out['node_feat'].numpy(), g['node_feat'])
  np.testing.assert_array_equal(out['node_mask'].numpy(), g['node_mask'])
  np.testing.assert_array_equal(out['label'].numpy(), g['label'])
  np.testing.assert_allclose(out['L'].numpy(), g['L'], atol=1e-7)
  np.testing.assert_allclose(out['D'].numpy(), g['D'], atol=1e-6)
  assert out['V'].shape == g['V'].shape


def test_fold_classes_from_comparison_bits():
  """model/lanczos_net.py `_classes_from_bits`: packed channels that never differed from an earlier
  packed channel join its class; folded channels follow their representative."""
  from lanczosnet_amd.model.lanczos_net import LanczosNet
  f = LanczosNet._classes_from_bits
  bit = lambda c, c2: 1 << (8 * c + c2)  # noqa: E731
  assert f(0, (0, 1)) == (0, 0)
  assert f(bit(1, 0), (0, 1)) == (0, 1)
  assert f(bit(1, 0) | bit(2, 1), (0, 1, 2)) == (0, 1, 0)          # 2 == 0, 1 differs from both
  assert f(bit(1, 0) | bit(2, 0), (0, 1, 2)) == (0, 1, 1)
  assert f(bit(2, 0), (0, 0, 2)) == (0, 0, 2)                       # already folded 1 stays with 0
  assert f(0, (0, 0, 2)) == (0, 0, 0)
  assert f(bit(1, 0) | bit(2, 0) | bit(2, 1) | bit(3, 0) | bit(3, 2), (0, 1, 2, 3)) == (0, 1, 2, 1)


def test_ada_long_diffusion_dist_is_put_in_the_reference_order():
  """The reference walks the powers in ascending order whatever the list says
  (model/ada_lanczos_net.py:262-270): an unsorted list is sorted once, a duplicate refused."""
  import pytest
  import oracle
  from lanczosnet_amd.model import AdaLanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config

  class Small(AdaLanczosNet):
    _spectral_hidden = 8
  cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1], long_diffusion_dist=[7, 5, 30],
             hidden_dim=[128], num_layer=1)
  net = Small(make_model_config(cfg, name='AdaLanczosNet'))
  assert net.long_diffusion_dist == [5, 7, 30]
  cfg['long_diffusion_dist'] = [5, 7, 5]
  with pytest.raises(ValueError):
    Small(make_model_config(cfg, name='AdaLanczosNet'))


def test_torch_extension_registers_every_entry_point_and_is_the_only_binding_of_the_product():
  """csrc/torch_ext_abi.inc is current with the header (tools/gen_torch_ext.py --check); every
  entry point the header declares is reachable as a dispatcher op (raw_<name>, or fused_launch for
  the four argument-block launches); host-side size queries answer without a GPU; and no module of
  the product path marshals through ctypes any more — only lanczosnet_amd/_lib.py (the raw-ABI
  binding kept for these tests) does."""
  import subprocess
  import sys
  import torch
  from lanczosnet_amd import _torch_ext
  assert subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_torch_ext.py'), '--check']).returncode == 0
  _torch_ext.load()
  struct_launches = {'lnz_lanczosnet_forward', 'lnz_lanczosnet_input_grad', 'lnz_lanczosnet_messages',
                     'lnz_lanczosnet_gain_grad'}
  for sym in _header_symbols():
    if sym in ('lnz_abi_version', 'lnz_last_error', 'lnz_last_kernel', 'lnz_stream_create_cu_masked') or \
        sym in struct_launches:
      continue
    assert hasattr(torch.ops.lanczosnet, 'raw_' + sym[4:]), sym
  assert hasattr(torch.ops.lanczosnet, 'fused_launch') and hasattr(torch.ops.lanczosnet, 'cu_masked_stream')
  assert torch.ops.lanczosnet.raw_large_nk(100) == 128
  assert torch.ops.lanczosnet.raw_lanczos_ritz_workspace_bytes(3, 192) == 3 * 192 * 193 * 8
  assert torch.ops.lanczosnet.raw_f32_linear_splits(1024, 4096, 4096) == 1
  assert torch.ops.lanczosnet.raw_f32_linear_splits(1024, 1056, 4096) == 256
  pkg = os.path.join(ROOT, 'lanczosnet_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith('.py') and f != '_lib.py':
        assert 'ctypes' not in open(os.path.join(dirpath, f)).read(), os.path.join(dirpath, f)


def test_no_process_wide_mutable_state_in_the_kernels_sources():
  """no global mutable state in the extension.  A function attribute (dynamic LDS
  above 64 KiB) is PER DEVICE, and the reference's own multi-GPU mechanism drives several devices
  from one process in one thread each (`nn.DataParallel`, runner/qm8_runner.py:62): a
  `static bool attr_set` guard configures the first device only and is written without
  synchronisation.  Any non-const `static` / namespace-scope variable in csrc/ fails here; the one
  allowed objects are the thread-local strings of `lnz_last_error()` / `lnz_last_kernel()`."""
  import re
  csrc = os.path.join(ROOT, 'lanczosnet_amd', 'csrc')
  decl = re.compile(r'^\s*static\s+(?!const\b|constexpr\b|inline\b|__device__|__global__|__host__|'
                    r'__forceinline__|thread_local\b)[\w:<>,\s\*&]+?\b(\w+)\s*(\[[^\]]*\])*\s*(=|;|\{)')
  bad = []
  for f in sorted(os.listdir(csrc)):
    if not f.endswith(('.hip', '.hpp', '.cpp', '.inc')):
      continue
    for i, line in enumerate(open(os.path.join(csrc, f)), 1):
      code = line.split('//')[0]
      if 'attr_set' in code or decl.match(code):
        bad.append('%s:%d: %s' % (f, i, line.strip()))
  assert not bad, '\n'.join(bad)
  tl = [l for f in os.listdir(csrc) if f.endswith(('.hip', '.hpp', '.cpp'))
        for l in open(os.path.join(csrc, f)) if 'thread_local' in l.split('//')[0]]
  assert len(tl) == 2 and 'g_err' in tl[0] + tl[1] and 'g_kernel' in tl[0] + tl[1], tl


def test_package_alias_resolves_to_the_same_modules():
  """`lanczosnetwork_amd` (the name of the build brief) is an alias of `lanczosnet_amd`, not a copy."""
  import lanczosnetwork_amd.model as m1
  import lanczosnet_amd.model as m2
  import lanczosnetwork_amd.ops as o1
  import lanczosnet_amd.ops as o2
  assert m1 is m2 and o1 is o2 and m1.LanczosNet is m2.LanczosNet


def test_round6_host_side_queries_and_argument_checks_without_gpu():
  """Host-side size queries and argument validation of the r06 entries (no launch happens): the K-step
  workspace layout, the image capacity rule, the head-backward workspace, and the refusals."""
  import torch
  from lanczosnet_amd import _lib, _torch_ext, ops
  _torch_ext.load()
  ns = torch.ops.lanczosnet
  B, N = 256, 2048
  basis = B * 64 * N * 8
  assert ns.raw_lanczos_ritz_kstep_workspace_bytes(B, N, 0, 0) == basis
  assert ns.raw_lanczos_ritz_kstep_workspace_bytes(B, N, 1, 0) == basis            # symmetric: same workspace
  img = ns.raw_lanczos_ritz_kstep_workspace_bytes(B, N, 3, 64) - basis
  # (image values + columns, slab widths, per-graph flags, per-row counts; 256-byte aligned regions)
  assert img >= B * (N // 64) * 64 * 64 * 6 and img < B * (N // 64) * 64 * 64 * 6 + B * (N // 64) * 4 + B * 4 + B * N * 4 + 5 * 256
  assert ns.raw_lanczos_ritz_kstep_workspace_bytes(B, N, 3, 256) > ns.raw_lanczos_ritz_kstep_workspace_bytes(B, N, 3, 64)
  assert [ops.kstep_row_cap(n) for n in (200, 512, 1024, 2048)] == [64, 64, 128, 256]
  assert ns.raw_head_backward_workspace_floats(16, 256) == 256 * (17 * 128 + 32 + 128)
  assert ns.raw_head_backward_workspace_floats(32, 256) == 0                         # head width <= 31
  lib = _lib.load()
  null = C.c_void_p(None)
  # the K-step entry refuses before it launches: missing pointers, a capacity that is not a multiple of 8
  rc = lib.lnz_lanczos_ritz_kstep(null, 0, 0, 1, null, 1, 256, 8, 8, 2, 12, null, 0, null, null, null, null, null)
  assert rc == _lib.LNZ_EINVAL
  assert b'row_cap' in lib.lnz_last_error()
  assert lib.lnz_head_backward(null, null, null, null, null, null, null, null, 4, 20, 16, 128, 256,
                               null, null, null, null, null, null, null) == _lib.LNZ_EINVAL
  assert lib.lnz_node_extents(null, 4, 20, null, null, null, null) == _lib.LNZ_EINVAL
  assert lib.lnz_last_kernel() is not None


def test_sparse_large_graph_entries_refuse_before_they_launch():
  """The round's large-graph additions answer bad requests on the host (no GPU needed): missing
  pointers, row capacities that are not multiples of 8 / below 32, too many readout columns, a
  column stride the K-step entry cannot read, an image request without the compact mode; and the
  row-capacity rule of the Python side."""
  import torch
  from lanczosnet_amd import _lib, ops
  lib = _lib.load()
  null = C.c_void_p(None)
  one = C.c_void_p(16)   # (a non-NULL, aligned "pointer": every call below fails before anything is touched)
  assert lib.lnz_large_sparse_image(null, 0, 0, 0, 0, 1, 64, 2, 64, null, null, null, null, null) == _lib.LNZ_EINVAL
  assert lib.lnz_large_sparse_image(one, 0, 0, 0, 0, 1, 64, 2, 20, one, null, one, one, null) == _lib.LNZ_EINVAL
  assert b'row_cap' in lib.lnz_last_error()
  assert lib.lnz_large_sparse_image(one, 0, 0, 0, 0, 1, 70000, 2, 64, one, null, one, one, null) == _lib.LNZ_ENOTSUP
  assert lib.lnz_large_sparse_conv(one, one, 24, one, 1, 64, 1, one, null) == _lib.LNZ_EINVAL
  assert lib.lnz_large_sparse_conv(null, one, 64, one, 1, 64, 1, one, null) == _lib.LNZ_EINVAL
  assert lib.lnz_large_sparse_conv_f32(one, null, one, 64, one, 1, 64, 1, one, null) == _lib.LNZ_EINVAL
  assert lib.lnz_large_gemm1_rows(one, 8, 10, one, 1, 64, one, null) == _lib.LNZ_EINVAL      # ldx < din
  assert lib.lnz_large_gemm1_rows(one, 200, 200, one, 1, 64, one, null) == _lib.LNZ_ENOTSUP   # width > 128
  assert lib.lnz_large_pack_vectors(one, 1, 64, 65, 1, one, null) == _lib.LNZ_ENOTSUP         # K > 64
  assert lib.lnz_large_head(one, one, one, one, 1, 64, 17, one, null) == _lib.LNZ_ENOTSUP
  assert lib.lnz_large_head(null, one, one, one, 1, 64, 2, one, null) == _lib.LNZ_EINVAL
  assert lib.lnz_stream_create_cu_masked(0, 64, null) == _lib.LNZ_EINVAL
  # column stride 3; stride 2 without the compact image / without the fallback flags
  args = lambda sc, flags, fb: (one, 0, 0, sc, null, 1, 256, 8, 8, flags, 64, one, 1 << 40, one, one, null, fb, null)  # noqa: E731
  assert lib.lnz_lanczos_ritz_kstep(*args(3, 2, one)) == _lib.LNZ_ENOTSUP
  assert lib.lnz_lanczos_ritz_kstep(*args(2, 1, one)) == _lib.LNZ_ENOTSUP
  assert lib.lnz_lanczos_ritz_kstep(*args(2, 2, null)) == _lib.LNZ_ENOTSUP
  rc = lib.lnz_lanczos_ritz_kstep_image(one, 0, 0, 1, null, 1, 256, 8, 8, 1, 64, one, 1 << 40, one, one, null, null,
                                        one, null, one, 64, one, null)
  assert rc == _lib.LNZ_EINVAL and b'COMPACT' in lib.lnz_last_error()
  assert [ops.large_sparse_row_cap(n) for n in (100, 256, 1024, 2048, 8192, 65536)] == [32, 32, 32, 64, 256, 256]
