"""ctypes binding of the C ABI in include/lanczosnet_hip.h.

The shared library is built in-tree (`lanczosnet_amd/csrc/liblanczosnet_hip.so`, see
`__graft_entry__.build()` / `make -C lanczosnet_amd/csrc`).  There is NO fallback: if the
library is missing or a symbol does not resolve, importing the product path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LANCZOSNET_HIP_LIB overrides the path (A/B builds of the same ABI); default = in-tree build
LIB_PATH = os.environ.get('LANCZOSNET_HIP_LIB') or os.path.join(_HERE, 'csrc', 'liblanczosnet_hip.so')

LNZ_OK, LNZ_EINVAL, LNZ_ELAUNCH, LNZ_ENOTSUP = 0, -1, -2, -3
ABI_VERSION = 7


class LnzError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__('lanczosnet_hip error %d: %s' % (code, msg))
    self.code = code


class NotSupported(LnzError, NotImplementedError):
  pass


class ForwardArgs(C.Structure):
  """struct lnz_forward_args (include/lanczosnet_hip.h)."""
  _fields_ = [
      ('B', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
      ('num_layer', C.c_int32),
      ('din0', C.c_int32), ('dhid', C.c_int32), ('dout', C.c_int32),
      ('n_short', C.c_int32), ('n_long', C.c_int32), ('n_edge', C.c_int32),
      ('short_dist', C.c_int32 * 8),
      ('node_feat', C.c_void_p), ('node_feat_f', C.c_void_p), ('embedding', C.c_void_p),
      ('num_atom', C.c_int32), ('filter_kind', C.c_int32),
      ('mask', C.c_void_p), ('Lp', C.c_void_p), ('V', C.c_void_p), ('G', C.c_void_p),
      ('Wp', C.c_void_p), ('bias', C.c_void_p),
      ('w_off', C.c_int64 * 16), ('b_off', C.c_int64 * 16),
      ('Wp_head', C.c_void_p), ('bias_head', C.c_void_p),
      ('score', C.c_void_p), ('state_out', C.c_void_p),
      ('gemm_mode', C.c_int32), ('plan', C.c_void_p), ('n_wg', C.c_void_p), ('plan_wg_cap', C.c_int),
      ('act_out', C.c_void_p), ('act', C.c_void_p), ('dy', C.c_void_p), ('dx0', C.c_void_p),
      ('bwd_din0', C.c_int32), ('x0', C.c_void_p), ('msg', C.c_void_p), ('msg_layer', C.c_int32), ('ident', C.c_void_p), ('row_off', C.c_void_p), ('dgains', C.c_void_p),
      ('dy_compact', C.c_void_p), ('dy_compact_rows', C.c_int64), ('dbias_part', C.c_void_p),
      ('dbias_part_cap', C.c_int32),
      ('strips', C.c_void_p), ('n_strips', C.c_void_p), ('strip_cap', C.c_int),
  ]


# name -> (restype, argtypes); every symbol include/lanczosnet_hip.h declares
_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
SIGNATURES = {
    'lnz_abi_version': (C.c_int, []),
    'lnz_last_error': (C.c_char_p, []),
    'lnz_last_kernel': (C.c_char_p, []),
    'lnz_laplacian_l4': (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    'lnz_laplacian': (C.c_int, [_P, _P, _I, _I, _I, _I, C.c_double, _P, _P]),
    'lnz_lanczos_ritz': (C.c_int, [_P, _L, _L, _L, _P, _I, _I, _I, _P, _P, _P, _P]),
    'lnz_lanczos_ritz_workspace_bytes': (C.c_int64, [_I, _I]),
    'lnz_lanczos_ritz_ws': (C.c_int, [_P, _L, _L, _L, _P, _I, _I, _I, _P, _P, _P, _P, _L, _I, _P]),
    'lnz_tridiag_eigh': (C.c_int, [_P, _P, _I, _I, _P, _P, _P]),
    'lnz_lanczos_ritz_large_workspace_bytes': (C.c_int64, [_I, _I]),
    'lnz_lanczos_ritz_large': (C.c_int, [_P, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'lnz_lanczos_ritz_large_sym': (C.c_int, [_P, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'lnz_lanczos_ritz_kstep_workspace_bytes': (C.c_int64, [_I, _I, _I, _I]),
    'lnz_lanczos_ritz_kstep': (C.c_int, [_P, _L, _L, _L, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P]),
    'lnz_lanczos_ritz_kstep_image': (C.c_int, [_P, _L, _L, _L, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'lnz_stream_create_cu_masked': (C.c_int, [_I, _I, _P]),
    'lnz_head_backward_workspace_floats': (C.c_int64, [_I, _I]),
    'lnz_head_backward': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'lnz_node_extents': (C.c_int, [_P, _I, _I, _P, _P, _P, _P]),
    'lnz_large_nk': (C.c_int64, [_I]),
    'lnz_large_pack_operators': (C.c_int, [_P, _L, _L, _L, _L, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'lnz_large_pack_operators_fold': (C.c_int, [_P, _L, _L, _L, _L, _P, _I, _I, _I, _I, _I,
                                                C.POINTER(C.c_int32), _I, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32), _P, _P, _P, _P]),
    'lnz_large_gemm1': (C.c_int, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _P]),
    'lnz_large_spectral': (C.c_int, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'lnz_large_conv': (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'lnz_large_sparse_image': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'lnz_large_spectral_gemm1_rows': (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    'lnz_large_head': (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    'lnz_large_sparse_conv_f32': (C.c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _P]),
    'lnz_large_pack_vectors': (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    'lnz_large_gemm1_rows': (C.c_int, [_P, _I, _I, _P, _I, _I, _P, _P]),
    'lnz_large_sparse_conv': (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _P, _P]),
    'lnz_midgraph_workspace_floats': (C.c_int64, [_I, _I, _I]),
    'lnz_midgraph_forward': (C.c_int, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                                       _I, _I, _I, _P, _P, _P, _P]),
    'lnz_packed_rows_k8_size': (C.c_int64, [_I, _I]),
    'lnz_pack_rows_k8': (C.c_int, [_P, _I, _I, _L, _P, _P]),
    'lnz_pack_rows_k8_split': (C.c_int, [_P, _I, _I, _L, _P, _P]),
    'lnz_pack_bias_rows': (C.c_int, [_P, _I, _P, _P]),
    'lnz_pack_laplacian': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _P, _P]),
    'lnz_collate_qm8': (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    'lnz_plan_wg_cap': (C.c_int, [_I, _I]),
    'lnz_pack_laplacian_plan': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'lnz_pack_laplacian_ident': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _P, _P, _P]),
    'lnz_prepare_batch_prev_gains': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, C.POINTER(C.c_int32), _I, _I, _P, _P, _P, _P, _P]),
    'lnz_prepare_batch': (C.c_int, [_P, _L, _L, _L, _L, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'lnz_plan_batch': (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    'lnz_plan_strips': (C.c_int, [_P, _I, _I, _I, _P, _P, _P]),
    'lnz_strip_cap': (C.c_int, [_I]),
    'lnz_plan_tiles': (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    'lnz_spectral_mlp_pack_size': (C.c_int64, [_I]),
    'lnz_pack_spectral_mlp': (C.c_int, [_P] * 8 + [_I, _P, _P]),
    'lnz_pack_spectral_mlp_layers': (C.c_int, [_P, _I, _I, _P, _P]),
    'lnz_spectral_mlp_grad_parts': (C.c_int, [_I, _I, _I]),
    'lnz_spectral_gains_rows_split': (C.c_int, [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    'lnz_spectral_gains_rows_split_to': (C.c_int, [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P]),
    'lnz_split_laplacian_pack': (C.c_int, [_P, _L, _P]),
    'lnz_split_laplacian_pack_to': (C.c_int, [_P, _L, _P, _P]),
    'lnz_spectral_mlp_grad_floats': (C.c_int, [_I]),
    'lnz_embedding_grad': (C.c_int, [_P, _I, _I, _P, _L, _L, _I, _I, _I, _P, _P]),
    'lnz_spectral_mlp_grad': (C.c_int, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    'lnz_spectral_gains': (C.c_int, [_P, _I, _I, C.POINTER(C.c_int32), _I, _I, _I, _P, _P, _P]),
    'lnz_spectral_gains_rows': (C.c_int, [_P, _I, _I, C.POINTER(C.c_int32), _I, _I, _I, _P, _P, _P, _P, _P]),
    'lnz_lanczosnet_input_grad': (C.c_int, [C.POINTER(ForwardArgs), _P]),
    'lnz_lanczosnet_messages': (C.c_int, [C.POINTER(ForwardArgs), _P]),
    'lnz_lanczosnet_gain_grad': (C.c_int, [C.POINTER(ForwardArgs), _P]),
    'lnz_lanczosnet_forward': (C.c_int, [C.POINTER(ForwardArgs), _P]),
    'lnz_forward_args_size': (C.c_int64, []),
    'lnz_ada_graph_laplacian': (C.c_int, [_P, _P, _I, _P, _I, _P, _L, _L, _L, _I, _I, _P, _P]),
    'lnz_ada_lanczos_layer': (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    'lnz_ada_t_powers': (C.c_int, [_P, _I, _I, C.POINTER(C.c_int32), _I, _P, _P]),
    'lnz_ada_symmetrize_filters': (C.c_int, [_P, _I, _I, _I, _P, _P]),
    'lnz_split_f16x3': (C.c_int, [_P, _I, _I, _I, _P, C.c_float, _I, _I, _P, _P]),
    'lnz_f16x3_split': (C.c_int, [_P, _I, _I, _L, C.c_float, _I, _I, _P, _P, _P]),
    'lnz_f16x3_linear': (C.c_int, [_P, _P, _I, _P, _P, _I, _P, C.c_float, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    'lnz_f16x3_linear_splits': (C.c_int, [_I, _I, _I]),
    'lnz_ada_lanczos_f64_workspace_doubles': (C.c_int64, [_I]),
    'lnz_ada_laplacian_f64_state_doubles': (C.c_int64, [_I, _I]),
    'lnz_ada_graph_laplacian_f64': (C.c_int, [_P, _I, _P, C.c_int64, C.c_int64, C.c_int64, _I, _I, _P, _P, _P]),
    'lnz_ada_graph_laplacian_f64_backward': (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P]),
    'lnz_ada_t_powers_f64': (C.c_int, [_P, _I, _I, _P, _I, _P, _P, _P]),
    'lnz_ada_t_powers_f64_backward': (C.c_int, [_P, _I, _I, _P, _I, _P, _P, _P, _P]),
    'lnz_ada_lanczos_layer_f64': (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'lnz_ada_lanczos_layer_f64_backward': (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'lnz_f32_linear': (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    'lnz_f32_linear_splits': (C.c_int, [_I, _I, _I]),
    'lnz_f32_linear_workspace_floats': (C.c_int64, [_I, _I, _I]),
    'lnz_unsorted_segment_sum_forward': (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'lnz_unsorted_segment_sum_backward': (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
}

_lib = None


def load():
  """Load (once) and type the shared library.  Raises ImportError if it is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        'lanczosnet_amd: HIP library not built: %s is missing. Build it with '
        '`python -c "import __graft_entry__ as g; g.build()"` or '
        '`make -C lanczosnet_amd/csrc` (needs hipcc, --offload-arch=gfx950). '
        'There is no CPU fallback for this path.' % LIB_PATH)
  # One HIP runtime per process: torch ships its own libamdhip64.so.7; importing torch first
  # makes the dynamic loader resolve this library's DT_NEEDED libamdhip64.so.7 to that same,
  # already-initialised runtime (otherwise /opt/rocm's copy is loaded beside it and sees no
  # device / no torch stream).
  import torch  # noqa: F401
  lib = C.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError (loud) if the symbol is missing
    fn.restype = res
    fn.argtypes = args
  ver = lib.lnz_abi_version()
  if ver != ABI_VERSION:
    raise ImportError('lanczosnet_amd: ABI version mismatch: library %d, binding %d (rebuild)' %
                      (ver, ABI_VERSION))
  if lib.lnz_forward_args_size() != C.sizeof(ForwardArgs):
    raise ImportError('lanczosnet_amd: lnz_forward_args layout mismatch: library %d B, binding %d B'
                      % (lib.lnz_forward_args_size(), C.sizeof(ForwardArgs)))
  _lib = lib
  return lib


def check(code):
  if code == LNZ_OK:
    return
  msg = load().lnz_last_error().decode('utf-8', 'replace')
  if code == LNZ_ENOTSUP:
    raise NotSupported(code, msg)
  raise LnzError(code, msg)
