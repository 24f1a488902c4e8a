"""Torch-tensor front end of the C ABI (device memory + streams are torch's; the compute is the
HIP library).  Every function requires CUDA(HIP) tensors and raises otherwise — no CPU path.

Reference interfaces replaced (paths relative to the reference repo):
  laplacian_l4        utils/data_helper.py:92-116,155-156 ; dataset/get_qm8_data.py:62-75
  lanczos_ritz        utils/data_helper.py:197-223 (eigh + |lambda| sort) ; dataset/qm8.py:264-291
  spectral_gains      model/lanczos_net.py:110-113,118-121,146-149
  lanczosnet_forward  model/lanczos_net.py:114-117,154-194
  unsorted_segment_sum operators/functions/unsorted_segment_sum.py:8-44
"""
import torch

import os

from . import _lib, _torch_ext

# Every kernel is reached through the torch extension (csrc/torch_ext.cpp -> torch.ops.lanczosnet.*:
# ATen dtype / device checks, device guard, the CURRENT HIP stream of the calling thread, no Python-
# side pointer marshalling): the high-level ops that allocate their outputs (laplacian_l4,
# lanczos_ritz, prepare_batch, spectral_gains, forward, unsorted_segment_sum_*), `fused_launch` for
# the four launches that take an argument block, and one `raw_<name>` op per remaining C entry point
# (generated from include/lanczosnet_hip.h by tools/gen_torch_ext.py).  The raw C ABI itself is what
# a host without torch binds (examples/ritz_pairs.c; lanczosnet_amd/_lib.py for the ABI tests).


def _map_error(e):
  """'lanczosnet_hip error <code>: <message>' -> the binding's exception types."""
  msg = str(e)
  if 'lanczosnet_hip error ' not in msg:
    return e
  tail = msg.split('lanczosnet_hip error ', 1)[1]
  try:
    code = int(tail.split(':', 1)[0])
  except ValueError:
    return e
  text = tail.split(':', 1)[1].strip() if ':' in tail else tail
  text = text.split('\nException raised from', 1)[0].strip()
  if code == _lib.LNZ_ENOTSUP:
    return _lib.NotSupported(code, text)
  return _lib.LnzError(code, text)


class _ExtNamespace(object):
  """torch.ops.lanczosnet (optionally with a name prefix) with the C ABI's error codes mapped back
  to LnzError / NotSupported."""

  def __init__(self, prefix=''):
    self._prefix = prefix

  def __getattr__(self, name):
    op = getattr(torch.ops.lanczosnet, self._prefix + name)

    def call(*args):
      try:
        return op(*args)
      except RuntimeError as e:
        m = _map_error(e)
        if m is e:
          raise
        raise m from None
    self.__dict__[name] = call   # resolved once
    return call


_EXT_NS = _ExtNamespace()
_RAW_NS = _ExtNamespace('raw_')


def _ext():
  _torch_ext.load()   # ImportError (loud) if the extension is not built
  return _EXT_NS


def _abi():
  """The C entry points as dispatcher ops: _abi().<name>(...) = lnz_<name>(..., current stream);
  tensors go in as tensors (None = NULL), host int arrays as lists."""
  _torch_ext.load()
  return _RAW_NS


NotSupported = _lib.NotSupported
LnzError = _lib.LnzError


def last_kernel():
  """lnz_last_kernel(): the kernel (with template arguments) the calling thread's last fused
  forward / input-gradient launch selected."""
  return torch.ops.lanczosnet.last_kernel()


def _need_cuda(*tensors):
  for t in tensors:
    if t is None:
      continue
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
      raise RuntimeError(
          'lanczosnet_amd: this path runs only on an AMD GPU (HIP); got a %s tensor. '
          'There is no CPU fallback.' % (getattr(t, 'device', type(t)),))


def _f32c(t):
  return t.to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------ R1
def laplacian_l4(adjs, n_nodes):
  """adjs [B,N,N,E] (bond-type adjacency, zero padded), n_nodes [B] -> L [B,N,N,E+1] float32."""
  _need_cuda(adjs, n_nodes)
  adjs = _f32c(adjs)
  n_nodes = n_nodes.to(torch.int32).contiguous()
  B, N, N2, E = adjs.shape
  assert N == N2 and n_nodes.shape == (B,)
  return _ext().laplacian_l4(adjs, n_nodes)


def laplacian(adjs, n_nodes, kind='L4', alpha=0.5):
  """get_laplacian of utils/data_helper.py:119-166 on the device, every kind 'L1' .. 'L7'
  (alpha: the 'L6' exponent).  Same layouts as laplacian_l4: adjs [B,N,N,E], n_nodes [B] ->
  L [B,N,N,E+1] float32, channel 0 = simple graph, 1 + e = bond type e."""
  _need_cuda(adjs, n_nodes)
  kinds = ('L1', 'L2', 'L3', 'L4', 'L5', 'L6', 'L7')
  if kind not in kinds:
    raise ValueError('Unsupported Graph Laplacian!')   # (the reference's message, :164)
  adjs = _f32c(adjs)
  n_nodes = n_nodes.to(torch.int32).contiguous()
  B, N, N2, E = adjs.shape
  assert N == N2 and n_nodes.shape == (B,)
  L = torch.empty((B, N, N, E + 1), dtype=torch.float32, device=adjs.device)
  with torch.cuda.device(adjs.device):
    _abi().laplacian(adjs, n_nodes, B, N, E, kinds.index(kind) + 1,
                                 float(alpha), L)
  return L


# ------------------------------------------------------------------------------------- R2 + R6
def lanczos_ritz(A, n_nodes, K, return_info=False, kernel='auto'):
  """Batched Lanczos -> tridiagonal eigensolve -> Ritz select.

  A: [B,N,N] float32 symmetric (any strides: pass `L[..., 0]` of a channels-last Laplacian
  without copying), N <= 192.  n_nodes: [B] real node counts (rows/cols >= n are ignored).
  Returns D [B,K], V [B,N,K] exactly like the collated `D`, `V` of dataset/qm8.py:264-291 /
  dataset/graph_data.py:262-287.
  kernel: 'auto' (wavefront per graph up to N = 32, workgroup per graph above), 'workgroup'
  (workgroup per graph at any N), 'workgroup_ws' (the same with the fp64 basis in a device
  workspace instead of LDS — what 'auto' does for N > 108), 'workgroup_ql' (the workgroup kernel
  with the QL sweep instead of its parallel tridiagonal eigensolver — its fallback, forced),
  'workgroup_mw' (the Lanczos phase on all eight waves where the wave-level form would run — the
  arithmetic of 'workgroup_ws' in the same order), 'workgroup_p1' / '_p2' / '_p4' (the wave-level
  Lanczos phase with one, two, four parts per row group instead of the number chosen by size)."""
  _need_cuda(A, n_nodes)
  assert A.dim() == 3 and A.shape[1] == A.shape[2] and A.dtype == torch.float32
  assert kernel in ('auto', 'workgroup', 'workgroup_ws', 'workgroup_ql', 'workgroup_mw', 'workgroup_p1',
                    'workgroup_p2', 'workgroup_p4')
  B, N, _ = A.shape
  n_nodes = n_nodes.to(torch.int32).contiguous()
  if kernel == 'auto' and N > RITZ_FULL_MAX_N:
    # beyond one workgroup's reach the full-length decomposition is not offered: the K-step Krylov
    # method — the reference's OTHER branch, use_eigen_decomp=False (utils/data_helper.py:205-208)
    _warn_once('lanczos_ritz: %d > %d nodes — Ritz pairs of the K-step Lanczos recurrence (the '
               'reference\'s use_eigen_decomp=False / eigsh branch, utils/data_helper.py:205-208), not of '
               'the full decomposition; converged leading pairs agree' % (N, RITZ_FULL_MAX_N))
    return lanczos_ritz_kstep(A, n_nodes, K, K, return_info=return_info)
  if kernel == 'auto':
    D, V, info = _ext().lanczos_ritz(A, n_nodes, K)
    return (D, V, info) if return_info else (D, V)
  D = torch.empty((B, K), dtype=torch.float32, device=A.device)
  V = torch.empty((B, N, K), dtype=torch.float32, device=A.device)
  info = torch.empty((B,), dtype=torch.int32, device=A.device) if return_info else None
  sb, sr, sc = A.stride()
  with torch.cuda.device(A.device):
    if kernel == 'auto' and N <= 32:
      _abi().lanczos_ritz(A, sb, sr, sc, n_nodes, B, N, K, D,
                                      V, info)
    else:
      # the workspace (if any) comes from torch's caching allocator, not from a hipMallocAsync
      flags = {'workgroup_ws': 1, 'workgroup_ql': 2, 'workgroup_mw': 4, 'workgroup_p1': 8, 'workgroup_p2': 16,
               'workgroup_p4': 24}.get(kernel, 0)
      need = B * N * (N | 1) * 8 if flags & 1 else _abi().lanczos_ritz_workspace_bytes(B, N)
      ws = torch.empty((need,), dtype=torch.uint8, device=A.device) if need else None
      _abi().lanczos_ritz_ws(A, sb, sr, sc, n_nodes, B, N, K, D,
                                         V, info, ws, need, flags)
  return (D, V, info) if return_info else (D, V)


RITZ_FULL_MAX_N = 192     # lnz_lanczos_ritz: one wavefront (N <= 32) / one workgroup per graph



def kstep_row_cap(N):
  """Default sliced-ELL row capacity of the compacted K-step path: N / 8 entries per row, between
  64 and 256 (a multiple of 8).  The capacity only sizes the workspace (6 bytes x N x capacity per
  graph: 3 MB at N = 2048) — a step reads the slabs' real widths; a graph with a longer row takes
  the dense stream."""
  return max(64, min(256, (N // 8 + 7) // 8 * 8))
_WARNED = set()


def _warn_once(msg):
  if msg not in _WARNED:
    _WARNED.add(msg)
    import warnings
    warnings.warn(msg, UserWarning, stacklevel=3)


def lanczos_ritz_kstep(A, n_nodes, M, K, symmetric=True, compact=True, row_cap=None,
                       workspace=None, return_info=False, return_fallback=False, conv_image=None):
  """lnz_lanczos_ritz_kstep: M-step Lanczos Ritz pairs (top K by |theta|) of a ragged batch of
  large dense-stored graphs — the reference's `eigsh` branch (utils/data_helper.py:205-208).
  A [B,N,N] float32, any N <= 2048, n_nodes [B] or None.  Read in place: contiguous rows, or (compact)
  channel 0 of a channels-last pair — `L[..., 0]` of the collated [B,N,N,2] — with 16-byte aligned
  rows and N a multiple of 4; anything else is copied into an aligned buffer first.
  compact: read A once and run the steps on its sliced-ELL image (graphs with a row of more than
  row_cap nonzeros take the dense stream in the same call — for the channels-last view after a look
  at the fallback flags and a copy); symmetric: the dense stream reads the upper chunk blocks only.
  conv_image: None, or the row capacity of a LargeSparseImage to leave behind (lnz_lanczos_ritz_kstep_image:
  the conv layers' image from the SAME pass over A; only when A is read in place — with a
  contiguous A the caller vouches that A is the batch's one operator).  The image is appended to
  the result (None when A had to be copied or the batch fell back to the dense stream).
  Returns D [B,K], V [B,N,K] (+ info [B] = steps taken) (+ fallback [B]) (+ image)."""
  _need_cuda(A, n_nodes, workspace)
  assert A.dim() == 3 and A.shape[1] == A.shape[2] and A.dtype == torch.float32
  B, N, _ = A.shape
  Np = (N + 3) // 4 * 4
  aligned = Np == N and A.stride(1) % 4 == 0 and A.stride(0) % 4 == 0 and A.data_ptr() % 16 == 0
  in_place = aligned and (A.stride(2) == 1 or (compact and A.stride(2) == 2))
  if in_place and A.stride(2) == 2:
    # the pair form reads float4s: the last one of a row ends one element behind the row's last
    # entry — inside the storage for channel 0 of a [.., N, 2] block, not for every stride-2 view
    last = A.storage_offset() + (B - 1) * A.stride(0) + (N - 1) * A.stride(1) + 2 * N
    in_place = last * 4 <= A.untyped_storage().nbytes()
  if not in_place:
    Ap = torch.zeros((B, Np, Np), dtype=torch.float32, device=A.device)
    Ap[:, :N, :N] = A
    if n_nodes is None:
      n_nodes = torch.full((B,), N, dtype=torch.int32, device=A.device)
    A = Ap
  if n_nodes is not None:
    n_nodes = n_nodes.to(torch.int32).contiguous()
  if not (0 < K <= M <= min(Np, 64)):
    raise _lib.NotSupported(_lib.LNZ_ENOTSUP, 'lanczos_ritz_kstep: K=%d <= M=%d <= 64 Lanczos steps (and M <= N=%d) '
                            'required: the K-step branch serves up to 64 Ritz pairs' % (K, M, N))
  flags = (1 if symmetric else 0) | (2 if compact else 0)
  cap = int(row_cap if row_cap is not None else kstep_row_cap(Np)) if compact else 0
  need = _abi().lanczos_ritz_kstep_workspace_bytes(B, Np, flags, cap)
  if workspace is None or workspace.numel() * workspace.element_size() < need:
    workspace = torch.empty((need,), dtype=torch.uint8, device=A.device)
  D = torch.empty((B, K), dtype=torch.float32, device=A.device)
  V = torch.empty((B, Np, K), dtype=torch.float32, device=A.device)
  info = torch.empty((B,), dtype=torch.int32, device=A.device) if return_info else None
  strided = A.stride(2) != 1
  fb = torch.empty((B,), dtype=torch.int32, device=A.device) if ((return_fallback or strided) and compact) else None
  img = None
  with torch.cuda.device(A.device):
    if conv_image is not None and compact and in_place:
      ccap = int(conv_image)
      img = LargeSparseImage(torch.empty((B, N, ccap), dtype=torch.int32, device=A.device),
                             torch.empty((B, N), dtype=torch.int32, device=A.device),
                             torch.empty((1,), dtype=torch.int32, device=A.device), ccap,
                             torch.empty((B, N, ccap), dtype=torch.float32, device=A.device))
      _abi().lanczos_ritz_kstep_image(A, A.stride(0), A.stride(1), A.stride(2), n_nodes, B, Np, M, K, flags,
                                      cap, workspace, workspace.numel() * workspace.element_size(), D, V,
                                      info, fb, img.entries, img.values, img.counts, ccap, img.flags)
    else:
      _abi().lanczos_ritz_kstep(A, A.stride(0), A.stride(1), A.stride(2), n_nodes, B, Np, M, K, flags, cap,
                                workspace, workspace.numel() * workspace.element_size(), D, V, info, fb)
  if strided and bool(fb.any()):
    # the dense streams read contiguous rows: a batch with a graph beyond the image's row capacity
    # is copied after all (the flags are the only host read of this path)
    return lanczos_ritz_kstep(A.contiguous(), n_nodes, M, K, symmetric=symmetric, compact=compact,
                              row_cap=row_cap, workspace=workspace, return_info=return_info,
                              return_fallback=return_fallback) + ((None,) if conv_image is not None else ())
  if Np != N:
    V = V[:, :N, :].contiguous()
  out = (D, V)
  if return_info:
    out += (info,)
  if return_fallback:
    out += (fb,)
  if conv_image is not None:
    out += (img,)
  return out


def tridiag_eigh(diag, offdiag):
  """Batched symmetric tridiagonal eigensolver: diag [B,M], offdiag [B,M-1] (float64)
  -> R [B,M] ascending, Bm [B,M,M] (columns = eigenvectors), like numpy.linalg.eigh."""
  _need_cuda(diag, offdiag)
  d = diag.to(torch.float64).contiguous()
  e = offdiag.to(torch.float64).contiguous()
  B, M = d.shape
  R = torch.empty((B, M), dtype=torch.float64, device=d.device)
  Bm = torch.empty((B, M, M), dtype=torch.float64, device=d.device)
  with torch.cuda.device(d.device):
    _abi().tridiag_eigh(d, e, B, M, R, Bm)
  return R, Bm


def lanczos_ritz_large(A, M, K, workspace=None, return_info=False, symmetric=False):
  """M-step Lanczos Ritz pairs for large dense graphs (N <= 2048, rows contiguous).
  A [B,N,N] float32 -> D [B,K], V [B,N,K].  `workspace`: optional reusable uint8 CUDA tensor of
  lnz_lanczos_ritz_large_workspace_bytes(B, N) bytes.  symmetric=True reads only the upper
  256 x 256 chunk blocks of A (lnz_lanczos_ritz_large_sym: A must equal its transpose)."""
  _need_cuda(A, workspace)
  assert A.dim() == 3 and A.dtype == torch.float32 and A.stride(2) == 1
  B, N, _ = A.shape
  need = _abi().lanczos_ritz_large_workspace_bytes(B, N)
  if workspace is None or workspace.numel() * workspace.element_size() < need:
    workspace = torch.empty((need,), dtype=torch.uint8, device=A.device)
  D = torch.empty((B, K), dtype=torch.float32, device=A.device)
  V = torch.empty((B, N, K), dtype=torch.float32, device=A.device)
  info = torch.empty((B,), dtype=torch.int32, device=A.device) if return_info else None
  with torch.cuda.device(A.device):
    fn = _abi().lanczos_ritz_large_sym if symmetric else _abi().lanczos_ritz_large
    fn(A, A.stride(0), A.stride(1), B, N, M, K, workspace, D, V, info)
  return (D, V, info) if return_info else (D, V)


# --------------------------------------------------------------- R9 / R11 for graphs beyond 32 nodes
LARGE_F16_A_SCALE = 1024.0   # csrc/conv_large.hip ElemTraits<2>::kAScale


def large_plane_dtype(planes):
  """Element type of the lnz_large_* operand planes: two planes are fp16 pieces, one / three bf16."""
  return torch.float16 if planes == 2 else torch.bfloat16


def split_bf16_planes(x, planes):
  """fp32 tensor -> [planes, ...] pieces with x ~= sum of the pieces (each piece the rounding of
  the remainder).  planes = 1: the plain bf16 cast; 3: bf16 pieces; 2: fp16 pieces of
  LARGE_F16_A_SCALE * x (the A-operand convention of the two-plane mode, csrc/conv_large.hip)."""
  dt = large_plane_dtype(planes)
  r = x.to(torch.float32) * (LARGE_F16_A_SCALE if planes == 2 else 1.0)
  out = []
  for _ in range(planes):
    p = r.to(dt)
    out.append(p)
    r = r - p.to(torch.float32)
  return torch.stack(out).contiguous()


def large_weight_fragments(Wb):
  """[planes, C*128, dinp] bf16 (row-major channel blocks of the mix weight) -> [planes, C, 4,
  dinp/16, 64, 8]: v_mfma_f32_32x32x16_bf16 A-fragment order (the Wf of lnz_large_gemm1): lane l of
  fragment (c, mt, ks) holds W_c[32 mt + (l & 31)][16 ks + 8 (l >> 5) .. + 7]."""
  P, rows, dinp = Wb.shape
  Cn = rows // 128
  x = Wb.view(P, Cn, 4, 32, dinp // 16, 2, 8)        # p, c, mt, l31, ks, h, u
  return x.permute(0, 1, 2, 4, 5, 3, 6).contiguous()  # p, c, mt, ks, h, l31, u


def large_pack_operators(L, V, planes=1, chan_src=None, chan_rep=None, chan_check=None, neq=None):
  """lnz_large_pack_operators[_fold]: L [B,N,N,C] fp32 (any strides), V [B,N,K] -> Lb [planes,B,Cd,
  RT,Nk/64,4,64,8] and Vb [planes,B,RT,4,64,8] (bf16 pieces; planes = 2: fp16 pieces of 1024 x the
  entries) in fragment-tile order (RT = ceil(N/32), Nk = N rounded up to 64; see
  include/lanczosnet_hip.h).  Lb.dims = (N, Nk).
  Channel folding: chan_src = the Cd distinct source channels that are packed (ascending; default
  all C), chan_rep[c] = packed slot channel c is claimed to equal, neq = a zeroed uint64 (int64
  tensor of one element) that receives the pairwise "differs somewhere" bits 8 c + c' of the
  compared channels (chan_check[c] = 0 exempts channel c), see the header."""
  _need_cuda(L, V, neq)
  assert L.dim() == 4 and L.dtype == torch.float32 and V.dim() == 3
  V = _f32c(V)
  B, N, _, Cn = L.shape
  K = V.shape[2]
  Nk = _abi().large_nk(N)
  RT = (N + 31) // 32
  dt = large_plane_dtype(planes)
  Cd = Cn if chan_src is None else len(chan_src)
  Lb = torch.empty((planes, B, Cd, RT, Nk // 64, 4, 64, 8), dtype=dt, device=L.device)
  Vb = torch.empty((planes, B, RT, 4, 64, 8), dtype=dt, device=L.device)
  Lb.dims = (N, Nk)
  sb, sr, sc, sch = L.stride()
  i32 = lambda xs: [int(x) for x in xs] if xs is not None else None  # noqa: E731
  if neq is not None:
    assert neq.dtype == torch.int64 and neq.numel() == 1
  with torch.cuda.device(L.device):
    _abi().large_pack_operators_fold(
        L, sb, sr, sc, sch, V, B, N, Cn, K, planes, i32(chan_src), Cd,
        i32(chan_rep if chan_src is not None else None), i32(chan_check), neq, Lb,
        Vb)
  return Lb, Vb


def large_gemm1(X, din, Lb, Wf, Zt):
  """lnz_large_gemm1 on the current stream: Zt <- (X W_c^T)^T for the node-space channels."""
  _need_cuda(X, Lb, Wf, Zt)
  planes, B, Cn = Lb.shape[:3]
  N, _ = Lb.dims
  assert X.dtype == torch.float32 and X.is_contiguous() and X.shape[0] == B and X.shape[1] == N
  with torch.cuda.device(X.device):
    _abi().large_gemm1(X, X.shape[2], din, Wf, B, N, Cn, planes, Zt)


def large_spectral(X, din, Lb, V, G, Wt, Ybuf, Tt):
  """lnz_large_spectral on the current stream: Tt <- (sum_s diag(g_s) (V^T X) W_s^T)^T."""
  _need_cuda(X, Lb, V, G, Wt, Ybuf, Tt)
  planes, B = Lb.shape[:2]
  N, _ = Lb.dims
  K, S = V.shape[2], G.shape[1]
  assert V.dtype == torch.float32 and V.is_contiguous()
  assert G.is_contiguous() and G.dtype == torch.float32 and tuple(G.shape) == (B, S, K)
  with torch.cuda.device(X.device):
    _abi().large_spectral(X, X.shape[2], din, V, G, Wt, B, N, K,
                                      S, planes, Ybuf, Tt)


def large_conv(Lb, Vb, Zt, Tt, bias, relu=True, out=None):
  """lnz_large_conv on the current stream: X' [B,N,128] = act(sum_c Lb_c Zt_c^T + Vb Tt^T + bias)."""
  _need_cuda(Lb, Vb, Zt, Tt, bias)
  planes, B, Cn = Lb.shape[:3]
  N, _ = Lb.dims
  if out is None:
    out = torch.empty((B, N, 128), dtype=torch.float32, device=Lb.device)
  with torch.cuda.device(Lb.device):
    _abi().large_conv(Lb, Vb, Zt, Tt, bias, B, N, Cn,
                                  planes, int(bool(relu)), out)
  return out


def large_conv_layer(X, din, Lb, Vb, V, Wf, Wt, G, bias, work, relu=True, out=None, side=None):
  """One conv layer on packed large-graph operators: lnz_large_gemm1 + lnz_large_spectral +
  lnz_large_conv.  X [B,N,ldx] fp32 (first `din` columns are the layer input); V [B,N,K] fp32 (the
  Ritz vectors Vb was packed from); Wf = large_weight_fragments(...) of the channel blocks;
  Wt = pack_rows_k8 of the long-scale blocks [128, S*dinp] and G [B,S,K] fp32 (or None without
  long scales); work = (Zt, Tt, Ybuf) from large_work_buffers().  side: optional torch.cuda.Stream —
  the eigen-space block (which only needs X) is then issued beside gemm1 instead of after it
  (measured: no gain at config 5's size, both launches fill the chip; off by default).
  Returns X' [B,N,128]."""
  Zt, Tt, Ybuf = work
  if G is not None and side is not None:
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
      large_spectral(X, din, Lb, V, G, Wt, Ybuf, Tt)
    large_gemm1(X, din, Lb, Wf, Zt)
    main.wait_stream(side)
  else:
    large_gemm1(X, din, Lb, Wf, Zt)
    if G is not None:
      large_spectral(X, din, Lb, V, G, Wt, Ybuf, Tt)
  return large_conv(Lb, Vb, Zt, Tt, bias, relu=relu, out=out)


def midgraph_forward(X0, L, V, G, mask_u8, W, bias, Whead, bhead, num_layer):
  """lnz_midgraph_forward: every conv layer, the head and the gated masked mean of a batch of
  graphs with 33..128 nodes in ONE launch (csrc/conv_mid.hip; config/graph_lanczos_net.yaml).
  X0 [B,N,din0] fp32 (din0 a multiple of 16), L [B,N,N,C] (any strides), V [B,N,K], G
  [num_layer,B,S,K] or None, W: the layers' [128][S + C][din_l] weight blocks behind each other,
  bias [num_layer,128], Whead [dout + 1,128], bhead [dout + 1].  Returns score [B,dout]."""
  _need_cuda(X0, L, V, G, mask_u8, W, bias, Whead, bhead)
  B, N, din0 = X0.shape
  K, Cn = V.shape[2], L.shape[3]
  S = 0 if G is None else G.shape[2]
  dout = Whead.shape[0] - 1
  dev = X0.device
  Xwork = torch.empty((int(_abi().midgraph_workspace_floats(B, N, num_layer)),), dtype=torch.float32, device=dev)
  sync = torch.zeros((B * (num_layer + 1) + 16,), dtype=torch.int32, device=dev)   # (+ 16: phase stamps of profiling builds)
  midgraph_forward.last_sync = sync
  score = torch.empty((B, dout), dtype=torch.float32, device=dev)
  sb, sr, sc, sch = L.stride()
  with torch.cuda.device(dev):
    _abi().midgraph_forward(X0, L, sb, sr, sc, sch, V, G, mask_u8, W, bias, Whead, bhead, B, N, K, Cn, S,
                            num_layer, din0, dout, Xwork, sync, score)
  return score


def large_work_buffers(Lb):
  """(Zt, Tt, Ybuf) work buffers for large_conv_layer: Zt [planes,B,C,128,Nk] and Tt [planes,B,128,
  64] bf16 and Ybuf [B,64,128] fp32, zero-initialised (gemm1 never writes the k padding; Tt stays
  zero without long scales; the spectral kernels keep Ybuf zero between layers)."""
  planes, B, Cn = Lb.shape[:3]
  N, Nk = Lb.dims
  Zt = torch.zeros((planes, B, Cn, 128, Nk), dtype=Lb.dtype, device=Lb.device)
  Tt = torch.zeros((planes, B, 128, 64), dtype=Lb.dtype, device=Lb.device)
  Ybuf = torch.zeros((B, 64, 128), dtype=torch.float32, device=Lb.device)
  return Zt, Tt, Ybuf


# ---- the same layer on the nonzeros of the Laplacian (csrc/conv_sparse.hip) ----------------------
class LargeSparseImage:
  """Row-by-row nonzeros of channel 0 of L [B,N,N,C] (lnz_large_sparse_image): entries [B,N,cap]
  int32 = bf16(value) << 16 | column, counts [B,N], flags (one int32 on the device: bit 0 = a
  channel differs from channel 0, bit 1 = a row overflowed `cap`; non-zero = the image must not be
  used)."""
  __slots__ = ('entries', 'counts', 'flags', 'cap', 'B', 'N', 'version', 'values')

  def __init__(self, entries, counts, flags, cap, values=None):
    self.entries, self.counts, self.flags, self.cap = entries, counts, flags, cap
    self.values = values   # [B,N,cap] fp32: the unrounded entries (the exact-fp32 form), or None
    self.B, self.N = counts.shape
    self.version = None   # the source tensor's version counter when the image rides on it (attach_sparse_image)


def attach_sparse_image(L, img):
  """Let the image ride on the tensor it was built from: `L._lnz_sparse_image`, stamped with L's
  version counter — an in-place write to L (or any of its views) afterwards invalidates it, a copy
  of L (`.to()`, `.clone()`, a DataParallel scatter) does not carry it."""
  img.version = L._version
  L._lnz_sparse_image = img


def attached_sparse_image(L):
  """The image riding on L (attach_sparse_image), or None when there is none or L changed since."""
  img = getattr(L, '_lnz_sparse_image', None)
  if img is None or img.version != L._version or (img.B, img.N) != tuple(L.shape[:2]) or \
      img.entries.device != L.device:
    return None
  return img


def lanczos_ritz_collated(L, n_nodes, K):
  """The Ritz pairs of the collate: lanczos_ritz(L[:, :, :, 0], n_nodes, K) on the collated
  L [B,N,N,C] (dataset/graph_data.py:262-287).  Beyond RITZ_FULL_MAX_N nodes the K-step entry reads
  channel 0 in place, and where the layout allows (the channels-last pair of a single-edge-type
  collate, or one operator in contiguous rows) the SAME pass over L leaves the large-graph conv's
  image riding on L (attach_sparse_image): L is then read from HBM once per batch."""
  B, N, _, Cn = L.shape
  A = L[:, :, :, 0]
  pair = Cn == 2 and L.stride(3) == 1 and L.stride(2) == 2
  single = L.stride(2) == 1 and (Cn == 1 or L.stride(3) == 0)
  if N <= RITZ_FULL_MAX_N or L.dtype != torch.float32 or not (pair or single) or N > 2048:
    return lanczos_ritz(A, n_nodes, K)
  _warn_once('lanczos_ritz: %d > %d nodes — Ritz pairs of the K-step Lanczos recurrence (the '
             'reference\'s use_eigen_decomp=False / eigsh branch, utils/data_helper.py:205-208), not of '
             'the full decomposition; converged leading pairs agree' % (N, RITZ_FULL_MAX_N))
  D, V, img = lanczos_ritz_kstep(A, n_nodes.to(torch.int32).contiguous() if n_nodes is not None else None,
                                 K, K, conv_image=large_sparse_row_cap(N))
  if img is not None:
    attach_sparse_image(L, img)
  return D, V


def large_sparse_row_cap(N):
  """Entries kept per row of the sparse image: N / 32, at least 32, at most 256, a multiple of 8.
  The gather costs ~0.28 ms per layer at 21 entries per row (B = 256, N = 2048) against the streamed
  form's 0.47 ms whatever the density: beyond ~30 entries per row the streamed form is the faster
  one, so a graph with a row twice that long is sent there (flags bit 1)."""
  return int(min(256, max(32, (N // 32 + 7) // 8 * 8)))


def large_sparse_image(L, row_cap=None, values=False):
  """lnz_large_sparse_image on the current stream.  L [B,N,N,C] fp32 (any strides).  values: also
  the entries' unrounded fp32 values (the exact-fp32 form of the gather)."""
  _need_cuda(L)
  assert L.dim() == 4 and L.dtype == torch.float32
  B, N, _, Cn = L.shape
  cap = large_sparse_row_cap(N) if row_cap is None else int(row_cap)
  dev = L.device
  entries = torch.empty((B, N, cap), dtype=torch.int32, device=dev)
  vals = torch.empty((B, N, cap), dtype=torch.float32, device=dev) if values else None
  counts = torch.empty((B, N), dtype=torch.int32, device=dev)
  flags = torch.empty((1,), dtype=torch.int32, device=dev)
  sb, sr, sc, sch = L.stride()
  with torch.cuda.device(dev):
    _abi().large_sparse_image(L, sb, sr, sc, sch, B, N, Cn, cap, entries, vals, counts, flags)
  return LargeSparseImage(entries, counts, flags, cap, vals)


def large_pack_vectors(V, planes=1):
  """lnz_large_pack_vectors: V [B,N,K] fp32 -> Vb [planes,B,RT,4,64,8] (see large_pack_operators)."""
  _need_cuda(V)
  V = _f32c(V)
  B, N, K = V.shape
  Vb = torch.empty((planes, B, (N + 31) // 32, 4, 64, 8), dtype=large_plane_dtype(planes), device=V.device)
  with torch.cuda.device(V.device):
    _abi().large_pack_vectors(V, B, N, K, planes, Vb)
  return Vb


def large_head(X, mask_u8, Whead, bhead):
  """lnz_large_head: score [B,P] = masked mean of (W_h x + b_h) * sigmoid(w_g x + b_g) over the
  last conv state X [B,N,128]; Whead [P + 1,128] (the gate's row last), bhead [P + 1]."""
  _need_cuda(X, mask_u8, Whead, bhead)
  assert X.dtype == torch.float32 and X.is_contiguous() and X.shape[2] == 128
  B, N, _ = X.shape
  P = Whead.shape[0] - 1
  score = torch.empty((B, P), dtype=torch.float32, device=X.device)
  with torch.cuda.device(X.device):
    _abi().large_head(X, mask_u8.to(torch.uint8).contiguous(), _f32c(Whead), _f32c(bhead), B, N, P, score)
  return score


_FUSED_PROJECT_GEMM1 = os.environ.get('LNZ_FUSED_PROJECT_GEMM1', '1') != '0'


def large_sparse_work_buffers(B, N, device):
  """(Z, Tt, Ybuf) of large_sparse_conv_layer: Z [B,N,128] bf16, Tt [1,B,128,64] bf16 (zero: stays
  zero without long scales), Ybuf [B,64,128] fp32 (zero; the spectral kernels keep it zero)."""
  Z = torch.empty((B, N, 128), dtype=torch.bfloat16, device=device)
  Tt = torch.zeros((1, B, 128, 64), dtype=torch.bfloat16, device=device)
  Ybuf = torch.zeros((B, 64, 128), dtype=torch.float32, device=device)
  return Z, Tt, Ybuf


def large_sparse_conv_layer(X, din, img, Vb, V, Wf, Wt, G, bias, work, relu=True, out=None):
  """One conv layer with the node-space term on the sparse image: lnz_large_gemm1_rows +
  lnz_large_spectral + lnz_large_conv (C = 0: the lift + bias) + lnz_large_sparse_conv.  X [B,N,ldx]
  fp32 (first `din` columns are the layer input); Vb = large_pack_vectors(V); V [B,N,K] fp32; Wf: the
  one-channel, one-plane fragments of the summed weight blocks (large_weight_fragments); Wt / G as
  large_conv_layer; work from large_sparse_work_buffers().  Returns X' [B,N,128]."""
  Z, Tt, Ybuf = work
  _need_cuda(X, img.entries, Vb, Wf, bias, Z, Tt)
  B, N = img.B, img.N
  assert X.dtype == torch.float32 and X.is_contiguous() and X.shape[0] == B and X.shape[1] == N
  assert Vb.shape[0] == 1 and Wf.shape[0] == 1 and Wf.shape[1] == 1
  if out is None:
    out = torch.empty((B, N, 128), dtype=torch.float32, device=X.device)
  with torch.cuda.device(X.device):
    abi = _abi()
    if G is not None:
      K, S = V.shape[2], G.shape[1]
      assert V.dtype == torch.float32 and V.is_contiguous()
      assert G.is_contiguous() and G.dtype == torch.float32 and tuple(G.shape) == (B, S, K)
      if _FUSED_PROJECT_GEMM1:
        abi.large_spectral_gemm1_rows(X, X.shape[2], din, V, G, Wt, Wf, B, N, K, S, Ybuf, Tt, Z)
      else:
        abi.large_gemm1_rows(X, X.shape[2], din, Wf, B, N, Z)
        abi.large_spectral(X, X.shape[2], din, V, G, Wt, B, N, K, S, 1, Ybuf, Tt)
    else:
      abi.large_gemm1_rows(X, X.shape[2], din, Wf, B, N, Z)
    abi.large_conv(None, Vb, None, Tt, bias, B, N, 0, 1, 0, out)
    abi.large_sparse_conv(img.entries, img.counts, img.cap, Z, B, N, int(bool(relu)), out)
  return out


def large_sparse_conv_layer_f32(X, din, img, Vb, V, Wn, Wt, G, bias, work, planes, relu=True, out=None):
  """The split-precision modes' layer with the node-space term in EXACT fp32 on the sparse image:
  lnz_f32_linear (Zf = X Wn^T) + lnz_large_spectral + lnz_large_conv (C = 0: the lift, `planes`
  pieces) + lnz_large_sparse_conv_f32.  X [B,N,ldx] fp32 with ldx a multiple of 32 (columns >= din
  zero); Wn [128, ldx] fp32 (the class's summed weight blocks, zero padded); Vb =
  large_pack_vectors(V, planes); img with `values`; work = (Zf [B,N,128] fp32, Tt [planes,B,128,64],
  Ybuf)."""
  Zf, Tt, Ybuf = work
  _need_cuda(X, img.entries, img.values, Vb, Wn, bias, Zf, Tt)
  B, N = img.B, img.N
  ldx = X.shape[2]
  assert X.dtype == torch.float32 and X.is_contiguous() and X.shape[0] == B and X.shape[1] == N
  assert ldx % 32 == 0 and tuple(Wn.shape) == (128, ldx) and Vb.shape[0] == planes == Tt.shape[0]
  if out is None:
    out = torch.empty((B, N, 128), dtype=torch.float32, device=X.device)
  f32_linear(X.view(B * N, ldx), Wn, out=Zf.view(B * N, 128))
  with torch.cuda.device(X.device):
    abi = _abi()
    if G is not None:
      K, S = V.shape[2], G.shape[1]
      assert V.dtype == torch.float32 and V.is_contiguous()
      assert G.is_contiguous() and G.dtype == torch.float32 and tuple(G.shape) == (B, S, K)
      abi.large_spectral(X, ldx, din, V, G, Wt, B, N, K, S, planes, Ybuf, Tt)
    abi.large_conv(None, Vb, None, Tt, bias, B, N, 0, planes, 0, out)
    abi.large_sparse_conv_f32(img.entries, img.values, img.counts, img.cap, Zf, B, N, int(bool(relu)), out)
  return out


# ------------------------------------------------------------------------------------- packing
def pack_rows_k8(W):
  """[rows, cols] -> MFMA fragment order (see include/lanczosnet_hip.h)."""
  _need_cuda(W)
  W = _f32c(W)
  rows, cols = W.shape
  out = torch.empty((_abi().packed_rows_k8_size(rows, cols),), dtype=torch.float32,
                    device=W.device)
  with torch.cuda.device(W.device):
    _abi().pack_rows_k8(W, rows, cols, cols, out)
  return out


def pack_rows_k8_split(W):
  """[rows, cols] (cols a multiple of 32) -> the weight stream of the split-precision strip kernel
  (gemm_mode 1): fp16 hi / lo pieces at pack_rows_k8's size and offsets (include/lanczosnet_hip.h)."""
  _need_cuda(W)
  W = _f32c(W)
  rows, cols = W.shape
  out = torch.empty((_abi().packed_rows_k8_size(rows, cols),), dtype=torch.float32,
                    device=W.device)
  with torch.cuda.device(W.device):
    _abi().pack_rows_k8_split(W, rows, cols, cols, out)
  return out


def pack_bias_rows(bias):
  _need_cuda(bias)
  bias = _f32c(bias)
  rows = bias.shape[0]
  out = torch.empty((((rows + 31) // 32) * 1024,), dtype=torch.float32, device=bias.device)
  with torch.cuda.device(bias.device):
    _abi().pack_bias_rows(bias, rows, out)
  return out


def pack_laplacian(L):
  """L [B,N,N,C] float32 (any strides) -> Lp [B,C,4,64,4] fragment order, N <= 32."""
  _need_cuda(L)
  assert L.dim() == 4 and L.dtype == torch.float32
  B, N, _, Cn = L.shape
  Lp = torch.empty((B, Cn, 4, 64, 4), dtype=torch.float32, device=L.device)
  # identity-channel bits ride along with the pack (read by lanczosnet_forward)
  Lp.ident = torch.empty((B,), dtype=torch.int32, device=L.device)
  sb, sr, sc, sch = L.stride()
  with torch.cuda.device(L.device):
    _abi().pack_laplacian_ident(L, sb, sr, sc, sch, B, N, Cn, Lp,
                                            Lp.ident)
  return Lp


def pack_laplacian_for(plan, L):
  """The Laplacian pack the fused forward expects for `plan`: fp32 fragments, or their fp16 hi | lo
  pieces as a float16 tensor for a split-precision plan (gemm_mode 1)."""
  Lf = L if L.dtype == torch.float32 else L.float()
  Lp = pack_laplacian(Lf)
  return split_laplacian_pack(Lp) if plan.get('gemm_mode', 0) == 1 else Lp


def pack_and_plan(plan, L, mask_u8, K, n_cu=None):
  """Everything the fused forward needs besides (V, G), in the fewest launches: the Laplacian pack
  for `plan`, the tile plan and the live-eigen-slot list: the single lnz_pack_laplacian_plan launch
  (the planner runs under the packing) while a molecule's block fits the pack workgroup's LDS.
  Returns (Lp, tiles, rows) for lanczosnet_forward / spectral_gains."""
  Lf = L if L.dtype == torch.float32 else L.float()
  if Lf.shape[1] * Lf.shape[1] * Lf.shape[3] * 4 > 48 * 1024:
    tiles, rows = plan_batch(mask_u8, pairing_supported(plan), K, n_cu)
    return pack_laplacian_for(plan, Lf), tiles, rows
  _need_cuda(Lf, mask_u8)
  B, N, _, Cn = Lf.shape
  n_cu = n_cu or _n_cu(Lf.device)
  cap = _abi().plan_wg_cap(B, n_cu)
  Lp = torch.empty((B, Cn, 4, 64, 4), dtype=torch.float32, device=Lf.device)
  Lp.ident = torch.empty((B,), dtype=torch.int32, device=Lf.device)
  buf = torch.empty((12 * cap + 2 + B * K,), dtype=torch.int32, device=Lf.device)
  n_wg, n_rows, rows = buf[12 * cap:12 * cap + 1], buf[12 * cap + 1:12 * cap + 2], buf[12 * cap + 2:]
  strips, ns = None, None
  if strip_plan_wanted(B, N):
    strips, scap = _strip_buf(B, Lf.device)
    ns = strips[scap * STRIP_INTS:]
    buf.strips = strips
  sb, sr, sc, sch = Lf.stride()
  with torch.cuda.device(Lf.device):
    _abi().pack_laplacian_plan(
        Lf, sb, sr, sc, sch, B, N, Cn, Lp, mask_u8, n_cu,
        int(pairing_supported(plan)), buf, n_wg, K, rows, n_rows,
        Lp.ident, strips, ns)
  return Lp, (buf, cap), (rows, n_rows)


def prepare_batch(plan, L, mask_u8, n_nodes, K, n_cu=None, pack_stream=None):
  """lnz_prepare_batch: Laplacian pack, batch plan and the Ritz pairs of L[..., 0] in one launch
  (exact-fp32 plans, N <= 32).  Returns (Lp, tiles, rows, D, V).
  pack_stream: a second stream.  The launch then carries the plan and the Ritz pairs only, and the
  pack is enqueued on `pack_stream` BEHIND it — it runs under whatever the caller launches next
  on the current stream (the spectral gains: matrix-pipe work, the memory path is idle) instead of
  next to the latency-bound Lanczos wavefronts.  `Lp.ready` is the event the pack's consumer has
  to wait for; lanczosnet_forward does (on its own stream)."""
  Lf = L if L.dtype == torch.float32 else L.float()
  B, N, _, Cn = Lf.shape
  # The single launch pays when the Ritz wavefronts are latency bound (about one per SIMD); with
  # several waves per SIMD they are throughput bound and the 64-thread standalone kernel, which
  # does not carry three idle waves and a pack tile per workgroup, is faster (B = 8192: 0.55 vs
  # 0.80 ms).
  if N > 32 or N * N * Cn * 4 > 40 * 1024 or \
      B > 8 * (n_cu or _n_cu(Lf.device)):
    Lp, tiles, rows = pack_and_plan(plan, Lf, mask_u8, K, n_cu)
    D, V = lanczos_ritz(Lf[:, :, :, 0], n_nodes, K)
    return Lp, tiles, rows, D, V
  _need_cuda(Lf, mask_u8, n_nodes)
  n_cu = n_cu or _n_cu(Lf.device)
  nn = n_nodes if n_nodes.dtype == torch.int32 and n_nodes.is_contiguous() else \
      n_nodes.to(torch.int32).contiguous()
  if pack_stream is not None and not torch.cuda.is_current_stream_capturing():
    buf, D, V = _ext().plan_ritz(Lf, mask_u8, nn, K, n_cu, bool(pairing_supported(plan)))
    # the pack's buffers come from the CURRENT stream's pool (ordered behind the last launch that
    # read their previous contents); the side stream starts behind the launch above
    Lp = torch.empty((B, Cn, 4, 64, 4), dtype=torch.float32, device=Lf.device)
    Lp.ident = torch.empty((B,), dtype=torch.int32, device=Lf.device)
    pack_stream.wait_stream(torch.cuda.current_stream(Lf.device))
    sb, sr, sc, sch = Lf.stride()
    with torch.cuda.stream(pack_stream):
      _abi().pack_laplacian_ident(Lf, sb, sr, sc, sch, B, N, Cn, Lp, Lp.ident)
      Lp.ready = torch.cuda.Event()
      Lp.ready.record(pack_stream)
    for t_ in (Lp, Lp.ident, Lf):
      t_.record_stream(pack_stream)
  else:
    Lp, ident, buf, D, V = _ext().prepare_batch(Lf, mask_u8, nn, K, n_cu,
                                                bool(pairing_supported(plan)))
    Lp.ident = ident
  cap = _abi().plan_wg_cap(B, n_cu)
  soff = 12 * cap + 2 + B * K
  if buf.numel() > soff:  # the strip plan rides behind the live-slot list
    buf.strips = buf[soff:]
  return Lp, (buf, cap), (buf[12 * cap + 2:soff], buf[12 * cap + 1:12 * cap + 2]), D, V


def prepare_batch_prev_gains(plan, L, mask_u8, n_nodes, K, prev, gains, n_cu=None):
  """lnz_prepare_batch_prev_gains: the preparation of THIS batch and the spectral gains of the
  PREVIOUS one in a single launch (software pipeline over a stream of batches).
  prev = (D_prev, rows_prev) from the previous prepare_batch*/ call; gains = (dist, num_layer,
  mlp_pack).  Returns (Lp, tiles, rows, D, V, G_prev)."""
  Lf = L if L.dtype == torch.float32 else L.float()
  B, N, _, Cn = Lf.shape
  _need_cuda(Lf, mask_u8, n_nodes, prev[0])
  assert N <= 32 and N * N * Cn * 4 <= 20480
  n_cu = n_cu or _n_cu(Lf.device)
  cap = _abi().plan_wg_cap(B, n_cu)
  dev = Lf.device
  Lp = torch.empty((B, Cn, 4, 64, 4), dtype=torch.float32, device=dev)
  Lp.ident = torch.empty((B,), dtype=torch.int32, device=dev)
  buf = torch.empty((12 * cap + 2 + B * K,), dtype=torch.int32, device=dev)
  n_wg, n_rows, rows = buf[12 * cap:12 * cap + 1], buf[12 * cap + 1:12 * cap + 2], buf[12 * cap + 2:]
  D = torch.empty((B, K), dtype=torch.float32, device=dev)
  V = torch.empty((B, N, K), dtype=torch.float32, device=dev)
  nn = n_nodes.to(torch.int32).contiguous()
  D_prev, (rows_prev, n_rows_prev) = prev
  Bp = D_prev.shape[0]
  dist, num_layer, mlp_pack = gains
  S = len(dist)
  Gbuf = torch.empty((num_layer * Bp * S * K + 16,), dtype=torch.float32, device=dev)
  G = Gbuf[:num_layer * Bp * S * K].view(num_layer, Bp, S, K)
  darr = [int(x) for x in dist]
  strips, ns = None, None
  if strip_plan_wanted(B, N):
    strips, scap = _strip_buf(B, dev)
    ns = strips[scap * STRIP_INTS:]
    buf.strips = strips
  sb, sr, sc, sch = Lf.stride()
  with torch.cuda.device(dev):
    _abi().prepare_batch_prev_gains(
        Lf, sb, sr, sc, sch, B, N, Cn, Lp, mask_u8, nn, n_cu,
        int(pairing_supported(plan)), buf, n_wg, K, rows, n_rows,
        D, V, Lp.ident, D_prev, Bp, rows_prev, n_rows_prev,
        darr, S, num_layer, mlp_pack, G, strips, ns)
  return Lp, (buf, cap), (rows, n_rows), D, V, G


def pack_spectral_mlp(linears, S, out=None):
  """linears: 4 (weight, bias) pairs of one `spectral_filter[l]` Sequential -> packed buffer."""
  size = _abi().spectral_mlp_pack_size(S)
  ws = []
  for (w, b) in linears:
    _need_cuda(w, b)
    ws += [_f32c(w), _f32c(b)]
  if out is None:
    out = torch.empty((size,), dtype=torch.float32, device=ws[0].device)
  with torch.cuda.device(ws[0].device):
    _abi().pack_spectral_mlp(*[t for t in ws], S, out)
  return out


def pack_spectral_mlp_layers(layers, S):
  """layers: per conv layer, the 4 (weight, bias) pairs of its `spectral_filter[l]` -> one
  [num_layer, pack_size] buffer, packed by ONE launch (lnz_pack_spectral_mlp_layers)."""
  size = _abi().spectral_mlp_pack_size(S)
  keep, ptrs = [], []
  for lins in layers:
    assert len(lins) == 4
    for (w, b) in lins:
      _need_cuda(w, b)
      keep += [_f32c(w), _f32c(b)]
  dev = keep[0].device
  out = torch.empty((len(layers), size), dtype=torch.float32, device=dev)
  with torch.cuda.device(dev):
    _abi().pack_spectral_mlp_layers(keep, len(layers), S, out)
  return out


# ------------------------------------------------------------------------------------------ R7
def spectral_gains(D, dist, num_layer, mlp_pack=None, rows=None, zero_fill=True, split_pack=None):
  """D [B,K] -> G [num_layer,B,S,K].  mlp_pack=None selects the plain-power branch.
  rows: optional (gain_rows, n_gain_rows) int32 device tensors from plan_batch(): the MLP runs
  only on the eigen slots that carry a Ritz pair, every other entry of G is zero — or left
  uninitialised with zero_fill=False, which is what the exact-fp32 forward kernel needs (it never
  reads the slots k >= n)).  split_pack: the batch's packed Laplacian (fp32) of a gemm_mode-1 plan —
  the call then returns (G, Lh), Lh the pack in the split-precision forward's form: a NEW float16
  tensor written by workgroups that ride along with the MLP launch (split_laplacian_pack() as part
  of this launch; the fp32 pack is left as it was)."""
  _need_cuda(D, mlp_pack, split_pack)
  D = _f32c(D)
  use_rows = rows is not None and mlp_pack is not None
  ride = dst = None
  if split_pack is not None:
    if split_pack.dtype == torch.float16 or mlp_pack is None:
      dst = split_laplacian_pack(split_pack)   # (already converted, or no MLP launch to ride along with)
    else:
      assert split_pack.dtype == torch.float32 and split_pack.is_contiguous()
      ready = getattr(split_pack, 'ready', None)   # a pack on a second stream
      if ready is not None:
        torch.cuda.current_stream(split_pack.device).wait_event(ready)
      ride, dst = split_pack, _split_pack_like(split_pack)
  G = _ext().spectral_gains(D, [int(x) for x in dist], num_layer, mlp_pack,
                            rows[0] if use_rows else None, rows[1] if use_rows else None,
                            bool(zero_fill), ride, dst if ride is not None else None)
  return G if split_pack is None else (G, dst)


def _split_pack_like(Lp):
  """The destination of a pack's split-precision form: float16, the same bytes ([..., 2 x last]); the
  identity-channel bits of the pack go along."""
  Lh = torch.empty(tuple(Lp.shape[:-1]) + (2 * Lp.shape[-1],), dtype=torch.float16, device=Lp.device)
  ident = getattr(Lp, 'ident', None)
  if ident is not None:
    Lh.ident = ident
  return Lh


def split_laplacian_pack(Lp):
  """lnz_split_laplacian_pack_to: the fp32 Laplacian pack -> its split-precision form (every fragment
  float4 = 4 fp16 hi pieces | 4 lo pieces) as a NEW float16 tensor; the fp32 pack is not touched.  The
  element type IS the format: the exact kernels take float32 packs only, the gemm_mode-1 forward
  float16 ones — a clone, view or detach keeps it.  A float16 pack is returned as it is."""
  _need_cuda(Lp)
  if Lp.dtype == torch.float16:
    return Lp
  assert Lp.dtype == torch.float32 and Lp.is_contiguous()
  ready = getattr(Lp, 'ready', None)
  if ready is not None:
    torch.cuda.current_stream(Lp.device).wait_event(ready)
  Lh = _split_pack_like(Lp)
  with torch.cuda.device(Lp.device):
    _abi().split_laplacian_pack_to(Lp, Lp.numel(), Lh)
  return Lh


def spectral_mlp_grad(D, dist, layers, dG, rows=None, rows_max=None):
  """lnz_spectral_mlp_grad: parameter gradients of the spectral-filter MLPs of every conv layer from
  dG [L, B*K, S] in one launch.  layers: per conv layer the 4 (weight, bias) pairs of its
  `spectral_filter[l]` (raw parameters).  rows: (gain_rows, n_gain_rows) of plan_batch() or None;
  rows_max: host-side upper bound of the live row count (sizes the grid; default B*K).  Returns
  [(dW0, db0), (dW2, db2), (dW4, db4), (dW6, db6)], each stacked over the layers ([L, ...])."""
  _need_cuda(D, dG)
  D = _f32c(D)
  B, K = D.shape
  S, L = len(dist), len(layers)
  assert dG.is_contiguous() and dG.dtype == torch.float32 and tuple(dG.shape) == (L, B * K, S)
  keep = []
  for lins in layers:
    assert [tuple(w.shape) for (w, _) in lins] == [(128, S), (128, 128), (128, 128), (S, 128)], \
        'lnz_spectral_mlp_grad is built for the reference\'s S-128-128-128-S filter MLP'
    for (w, b) in lins:
      keep += [_f32c(w.detach()), _f32c(b.detach())]
  dev = D.device
  parts = _abi().spectral_mlp_grad_parts(int(rows_max or B * K), L, _n_cu(dev))
  T = _abi().spectral_mlp_grad_floats(S)
  partials = torch.empty((L, parts, T), dtype=torch.float32, device=dev)
  with torch.cuda.device(dev):
    _abi().spectral_mlp_grad(D, B, K, [int(x) for x in dist], S, L,
                             rows[0] if rows is not None else None,
                             rows[1] if rows is not None else None, dG, keep, parts, partials)
  g = partials.sum(dim=1)   # [L, T]: the partials, added in a fixed order — one reduction for all eight
  o2, o4, o6, ob = 128 * S, 128 * S + 16384, 128 * S + 32768, 128 * S + 32768 + S * 128
  return [(g[:, :o2].view(L, 128, S), g[:, ob:ob + 128]),
          (g[:, o2:o4].view(L, 128, 128), g[:, ob + 128:ob + 256]),
          (g[:, o4:o6].view(L, 128, 128), g[:, ob + 256:ob + 384]),
          (g[:, o6:ob].view(L, S, 128), g[:, ob + 384:ob + 384 + S])]


def node_extents_block(mask_u8):
  """lnz_node_extents: mask [B,N] uint8 -> ONE int64 tensor [2 B + 1]: extents | row offsets (their
  exclusive prefix sums) | total.  One launch."""
  _need_cuda(mask_u8)
  B, N = mask_u8.shape
  out = torch.empty((2 * B + 1,), dtype=torch.int64, device=mask_u8.device)
  with torch.cuda.device(mask_u8.device):
    _abi().node_extents(mask_u8.contiguous(), B, N, out[:B], out[B:2 * B], out[2 * B:])
  return out


def node_extents(mask_u8):
  """(extent [B], row_off [B], total [1]) views of node_extents_block."""
  B = mask_u8.shape[0]
  out = node_extents_block(mask_u8)
  return out[:B], out[B:2 * B], out[2 * B:]


def head_backward(X_last, mask_u8, grad_score, Whead, bhead, N, dY, row_off=None, dY_compact=None, n_wg=256,
                  Wgate=None, bgate=None):
  """lnz_head_backward: the readout head's backward in one launch (+ a tiny fixed-order reduction).
  X_last [B,32,128] (last conv state), mask [B,N] uint8, grad_score [B,P], Whead [P+1,128] / bhead
  [P+1] (output rows, then the gate row); dY [B,32,128] is written in place (and dY_compact [R,128]
  at row_off[b] + r for the rows below the node extent).  Wgate / bgate: the gate row on its own
  (Whead then holds the P output rows only).  Returns dWhead [P+1,128], dbhead [P+1],
  dbias_last [128] (column sums of dY)."""
  _need_cuda(X_last, mask_u8, grad_score, Whead, bhead, dY, row_off, dY_compact)
  B, P = grad_score.shape
  assert X_last.shape == (B, 32, 128) and X_last.is_contiguous() and dY.shape == (B, 32, 128) and dY.is_contiguous()
  assert Whead.shape == (P + (Wgate is None), 128) and Whead.is_contiguous() and mask_u8.dtype == torch.uint8
  assert Wgate is None or (Wgate.shape == (1, 128) and Wgate.is_contiguous())
  dev = X_last.device
  ws = torch.empty((int(_abi().head_backward_workspace_floats(P, n_wg)),), dtype=torch.float32, device=dev)
  dW = torch.empty((P + 1, 128), dtype=torch.float32, device=dev)
  db = torch.empty((P + 1,), dtype=torch.float32, device=dev)
  dbl = torch.empty((128,), dtype=torch.float32, device=dev)
  with torch.cuda.device(dev):
    _abi().head_backward(X_last, mask_u8.contiguous(), grad_score.float().contiguous(), Whead, bhead.contiguous(),
                         Wgate, bgate, row_off, B, N, P, 128, n_wg, ws, dY, dY_compact, dW, db, dbl)
  return dW, db, dbl


def embedding_grad(ids, dx, width, num_atom, chunks=64):
  """lnz_embedding_grad: dE [num_atom, width] = sum of the rows dx[b, i, :width] by atom id ids[b, i]
  (ids [B, N] int64; dx [B, >= N, >= width] fp32 with a contiguous last dimension) — the embedding
  table's gradient without atomics (partials per row chunk, added in a fixed order)."""
  _need_cuda(ids, dx)
  B, N = ids.shape
  assert ids.dtype == torch.int64 and ids.is_contiguous() and dx.dtype == torch.float32
  assert dx.dim() == 3 and dx.shape[0] == B and dx.shape[1] >= N and dx.shape[2] >= width and dx.stride(2) == 1
  part = torch.empty((chunks, num_atom, width), dtype=torch.float32, device=dx.device)
  with torch.cuda.device(dx.device):
    _abi().embedding_grad(ids, B, N, dx, dx.stride(0), dx.stride(1), width, num_atom, chunks, part)
  return part.sum(dim=0)


def collate_qm8(shard, ids, N, E, P):
  """lnz_collate_qm8: device arrays of a packed shard (dataset/packed.py) + molecule ids [B] ->
  padded batch dict (node_feat, node_mask, label, L [B,N,N,E+1], n_nodes)."""
  _need_cuda(shard['mol_off'], shard['edge_off'], shard['atoms'], shard['edges'], shard['labels'],
             ids)
  assert ids.dtype == torch.int64 and ids.is_contiguous()
  dev = ids.device
  B = ids.numel()
  n_mol = shard['mol_off'].numel() - 1
  node_feat = torch.empty((B, N), dtype=torch.int64, device=dev)
  mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
  label = torch.empty((B, P), dtype=torch.float32, device=dev)
  L = torch.empty((B, N, N, E + 1), dtype=torch.float32, device=dev)
  n_nodes = torch.empty((B,), dtype=torch.int32, device=dev)
  _abi().collate_qm8(shard['mol_off'], shard['edge_off'],
                                 shard['atoms'], shard['edges'], shard['labels'],
                                 ids, n_mol, B, N, E, P, node_feat, mask,
                                 label, L, n_nodes)
  return dict(node_feat=node_feat, node_mask=mask, label=label, L=L, n_nodes=n_nodes)


_N_CU = {}


def _n_cu(device):
  idx = device.index if device.index is not None else torch.cuda.current_device()
  if idx not in _N_CU:
    _N_CU[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
  return _N_CU[idx]


def plan_batch(mask_u8, allow_pairs, K, n_cu=None):
  """lnz_plan_batch: the tile plan of plan_tiles() plus the list of eigen slots that carry a Ritz
  pair.  Returns ((buf, cap), (gain_rows, n_gain_rows)); pass the first to lanczosnet_forward
  (tiling=) and the second to spectral_gains (rows=)."""
  B, N = mask_u8.shape
  n_cu = n_cu or _n_cu(mask_u8.device)
  cap = _abi().plan_wg_cap(B, n_cu)
  buf = torch.empty((12 * cap + 2 + B * K,), dtype=torch.int32, device=mask_u8.device)
  n_wg, n_rows, rows = buf[12 * cap:12 * cap + 1], buf[12 * cap + 1:12 * cap + 2], buf[12 * cap + 2:]
  strips, ns = None, None
  if strip_plan_wanted(B, N):
    strips, scap = _strip_buf(B, mask_u8.device)
    ns = strips[scap * STRIP_INTS:]
    buf.strips = strips
  _abi().plan_batch(mask_u8, B, N, n_cu, int(bool(allow_pairs)), buf,
                                n_wg, K, rows, n_rows, strips, ns)
  return (buf, cap), (rows, n_rows)


def plan_tiles(mask_u8, allow_pairs, n_cu=None):
  """lnz_plan_tiles: small molecules share a 32-row tile; tiles are dealt evenly over one workgroup
  per CU.  Returns (buf, cap): int32 tensor [12*cap + 1] = cap workgroup entries of 4 slots x
  (molecule A, molecule B | -1, split row), followed by the number of workgroups in use."""
  B, N = mask_u8.shape
  n_cu = n_cu or _n_cu(mask_u8.device)
  cap = _abi().plan_wg_cap(B, n_cu)
  buf = torch.empty((12 * cap + 1,), dtype=torch.int32, device=mask_u8.device)
  if allow_pairs and strip_plan_wanted(B, N):
    # (a plan that may share tiles also carries the strip plan; 'single' plans keep one molecule
    # per tile for the kernels and tests that ask for exactly that)
    strips, scap = _strip_buf(B, mask_u8.device)
    buf.strips = strips
    _abi().plan_batch(mask_u8, B, N, n_cu, 1, buf, buf[12 * cap:], 0, None, None, strips,
                      strips[scap * STRIP_INTS:])
    return buf, cap
  _abi().plan_tiles(mask_u8, B, N, n_cu, int(bool(allow_pairs)), buf,
                                buf[12 * cap:])
  return buf, cap


def pairing_supported(plan):
  """Pair tiles exist in every forward kernel there is (the spectral channels run in eigen space:
  diagonal gains of LanczosNet, dense K x K filters of AdaLanczosNet); kept for the callers that ask."""
  return True


def lanczosnet_forward(plan, node_feat, Lp, V, G, mask, return_state=False, tiling='auto',
                       act_out=None, use_ident=True):
  """Launch the fused forward.  `plan` is a dict made by LanczosNet._plan() holding the packed
  parameters and the static sizes.  tiling: 'auto' = lnz_plan_tiles with pairing where the kernel
  supports it, 'single' = planned but one molecule per tile, 'none' = no plan (batch order), or
  the (buf, cap) pair returned by plan_tiles() for this mask (pairs only for the exact kernel).
  act_out: optional zero-initialised [num_layer,B,32,dhid] that receives every layer's activations
  (training forward).  use_ident=False ignores the identity-channel bits of the pack (every
  channel goes through its Laplacian fragments)."""
  _need_cuda(node_feat, Lp, V, G, mask)
  B, N, K = V.shape
  ready = getattr(Lp, 'ready', None)   # a pack on a second stream (prepare_batch(pack_stream=...))
  if ready is not None:
    torch.cuda.current_stream(Lp.device).wait_event(ready)
  if plan.get('gemm_mode', 0) == 1:
    Lp = split_laplacian_pack(Lp)   # (a float16 pack — spectral_gains(split_pack=...) made it on the way — passes)
  elif Lp.dtype != torch.float32:
    raise RuntimeError('this Laplacian pack is the float16 form of a split-precision plan (gemm_mode 1): '
                       'the exact kernels read the fp32 pack')
  if act_out is None and not return_state:
    return _forward_ext(plan, node_feat, Lp, V, G, mask, tiling, use_ident)
  ops_, dims = _fused_operands(plan, V)
  emb = None
  if node_feat.dtype in (torch.int64,):
    ops_[_IN['node_feat']] = node_feat.contiguous()
    emb = ops_[_IN['embedding']] = plan['embedding']
    dims[_DIM['num_atom']] = emb.shape[0]
  else:
    nf = node_feat.to(torch.float32)
    if nf.shape[-1] != plan['din0']:  # zero-pad feature columns to the kernel's 32-column groups
      assert nf.shape[-1] == plan['din0_raw']
      nf = torch.nn.functional.pad(nf, (0, plan['din0'] - nf.shape[-1]))
    ops_[_IN['node_feat_f']] = nf.contiguous()
  mask_u8 = mask.to(torch.uint8).contiguous()
  ops_[_IN['mask']] = mask_u8
  ops_[_IN['Lp']] = Lp
  ident = getattr(Lp, 'ident', None)  # identity-channel bits written by the pack kernels
  if ident is not None and use_ident:
    ops_[_IN['ident']] = ident
  fk = int(plan.get('filter_kind', 0))
  if G is not None:
    want = (plan['num_layer'], B, plan['n_long'], K) + ((K,) if fk == 1 else ())
    assert tuple(G.shape) == want and G.is_contiguous() and G.dtype == torch.float32, \
        (tuple(G.shape), want)
  ops_[_IN['G']] = G
  ops_[_IN['Wp']], ops_[_IN['bias']] = plan['Wp'], plan['bias']
  ops_[_IN['Wp_head']], ops_[_IN['bias_head']] = plan['Wp_head'], plan['bias_head']
  gemm_mode = int(plan.get('gemm_mode', 0))
  dims[_DIM['gemm_mode']] = gemm_mode
  # Tile plan: small molecules share a 32-row tile and the tiles are dealt, balanced by cost, over
  # one workgroup per CU (the launch is a single round: it lasts as long as its busiest CU).
  if isinstance(tiling, tuple):
    tiles, cap = tiling
  elif tiling != 'none':
    assert tiling in ('auto', 'single')
    tiles, cap = plan_tiles(mask_u8, allow_pairs=(tiling == 'auto' and pairing_supported(plan)))
  if tiling != 'none':
    _set_plan(ops_, dims, tiles, cap)
  if gemm_mode == 1 and ops_[_IN['strips']] is None:   # this mode exists on the strip plan only
    strips = plan_strips(mask_u8)
    scap = (strips.numel() - 1) // STRIP_INTS
    ops_[_IN['strips']], ops_[_IN['n_strips']] = strips, strips[scap * STRIP_INTS:]
    dims[_DIM['strip_cap']] = scap
  score = torch.empty((B, plan['dout']), dtype=torch.float32, device=V.device)
  if act_out is not None:
    assert tuple(act_out.shape) == (plan['num_layer'], B, 32, plan['dhid']) and \
        act_out.is_contiguous() and act_out.dtype == torch.float32
  state = None
  if return_state:
    state = torch.zeros((B, 32, plan['dhid']), dtype=torch.float32, device=V.device)
  nl = plan['num_layer']
  _ext().fused_launch(0, ops_, dims, [int(x) for x in plan['w_off'][:nl]],
                      [int(x) for x in plan['b_off'][:nl]], [int(p) for p in plan['short']],
                      score, state, act_out, None, None, None, None, None, None)
  return (score, state) if return_state else score


# operand / scalar slots of torch.ops.lanczosnet.fused_launch (csrc/torch_ext.cpp: kIn / kDim)
_IN = {k: i for i, k in enumerate(
    ['node_feat', 'node_feat_f', 'embedding', 'mask', 'Lp', 'V', 'G', 'Wp', 'bias', 'Wp_head',
     'bias_head', 'plan', 'n_wg', 'act', 'x0', 'ident', 'row_off',
     'strips', 'n_strips'])}
_DIM = {k: i for i, k in enumerate(
    ['B', 'N', 'K', 'num_layer', 'din0', 'dhid', 'dout', 'n_long', 'n_edge', 'num_atom', 'filter_kind',
     'gemm_mode', 'plan_cap', 'bwd_din0', 'msg_layer', 'dy_compact_rows', 'strip_cap'])}


def _fused_operands(plan, V):
  """Operand list and scalar fields common to the four argument-block launches."""
  B, N, K = V.shape
  ops_ = [None] * len(_IN)
  ops_[_IN['V']] = _f32c(V)
  dims = [0] * len(_DIM)
  for k, v in (('B', B), ('N', N), ('K', K), ('num_layer', plan['num_layer']), ('din0', plan['din0']),
               ('dhid', plan['dhid']), ('dout', plan['dout']), ('n_long', plan['n_long']),
               ('n_edge', plan['n_edge']), ('filter_kind', int(plan.get('filter_kind', 0)))):
    dims[_DIM[k]] = int(v)
  return ops_, dims


def _set_plan(ops_, dims, tiles, cap):
  ops_[_IN['plan']] = tiles
  ops_[_IN['n_wg']] = tiles[12 * cap:]
  dims[_DIM['plan_cap']] = int(cap)
  strips = getattr(tiles, 'strips', None)  # strip plan made with the tile plan (_attach_strips)
  if strips is not None:
    scap = (strips.numel() - 1) // STRIP_INTS
    ops_[_IN['strips']], ops_[_IN['n_strips']] = strips, strips[scap * STRIP_INTS:]
    dims[_DIM['strip_cap']] = scap


STRIP_INTS = 80   # LNZ_STRIP_INTS


def strip_plan_wanted(B, N):
  """The strip plan (lnz_plan_strips) is made next to the tile plan for every batch it takes."""
  return N <= 32


def _strip_buf(B, device):
  scap = _abi().strip_cap(B)
  return torch.empty((scap * STRIP_INTS + 1,), dtype=torch.int32, device=device), scap


def plan_strips(mask_u8, n_cu=None):
  """lnz_plan_strips: molecules packed at 4-row granularity into strips of 16-row subtiles, one
  workgroup of the 16 x 16-tile inference forward each.  Returns the int32 tensor
  [scap * 80 + 1] = scap entries (molecules, subtiles, then (molecule, first row, extent) triples)
  followed by the number of strips in use."""
  B, N = mask_u8.shape
  n_cu = n_cu or _n_cu(mask_u8.device)
  buf, scap = _strip_buf(B, mask_u8.device)
  with torch.cuda.device(mask_u8.device):
    _abi().plan_strips(mask_u8, B, N, n_cu, buf, buf[scap * STRIP_INTS:])
  return buf


def _forward_ext(plan, node_feat, Lp, V, G, mask, tiling, use_ident):
  """lanczosnet_forward through torch.ops.lanczosnet.forward (exact-fp32 kernel, scores only)."""
  B, N, K = V.shape
  mask_u8 = mask if mask.dtype == torch.uint8 and mask.is_contiguous() else \
      mask.to(torch.uint8).contiguous()
  emb = None
  if node_feat.dtype == torch.int64:
    nf, emb = node_feat.contiguous(), plan['embedding']
  else:
    nf = node_feat.to(torch.float32)
    if nf.shape[-1] != plan['din0']:  # zero-pad feature columns to the kernel's 32-column groups
      assert nf.shape[-1] == plan['din0_raw']
      nf = torch.nn.functional.pad(nf, (0, plan['din0'] - nf.shape[-1]))
    nf = nf.contiguous()
  fk = int(plan.get('filter_kind', 0))
  if G is not None:
    want = (plan['num_layer'], B, plan['n_long'], K) + ((K,) if fk == 1 else ())
    assert tuple(G.shape) == want and G.is_contiguous() and G.dtype == torch.float32, \
        (tuple(G.shape), want)
  tiles, cap = None, 0
  if isinstance(tiling, tuple):
    tiles, cap = tiling
  elif tiling != 'none':
    assert tiling in ('auto', 'single')
    tiles, cap = plan_tiles(mask_u8, allow_pairs=(tiling == 'auto' and pairing_supported(plan)))
  consts = plan.get('_ext_consts')
  if consts is None:   # static per plan: built once
    consts = plan['_ext_consts'] = (
        [int(x) for x in plan['w_off'][:plan['num_layer']]],
        [int(x) for x in plan['b_off'][:plan['num_layer']]],
        [plan['num_layer'], plan['din0'], plan['dhid'], plan['dout'], plan['n_long'], plan['n_edge'], fk,
         int(plan.get('gemm_mode', 0))],
        [int(p) for p in plan['short']])
  w_off, b_off, dims, short = consts
  ident = getattr(Lp, 'ident', None) if use_ident else None
  strips = getattr(tiles, 'strips', None)
  if strips is None and plan.get('gemm_mode', 0) == 1:
    strips = plan_strips(mask_u8)   # the split-precision GEMM1 exists on the strip plan only
  scap = (strips.numel() - 1) // STRIP_INTS if strips is not None else 0
  return _ext().forward(nf, emb, Lp, ident, _f32c(V), G, mask_u8, plan['Wp'], plan['bias'], w_off,
                        b_off, plan['Wp_head'], plan['bias_head'], tiles, cap, dims, short,
                        strips, scap)


def _training_args(plan, Lp, V, G, mask_u8, tiling):
  """Common part of the training launches: sizes, operators, gains, tile plan."""
  ops_, dims = _fused_operands(plan, V)
  ops_[_IN['mask']], ops_[_IN['Lp']], ops_[_IN['G']] = mask_u8, Lp, G
  tiles, cap = tiling
  _set_plan(ops_, dims, tiles, cap)
  return ops_, dims


def lanczosnet_input_grad(plan, Lp, V, G, mask_u8, act, dy, dx0, tiling, row_off=None,
                          dy_compact=None, dbias_part=None):
  """lnz_lanczosnet_input_grad: dy[num_layer-1] holds dLoss/dY of the last conv layer on entry;
  fills dy[0..num_layer-2] and dx0 [B,32,din0].  plan['Wp_t'] / plan['wt_off']: transposed packs in
  kernel-layer order (LanczosNet._plan_backward).  All buffers zero-initialised by the caller.
  Optional: dy_compact [num_layer, R, dhid] with row_off [B] int64 — slots 0 .. num_layer-2 receive
  the same gradients in the compact row numbering of the message matrix; dbias_part [>= 2 * cap (with
  max(2 * cap, strip entries) the pass runs on the strip plan),
  num_layer, dhid] (zero-initialised) — per workgroup half the column sums of dY_l, l <= num_layer-2."""
  _need_cuda(Lp, V, G, mask_u8, act, dy, dx0, row_off, dy_compact, dbias_part)
  ops_, dims = _training_args(plan, Lp, V, G, mask_u8, tiling)
  if dy_compact is not None:
    assert row_off is not None and row_off.dtype == torch.int64 and row_off.is_contiguous()
    assert dy_compact.dim() == 3 and dy_compact.is_contiguous() and dy_compact.dtype == torch.float32 \
        and dy_compact.shape[0] == plan['num_layer'] and dy_compact.shape[2] == plan['dhid']
    ops_[_IN['row_off']] = row_off
    dims[_DIM['dy_compact_rows']] = int(dy_compact.shape[1])
  if dbias_part is not None:
    assert dbias_part.shape[0] >= 2 * tiling[1] and \
        tuple(dbias_part.shape[1:]) == (plan['num_layer'], plan['dhid']) and \
        dbias_part.is_contiguous() and dbias_part.dtype == torch.float32
  B = V.shape[0]
  L, dh = plan['num_layer'], plan['dhid']
  assert tuple(dy.shape) == (L, B, 32, dh) and dy.is_contiguous() and dy.dtype == torch.float32
  assert tuple(act.shape) == (L, B, 32, dh) and act.is_contiguous() and act.dtype == torch.float32
  assert tuple(dx0.shape) == (B, 32, plan['din0']) and dx0.is_contiguous()
  dims[_DIM['din0']], dims[_DIM['bwd_din0']] = dh, plan['din0']
  ops_[_IN['Wp']], ops_[_IN['act']] = plan['Wp_t'], act
  _ext().fused_launch(1, ops_, dims, [int(x) for x in plan['wt_off'][:L]], [],
                      [int(p) for p in plan['short']], None, None, None, dy, dx0, None, None, dy_compact, dbias_part)


def lanczosnet_messages(plan, Lp, V, G, mask_u8, act, x0, layer, msg, tiling, row_off=None):
  """lnz_lanczosnet_messages: msg [B*32, C*d] = cat_c(M_c X_layer) (zero-initialised by the
  caller); x0 [B,32,din0] is X_0, act the stored activations of the forward.  With row_off [B]
  int64 (exclusive scan of the node counts) msg is compact, [sum(n), C*d], real nodes only, and
  needs no initialisation."""
  _need_cuda(Lp, V, G, mask_u8, act, x0, msg)
  ops_, dims = _training_args(plan, Lp, V, G, mask_u8, tiling)
  B = V.shape[0]
  d = plan['din0'] if layer == 0 else plan['dhid']
  Cn = len(plan['short']) + plan['n_long'] + plan['n_edge']
  assert msg.shape[1] == Cn * d and msg.is_contiguous() and msg.dtype == torch.float32
  assert row_off is not None or msg.shape[0] == B * 32
  assert tuple(x0.shape) == (B, 32, plan['din0']) and x0.is_contiguous()
  dims[_DIM['msg_layer']] = int(layer)
  ops_[_IN['act']], ops_[_IN['x0']] = act, x0
  if row_off is not None:
    assert row_off.dtype == torch.int64 and row_off.is_contiguous() and row_off.numel() == B
    ops_[_IN['row_off']] = row_off
  _ext().fused_launch(2, ops_, dims, [], [], [int(p) for p in plan['short']], None, None, None,
                      None, None, msg, None, None, None)


def lanczosnet_gain_grad(plan, Lp, V, G, mask_u8, act, x0, dy, tiling):
  """lnz_lanczosnet_gain_grad: dG [num_layer,B,K,S] = dLoss/d(spectral gains) from the stored
  activations (act, x0) and the pre-activation gradients dy left by lanczosnet_input_grad —
  sum_o (V^T dY_l)[k][o] ((V^T X_l) W_{l,s}^T)[k][o], evaluated per node tile in eigen space.
  Uses the FORWARD weight packs of `plan`."""
  _need_cuda(V, mask_u8, act, x0, dy)
  ops_, dims = _training_args(plan, Lp, V, G, mask_u8, tiling)
  B, _, K = V.shape
  L, dh, S = plan['num_layer'], plan['dhid'], plan['n_long']
  assert tuple(dy.shape) == (L, B, 32, dh) and dy.is_contiguous() and dy.dtype == torch.float32
  assert tuple(act.shape) == (L, B, 32, dh) and act.is_contiguous() and act.dtype == torch.float32
  assert tuple(x0.shape) == (B, 32, plan['din0']) and x0.is_contiguous()
  ops_[_IN['Wp']], ops_[_IN['act']], ops_[_IN['x0']] = plan['Wp'], act, x0
  # zero-initialised: eigen slots beyond a molecule's row block (k >= split of a shared tile) are
  # dead (k >= n) and are not written
  dG = torch.zeros((L, B, K, S), dtype=torch.float32, device=V.device)
  _ext().fused_launch(3, ops_, dims, [int(x) for x in plan['w_off'][:L]], [],
                      [int(p) for p in plan['short']], None, None, None, dy, None, None, dG, None, None)
  return dG


# ------------------------------------------------------------------------------ R4, R5, R8 (Ada)
def ada_graph_laplacian(node_feat, embedding, L0):
  """Learned Laplacian (model/ada_lanczos_net.py:101-137).  node_feat: [B,N] int64 ids (with
  `embedding` [num_atom, D]) or [B,N,D] float features (embedding=None).  L0: [B,N,N] view of the
  simple-graph Laplacian (adjacency mask = L0 != 0, :310-311).  Returns Le [B,N,N]."""
  _need_cuda(node_feat, embedding, L0)
  B, N = L0.shape[0], L0.shape[1]
  assert L0.dtype == torch.float32
  Le = torch.empty((B, N, N), dtype=torch.float32, device=L0.device)
  sb, sr, sc = L0.stride()
  if node_feat.dtype == torch.int64:
    emb = _f32c(embedding)
    nf = node_feat.contiguous()
    args = (nf, emb, emb.shape[0], None, emb.shape[1])
  else:
    nf = _f32c(node_feat)
    args = (None, None, 0, nf, nf.shape[2])
  with torch.cuda.device(L0.device):
    _abi().ada_graph_laplacian(*args, L0, sb, sr, sc, B, N, Le)
  return Le


def ada_lanczos_layer(A, mask, q1, K):
  """Reference-exact in-model Lanczos layer (model/ada_lanczos_net.py:139-247).
  A [B,N,N], mask [B,N] (or None), q1 [B,N] raw start vector -> T [B,K,K], Q [B,N,K]."""
  _need_cuda(A, mask, q1)
  A = _f32c(A)
  B, N, _ = A.shape
  q1 = _f32c(q1.reshape(B, N))
  m = mask.to(torch.uint8).contiguous() if mask is not None else None
  T = torch.empty((B, K, K), dtype=torch.float32, device=A.device)
  Q = torch.empty((B, N, K), dtype=torch.float32, device=A.device)
  with torch.cuda.device(A.device):
    _abi().ada_lanczos_layer(A, m, q1, B, N, K, T, Q)
  return T, Q


def ada_lanczos_layer_f64(Le, mask, q1, K):
  """lnz_ada_lanczos_layer_f64: the Lanczos layer on an fp64 Laplacian Le [B,N,N]; returns
  (T [B,K,K], Q [B,N,K], ws) in fp64 — ws is the state ada_lanczos_layer_f64_backward needs."""
  _need_cuda(Le, mask, q1)
  assert Le.dtype == torch.float64 and Le.is_contiguous() and Le.dim() == 3
  B, N = Le.shape[0], Le.shape[1]
  mask_u8 = None if mask is None else mask.to(torch.uint8).contiguous()
  q = _f32c(q1).reshape(B, N)
  T = torch.empty((B, K, K), dtype=torch.float64, device=Le.device)
  Q = torch.empty((B, N, K), dtype=torch.float64, device=Le.device)
  ws = torch.empty((int(_abi().ada_lanczos_f64_workspace_doubles(B)),), dtype=torch.float64,
                   device=Le.device)
  with torch.cuda.device(Le.device):
    _abi().ada_lanczos_layer_f64(Le, mask_u8, q, B, N, K, T, Q,
                                             ws)
  return T, Q, ws


def ada_lanczos_layer_f64_backward(Le, ws, dT, dQ):
  """dLoss/dLe [B,N,N] fp64 from dLoss/dT [B,K,K], dLoss/dQ [B,N,K] (fp64) and the forward's state."""
  _need_cuda(Le, ws, dT, dQ)
  B, N = Le.shape[0], Le.shape[1]
  K = dT.shape[1]
  dT = dT.to(torch.float64).contiguous()
  dQ = dQ.to(torch.float64).contiguous()
  assert tuple(dT.shape) == (B, K, K) and tuple(dQ.shape) == (B, N, K)
  dLe = torch.empty_like(Le)
  with torch.cuda.device(Le.device):
    _abi().ada_lanczos_layer_f64_backward(Le, B, N, K, ws, dT, dQ,
                                                      dLe)
  return dLe


def ada_graph_laplacian_f64(state, L0):
  """lnz_ada_graph_laplacian_f64: learned Laplacian in fp64 from the embedded node states
  [B,N,D] fp32 and the adjacency mask L0 != 0 ([B,N,N] view, any strides).  Returns (Le [B,N,N]
  fp64, saved state for ada_graph_laplacian_f64_backward)."""
  _need_cuda(state, L0)
  X = _f32c(state)
  B, N, D = X.shape
  assert L0.dtype == torch.float32 and tuple(L0.shape) == (B, N, N)
  Le = torch.empty((B, N, N), dtype=torch.float64, device=X.device)
  sv = torch.empty((int(_abi().ada_laplacian_f64_state_doubles(B, N)),), dtype=torch.float64,
                   device=X.device)
  with torch.cuda.device(X.device):
    _abi().ada_graph_laplacian_f64(X, D, L0, L0.stride(0), L0.stride(1),
                                               L0.stride(2), B, N, Le, sv)
  return Le, (X, sv)


def ada_graph_laplacian_f64_backward(saved, dLe):
  """dLoss/dstate [B,N,D] fp64 from dLoss/dLe [B,N,N] fp64."""
  X, sv = saved
  _need_cuda(X, sv, dLe)
  B, N, D = X.shape
  dLe = dLe.to(torch.float64).contiguous()
  dX = torch.empty((B, N, D), dtype=torch.float64, device=X.device)
  with torch.cuda.device(X.device):
    _abi().ada_graph_laplacian_f64_backward(X, D, B, N, sv, dLe, dX)
  return dX


def ada_t_powers_f64(T, dist):
  """lnz_ada_t_powers_f64: T [B,K,K] fp64 -> (Tcat [B, K, S*K] fp32, saved powers)."""
  _need_cuda(T)
  assert T.dtype == torch.float64 and T.is_contiguous()
  B, K, _ = T.shape
  S = len(dist)
  pmax = max(int(x) for x in dist)
  out = torch.empty((B, K, S * K), dtype=torch.float32, device=T.device)
  P = torch.empty((B, pmax, K, K), dtype=torch.float64, device=T.device)
  darr = [int(x) for x in dist]
  with torch.cuda.device(T.device):
    _abi().ada_t_powers_f64(T, B, K, darr, S, out, P)
  return out, (T, P, tuple(int(x) for x in dist))


def ada_t_powers_f64_backward(saved, dTcat):
  """dLoss/dT [B,K,K] fp64 from dLoss/dTcat [B, K, S*K] (or [B, K*S*K]) fp32."""
  T, P, dist = saved
  _need_cuda(T, P, dTcat)
  B, K, _ = T.shape
  S = len(dist)
  g = _f32c(dTcat).reshape(B, K, S * K)
  dT = torch.empty_like(T)
  darr = [int(x) for x in dist]
  with torch.cuda.device(T.device):
    _abi().ada_t_powers_f64_backward(T, B, K, darr, S, g, P, dT)
  return dT


def ada_t_powers(T, dist):
  """T [B,K,K] -> Tcat [B, K, S*K] = cat([T^p for p in dist], dim=2) (:262-270)."""
  _need_cuda(T)
  T = _f32c(T)
  B, K, _ = T.shape
  S = len(dist)
  out = torch.empty((B, K, S * K), dtype=torch.float32, device=T.device)
  darr = [int(x) for x in dist]
  with torch.cuda.device(T.device):
    _abi().ada_t_powers(T, B, K, darr, S, out)
  return out


def ada_symmetrize_filters(DD, K, S, out=None):
  """DD [B, K*K*S] (MLP output) -> DDp [B,S,K,K] = (DD + DD^T)/2 (:274-278)."""
  _need_cuda(DD)
  DD = _f32c(DD)
  B = DD.shape[0]
  if out is None:
    out = torch.empty((B, S, K, K), dtype=torch.float32, device=DD.device)
  with torch.cuda.device(DD.device):
    _abi().ada_symmetrize_filters(DD, B, K, S, out)
  return out


def split_f16x3(X, bias=None, alpha=1.0, relu=False, Kp=None, out=None):
  """[relu](alpha * X + bias), X [M, K] fp32 -> [M, 3 Kp] fp16 = [hi | hi | lo] (lnz_split_f16x3):
  the activation operand of a split-precision fp16 GEMM against weights [w_hi | w_lo | w_hi]."""
  _need_cuda(X, bias)
  assert X.dim() == 2 and X.dtype == torch.float32 and X.stride(1) == 1
  M, K = X.shape
  Kp = Kp or (K + 3) // 4 * 4
  if out is None:
    out = torch.empty((M, 3 * Kp), dtype=torch.float16, device=X.device)
  b = None if bias is None else _f32c(bias)
  with torch.cuda.device(X.device):
    _abi().split_f16x3(X, M, K, X.stride(0), b, float(alpha), int(relu), Kp,
                                   out)
  return out


F16X3_WEIGHT_SCALE = 1024.0


def split_weight_f16x3(W, scale=F16X3_WEIGHT_SCALE, Kp=None):
  """W [N, K] fp32 -> [N, 3 Kp] fp16 = [w_hi | w_lo | w_hi] of scale * W (a power-of-two scale
  keeps the low pieces of Xavier-sized weights out of fp16's subnormal range; undo it with
  alpha = 1 / scale in the next split_f16x3)."""
  W = W.detach().float() * scale
  N, K = W.shape
  Kp = Kp or (K + 3) // 4 * 4
  hi = W.half()
  lo = (W - hi.float()).half()
  out = torch.zeros((N, 3 * Kp), dtype=torch.float16, device=W.device)
  out[:, :K] = hi
  out[:, Kp:Kp + K] = lo
  out[:, 2 * Kp:2 * Kp + K] = hi
  return out


# ------------------------------------------------- R8, hand-written split-precision Linear chain
def f16x3_pack_weight(W, scale=F16X3_WEIGHT_SCALE):
  """W [N, K] fp32 -> planes [2, Np, Kp] fp16 (hi, lo) of scale * W, Np = ceil(N / 128) * 128,
  Kp = ceil(K / 64) * 64, zero padded: the weight operand of lnz_f16x3_linear."""
  W = W.detach().float() * scale
  N, K = W.shape
  Np, Kp = (N + 127) // 128 * 128, (K + 63) // 64 * 64
  out = torch.zeros((2, Np, Kp), dtype=torch.float16, device=W.device)
  hi = W.half()
  out[0, :N, :K] = hi
  out[1, :N, :K] = (W - hi.float()).half()
  return out


def f16x3_split(X, Kp=None, scale=1.0, out=None):
  """X [M, K] fp32 -> planes [2, Mp, Kp] fp16 (hi, lo), Mp = ceil(M / 128) * 128 (lnz_f16x3_split)."""
  _need_cuda(X)
  assert X.dim() == 2 and X.dtype == torch.float32 and X.stride(1) == 1
  M, K = X.shape
  Mp = (M + 127) // 128 * 128
  Kp = Kp or (K + 63) // 64 * 64
  if out is None:
    out = torch.empty((2, Mp, Kp), dtype=torch.float16, device=X.device)
  with torch.cuda.device(X.device):
    _abi().f16x3_split(X, M, K, X.stride(0), float(scale), Mp, Kp, out[0],
                                   out[1])
  return out


def f16x3_linear(xp, wp, bias, M, N, alpha=1.0 / F16X3_WEIGHT_SCALE, relu=True, out_planes=None,
                 out_f32=None):
  """[relu](alpha * X W^T + bias) by lnz_f16x3_linear.  xp [2, Mp, K] / wp [2, Np, K] are (hi, lo)
  fp16 planes; the result goes to `out_planes` [2, Mp, >= N] (the next Linear's operand; allocated
  zeroed when None and out_f32 is None) or to the fp32 matrix `out_f32` [M, >= N]."""
  _need_cuda(xp, wp, bias, out_planes, out_f32)
  assert xp.dtype == torch.float16 and wp.dtype == torch.float16 and xp.shape[2] == wp.shape[2]
  K = xp.shape[2]
  assert xp.shape[1] >= (M + 127) // 128 * 128 and wp.shape[1] >= (N + 127) // 128 * 128
  if out_f32 is None and out_planes is None:
    out_planes = torch.zeros((2, xp.shape[1], (N + 63) // 64 * 64), dtype=torch.float16,
                             device=xp.device)
  b = None if bias is None else _f32c(bias)
  ns = _abi().f16x3_linear_splits(M, N, K)   # split-K scratch for shapes with few output tiles
  part = torch.empty((ns, M, N), dtype=torch.float32, device=xp.device) if ns > 1 else None
  with torch.cuda.device(xp.device):
    if out_f32 is not None:
      assert out_f32.dtype == torch.float32 and out_f32.stride(1) == 1 and out_f32.shape[1] >= N
      _abi().f16x3_linear(xp[0], xp[1], xp.stride(1), wp[0], wp[1],
                                      wp.stride(1), b, float(alpha), int(relu), M, N, K,
                                      None, None, out_f32, out_f32.stride(0),
                                      part)
      return out_f32
    assert out_planes.dtype == torch.float16 and out_planes.shape[2] >= N
    _abi().f16x3_linear(xp[0], xp[1], xp.stride(1), wp[0], wp[1],
                                    wp.stride(1), b, float(alpha), int(relu), M, N, K,
                                    out_planes[0], out_planes[1], None,
                                    out_planes.stride(1), part)
  return out_planes


def f32_linear_supported(x, w):
  """Shapes lnz_f32_linear takes: fp32 row-major operands, K a multiple of 32, 16-byte rows."""
  return (x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and w.dim() == 2 and
          x.shape[1] == w.shape[1] and x.shape[1] % 32 == 0 and x.stride(1) == 1 and w.stride(1) == 1
          and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and
          w.data_ptr() % 16 == 0)


def f32_linear(x, w, bias=None, relu=False, out=None):
  """[relu](x w^T + bias) by lnz_f32_linear (hand-written exact-fp32 MFMA kernel): x [M, K],
  w [N, K] (nn.Linear layout), result [M, N] fp32."""
  _need_cuda(x, w, bias, out)
  assert f32_linear_supported(x, w), (x.shape, w.shape, x.stride(), w.stride())
  M, K = x.shape
  N = w.shape[0]
  if out is None:
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
  assert out.dtype == torch.float32 and out.stride(1) == 1 and tuple(out.shape) == (M, N)
  b = None if bias is None else _f32c(bias)
  part = _f32_linear_workspace(M, N, K, x.device)
  with torch.cuda.device(x.device):
    _abi().f32_linear(x, x.stride(0), w, w.stride(0), b, int(relu), M, N,
                                  K, out, out.stride(0), part)
  return out


import collections

_F32_LINEAR_WS = collections.OrderedDict()   # (device, stream, size) -> zero-initialised buffer
_F32_LINEAR_WS_PINNED = {}                    # id -> buffer a captured HIP graph holds the address of
_F32_LINEAR_WS_ENTRIES = 32                   # entries kept; the least recently used goes


def _f32_linear_workspace(M, N, K, device):
  """Stream-K workspace of lnz_f32_linear (None for shapes that run one workgroup per tile): the
  partial tiles + the tile counters.  The counters have to be zero on entry and are zero again on
  return, the partial tiles are not — so a buffer serves ONE layout only: the key carries the size
  (a buffer shared between two shapes would hand the second one the first one's partial tiles as
  counters, and a stream-K workgroup spins on its counter).  One zero-initialised buffer per
  (device, stream, size), no fill launch per Linear; kernels on one stream run in order, another
  stream gets its own buffer.  Bounded: the 32 most recently used entries are kept (a few MB each);
  buffers handed out under HIP-graph capture (train.GraphedTrainStep) are pinned for the life of the
  process — the graph holds their addresses."""
  need = int(_abi().f32_linear_workspace_floats(M, N, K))
  if need == 0:
    return None
  key = (device.index, torch.cuda.current_stream(device).cuda_stream, need)
  ws = _F32_LINEAR_WS.get(key)
  if ws is None:
    ws = torch.zeros((need,), dtype=torch.float32, device=device)
  _F32_LINEAR_WS[key] = ws
  _F32_LINEAR_WS.move_to_end(key)
  if torch.cuda.is_current_stream_capturing():
    _F32_LINEAR_WS_PINNED[id(ws)] = ws
  while len(_F32_LINEAR_WS) > _F32_LINEAR_WS_ENTRIES:
    _F32_LINEAR_WS.popitem(last=False)
  return ws


# ----------------------------------------------------------------------------------------- R12
def unsorted_segment_sum_forward(data, segment_ids, num_segments):
  _need_cuda(data, segment_ids)
  data = _f32c(data)
  ids = segment_ids.to(torch.int64).contiguous()
  B, D1, D2 = data.shape
  return _ext().unsorted_segment_sum_forward(data, ids, num_segments)


def unsorted_segment_sum_backward(grad_out, segment_ids, dim1):
  _need_cuda(grad_out, segment_ids)
  g = _f32c(grad_out)
  ids = segment_ids.to(torch.int64).contiguous()
  B, S, D2 = g.shape
  return _ext().unsorted_segment_sum_backward(g, ids, dim1)
