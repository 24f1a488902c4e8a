"""GPU parity on the reference's OWN graph configuration (config/graph_lanczos_net.yaml: graphs of
20..100 nodes, K = 20, one edge type, LanczosNetGeneral 10 -> 7 x 128 -> 2; batch sizes 10 and 64).

Fixture: tests/golden/graph_config.npz — the unmodified reference run end to end in the build
container (dataset/get_graph_data.py -> dataset/graph_data.py -> model/lanczos_net_general.py), see
tests/golden/make_golden_graph.py.  Everything here goes through the C ABI:

  raw adjacency --lnz_laplacian_l4--> L --lnz_lanczos_ritz (workgroup-per-graph kernel)--> (D, V)
               --LanczosNetGeneral (lnz_large_* streamed kernels, split-precision planes)--> score

Tolerances: L 1e-7 abs, sorted D 1e-6 abs, V diag(D^p) V^T 1e-5 rel for
p in {1, 5, 30}, score 1e-5 rel per graph.
"""
import numpy as np
import pytest
import torch

import oracle
from graph_fixture import GRAPH_CFG, check_ritz, load_split, pad_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
K = GRAPH_CFG['num_eig_vec']


def _t(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _net(seed):
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  P = oracle.make_lanczosnet_params(GRAPH_CFG, seed, general=True)
  net = LanczosNetGeneral(make_model_config(GRAPH_CFG, general=True)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return net.to(DEV)


def _eigh_ref(A, ns, N, Kk):
  Dl, Vl, full = [], [], np.zeros((len(ns), N))
  for b, n in enumerate(ns):
    e, v = np.linalg.eigh(A[b, :n, :n].astype(np.float64))
    idx = np.argsort(-np.abs(e), kind='mergesort')
    Dl.append(e[idx])
    Vl.append(v[:, idx])
    full[b, :n] = e[idx]
  Dr, Vr = oracle.collate_eigs(Dl, Vl, N, Kk)
  return Dr, Vr, full


@pytest.mark.parametrize('split', ['train', 'test'])
def test_graph_config_device_pipeline_matches_reference(split):
  from lanczosnet_amd import ops
  items, ref, seed, _ = load_split(split)
  adjs, X, mask, n = pad_batch(items)
  N = mask.shape[1]
  assert N > 32   # beyond the wavefront-per-molecule kernel of the QM8 regime
  nd = _t(n)
  L = ops.laplacian_l4(_t(adjs), nd)
  if split == 'train':
    assert np.abs(L[..., 0].cpu().numpy() - ref['L0']).max() < 1e-7
  assert torch.equal(L[..., 0], L[..., 1])
  D, V, info = ops.lanczos_ritz(L[:, :, :, 0], nd, K, return_info=True)   # strided view, no copy
  Dn, Vn = D.cpu().numpy(), V.cpu().numpy()
  wd, wp, checked = check_ritz(Dn, Vn, ref['D'], ref['V'], n, ref['D_full'], K)
  assert checked == len(n)
  print('graph config %s: B=%d N=%d  max|dD| %.2e  worst projector %.2e  restarts %d'
        % (split, len(n), N, wd, wp, int((info % 256).sum())))
  # every graph of the reference's configuration takes the parallel tridiagonal eigensolver
  assert int((info >= 256).sum()) == 0
  net = _net(seed)
  lab = _t(ref['label'])
  with torch.no_grad():
    score, loss = net(_t(X), L, D, V, label=lab, mask=_t(mask))
    score_ref_dv = net(_t(X), L, _t(ref['D']), _t(ref['V']), mask=_t(mask))
  for s in (score, score_ref_dv):
    per = np.abs(s.cpu().numpy() - ref['score']).max(axis=1) / np.abs(ref['score']).max(axis=1)
    assert per.max() < 1e-5, per
  assert abs(float(loss) - ref['loss']) < 1e-5 * ref['loss']


def test_graph_collate_adjacency_is_the_device_pipeline():
  from lanczosnet_amd.dataset.graph_data import collate_graph_adjacency
  items, ref, seed, _ = load_split('train')
  data = collate_graph_adjacency(items, K, device=DEV)
  n = ref['n_nodes']
  np.testing.assert_array_equal(data['n_nodes'].cpu().numpy(), n)
  np.testing.assert_array_equal(data['label'].cpu().numpy(), ref['label'])
  assert np.abs(data['L'][..., 0].cpu().numpy() - ref['L0']).max() < 1e-7
  check_ritz(data['D'].cpu().numpy(), data['V'].cpu().numpy(), ref['D'], ref['V'], n,
             ref['D_full'], K)
  net = _net(seed)
  with torch.no_grad():
    score = net(data['node_feat'], data['L'], data['D'], data['V'], mask=data['node_mask'])
  per = np.abs(score.cpu().numpy() - ref['score']).max(axis=1) / np.abs(ref['score']).max(axis=1)
  assert per.max() < 1e-5, per


@pytest.mark.parametrize('model_name,kind,sign', [('DCNN', 'L7', 1.0), ('ChebyNet', 'L6', -1.0)])
def test_graph_collate_adjacency_baseline_branches(model_name, kind, sign):
  """The DCNN / ChebyNet branches of the reference's graph collate (dataset/graph_data.py:247-260:
  simple-graph channel = L_simple_7, resp. MINUS L_simple_6, both written by get_graph_data.py:70-75
  with get_laplacian on the summed adjacency) from the raw graphs on the device; bond-type channels
  and Ritz pairs stay those of the default branch."""
  from lanczosnet_amd.dataset.graph_data import collate_graph_adjacency
  items, ref, _, _ = load_split('train')
  base = collate_graph_adjacency(items, K, device=DEV)
  data = collate_graph_adjacency(items, K, device=DEV, model_name=model_name)
  L = data['L'].cpu().numpy()
  assert np.array_equal(L[..., 1:], base['L'][..., 1:].cpu().numpy())
  assert torch.equal(data['D'], base['D']) and torch.equal(data['V'], base['V'])
  for b, it in enumerate(items):
    n = int(ref['n_nodes'][b])
    want = sign * oracle.get_laplacian(np.asarray(it['adjs'], np.float64).sum(axis=2), kind)
    assert np.abs(L[b, :n, :n, 0] - want).max() < 1e-7
    assert (L[b, n:, :, 0] == 0).all() and (L[b, :, n:, 0] == 0).all()


def _random_laplacians(rs, B, N, n_lo, n_hi, p):
  A = np.zeros((B, N, N), np.float32)
  ns = rs.randint(n_lo, n_hi + 1, size=B).astype(np.int32)
  ns[0] = N
  for b in range(B):
    n = ns[b]
    adj = np.triu((rs.rand(n, n) < p).astype(np.float64), 1)
    A[b, :n, :n] = oracle.laplacian_l4(adj + adj.T)
  return A, ns


def _lds_boundary():
  from lanczosnet_amd import _lib
  lib = _lib.load()
  return max(N for N in range(33, 193) if lib.lnz_lanczos_ritz_workspace_bytes(8, N) == 0)


@pytest.mark.parametrize('N,p', [(70, 0.5), (100, 0.08), ('fit', 0.3), ('fit+1', 0.3), (150, 0.5),
                                 (192, 0.012)])
def test_workgroup_ritz_kernel_matches_eigh(N, p):
  """Both placements of the basis (LDS up to the boundary lnz_lanczos_ritz_workspace_bytes reports, device workspace above) against
  numpy.linalg.eigh + the reference's |lambda| sort, incl. sparse graphs with isolated nodes /
  several components (exactly degenerate eigenvalues -> Lanczos restarts)."""
  from lanczosnet_amd import ops
  if isinstance(N, str):   # the largest N whose basis still lives in LDS, and the first beyond it
    N = _lds_boundary() + (1 if N.endswith('+1') else 0)
  rs = np.random.RandomState(N)
  # sparse graphs are full of exactly degenerate eigenvalues (isolated nodes, twin leaves): a top-K
  # cut would split a cluster in nearly every graph, so nothing is cut there
  Kk = 24 if p >= 0.1 else N
  A, ns = _random_laplacians(rs, 5, N, max(2, N // 3), N, p)
  Dr, Vr, full = _eigh_ref(A, ns, N, Kk)
  D, V, info = ops.lanczos_ritz(_t(A), _t(ns), Kk, return_info=True)
  wd, wp, c = check_ritz(D.cpu().numpy(), V.cpu().numpy(), Dr, Vr, ns, full, Kk, powers=(1, 5))
  assert c >= 3
  if p < 0.1:
    assert int((info % 256).sum()) > 0   # the restart branch ran
  # the eight-wave Lanczos phase (what runs with the basis in the workspace) with the basis in LDS
  # and in the workspace: the same arithmetic in the same order, bit-identical results; the
  # one-wave phase of the LDS placement sums in another order
  D2, V2 = ops.lanczos_ritz(_t(A), _t(ns), Kk, kernel='workgroup_ws')
  Dm, Vm = ops.lanczos_ritz(_t(A), _t(ns), Kk, kernel='workgroup_mw')
  assert torch.equal(Dm, D2) and torch.equal(Vm, V2)
  assert (D - D2).abs().max().item() < 1e-6
  for parts in ('workgroup_p1', 'workgroup_p2', 'workgroup_p4'):   # every split of the wave-level phase
    Dp, Vp = ops.lanczos_ritz(_t(A), _t(ns), Kk, kernel=parts)
    assert (Dp - D2).abs().max().item() < 1e-6, parts
    check_ritz(Dp.cpu().numpy(), Vp.cpu().numpy(), Dr, Vr, ns, full, Kk, powers=(1, 5))
  check_ritz(D2.cpu().numpy(), V2.cpu().numpy(), Dr, Vr, ns, full, Kk, powers=(1, 5))
  # the QL sweep (the fallback of the parallel tridiagonal eigensolver), forced: same function
  Dq, Vq, iq = ops.lanczos_ritz(_t(A), _t(ns), Kk, return_info=True, kernel='workgroup_ql')
  assert (iq.cpu().numpy() >= 256).all()
  check_ritz(Dq.cpu().numpy(), Vq.cpu().numpy(), Dr, Vr, ns, full, Kk, powers=(1, 5))
  assert np.abs(Dq.cpu().numpy() - D.cpu().numpy()).max() < 1e-6
  # K >= n (nothing cut): every slot is checked, zero padded beyond n
  D3, V3 = ops.lanczos_ritz(_t(A[:2]), _t(ns[:2]), N)
  Dr3, Vr3, full3 = _eigh_ref(A[:2], ns[:2], N, N)
  assert np.abs(D3.cpu().numpy() - Dr3).max() < 1e-6
  assert (V3.cpu().numpy()[1, :, ns[1]:] == 0).all()


def test_workgroup_ritz_kernel_small_graphs_and_edge_cases():
  """The workgroup kernel at the sizes the wavefront kernels own (cross-check of two independent
  implementations of the same function), with an empty graph, single nodes, a star, a ring, a
  complete graph and disconnected pieces."""
  from lanczosnet_amd import ops
  N = 40
  mats, ns = [], []

  def add(adj):
    n = adj.shape[0]
    A = np.zeros((N, N), np.float32)
    A[:n, :n] = oracle.laplacian_l4(adj)
    mats.append(A)
    ns.append(n)
  n = 33
  star = np.zeros((n, n)); star[0, 1:] = 1; star[1:, 0] = 1
  ring = np.zeros((n, n))
  for i in range(n):
    ring[i, (i + 1) % n] = ring[(i + 1) % n, i] = 1
  two = np.zeros((n, n)); two[:16, :16] = ring[:16, :16]; two[16:, 16:] = star[:17, :17]
  add(star); add(ring); add(np.ones((n, n)) - np.eye(n)); add(np.zeros((7, 7))); add(two)
  add(np.zeros((1, 1))); add(np.ones((2, 2)) - np.eye(2))
  rs = np.random.RandomState(3)
  for _ in range(4):
    m = int(rs.randint(5, N + 1))
    a = np.triu((rs.rand(m, m) < 0.2).astype(np.float64), 1)
    add(a + a.T)
  A = np.stack(mats + [np.zeros((N, N), np.float32)])
  ns = np.array(ns + [0], np.int32)
  Kk = 20
  for kern in ('workgroup', 'workgroup_ws'):
    D, V = ops.lanczos_ritz(_t(A), _t(ns), Kk, kernel=kern)
    D, V = D.cpu().numpy(), V.cpu().numpy()
    assert np.isfinite(D).all() and np.isfinite(V).all()
    assert (D[-1] == 0).all() and (V[-1] == 0).all()
    Dr, Vr, full = _eigh_ref(A[:-1], ns[:-1], N, Kk)
    check_ritz(D[:-1], V[:-1], Dr, Vr, ns[:-1], full, Kk, powers=(1, 5, 30))
  # against the wavefront-per-graph kernel (N <= 64): same function, independent schedule
  Dw, Vw = ops.lanczos_ritz(_t(A), _t(ns), Kk, kernel='auto')
  assert np.abs(Dw.cpu().numpy() - D).max() < 1e-6


def test_graphs_beyond_one_workgroup_take_the_kstep_branch_and_say_so():
  """Beyond 192 nodes the hand-written full-length decomposition is not offered: the workgroup kernel
  refuses (use_eigen_decomp=True goes to the vendor eigensolver, next test), the default entry answers
  with the reference's OTHER branch
  (use_eigen_decomp=False: a K-dimensional Krylov method, utils/data_helper.py:205-208) and warns.
  For a graph whose Krylov space from the start vector is smaller than K the two branches coincide:
  the pairs are exact eigenpairs."""
  from lanczosnet_amd import ops, _lib
  from lanczosnet_amd.utils.data_helper import get_graph_laplacian_eigs_batched
  n = 200
  a = np.zeros((n, n))
  a[0, 1:] = a[1:, 0] = 1.0                     # a star: L4 has three distinct eigenvalues
  A = torch.from_numpy(oracle.laplacian_l4(a)[None].astype(np.float32)).to(DEV)
  nn = torch.tensor([n], dtype=torch.int32, device=DEV)
  with pytest.raises(_lib.NotSupported):
    ops.lanczos_ritz(A, nn, 20, kernel='workgroup')
  ops._WARNED.clear()
  with pytest.warns(UserWarning, match='use_eigen_decomp=False'):
    D, V = get_graph_laplacian_eigs_batched(A, nn, 20)
  D2, V2 = get_graph_laplacian_eigs_batched(A, nn, 20, use_eigen_decomp=False)
  assert torch.equal(D, D2) and torch.equal(V, V2)
  lam = np.linalg.eigvalsh(A[0].double().cpu().numpy())
  # the start vector is not orthogonal to the lambda = 1 and the two star eigenvectors; of the
  # (n - 2)-fold eigenvalue it meets one direction: four Ritz values, all exact
  got = D[0].cpu().numpy()
  live = got[np.abs(got) > 0]
  assert 3 <= len(live) <= 4
  assert all(np.abs(lam - x).min() < 1e-6 for x in live) and abs(live[0] - 1.0) < 1e-6
  Vd = V[0].double().cpu().numpy()[:, :len(live)]
  Ad = A[0].double().cpu().numpy()
  assert np.abs(Ad @ Vd - Vd * live).max() < 1e-6


def test_full_decomposition_beyond_192_nodes_runs_on_the_vendor_eigensolver_and_says_so():
  """use_eigen_decomp=True has no size limit in the reference (np.linalg.eigh, utils/data_helper.py:
  199-201).  Beyond the kernels' 192 nodes it is served by torch.linalg.eigh (fp64, device) with a
  UserWarning: ragged batch, the reference's |eigenvalue| order / pad / cut (oracle.graph_eigs =
  numpy on the unpadded blocks), eigenvalues to 1e-6, the selected invariant subspaces to 1e-5 where
  the cut does not fall inside a cluster; V zero on the padding, D / V zero beyond n."""
  from lanczosnet_amd.utils.data_helper import get_graph_laplacian_eigs_batched
  rs = np.random.RandomState(3)
  N, K = 300, 40
  sizes = [300, 257, 257, 30]
  A = np.zeros((len(sizes), N, N), np.float32)
  for b, n in enumerate(sizes):
    a = np.triu((rs.rand(n, n) < 0.05).astype(np.float64), 1)
    A[b, :n, :n] = oracle.laplacian_l4(a + a.T)
  nn = torch.tensor(sizes, dtype=torch.int32, device=DEV)
  with pytest.warns(UserWarning, match='vendor eigensolver'):
    D, V = get_graph_laplacian_eigs_batched(torch.from_numpy(A).to(DEV), nn, K, use_eigen_decomp=True)
  D, V = D.cpu().numpy(), V.double().cpu().numpy()
  assert D.shape == (len(sizes), K) and V.shape == (len(sizes), N, K)
  for b, n in enumerate(sizes):
    w, U = np.linalg.eigh(A[b, :n, :n].astype(np.float64))
    idx = np.argsort(-np.abs(w), kind='mergesort')
    kk = min(K, n)
    assert np.abs(D[b, :kk] - w[idx[:kk]]).max() < 1e-6 and (D[b, kk:] == 0).all()
    assert (V[b, n:] == 0).all() and (V[b, :, kk:] == 0).all()
    Vb = V[b, :n, :kk]
    assert np.abs(Vb.T @ Vb - np.eye(kk)).max() < 1e-5
    assert np.abs(A[b, :n, :n].astype(np.float64) @ Vb - Vb * D[b, :kk]).max() < 1e-5      # eigenpairs
    big = np.abs(Vb).argmax(axis=0)
    assert (Vb[big, np.arange(kk)] > 0).all()                                            # the kernels' sign rule
    mags = np.abs(w[idx])
    if kk == n or mags[kk - 1] - mags[kk] > 1e-6:                                          # cut outside a cluster
      Ur = U[:, idx[:kk]]
      assert np.abs(Vb @ Vb.T - Ur @ Ur.T).max() < 1e-5


def _structured_graphs():
  """Graphs whose Laplacians have highly degenerate spectra (every multiple eigenvalue is a Lanczos
  breakdown + restart, and a stress test for the block splitting of the tridiagonal eigensolver)."""
  import itertools
  gs = {}
  n = 64
  a = np.zeros((n, n)); i = np.arange(n - 1); a[i, i + 1] = a[i + 1, i] = 1
  gs['path64'] = a
  c = a.copy(); c[0, n - 1] = c[n - 1, 0] = 1
  gs['cycle64'] = c
  g = np.zeros((81, 81))
  for x, y in itertools.product(range(9), range(9)):
    if x + 1 < 9: g[9 * x + y, 9 * (x + 1) + y] = g[9 * (x + 1) + y, 9 * x + y] = 1
    if y + 1 < 9: g[9 * x + y, 9 * x + y + 1] = g[9 * x + y + 1, 9 * x + y] = 1
  gs['grid9x9'] = g
  h = np.zeros((64, 64))
  for u in range(64):
    for bit in range(6):
      h[u, u ^ (1 << bit)] = 1
  gs['hypercube6'] = h
  kb = np.zeros((70, 70)); kb[:30, 30:] = 1; kb[30:, :30] = 1
  gs['K30,40'] = kb
  bb = np.zeros((90, 90)); bb[:40, :40] = 1; bb[50:, 50:] = 1; np.fill_diagonal(bb, 0)
  for u in range(39, 50): bb[u, u + 1] = bb[u + 1, u] = 1
  gs['barbell'] = bb
  st = np.zeros((100, 100)); st[0, 1:] = st[1:, 0] = 1
  gs['star100'] = st
  pet = np.zeros((10, 10))
  for u in range(5):
    pet[u, (u + 1) % 5] = pet[(u + 1) % 5, u] = 1
    pet[5 + u, 5 + (u + 2) % 5] = pet[5 + (u + 2) % 5, 5 + u] = 1
    pet[u, 5 + u] = pet[5 + u, u] = 1
  gs['8xpetersen'] = np.kron(np.eye(8), pet)
  return gs


@pytest.mark.parametrize('kernel', ['auto', 'workgroup_ws', 'workgroup_ql', 'workgroup_mw', 'workgroup_p2', 'workgroup_p4'])
def test_workgroup_ritz_kernel_on_degenerate_spectra(kernel):
  """Paths, cycles, grids, a hypercube, complete bipartite, barbell, star and disjoint Petersen
  graphs: multiplicities up to 20.  K = N so that no top-K cut splits a cluster; eigenvalues to
  1e-6, and the invariant subspaces through V diag(D^p) V^T; V^T V = I to 1e-5."""
  from lanczosnet_amd import ops
  gs = _structured_graphs()
  N = max(a.shape[0] for a in gs.values())
  B = len(gs)
  A = np.zeros((B, N, N), np.float32)
  ns = np.zeros(B, np.int32)
  for b, (name, adj) in enumerate(gs.items()):
    n = adj.shape[0]
    A[b, :n, :n] = oracle.laplacian_l4(adj)
    ns[b] = n
  D, V, info = ops.lanczos_ritz(_t(A), _t(ns), N, return_info=True, kernel=kernel)
  D, V = D.cpu().numpy(), V.cpu().numpy()
  Dr, Vr, full = _eigh_ref(A, ns, N, N)
  assert np.isfinite(D).all() and np.isfinite(V).all()
  for b, name in enumerate(gs):
    n = ns[b]
    # as multisets: a bipartite graph's spectrum is symmetric, so +x and -x tie in |lambda| and the
    # |lambda| ordering between them is decided by the last bit (in the reference's eigh as well)
    assert np.abs(np.sort(D[b, :n]) - np.sort(Dr[b, :n])).max() < 1e-6, name
    assert (D[b, n:] == 0).all() and (V[b, n:] == 0).all() and (V[b, :, n:] == 0).all()
    assert np.abs(np.sort(np.abs(D[b, :n]))[::-1] - np.abs(D[b, :n])).max() < 1e-6, name  # |lambda| descending
    Vb = V[b, :n, :n].astype(np.float64)
    assert np.abs(Vb.T @ Vb - np.eye(n)).max() < 1e-5, name
    for p in (1, 5):
      a = oracle.spectral_projector(D[b], V[b], p)
      r = oracle.spectral_projector(Dr[b], Vr[b], p)
      assert np.abs(a - r).max() / np.abs(r).max() < 1e-5, (name, p)
  restarts = info.cpu().numpy() % 256
  assert restarts.sum() > 50   # the degenerate spectra really went through the restart path
  print('structured graphs (%s): restarts %s, QL fallbacks %d of %d'
        % (kernel, restarts.tolist(), int((info.cpu().numpy() >= 256).sum()), B))


@pytest.mark.parametrize('p', [0.4, 0.03])
def test_wave_level_lanczos_phase_at_its_size_boundaries(p):
  """The Lanczos phase of graphs with the basis in LDS runs on four waves (csrc/lanczos_ritz_wg.hip,
  lanczos_waves): one row group of 64 rows in four parts up to n = 64, two row groups in two parts
  above.  Graphs on both sides of every boundary (n = 33, 63..66, the row ranges of the parts, N)
  in one batch, dense and sparse (isolated nodes and twin leaves: breakdowns and restarts in every
  graph), every split against numpy.linalg.eigh and against the eight-wave form."""
  from lanczosnet_amd import ops
  N = 100
  sizes = [33, 40, 47, 48, 49, 63, 64, 65, 66, 71, 72, 73, 96, 97, 99, 100]
  rs = np.random.RandomState(7)
  A = np.zeros((len(sizes), N, N), np.float32)
  ns = np.array(sizes, np.int32)
  for b, n in enumerate(sizes):
    adj = np.triu((rs.rand(n, n) < p).astype(np.float64), 1)
    A[b, :n, :n] = oracle.laplacian_l4(adj + adj.T)
  Kk = 24 if p >= 0.1 else N
  Dr, Vr, full = _eigh_ref(A, ns, N, Kk)
  Dm, Vm, im = ops.lanczos_ritz(_t(A), _t(ns), Kk, return_info=True, kernel='workgroup_mw')
  if p < 0.1:
    assert int((im % 256).sum()) > 0   # the restart branch ran
  for kern in ('auto', 'workgroup_p1', 'workgroup_p2', 'workgroup_p4'):
    D, V, info = ops.lanczos_ritz(_t(A), _t(ns), Kk, return_info=True, kernel=kern)
    assert torch.isfinite(D).all() and torch.isfinite(V).all()
    check_ritz(D.cpu().numpy(), V.cpu().numpy(), Dr, Vr, ns, full, Kk, powers=(1, 5))
    assert (D - Dm).abs().max().item() < 1e-6, kern
    assert torch.equal(info % 256 > 0, im % 256 > 0), kern   # the same graphs broke down


@pytest.mark.parametrize('N', [128, 160, 192])
def test_workgroup_ritz_kernel_on_forests_of_equal_stars(N):
  """Massively degenerate spectra beyond 128 nodes (r05 fuzz, tools/experiments/ritz_wg_forest.py):
  caterpillars of equal stars with 112..192 nodes.  Their Ritz values converge within a few steps,
  and the eight-wave Lanczos form — which re-orthogonalised a second time only where the first pass
  had removed more than 99 % of the vector's squared length — lost orthogonality geometrically:
  half of the graphs of 150+ nodes came back with a non-orthonormal V (r04 and r05 alike).  Beyond
  128 nodes the second pass now runs whenever the first removed more than half ("twice is enough")."""
  from lanczosnet_amd import ops
  rs = np.random.RandomState(N)
  B = 48
  ns = rs.randint(N - 16, N + 1, size=B)
  adj = np.zeros((B, N, N, 1), np.float32)
  for b in range(B):
    n, m = int(ns[b]), rs.randint(3, 9)
    a = np.zeros((n, n), np.float32)
    for i in range(1, n):
      a[(i - 1) // m * m if i % m else max(i - m, 0), i] = 1.0
    adj[b, :n, :n, 0] = np.maximum(a, a.T)
  nd = _t(ns.astype(np.int32))
  L = ops.laplacian_l4(_t(adj), nd)
  D, V = ops.lanczos_ritz(L[:, :, :, 0], nd, K)
  A, Vd, Dd = L[:, :, :, 0].double(), V.double(), D.double()
  eye = torch.eye(K, device=DEV, dtype=torch.float64)[None]
  # (the broken form returned overlaps of 1e-2 .. 1; up to 128 nodes the 99 % rule stays — semi-
  # orthogonality: 3e-6 measured on this batch — and beyond it the classical rule gives 1e-6)
  assert (Vd.transpose(1, 2) @ Vd - eye).abs().max().item() < 1e-5
  assert (A @ Vd - Vd * Dd[:, None, :]).abs().max().item() < 1e-5
  for b in range(0, B, 9):
    n = int(ns[b])
    lam = np.linalg.eigvalsh(A[b, :n, :n].cpu().numpy())
    want = np.sort(lam[np.argsort(-np.abs(lam), kind='mergesort')][:K])
    assert np.abs(np.sort(Dd[b].cpu().numpy()) - want).max() < 1e-6


def test_two_stage_stream_of_batches_on_disjoint_compute_units():
  """utils/streams.cu_masked_stream: the graph configuration as a stream of batches — L4 + Ritz pairs
  of batch k+1 on 64 compute units beside the forward of batch k on the other 192, each half a
  captured HIP graph — gives the scores of the sequential step bitwise."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  from lanczosnet_amd.utils.streams import cu_masked_stream
  dev = torch.device(DEV)
  B, K = 16, 20
  cfg = dict(num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
             num_eig_vec=K, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7,
             output_dim=2, num_layer=7, num_atom=0)
  rs = np.random.RandomState(5)
  ns = rs.randint(20, 101, size=B).astype(np.int32)
  N = int(ns.max())
  torch.manual_seed(3)
  net = LanczosNetGeneral(make_model_config(cfg, general=True)).eval().to(dev)
  batches = []
  for _ in range(3):
    adjs = np.zeros((B, N, N, 1), np.float32)
    for b in range(B):
      a = np.triu((rs.rand(ns[b], ns[b]) < 0.5).astype(np.float32), 1)
      adjs[b, :ns[b], :ns[b], 0] = a + a.T
    batches.append(torch.from_numpy(adjs).to(dev))
  nd = torch.from_numpy(ns).to(dev)
  Xd = torch.from_numpy(rs.randn(B, N, 10).astype(np.float32)).to(dev)
  md = torch.from_numpy((np.arange(N)[None, :] < ns[:, None]).astype(np.uint8)).to(dev)
  with torch.no_grad():
    refs = []
    for ad in batches:
      L = ops.laplacian_l4(ad, nd)
      D, V = ops.lanczos_ritz(L[:, :, :, 0], nd, K)
      refs.append(net(Xd, L, D, V, mask=md))
    torch.cuda.synchronize()
    s_prep, s_fwd = cu_masked_stream(0, 64, dev), cu_masked_stream(64, 256, dev)
    ad_in = batches[0].clone()
    slots = []
    for i in range(2):
      sl = {'gp': torch.cuda.CUDAGraph(), 'gf': torch.cuda.CUDAGraph()}
      with torch.cuda.stream(s_prep):
        with torch.cuda.graph(sl['gp'], stream=s_prep):
          sl['L'] = ops.laplacian_l4(ad_in, nd)
          sl['D'], sl['V'] = ops.lanczos_ritz(sl['L'][:, :, :, 0], nd, K)
      with torch.cuda.stream(s_fwd):
        with torch.cuda.graph(sl['gf'], stream=s_fwd):
          sl['score'] = net(Xd, sl['L'], sl['D'], sl['V'], mask=md)
      slots.append(sl)
    torch.cuda.synchronize()
    got = []

    def prep(sl, ad):
      with torch.cuda.stream(s_prep):
        if 'done' in sl:
          s_prep.wait_event(sl['done'])
        ad_in.copy_(ad)
        sl['gp'].replay()
        sl['ready'] = torch.cuda.Event()
        sl['ready'].record(s_prep)

    def fwd(sl):
      with torch.cuda.stream(s_fwd):
        s_fwd.wait_event(sl['ready'])
        sl['gf'].replay()
        got.append(sl['score'].clone())
        sl['done'] = torch.cuda.Event()
        sl['done'].record(s_fwd)
    prep(slots[0], batches[0])
    for k in range(3):
      if k + 1 < 3:
        prep(slots[(k + 1) & 1], batches[k + 1])
      fwd(slots[k & 1])
    torch.cuda.synchronize()
  for k in range(3):
    assert torch.equal(got[k], refs[k])
