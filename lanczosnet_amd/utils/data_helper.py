"""Device-side counterparts of reference `utils/data_helper.py` for dense padded batches.

  check_dist                      utils/data_helper.py:9-14   (host logic, same behaviour)
  get_laplacian_l4_batched        :92-116,155-156  -> HIP `lnz_laplacian_l4`
  get_graph_laplacian_eigs_batched :169-258 (use_eigen_decomp=True, is_sym=True branch) +
                                  the pad/cut of dataset/qm8.py:264-291 -> HIP `lnz_lanczos_ritz`;
                                  the use_eigen_decomp=False branch (:205-208) -> `lnz_lanczos_ritz_kstep`
"""


def check_dist(dist):
    for dd in dist:
        if not isinstance(dd, int) and dd != 'inf':
            raise ValueError("Non-supported value of diffusion distance")
    return dist


def get_laplacian_l4_batched(adjs, n_nodes):
    """adjs [B,N,N,E] cuda float, n_nodes [B] -> L [B,N,N,E+1] (channel 0 = simple graph)."""
    from .. import ops
    return ops.laplacian_l4(adjs, n_nodes)


def get_graph_laplacian_eigs_batched(L_simple, n_nodes, k, use_eigen_decomp=None):
    """L_simple [B,N,N] (e.g. `L[..., 0]`), n_nodes [B] -> (D [B,k], V [B,N,k]) ordered by
    descending |eigenvalue| like `np.argsort(-|eigs|, kind='mergesort')` (:218-223).

    use_eigen_decomp (the reference's switch, :173,199-208):
      True   the pairs of the FULL decomposition (`np.linalg.eigh`, :201) — full-length Lanczos +
             tridiagonal eigensolver, `lnz_lanczos_ritz`, graphs of up to 192 nodes; larger ones
             raise (the reference itself calls this branch "computationally heavy for large size adj");
      False  the k-dimensional Krylov method (`eigsh(L, k, which='LM')`, :208) — the k-step
             Lanczos of `lnz_lanczos_ritz_kstep`, any N <= 2048, ragged batches;
      None   (default) True up to 192 nodes, False beyond, with a UserWarning naming the branch."""
    from .. import ops, _lib
    N = L_simple.shape[1]
    if use_eigen_decomp is None:
        return ops.lanczos_ritz(L_simple, n_nodes, k)
    if use_eigen_decomp:
        if N > ops.RITZ_FULL_MAX_N:
            raise _lib.NotSupported(_lib.LNZ_ENOTSUP,
                                    'use_eigen_decomp=True serves graphs of up to %d nodes (got %d); '
                                    'use_eigen_decomp=False is the k-step branch' % (ops.RITZ_FULL_MAX_N, N))
        return ops.lanczos_ritz(L_simple, n_nodes, k)
    return ops.lanczos_ritz_kstep(L_simple, n_nodes, min(k, 64), k)
