"""Device-side counterparts of reference `utils/data_helper.py` for dense padded batches.

  check_dist                      utils/data_helper.py:9-14   (host logic, same behaviour)
  get_laplacian_l4_batched        :92-116,155-156  -> HIP `lnz_laplacian_l4`
  get_graph_laplacian_eigs_batched :169-258 (use_eigen_decomp=True, is_sym=True branch) +
                                  the pad/cut of dataset/qm8.py:264-291 -> HIP `lnz_lanczos_ritz`
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


def get_graph_laplacian_eigs_batched(L_simple, n_nodes, k):
    """L_simple [B,N,N] (e.g. `L[..., 0]`), n_nodes [B] -> (D [B,k], V [B,N,k]) ordered by
    descending |eigenvalue| like `np.argsort(-|eigs|, kind='mergesort')` (:218-223)."""
    from .. import ops
    return ops.lanczos_ritz(L_simple, n_nodes, k)
