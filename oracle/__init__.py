"""CPU oracle for the LanczosNet hot path.  TEST INFRASTRUCTURE ONLY.

Everything in this package is a numpy restatement of the reference algorithm
(lrjconan/LanczosNetwork), each function citing the reference file:line it
follows.  It exists to check the HIP product path; it is never the thing that
is shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under
`lanczosnet_amd/` imports it.

Parity pin: the restatements are checked against outputs of the *unmodified*
reference classes/functions imported from /root/reference in the build
container; those outputs are committed as fixtures under `tests/golden/`
together with the generating script `tests/golden/make_golden.py`
(see tests/test_oracle_golden.py).
"""
from .laplacian import laplacian_l4, laplacian_multi_l4, get_laplacian  # noqa: F401
from .eigs import (graph_laplacian_eigs, collate_eigs, spectral_projector,  # noqa: F401
                   degenerate_cut, CUT_GAP)
from .lanczos_net import (lanczos_net_forward, spectral_gains, make_lanczosnet_params,  # noqa: F401
                          lanczosnet_dims, DEFAULT_QM8_CFG)
from .ada_lanczos import (ada_lanczos_layer, ada_graph_laplacian, ada_t_powers,  # noqa: F401
                          ada_spectral_filter_dd, ada_lanczos_net_forward, make_ada_params)
from .lanczos_net_torch import lanczos_net_forward_torch  # noqa: F401
from .lanczos_kstep import lanczos_kstep_fp64  # noqa: F401
from .collate import collate_packed, dense_from_edges  # noqa: F401
from .segment_sum import (unsorted_segment_sum_forward_gpu_semantics,  # noqa: F401
                          unsorted_segment_sum_backward_gpu_semantics,
                          unsorted_segment_sum_forward_cpu_semantics)
from .baselines import baseline_forward  # noqa: F401
