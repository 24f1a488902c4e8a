from .qm8 import collate_adjacency, collate_preprocessed  # noqa: F401
from .packed import PackedQM8, write_packed, edges_from_dense, edges_from_laplacians  # noqa: F401
