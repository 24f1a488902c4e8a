from .qm8 import QM8Data, collate_adjacency, collate_preprocessed  # noqa: F401
from .packed import (PackedQM8, PackedQM8Data, write_packed, edges_from_dense,  # noqa: F401
                     edges_from_laplacians)
