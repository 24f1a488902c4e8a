from .qm8 import collate_adjacency, collate_preprocessed  # noqa: F401
