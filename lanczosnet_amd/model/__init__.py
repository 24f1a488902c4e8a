from .lanczos_net import LanczosNet, LanczosNetGeneral  # noqa: F401

__all__ = ['LanczosNet', 'LanczosNetGeneral']
