from .lanczos_net import LanczosNet, LanczosNetGeneral, AdaLanczosNet  # noqa: F401

__all__ = ['LanczosNet', 'LanczosNetGeneral', 'AdaLanczosNet']
