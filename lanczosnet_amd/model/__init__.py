from .lanczos_net import LanczosNet, LanczosNetGeneral, AdaLanczosNet  # noqa: F401
from .baselines import GCN, DCNN, ChebyNet  # noqa: F401

__all__ = ['LanczosNet', 'LanczosNetGeneral', 'AdaLanczosNet', 'GCN', 'DCNN', 'ChebyNet']
