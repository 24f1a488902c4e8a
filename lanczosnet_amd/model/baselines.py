"""Dense message-passing baselines on the same fused kernel.

The reference's GCN (`model/gcn.py:17-113`) and DCNN (`model/dcnn.py:11-121`) are LanczosNet's conv
layer without spectral channels: `X' = relu(Linear(cat_c(M_c X)))` with `M_c` the per-bond-type
Laplacians (GCN) plus powers `L_0^k` of the simple-graph Laplacian (DCNN's diffusion scales — the
kernel's short-diffusion channels), and the same gated masked-mean head.  Same class names,
constructor, `forward(node_feat, L, label=None, mask=None)`, `state_dict` keys and init order as the
reference, so `runner/qm8_runner.py` picks them up through `from model import *` unchanged.
"""
import torch

from .lanczos_net import _LanczosNetBase

__all__ = ['GCN', 'DCNN']


class _NoSpectrum(_LanczosNetBase):
    """No Ritz pairs in the signature: the kernel gets an empty (K = 1, all-zero) spectrum."""

    def forward(self, node_feat, L, label=None, mask=None):
        B, N = L.shape[0], L.shape[1]
        D = torch.zeros((B, 1), dtype=torch.float32, device=L.device)
        V = torch.zeros((B, N, 1), dtype=torch.float32, device=L.device)
        return super().forward(node_feat, L, D, V, label=label, mask=mask)


class GCN(_NoSpectrum):
    """`model/gcn.py`: channels = the E+1 Laplacians of `L[..., e]` (:88-91)."""

    def _diffusion_conf(self, m):
        return [], [], 1, 'None'


class DCNN(_NoSpectrum):
    """`model/dcnn.py`: channels = the E+1 Laplacians, THEN `L_0^k X` for k in `diffusion_dist`
    (:82-97) — the kernel computes the powers first, so the weight's column blocks are permuted."""

    def _diffusion_conf(self, m):
        return list(m.diffusion_dist), [], 1, 'None'

    def _channel_order(self):
        S, E1 = self.num_scale_short, self.num_edgetype + 1
        return [E1 + c for c in range(S)] + list(range(E1))
