"""Dense message-passing baselines on the same fused kernel.

The reference's GCN (`model/gcn.py:17-113`), DCNN (`model/dcnn.py:11-121`) and ChebyNet
(`model/cheby_net.py:9-121`) are LanczosNet's conv layer without spectral channels: `X' = relu(Linear(cat_c(M_c X)))` with `M_c` the per-bond-type
Laplacians (GCN) plus powers `L_0^k` of the simple-graph Laplacian (DCNN's diffusion scales — the
kernel's short-diffusion channels) or Chebyshev polynomials of it (ChebyNet: a fixed linear
re-combination of the same power channels, folded into the weights), and the same gated
masked-mean head.  Same class names,
constructor, `forward(node_feat, L, label=None, mask=None)`, `state_dict` keys and init order as the
reference, so `runner/qm8_runner.py` picks them up through `from model import *` unchanged.
"""
import torch

from .lanczos_net import _LanczosNetBase

__all__ = ['GCN', 'DCNN', 'ChebyNet']


class _NoSpectrum(_LanczosNetBase):
    """No Ritz pairs in the signature: the kernel gets an empty (K = 1, all-zero) spectrum."""

    def forward(self, node_feat, L, label=None, mask=None):
        # inputs may arrive on the host (runner/qm8_runner.py:301-302 leaves L there): move them
        # to the module's device BEFORE anything is built from them
        dev = self._guard_forward(L, mask)
        t = self._to_module_device(dev, node_feat=node_feat, L=L, label=label, mask=mask)
        node_feat, L, label, mask = t['node_feat'], t['L'], t['label'], t['mask']
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


class ChebyNet(_NoSpectrum):
    """`model/cheby_net.py`: messages = `L[..., e] X` for e = 1..E, then the reference's Chebyshev
    states of the simple-graph Laplacian `L_0` (:88-93) — `S_0 = L_0 X`, `S_1 = 2 L_0 S_0 - X`,
    `S_k = 2 L_0 S_{k-1} - S_{k-2}` for k < polynomial_order — then `X` itself (:98).

    Every `S_k` is a polynomial `sum_p c[k][p] L_0^p X`, so
        sum_k S_k W_k^T + X W_I^T  =  sum_p L_0^p X (sum_k c[k][p] W_k + [p = 0] W_I)^T :
    the kernel runs its power channels (short diffusion p = 2..K, the edge-type channel of `L_0` as
    p = 1, and one identity channel appended to `L` as p = 0 — found and skipped by the
    identity-channel shortcut) with the re-combined weights; the map is linear, so the weight
    gradient goes back through its transpose.  Exact in real arithmetic; in fp32 the recombination
    (coefficients up to 20 at order 5) costs ~1e-6 relative, inside the 1e-5 parity bar."""

    def _diffusion_conf(self, m):
        self.polynomial_order = int(m.polynomial_order)
        if self.polynomial_order < 1:
            raise ValueError('polynomial_order must be >= 1')
        return list(range(2, self.polynomial_order + 1)), [], 1, 'None'

    def _override_dims(self):
        self.num_bond_type_ref = self.num_edgetype
        self.num_edgetype += 1  # + the identity channel appended in forward()

    def _recombination(self, device):
        """T [kernel channel, reference block]: kernel = (p = 2..K | L_0 (p = 1), edge 1..E,
        identity (p = 0)); reference = (edge 1..E | S_0..S_{K-1} | X)."""
        K, E = self.polynomial_order, self.num_bond_type_ref
        c = torch.zeros((K, K + 1), dtype=torch.float64)        # c[k][p]
        prev = torch.zeros(K + 1, dtype=torch.float64)          # S_{-1} = X
        prev[0] = 1.0
        c[0, 1] = 1.0                                           # S_0 = L X
        for k in range(1, K):
            c[k, 1:] = 2.0 * c[k - 1, :-1]
            c[k] -= c[k - 2] if k >= 2 else prev
        T = torch.zeros((K + E + 1, K + E + 1), dtype=torch.float64)
        chan_of_p = {p: p - 2 for p in range(2, K + 1)}         # short channels
        chan_of_p[1] = K - 1                                    # edge-type channel of L_0
        chan_of_p[0] = K + E                                    # identity channel
        for k in range(K):
            for p in range(K + 1):
                T[chan_of_p[p], E + k] += c[k, p]
        T[chan_of_p[0], E + K] += 1.0                           # the X block
        for e in range(E):
            T[K + e, e] = 1.0                                   # bond-type channels
        return T.to(device=device, dtype=torch.float32)

    def _mix_weight(self, t):
        w = self.filter[t].weight
        T = self._recombination(w.device)
        n = T.shape[0]
        d = w.shape[1] // n
        return torch.einsum('kr,ord->okd', T, w.view(w.shape[0], n, d)).reshape(w.shape[0], -1)

    def _to_reference_channel_order(self, dW):
        T = self._recombination(dW.device)
        n = T.shape[0]
        d = dW.shape[1] // n
        return torch.einsum('kr,okd->ord', T, dW.view(dW.shape[0], n, d)).reshape(dW.shape[0], -1)

    def forward(self, node_feat, L, label=None, mask=None):
        if mask is None:
            raise ValueError('forward needs `mask` (model/cheby_net.py:106)')
        dev = self._guard_forward(L, mask)
        t = self._to_module_device(dev, L=L, mask=mask)         # host L / device mask must not meet in cat
        L, mask = t['L'], t['mask']
        eye = torch.diag_embed((mask != 0).to(L.dtype))         # identity on the real nodes
        return super().forward(node_feat, torch.cat([L, eye.unsqueeze(3)], dim=3), label=label,
                               mask=mask)
