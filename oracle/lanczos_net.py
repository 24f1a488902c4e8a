"""Oracle (test infrastructure): numpy restatement of `LanczosNet.forward`.

Follows reference `model/lanczos_net.py`:
  * spectral filters  :95-123  (`_get_spectral_filters`, MLP and non-MLP branch)
  * forward           :125-199 (D powers :146-149, embedding :154, conv block
                                :157-182, head :185-188, masked mean :190-194)
and `model/lanczos_net_general.py:127-201` (same, `state = node_feat` at :156).

The association of the contractions is the reference's: explicit
`L_s = (V * g_s) V^T`, then `bmm(L_s, state)`, concat, Linear.
"""
import numpy as np

DEFAULT_QM8_CFG = dict(  # config/qm8_lanczos_net.yaml:9-24
    num_atom=70, num_bond_type=6,
    short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
    num_eig_vec=20, spectral_filter_kind='MLP',
    input_dim=64, hidden_dim=[128] * 7, output_dim=16, num_layer=7)


def lanczosnet_dims(cfg):
  dim_list = [cfg['input_dim']] + list(cfg['hidden_dim']) + [cfg['output_dim']]
  n_chan = (len(cfg['short_diffusion_dist']) + len(cfg['long_diffusion_dist'])
            + cfg['num_bond_type'] + 1)
  return dim_list, n_chan


def make_lanczosnet_params(cfg, seed, general=False, bias_scale=0.1):
  """Deterministic (numpy RandomState) parameters keyed like the reference
  state_dict (model/lanczos_net.py:36-60).  Xavier-uniform weights
  like `_init_param` (:74-93) but NON-zero biases so bias handling is exercised."""
  rs = np.random.RandomState(seed)
  dim_list, n_chan = lanczosnet_dims(cfg)
  S = len(cfg['long_diffusion_dist'])
  P = {}

  def lin(name, fan_out, fan_in):
    a = np.sqrt(6.0 / (fan_in + fan_out))
    P[name + '.weight'] = rs.uniform(-a, a, size=(fan_out, fan_in)).astype(np.float32)
    P[name + '.bias'] = (bias_scale * rs.uniform(-1, 1, size=(fan_out,))).astype(np.float32)

  for tt in range(cfg['num_layer']):
    lin('filter.%d' % tt, dim_list[tt + 1], dim_list[tt] * n_chan)
  lin('filter.%d' % cfg['num_layer'], dim_list[-1], dim_list[-2])
  if not general:
    P['embedding.weight'] = rs.randn(cfg['num_atom'], cfg['input_dim']).astype(np.float32)
  if cfg['spectral_filter_kind'] == 'MLP' and S > 0:
    for tt in range(cfg['num_layer']):
      lin('spectral_filter.%d.0' % tt, 128, S)
      lin('spectral_filter.%d.2' % tt, 128, 128)
      lin('spectral_filter.%d.4' % tt, 128, 128)
      lin('spectral_filter.%d.6' % tt, S, 128)
  lin('att_func.0', 1, dim_list[-2])
  return P


def _linear(x, P, name):
  return x @ P[name + '.weight'].T + P[name + '.bias']


def spectral_gains(P, cfg, D, layer_idx, dtype=np.float32):
  """Per-eigenvalue filter gains G[B,K,S] (model/lanczos_net.py:110-113 for 'MLP',
  :118-121 otherwise)."""
  D = np.asarray(D, dtype=dtype)
  pows = [np.power(D, ii) for ii in cfg['long_diffusion_dist']]  # :146-149
  DD = np.stack(pows, axis=2)  # B x K x S
  if cfg['spectral_filter_kind'] != 'MLP':
    return DD
  B, K, S = DD.shape
  h = DD.reshape(B * K, S)
  pre = 'spectral_filter.%d.' % layer_idx
  Pc = {k: v.astype(dtype) for k, v in P.items() if k.startswith(pre)}
  for ii in (0, 2, 4):
    h = np.maximum(_linear(h, Pc, pre + str(ii)), 0)
  h = _linear(h, Pc, pre + '6')
  return h.reshape(B, K, S)


def lanczos_net_forward(P, cfg, node_feat, L, D, V, mask, dtype=np.float32,
                        general=False, return_state=False):
  """score[B,P] of LanczosNet / LanczosNetGeneral (eval mode, dropout p=0).

  node_feat: int [B,N] (or float [B,N,d] when general); L: [B,N,N,E+1];
  D: [B,K]; V: [B,N,K]; mask: [B,N] (bool / uint8)."""
  P = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
  L = np.asarray(L, dtype=dtype)
  V = np.asarray(V, dtype=dtype)
  B, N = L.shape[0], L.shape[1]
  S = len(cfg['long_diffusion_dist'])
  short = list(cfg['short_diffusion_dist'])
  E1 = cfg['num_bond_type'] + 1

  if general:
    state = np.asarray(node_feat, dtype=dtype)  # lanczos_net_general.py:156
  else:
    state = P['embedding.weight'][np.asarray(node_feat)]  # lanczos_net.py:154

  for tt in range(cfg['num_layer']):
    msg = []
    if S > 0:
      G = spectral_gains(P, cfg, D, tt, dtype)  # B x K x S
    if short:  # :164-169
      tmp = state
      for ii in range(1, max(short) + 1):
        tmp = L[:, :, :, 0] @ tmp
        if ii in short:
          msg.append(tmp)
    for s in range(S):  # :114-117 + :172-174
      Ls = (V * G[:, None, :, s]) @ V.transpose(0, 2, 1)
      msg.append(Ls @ state)
    for e in range(E1):  # :177-178
      msg.append(L[:, :, :, e] @ state)
    msg = np.concatenate(msg, axis=2).reshape(B * N, -1)  # :180
    state = np.maximum(_linear(msg, P, 'filter.%d' % tt), 0).reshape(B, N, -1)  # :181

  flat = state.reshape(B * N, -1)
  y = _linear(flat, P, 'filter.%d' % cfg['num_layer'])  # :186
  att = 1.0 / (1.0 + np.exp(-_linear(flat, P, 'att_func.0')))  # :187
  y = (att * y).reshape(B, N, -1)
  m = np.asarray(mask).astype(bool)
  score = np.stack([y[b, m[b], :].mean(axis=0) for b in range(B)])  # :190-194
  if return_state:
    return score.astype(dtype), state
  return score.astype(dtype)
