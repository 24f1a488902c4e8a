"""Oracle (test infrastructure): the same restatement of `LanczosNet.forward` as
oracle/lanczos_net.py, on torch CPU tensors instead of numpy arrays.

Why a second copy: bench.py's `cpu_baseline` leg times "the reference's CPU path" on the GPU box,
where the reference tree does not exist.  The reference is a torch program (batched `torch.bmm`
on a multi-threaded CPU backend); the numpy restatement reaches only ~0.45x of its rate on the
same host (numpy's batched matmul walks the batch serially — profiles/cpu_port_vs_reference.json).
This one issues the reference's own operator sequence through the same library, so its rate IS
the reference's to within a few per cent (same file), while every line still cites what it follows:

  model/lanczos_net.py:95-123   _get_spectral_filters (MLP branch :110-117, power branch :118-121)
  model/lanczos_net.py:146-199  forward: D powers, embedding, conv block (short :164-169, long
                                :172-174, edge :177-178, cat + Linear + ReLU :180-182), head
                                :185-188, masked mean per molecule :190-194
  model/lanczos_net_general.py:156  state = node_feat for the general variant
"""
import numpy as np
import torch


def lanczos_net_forward_torch(P, cfg, node_feat, L, D, V, mask, general=False):
  """Same signature / result as oracle.lanczos_net_forward (float32).  P: dict name -> numpy."""
  T = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in P.items()}
  L = torch.as_tensor(np.asarray(L)).float()
  D = torch.as_tensor(np.asarray(D)).float()
  V = torch.as_tensor(np.asarray(V)).float()
  B, N = L.shape[0], L.shape[1]
  long_d, short_d = list(cfg['long_diffusion_dist']), list(cfg['short_diffusion_dist'])
  S, E1 = len(long_d), cfg['num_bond_type'] + 1
  lin = lambda x, name: torch.nn.functional.linear(x, T[name + '.weight'], T[name + '.bias'])  # noqa: E731
  with torch.no_grad():
    if general:
      state = torch.as_tensor(np.asarray(node_feat)).float()
    else:
      state = T['embedding.weight'][torch.as_tensor(np.asarray(node_feat)).long()]   # :154
    D_pow = [torch.pow(D, p) for p in long_d]                                            # :146-149
    Vt = V.transpose(1, 2)
    for tt in range(cfg['num_layer']):
      msg = []
      if short_d:                                                                        # :164-169
        tmp = state
        for ii in range(1, max(short_d) + 1):
          tmp = torch.bmm(L[:, :, :, 0], tmp)
          if ii in short_d:
            msg.append(tmp)
      if S > 0:
        if cfg['spectral_filter_kind'] == 'MLP':                                         # :110-117
          h = torch.stack(D_pow, dim=2).view(-1, S)
          pre = 'spectral_filter.%d.' % tt
          for ii in (0, 2, 4):
            h = torch.relu(lin(h, pre + str(ii)))
          G = lin(h, pre + '6').view(B, -1, S)
        else:
          G = torch.stack(D_pow, dim=2)                                                  # :118-121
        for s in range(S):                                                               # :172-174
          Ls = torch.bmm(V * G[:, :, s].unsqueeze(1), Vt)
          msg.append(torch.bmm(Ls, state))
      for e in range(E1):                                                                # :177-178
        msg.append(torch.bmm(L[:, :, :, e], state))
      m = torch.cat(msg, dim=2).view(B * N, -1)                                          # :180
      state = torch.relu(lin(m, 'filter.%d' % tt)).view(B, N, -1)                        # :181
    flat = state.view(B * N, -1)
    y = lin(flat, 'filter.%d' % cfg['num_layer'])                                        # :186
    att = torch.sigmoid(lin(flat, 'att_func.0'))                                         # :187
    y = (att * y).view(B, N, -1)
    mk = torch.as_tensor(np.asarray(mask)).bool()
    score = torch.stack([y[b][mk[b]].mean(dim=0) for b in range(B)])                     # :190-194
  return score.numpy()
