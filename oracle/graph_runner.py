"""Oracle (test infrastructure): restatement of the reference's GRAPH experiment loop
(`runner/graph_runner.py:24-353`, `config/graph_lanczos_net.yaml`) — the caller of the hot path for
`LanczosNetGeneral` on the synthetic-graph dataset.  Same role and same pinning as
`oracle/qm8_runner.py` (which it extends): the parity tests drive the product module with this
class on the GPU box, and `tests/test_graph_runner_dropin.py` runs the unmodified
`runner.graph_runner.GraphRunner` and this class side by side on the reference's CPU model and
dataset, asserting bit-identical loss trajectories.

Differences from the QM8 loop that matter for the numbers (`runner/graph_runner.py`):
  __init__   :26-34   no meta data, no const_factor
  train      :36-264  identical structure; the device keys / call shape of `LanczosNetGeneral` are
                      (L, D, V) / (node_feat, L, D, V) (:106-108,124-131,196-198,218-225); the
                      validation metric is the MEAN SQUARED error over batch entries and targets
                      (:154-157), not a weighted MAE
  test       :266-353 multiplies by `self.const_factor` (:348), which GraphRunner never defines:
                      the reference's `test()` raises AttributeError.  Restated as it is.
"""
import numpy as np
import torch

from . import qm8_runner as _q

_EXTRA_KEYS = {'LanczosNetGeneral': ('L', 'D', 'V'), 'GraphSAGE': ('nn_idx', 'nonempty_mask'),
               'GPNN': ('L', 'L_cluster', 'L_cut')}
_INPUTS = {'AdaLanczosNet': ('node_feat', 'L'), 'LanczosNetGeneral': ('node_feat', 'L', 'D', 'V'),
           'GraphSAGE': ('node_feat', 'nn_idx', 'nonempty_mask'),
           'GPNN': ('node_feat', 'L', 'L_cluster', 'L_cut')}


class GraphRunner(_q.QM8Runner):

  def __init__(self, config, namespace):
    self.config = config
    self.namespace = dict(namespace)
    self.dataset_conf, self.model_conf = config.dataset, config.model
    self.train_conf, self.test_conf = config.train, config.test
    self.use_gpu, self.gpus = config.use_gpu, config.gpus

  def _to_gpu(self, data, extra):
    if not self.use_gpu:
      return
    data['node_feat'], data['node_mask'], data['label'] = _q.data_to_gpu(
        data['node_feat'], data['node_mask'], data['label'])
    keys = _EXTRA_KEYS.get(self.model_conf.name, ('L',))
    for k, v in zip(keys, _q.data_to_gpu(*[data[k] for k in keys])):
      data[k] = v

  def _call(self, model, data):
    keys = _INPUTS.get(self.model_conf.name, ('node_feat', 'L'))
    return model(*[data[k] for k in keys], label=data['label'], mask=data['node_mask'])

  def _mae(self, model, loader, extra):
    """The validation metric of train(): mean squared error (:154-157)."""
    errs = []
    for data in loader:
      self._to_gpu(data, extra)
      with torch.no_grad():
        pred, _ = self._call(model, data)
      errs.append((pred - data['label']).pow(2).cpu().numpy())
    return float(np.mean(np.concatenate(errs)))

  def test(self):
    """:266-353.  `curr_loss = (pred - label).pow(2).cpu().numpy() * self.const_factor` (:346-348):
    GraphRunner never defines `const_factor`, so the first batch raises AttributeError in the
    reference — and here."""
    loader = self._loader('test', self.test_conf.batch_size, False, self.test_conf.num_workers)
    model = self.namespace[self.model_conf.name](self.config)
    _q.load_model(model, self.test_conf.test_model)
    if self.use_gpu:
      model = torch.nn.DataParallel(model, device_ids=self.gpus).cuda()
    model.eval()
    test_loss = []
    for data in loader:
      if self.use_gpu:
        data['node_feat'], data['node_mask'], data['label'] = _q.data_to_gpu(
            data['node_feat'], data['node_mask'], data['label'])
        keys = {'LanczosNetGeneral': ('D', 'V')}.get(self.model_conf.name, ('L',))  # :298-299
        for k, v in zip(keys, _q.data_to_gpu(*[data[k] for k in keys])):
          data[k] = v
      with torch.no_grad():
        pred, _ = self._call(model, data)
      test_loss += [(pred - data['label']).pow(2).cpu().numpy() * self.const_factor]
    return float(np.mean(np.concatenate(test_loss)))
