"""The training step of the reference runner as a replayed HIP graph.

`QM8Runner.train` (runner/qm8_runner.py:189-250) runs, per batch,
`optimizer.zero_grad(); _, loss = model(...); loss.backward(); optimizer.step()`.  On the HIP
module that is ~6 ms of GPU work at B = 1024 issued through ~400 Python-level operations (kernel
launches through the dispatcher, small torch ops of the backward, the optimizer's foreach kernels): the
eager step is bound by the host.  `GraphedTrainStep` captures exactly that
sequence — the fused forward (activations stored), the HIP input-gradient / message /
gain-gradient kernels, the library GEMMs, the parameter re-packing and the optimizer update —
into one HIP graph per batch shape and replays it: one launch per step, no Python on the critical
path, same kernels, same arithmetic, same parameter trajectory.

Nothing in the captured region may touch the host, so the module's backward switches to its
static-shape variant while capturing (message matrix sized by the padded row count, tail masked
on the device; `_LanczosNetFusedFunction`).  The optimizer must keep its step counters on the
device: build it with `capturable=True` (`make_adam` does).

    step = GraphedTrainStep(model, make_adam(model.parameters(), lr=1e-4))
    for data in loader:                      # device-resident batches (PackedQM8Data, or .cuda())
        loss = step(data['node_feat'], data['L'], data['D'], data['V'], data['label'],
                    data['node_mask'])       # 0-dim tensor, valid until the next call
"""
import torch

__all__ = ['GraphedTrainStep', 'make_adam']


def make_adam(params, lr=1e-4, weight_decay=0.0, **kw):
    """torch.optim.Adam as the runner builds it (runner/qm8_runner.py:72-74), with the step
    counters on the device so that `optimizer.step()` can be captured."""
    params = list(params)
    # lr as a device tensor: an LR scheduler then changes it in place and the captured step sees it
    lr_t = torch.tensor(float(lr), dtype=torch.float32, device=params[0].device)
    # (not fused=True: with torch 2.10 / ROCm 7 the fused multi-tensor Adam replayed from a graph
    # leaves the eager trajectory after the first step — loss off by 1e-3 at step 2 in
    # tests/test_gpu_train_graph.py — while the foreach form follows it to rounding; eager loops can
    # pass fused=True to torch.optim.Adam themselves: 2.3 instead of 5.6 ms per step for
    # AdaLanczosNet's 351 M parameters)
    return torch.optim.Adam(params, lr=lr_t, weight_decay=weight_decay, capturable=True, **kw)


class GraphedTrainStep(object):
    """Callable `(node_feat, L, D, V, label, mask) -> loss` for LanczosNet-signature modules
    (`names=('node_feat', 'L', 'label', 'mask')`-style modules pass `D = V = None`).

    The first `warmup` calls for a new input shape run eagerly (they are ordinary training steps);
    the next call captures the step into a graph for that shape and every later call with the same
    shapes replays it.  QM8 batches differ in their padded node count N, so a handful of graphs
    (one per N) coexist; they share one memory pool.

    If the module was trained eagerly before, drop every reference to those steps' losses first
    (`loss = float(loss)`): a live autograd graph keeps the parameters' AccumulateGrad nodes bound
    to the stream they were created on, and they must be re-created on the capture stream.

    Learning-rate schedules stay with the caller: the reference runner steps its MultiStepLR once
    per EPOCH (runner/qm8_runner.py:190, milestones = `lr_decay_steps` in epochs), never per batch.
    With `make_adam` the learning rate is a device tensor that the scheduler updates in place, so
    a replayed graph sees the new value (tests/test_gpu_train_graph.py).

    warmup >= 1: the first step must run eagerly — it creates the optimizer state (Adam's moments
    and step counter); captured inside the graph, that initialisation would be replayed, resetting
    the state at every step."""

    def __init__(self, model, optimizer, warmup=2, scheduler=None):
        if scheduler is not None:
            # the reference steps its scheduler once per EPOCH (see above); silently ignoring the
            # argument would train with a learning rate that never decays
            raise TypeError('GraphedTrainStep(scheduler=...) is not supported: step the LR scheduler '
                            'in the epoch loop like runner/qm8_runner.py:190 does')
        for group in optimizer.param_groups:
            # optimizers with host-side step counters (Adam & co.) expose the flag; plain SGD has
            # no such state and captures as it is
            if 'capturable' in group and not group['capturable']:
                raise ValueError('GraphedTrainStep needs an optimizer built with capturable=True '
                                 '(see lanczosnet_amd.train.make_adam)')
        if int(warmup) < 1:
            raise ValueError('GraphedTrainStep: warmup must be >= 1 (the optimizer state is created '
                             'by the first eager step; capturing it would reset it on every replay)')
        self.model, self.optimizer, self.warmup = model, optimizer, int(warmup)
        # the module behind a DataParallel / DDP wrapper owns the plans and the start-vector buffer
        self._core = model.module if hasattr(model, 'module') and \
            isinstance(getattr(model, 'module'), torch.nn.Module) else model
        self._graphs = {}
        self._seen = {}
        self._q1 = {}
        self._pool = None
        # warm-up steps and captures share one side stream (autograd's AccumulateGrad nodes
        # remember the stream they were created on)
        self._stream = torch.cuda.Stream()

    @staticmethod
    def _key(tensors):
        return tuple((tuple(t.shape), t.dtype) if t is not None else None for t in tensors)

    def _eager(self, *inputs):
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            self.optimizer.zero_grad(set_to_none=True)
            loss = self._eager_body(*inputs)
        cur.wait_stream(self._stream)
        return loss

    def _capture(self, key, inputs):
        static = [t.clone() if t is not None else None for t in inputs]
        graph = torch.cuda.CUDAGraph()
        # gradients must be allocated inside the graph's pool: drop the eager ones first
        self.optimizer.zero_grad(set_to_none=True)
        if hasattr(self._core, 'invalidate_plan'):
            self._core.invalidate_plan()
        kw = {'pool': self._pool} if self._pool is not None else {}
        with torch.cuda.graph(graph, stream=self._stream, **kw):
            loss = self._eager_body(*static)
        if self._pool is None:
            self._pool = graph.pool()
        self._graphs[key] = (graph, static, loss)

    def _eager_body(self, node_feat, L, D, V, label, mask):
        # (no zero_grad inside: backward writes fresh .grad tensors in the pool at capture and the
        # replays overwrite the same memory)
        if D is None:
            _, loss = self.model(node_feat, L, label=label, mask=mask)
        else:
            _, loss = self.model(node_feat, L, D, V, label=label, mask=mask)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _refresh_q1(self, node_feat):
        """AdaLanczosNet draws its Lanczos start vector from the CPU generator in every forward
        (model/ada_lanczos_net.py:161).  A HIP graph cannot: the draw happens HERE, once per step
        like the reference's (same generator, same shape, same order), and is copied into the static
        device buffer the captured forward reads."""
        if not hasattr(self._core, '_draw_q1'):
            return
        B, N = int(node_feat.shape[0]), int(node_feat.shape[1])
        buf = self._q1.get((B, N))
        if buf is None:
            buf = self._q1[(B, N)] = torch.empty((B, N, 1), device=node_feat.device)
        buf.copy_(torch.randn(B, N, 1))   # (pageable source: the host side of the copy is synchronous)
        self._core._static_q1 = buf

    def __call__(self, node_feat, L, D, V, label, mask):
        inputs = (node_feat, L, D, V, label, mask)
        key = self._key(inputs)
        self._refresh_q1(node_feat)
        entry = self._graphs.get(key)
        if entry is None:
            n = self._seen.get(key, 0)
            if n < self.warmup:
                self._seen[key] = n + 1
                loss = self._eager(*inputs)
                self._after_step()
                return loss
            # capture records the step without running it; the replay below executes it
            torch.cuda.synchronize()
            self._capture(key, inputs)
            entry = self._graphs[key]
        graph, static, loss = entry
        for dst, src in zip(static, inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        graph.replay()
        self._after_step()
        return loss

    def _after_step(self):
        # the packed-parameter plans are keyed on tensor versions, which a replay does not bump:
        # whoever runs the module eagerly next (validation, another shape's warm-up) must re-pack
        if hasattr(self._core, 'invalidate_plan'):
            self._core.invalidate_plan()
        if hasattr(self._core, '_draw_q1'):
            self._core._static_q1 = None   # eager forwards draw their own start vectors again
