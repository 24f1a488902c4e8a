"""Multi-GPU sharding of the LanczosNet forward: one process per GPU, `torch.distributed`
(backend `nccl` = RCCL over xGMI on ROCm; `gloo` in the CPU tests).

Molecules are independent (every tensor of the collate output has the batch on dim 0 and no op
mixes batch entries: dataset/qm8.py:71-90,262,289-291; model/lanczos_net.py:190-194), so the
path shards with NO data-path collective: each rank runs the HIP pipeline on a contiguous
dim-0 slice.  The only exchange is the result: one all-gather of the per-shard scores
(64 KiB per rank at 1024 x 16 fp32 — latency bound, far below the 7 x ~153 GB/s xGMI links) and,
when labels are given, one all-reduce of (sum of squared errors, count) so the loss is the
size-weighted mean the single-process `MSELoss` would have produced (the reference's
`nn.DataParallel`, runner/qm8_runner.py:62, gathers per-replica means instead, which is only
correct for equal shards).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous dim-0 split of n items over `world` ranks; the first n % world ranks get one
    extra item (same rule as torch.tensor_split)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_batch(batch, rank=None, world=None):
    """Slice every tensor of a collated batch (dict or tuple/list of tensors) on dim 0."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world

    def cut(t):
        if t is None:
            return None
        lo, hi = shard_bounds(t.shape[0], rank, world)
        return t[lo:hi]

    if isinstance(batch, dict):
        return {k: cut(v) for k, v in batch.items()}
    return type(batch)(cut(v) for v in batch)


def all_gather_scores(local_score, n_global, group=None):
    """Gather the per-shard scores [b_r, P] into the full [n_global, P] on every rank.
    Shards may be uneven (sizes follow shard_bounds): shorter shards are zero-padded for the
    collective and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_score
    sizes = [shard_bounds(n_global, r, world)[1] - shard_bounds(n_global, r, world)[0]
             for r in range(world)]
    bmax = max(sizes)
    P = local_score.shape[1]
    send = local_score
    if local_score.shape[0] != bmax:
        send = torch.zeros((bmax, P), dtype=local_score.dtype, device=local_score.device)
        send[:local_score.shape[0]] = local_score
    out = torch.empty((world * bmax, P), dtype=local_score.dtype, device=local_score.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax:r * bmax + sizes[r]] for r in range(world)], dim=0)


class AsyncScoreGather:
    """The per-step all-gather of shard scores for a STREAM of batches, kept off the critical path.

    `submit(score_k)` issues the collective asynchronously (on RCCL's own stream, ordered after
    the kernels that produced `score_k`) into one of `depth` result buffers and returns at once,
    so the compute stream goes on to the next batch instead of waiting for a latency-bound 64 KiB
    exchange — and a slow rank delays its peers only when it falls `depth` steps behind, not at
    every step.  `result(ticket)` / `drain()` wait for the gathers.  Equal shards (the bench and
    the runner's full batches); `all_gather_scores` handles a ragged tail batch."""

    def __init__(self, shard_rows, width, device, dtype=torch.float32, depth=2, group=None,
                 force_collective=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: issue the collective even in a one-rank group (a legal RCCL
        # communicator): how the exchange is exercised on its real backend on a one-GPU box
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.bufs = [torch.empty((self.world * shard_rows, width), dtype=dtype, device=device)
                     for _ in range(depth)]
        self.work = [None] * depth
        self.issued = 0

    def submit(self, local_score):
        i = self.issued % len(self.bufs)
        if self.work[i] is not None:
            self.work[i].wait()  # this buffer's previous gather (depth steps ago) has landed
            self.work[i] = None
        if not self.collective:
            self.bufs[i].copy_(local_score)
        else:
            self.work[i] = dist.all_gather_into_tensor(self.bufs[i], local_score.contiguous(),
                                                       group=self.group, async_op=True)
        self.issued += 1
        return self.issued - 1

    def result(self, ticket):
        """Full [world * shard_rows, width] scores of submit() number `ticket` (one of the last
        `depth` submissions)."""
        assert self.issued - len(self.bufs) <= ticket < self.issued, 'result buffer already reused'
        i = ticket % len(self.bufs)
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.bufs[i]

    def drain(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None


def global_mse(local_score, local_label, group=None):
    """Size-weighted global mean squared error == MSELoss over the unsharded batch
    (model/lanczos_net.py:66,197).  One all-reduce of two scalars."""
    sse = ((local_score.double() - local_label.double()) ** 2).sum()
    acc = torch.stack([sse, torch.tensor(float(local_score.numel()), dtype=torch.float64,
                                         device=local_score.device)])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return (acc[0] / acc[1]).to(local_score.dtype)


def forward_sharded(forward_fn, batch, n_global, label_key='label', group=None):
    """Run `forward_fn(shard) -> score [b_r, P]` on this rank's slice of `batch` (already
    sliced or sliced here when it still has n_global rows) and return (full_score, loss|None)."""
    first = next(v for v in (batch.values() if isinstance(batch, dict) else batch)
                 if v is not None)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    # slice by the GROUP's rank / size: with a sub-group the default-group rank would cut the
    # wrong rows
    shard = shard_batch(batch, dist.get_rank(group), dist.get_world_size(group)) \
        if multi and first.shape[0] == n_global else batch
    local = forward_fn(shard)
    full = all_gather_scores(local, n_global, group)
    loss = None
    if isinstance(shard, dict) and shard.get(label_key) is not None:
        loss = global_mse(local, shard[label_key], group)
    return full, loss


def all_reduce_gradients(params, local_count, group=None, bucket_bytes=256 << 20,
                         force_collective=False):
    """Data-parallel gradient exchange for training (runner/qm8_runner.py:216-248 under
    `nn.DataParallel`, :62): every rank holds d(mean loss over ITS shard)/dθ; the gradient of the
    mean over the whole batch is the shard-size-weighted average.  Gradients travel as flat fp32
    buckets of at most `bucket_bytes` (LanczosNet: 7.4 MB, one bucket; AdaLanczosNet: ~1.4 GB,
    six) — few large ring all-reduces, which on point-to-point xGMI are bound by one link, instead
    of one latency-bound collective per parameter; the buckets are issued asynchronously and
    collected in order, so the copy-back of bucket i overlaps the wire time of bucket i+1.

    params: iterable of parameters whose `.grad` is replaced in place; local_count: number of
    molecules (rows of the loss mean) this rank contributed.  The bucket layout is a function of
    the parameter list alone (every `requires_grad` parameter; a missing `.grad` travels as
    zeros): ranks whose shards left different parameters without a gradient still issue
    identically shaped collectives.  A parameter that NO rank has a gradient for keeps
    `.grad = None` afterwards, as in the single-process run (torch optimizers skip such
    parameters: no weight decay, no momentum update, no step count) — a has-gradient mask rides
    with the shard counts.  With one process (no group, or world size 1) the function leaves
    every gradient as it is — unless `force_collective` asks for the exchange in a one-rank group
    (the RCCL test on a one-GPU box)."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    multi = dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
    if not multi:
        return
    dev = params[0].device
    had = torch.tensor([1.0 if p.grad is not None else 0.0 for p in params], dtype=torch.float32,
                       device=dev)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    cnt = torch.tensor([float(local_count)], dtype=torch.float32, device=dev)
    head = torch.cat([cnt, had])
    dist.all_reduce(head, op=dist.ReduceOp.SUM, group=group)
    total, had_any = head[:1], (head[1:] > 0).tolist()
    # greedy buckets in parameter order
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.grad.numel() * 4
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    buckets.append(cur)
    pending = []
    for bk in buckets:
        flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in bk]) * cnt
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True) if multi else None
        pending.append((bk, flat, work))
    for bk, flat, work in pending:
        if work is not None:
            work.wait()
        flat = flat / total
        off = 0
        for p in bk:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    for p, any_rank in zip(params, had_any):
        if not any_rank:
            p.grad = None
