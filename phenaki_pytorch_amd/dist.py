"""Batch-sharded sampling across the GPUs of one node (new component: the reference has no inference
parallelism).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
Nothing in `Phenaki.sample` couples batch rows, so rank r samples its own contiguous slice of the batch with
zero communication inside the 18-step loop, and the decoded videos are exchanged with ONE all-gather at the end
(per rank (B/R, 3, F, H, W) f32; B = 32, R = 8: 53.5 MB per rank -- on the xGMI full mesh each shard crosses each
link once).  For make_video the gather happens once after the final concat, priming stays rank-local.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batch(n_items, rank=None, world_size=None):
    """contiguous [lo, hi) slice of a batch of n_items for `rank`; the first (n % R) ranks get one extra row."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_batch(local, n_items, group=None, force=False):
    """concatenate per-rank (b_r, ...) tensors along dim 0 into (n_items, ...); ONE collective (all_gather of
    equal-sized shards; ragged tails are padded to the largest shard and trimmed).  force: issue the collective even in a one-rank
    group (diagnostics: runs the RCCL path on a single-GPU box)."""
    rank, ws = world()
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return local
    sizes = [shard_batch(n_items, r, ws) for r in range(ws)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < bmax:
        pad = torch.cat((local, local.new_zeros((bmax - local.shape[0], *local.shape[1:]))), dim=0)
    pad = pad.contiguous()
    out = torch.empty((ws * bmax, *local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, pad, group=group)
    if all(hi - lo == bmax for lo, hi in sizes):
        return out
    chunks = [out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(chunks, dim=0)


def _slice(x, lo, hi):
    if x is None:
        return None
    if isinstance(x, (list, tuple)):
        return type(x)(x[lo:hi])
    return x[lo:hi]


def sample_sharded(phenaki, *, num_frames, texts=None, prime_frames=None, batch_size=1, gather=True, _force_collective=False, **kwargs):
    """`Phenaki.sample` with the batch split over the ranks; every rank returns the full (B, C, F, H, W) video
    (gather=True) or only its shard."""
    if isinstance(texts, str):
        texts = [texts]
    n_items = len(texts) if texts is not None else batch_size
    rank, ws = world()
    lo, hi = shard_batch(n_items, rank, ws)
    assert hi > lo, f'rank {rank} received an empty shard: batch {n_items} < world size {ws}'
    local = phenaki.sample(num_frames=num_frames, texts=_slice(texts, lo, hi), prime_frames=_slice(prime_frames, lo, hi),
                           batch_size=hi - lo, **kwargs)
    return all_gather_batch(local, n_items, force=_force_collective) if gather else local


def make_video_sharded(phenaki, texts_per_item, num_frames, prime_lengths, make_video_fn=None, gather=True):
    """`make_video` for a batch: texts_per_item is a list (batch) of per-scene text lists.  Scenes are sampled
    rank-locally (scene s of every local item in one batched `sample`), the final video is gathered once."""
    from .phenaki import cast_tuple
    n_items = len(texts_per_item)
    rank, ws = world()
    lo, hi = shard_batch(n_items, rank, ws)
    assert hi > lo
    mine = texts_per_item[lo:hi]
    num_scenes = len(mine[0])
    assert all(len(t) == num_scenes for t in mine)
    nf = cast_tuple(num_frames, num_scenes)
    pl = (*cast_tuple(prime_lengths, num_scenes - 1), 0)
    prime, scenes = None, []
    for s in range(num_scenes):
        video = phenaki.sample(texts=[t[s] for t in mine], prime_frames=prime, num_frames=nf[s])
        scenes.append(video)
        prime = video[:, :, -pl[s]:]
    local = torch.cat(scenes, dim=2)
    return all_gather_batch(local, n_items) if gather else local


# ---------------------------------------------------------------------------------------------- training: gradient exchange
# The reference trains data-parallel through HF accelerate -> torch DDP (cvivit_trainer.py:241-249, phenaki_trainer.py:378-386:
# `accelerator.backward(loss)` all-reduces the gradients before `clip_grad_norm_` / `opt.step()`).  The MI355X build keeps the exchange an
# explicit step between `loss.backward()` (train.py) and `opt.step()` (optim.py) instead of DDP's per-parameter autograd hooks: average the
# .grad of the given parameters over the ranks in BUCKETS.  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
# a ring all-reduce is bound by one link, ~2 (R-1)/R x bytes / 153 GB/s, so buckets are sized for link efficiency (64 MB default:
# ~0.8 ms each at 8 ranks, far above RCCL's per-call latency) and kept few -- the 134 MB of the vocabulary head's weight gradient is
# 3 buckets.  Each bucket is one flat contiguous buffer (one collective per bucket, not per tensor), reduced in f32.

def bucket_plan(numels, bucket_elems):
    """greedy, order-preserving: lists of tensor indices whose sizes sum to <= bucket_elems (a larger tensor gets its own bucket)"""
    plan, cur, fill = [], [], 0
    for i, n in enumerate(numels):
        if cur and fill + n > bucket_elems:
            plan.append(cur)
            cur, fill = [], 0
        cur.append(i)
        fill += n
    if cur:
        plan.append(cur)
    return plan


def _group_size(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def broadcast_parameters(tensors, src=0, group=None, bucket_mb=64.0, force=False):
    """in place: every tensor in `tensors` (parameters AND buffers of the trained modules) <- rank `src`'s copy, in flat buckets (one
    broadcast per bucket and dtype).  What DDP does when it wraps a module (the reference trains through accelerate -> DDP,
    cvivit_trainer.py / phenaki_trainer.py): averaging gradients is data-parallel training only if every replica STARTS from the same weights.
    `src` is a rank of `group`.  Returns the number of collectives."""
    if _group_size(group) == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return 0
    src_global = dist.get_global_rank(group, src) if group is not None else src
    ts = [t for t in tensors if t is not None and t.numel()]
    n = 0
    for dtype in sorted({t.dtype for t in ts}, key=str):
        same = [t for t in ts if t.dtype == dtype]
        elems = max(1, int(bucket_mb * (1 << 20) / max(1, same[0].element_size())))
        for idxs in bucket_plan([t.numel() for t in same], elems):
            flat = torch.cat([same[i].detach().reshape(-1) for i in idxs])
            dist.broadcast(flat, src=src_global, group=group)
            off = 0
            with torch.no_grad():
                for i in idxs:
                    k = same[i].numel()
                    same[i].copy_(flat[off:off + k].view_as(same[i]))
                    off += k
            n += 1
    return n


def broadcast_module(module, src=0, group=None):
    """parameters and buffers of `module` <- rank `src`'s (call once before training; packed-weight caches are dropped)"""
    n = broadcast_parameters(list(module.parameters()) + list(module.buffers()), src=src, group=group)
    if n:
        from .attention import invalidate_packed
        invalidate_packed(module)
    return n


def all_reduce_gradients(params, bucket_mb=64.0, average=True, group=None, force=False):
    """in place: p.grad <- mean (or sum) over the ranks of `group` of p.grad, for every parameter in `params` that has a gradient.  Identical
    parameter order on every rank is the caller's contract (as with DDP).  Returns the number of collectives issued.  force: run the
    collectives even in a one-rank group (diagnostics)."""
    ws = _group_size(group)
    grads = [p.grad for p in params if p.grad is not None and p.grad.numel()]
    if ws <= 0 or (ws == 1 and not (force and dist.is_available() and dist.is_initialized())) or not grads:
        return 0                                    # (ws < 0: this rank is not a member of `group`)
    bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
    plan = bucket_plan([g.numel() for g in grads], bucket_elems)
    for idxs in plan:
        flat = torch.cat([grads[i].reshape(-1).float() for i in idxs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= ws
        off = 0
        for i in idxs:
            n = grads[i].numel()
            grads[i].copy_(flat[off:off + n].view_as(grads[i]))
            off += n
    return len(plan)


class GradientReducer:
    """`all_reduce_gradients` OVERLAPPED with the backward pass, on a pre-allocated **flat gradient arena**: every bucket is ONE contiguous f32
    buffer and each parameter's `.grad` is a view into it, so a bucket's all-reduce runs IN PLACE on the gradients themselves -- no flatten copy
    before the collective, no copy back after it (torch DDP's gradient_as_bucket_view).  A bucket's all-reduce is started (async, on RCCL's own
    stream) the moment the last of its parameters has received its gradient, while the remaining backward kernels keep the compute stream
    busy; `finish()` -- called between `loss.backward()` and `opt.step()` -- launches what backward left incomplete (parameters that got no
    gradient on this rank this step contribute zeros and receive the reduced value as their `.grad`, so every replica steps them alike), waits,
    and averages.  Buckets are fixed at construction in REVERSE parameter
    order (the order backward produces gradients in), so every rank issues the same collectives in the same order.
    With `zero_grad(set_to_none=False)` the views persist and backward accumulates straight into the arena (zero copies per step); after
    `zero_grad(set_to_none=True)` autograd hands each parameter a fresh tensor, which the hook moves into its view (one copy, half of what
    flatten + copy-back cost).  At construction the parameters (and `buffers=`) are broadcast from rank 0 of the group, as DDP does at wrap time:
    the replicas must start from the same weights for the averaged gradient to be the data-parallel one.

        reducer = GradientReducer(params)            # once (collective: every rank must construct it)
        loss.backward(); reducer.finish(); opt.step()
        with reducer.no_sync(): loss.backward()      # gradient accumulation: no collectives for this backward
    """

    def __init__(self, params, bucket_mb=64.0, average=True, group=None, force=False, broadcast=True, buffers=()):
        """force: issue the collectives even in a one-rank group (diagnostics: the RCCL path on a single-GPU box)"""
        self.params = [p for p in params if p.requires_grad and p.numel()]
        self.average, self.group = average, group
        self.force = bool(force)
        order = list(range(len(self.params)))[::-1]
        bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = [[order[j] for j in idxs] for idxs in bucket_plan([self.params[i].numel() for i in order], bucket_elems)]
        self._bucket_of = {i: b for b, idxs in enumerate(self.buckets) for i in idxs}
        # the arena: one flat f32 buffer per bucket (allocated on the parameters' device), one view per parameter
        self.arena, self._views = [], [None] * len(self.params)
        for idxs in self.buckets:
            dev = self.params[idxs[0]].device
            assert all(self.params[i].device == dev and self.params[i].dtype == torch.float32 for i in idxs), 'GradientReducer: f32 parameters on one device'
            flat = torch.zeros(sum(self.params[i].numel() for i in idxs), device=dev, dtype=torch.float32)
            off = 0
            for i in idxs:
                k = self.params[i].numel()
                self._views[i] = flat[off:off + k].view_as(self.params[i])
                off += k
            self.arena.append(flat)
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._pending = []
        self._absent = []
        self._enabled = True
        self.collectives = 0
        self.broadcasts = broadcast_parameters(self.params + list(buffers), group=group) if broadcast and self._active() and not force else 0
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _adopt(self, i):
        """make parameter i's gradient the arena view (moving a freshly allocated gradient in); returns False when it has none"""
        p, v = self.params[i], self._views[i]
        g = p.grad
        if g is None:
            return False
        if g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
            v.copy_(g)
            p.grad = v
        return True

    def _make_hook(self, i):
        def hook(_param):
            if not self._enabled or not self._active():
                return
            b = self._bucket_of[i]
            if self._launched[b]:
                raise RuntimeError('GradientReducer: a second backward reached a bucket that is already being reduced -- call finish() after every '
                                   'backward (or run the accumulation backwards under no_sync())')
            self._adopt(i)
            self._ready[b] += 1
            if self._ready[b] == len(self.buckets[b]) and not self._launched[b]:
                self._launch(b)
        return hook

    def _active(self):
        return _group_size(self.group) > 1 or (self.force and dist.is_available() and dist.is_initialized())

    def _launch(self, b):
        self._launched[b] = True
        have = [self._adopt(i) for i in self.buckets[b]]
        for i, ok in zip(self.buckets[b], have):
            if not ok:
                self._views[i].zero_()              # no gradient on this rank this step: contributes zeros (another rank may have one)
                self._absent.append(i)
        work = dist.all_reduce(self.arena[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((b, work))
        self.collectives += 1

    def finish(self):
        """wait for the collectives in flight, reduce the incomplete buckets, average in place; returns #collectives"""
        if self._active() and self._enabled:
            ws = _group_size(self.group)
            for b in range(len(self.buckets)):                  # in bucket order on every rank
                if not self._launched[b]:
                    self._launch(b)
            for b, work in self._pending:
                work.wait()
                if self.average and ws > 1:
                    self.arena[b].div_(ws)
            # a parameter that got no gradient HERE may have got one on another rank: every rank must step it with the same averaged value
            # (an optimizer skips `.grad is None`), so the reduced view becomes its gradient.  A parameter NO rank produced a gradient for keeps
            # .grad = None -- DDP all-reduces a used-parameter bitmap for exactly this (reducer.cpp, find_unused_parameters) -- otherwise AdamW
            # would decay it and create optimizer state under data parallelism only.  One small int32 all-reduce of the flags;
            # only issued when some rank has an absent parameter... which no rank can know locally, so it is issued every step (a few KB).
            used = torch.ones(len(self.params), device=self.arena[0].device, dtype=torch.int32)
            if self._absent:
                used[torch.tensor(self._absent, device=used.device, dtype=torch.long)] = 0
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)      # (not counted in `collectives`: those are the gradient buckets)
            used = used.tolist()
            for i in self._absent:
                if used[i]:
                    self.params[i].grad = self._views[i]
                else:
                    self.params[i].grad = None
        self._absent.clear()
        n = self.collectives
        self._pending.clear()
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self.collectives = 0
        return n

    def no_sync(self):
        reducer = self

        class _NoSync:
            def __enter__(self_inner):
                reducer._enabled = False

            def __exit__(self_inner, *exc):
                reducer._enabled = True
        return _NoSync()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
