"""Data-parallel gradient exchange for the training slice: one process per GPU, RCCL over xGMI (torch.distributed backend
"nccl" on ROCm; gloo on CPU for the tests).  The reference trains through Lightning's DDP strategy (train.py:74-83).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one link's bandwidth and by
per-collective latency: InsMOS has ~6.5 M parameters (26 MB fp32) in ~250 tensors, i.e. hundreds of latency-bound tiny
collectives if reduced tensor by tensor.  Gradients are therefore packed into a few flat fp32 buckets (default 8 MB:
large enough to run at link bandwidth, small enough that the first bucket can start while later gradients are still
being produced) and each bucket is all-reduced once; `reduce()` returns after the last bucket is unpacked.

overlap=True (round 2): the exchange starts DURING backward.  Buckets are laid out in reverse parameter order (gradients
arrive roughly last layer first); a post-accumulate hook on every parameter copies its gradient into its bucket, and a bucket
is all-reduced (async) as soon as it is complete AND every earlier bucket has been launched -- the launch order is the bucket
order on every rank, whatever order the gradients arrive in and whether or not a rank produced all of them (a gradient that
never arrives counts as zero and its bucket goes out in `reduce()`), so ranks can never pair up different collectives.

Contract with overlap=True: exactly ONE backward() between two reduce() calls (no gradient accumulation, no retain_graph
second pass) -- a gradient that arrives for a bucket whose all-reduce is already in flight raises instead of racing with it;
call close() before building another reducer over the same parameters (the hooks are removed; a closed reducer raises).
"""

import torch
import torch.distributed as dist


class _Staged:
    """An async all-reduce on a host copy of a device bucket; wait() copies the result back."""

    def __init__(self, work, host, flat):
        self.work, self.host, self.flat = work, host, flat

    def wait(self):
        self.work.wait()
        self.flat.copy_(self.host)


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=8 << 20, group=None, overlap=False, force_collective=False):
        """force_collective: issue the all-reduce even in a world of one rank (the one-GPU RCCL self-check: the sum over one
        rank is the gradient itself, so the result is checkable bit for bit).
        params: dict name -> tensor (requires_grad) or list of tensors; the bucket layout is fixed at construction and the
        same on every rank: sorted names / list order, or -- overlap=True -- the reverse of the dict / list order (pass the
        parameters in forward order)."""
        if overlap:
            self.items = (list(params.items()) if isinstance(params, dict) else list(enumerate(params)))[::-1]
        else:
            self.items = sorted(params.items()) if isinstance(params, dict) else list(enumerate(params))
        self.group = group
        self.overlap = bool(overlap)
        self.force_collective = bool(force_collective)
        self.collectives_issued = 0      # all-reduces handed to the backend since construction
        self.buckets = []  # list of (flat buffer, [(tensor, offset, numel)])
        cur, off, cap = [], 0, max(1, bucket_bytes // 4)
        for _, p in self.items:
            n = p.numel()
            if cur and off + n > cap:
                self._close(cur, off)
                cur, off = [], 0
            cur.append((p, off, n))
            off += n
        if cur:
            self._close(cur, off)
        self._works, self._launched, self._seen = [], 0, [set() for _ in self.buckets]
        self.launched_in_backward = 0   # collectives of the last step that went out before reduce() was called
        self._hooks, self._closed = [], False
        if self.overlap:
            for bi, (_, members) in enumerate(self.buckets):
                for mi, (p, _, _) in enumerate(members):
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda t, bi=bi, mi=mi: self._on_grad(bi, mi)))

    def close(self):
        """Remove the backward hooks (overlap=True) -- required before another reducer is built over the same parameters, or
        both would launch collectives for every gradient.  Outstanding collectives are waited for."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for w in self._works:
            w.wait()
        self._works = []
        self._closed = True

    def _close(self, cur, total):
        dev = cur[0][0].device
        self.buckets.append((torch.zeros(total, dtype=torch.float32, device=dev), cur))

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _launch(self, bi):
        """Pack bucket bi (gradients the hooks did not see count as zero) and start its all-reduce."""
        flat, members = self.buckets[bi]
        for mi, (p, off, n) in enumerate(members):
            if mi in self._seen[bi]:
                continue                      # copied by its hook
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
        if self._world() > 1 or (self.force_collective and dist.is_initialized()):
            self.collectives_issued += 1
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":
                # two ranks on one GPU (the dry run RCCL refuses): gloo moves host memory -- stage the bucket through it
                host = flat.cpu()
                self._works.append(_Staged(dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True), host, flat))
            else:
                self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, bi, mi):
        if bi < self._launched or mi in self._seen[bi]:
            # the bucket's all-reduce is already in flight (or this gradient was already packed): a second backward before
            # reduce() -- gradient accumulation / retain_graph -- would write into a buffer RCCL is reading
            raise RuntimeError("BucketedGradReducer(overlap=True): a gradient arrived twice between two reduce() calls "
                               "(one backward per reduce(); use overlap=False to accumulate gradients over several; after a "
                               "backward that was aborted midway call reset() before the next one)")
        flat, members = self.buckets[bi]
        p, off, n = members[mi]
        flat[off:off + n].copy_(p.grad.reshape(-1))
        self._seen[bi].add(mi)
        while self._launched < len(self.buckets) and len(self._seen[self._launched]) == len(self.buckets[self._launched][1]):
            self._launch(self._launched)      # strictly in bucket order
            self._launched += 1

    def reset(self):
        """Forget a backward that did not reach reduce() (an exception or OOM midway): wait for the collectives already in
        flight (every rank must call this: they are collectives), drop what the hooks packed, start again at bucket 0."""
        for w in self._works:
            w.wait()
        self._works, self._launched, self._seen = [], 0, [set() for _ in self.buckets]

    def reduce(self, average=True):
        """All-reduce every parameter's .grad (missing grads count as zero) in place; returns the number of collectives.
        With overlap=True most of them are already in flight (or done) when this is called after backward()."""
        if self._closed:
            raise RuntimeError("BucketedGradReducer.reduce() after close()")
        world = self._world()
        self.launched_in_backward = self._launched
        while self._launched < len(self.buckets):
            self._launch(self._launched)
            self._launched += 1
        for w in self._works:
            w.wait()
        for flat, members in self.buckets:
            if average and world > 1:
                flat.div_(world)
            for p, off, n in members:
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
        self._works, self._launched, self._seen = [], 0, [set() for _ in self.buckets]
        return len(self.buckets)
