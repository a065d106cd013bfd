"""Data-parallel gradient exchange for the training slice: one process per GPU, RCCL over xGMI (torch.distributed backend
"nccl" on ROCm; gloo on CPU for the tests).  The reference trains through Lightning's DDP strategy (train.py:74-83).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one link's bandwidth and by
per-collective latency: InsMOS has ~6.5 M parameters (26 MB fp32) in ~250 tensors, i.e. hundreds of latency-bound tiny
collectives if reduced tensor by tensor.  Gradients are therefore packed into a few flat fp32 buckets (default 8 MB:
large enough to run at link bandwidth, small enough that the first bucket can start while later gradients are still
being produced) and each bucket is all-reduced once; `reduce()` returns after the last bucket is unpacked.
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=8 << 20, group=None):
        """params: dict name -> tensor (requires_grad) or list of tensors; the bucket layout is fixed at construction
        (same order on every rank: sorted names / list order)."""
        self.items = sorted(params.items()) if isinstance(params, dict) else list(enumerate(params))
        self.group = group
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

    def _close(self, cur, total):
        dev = cur[0][0].device
        self.buckets.append((torch.zeros(total, dtype=torch.float32, device=dev), cur))

    def reduce(self, average=True):
        """All-reduce every parameter's .grad (missing grads count as zero) in place; returns the number of collectives."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        works = []
        for flat, members in self.buckets:
            for p, off, n in members:
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
            if world > 1:
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
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
        return len(self.buckets)
