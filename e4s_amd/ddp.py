"""Data-parallel gradient averaging for the joint train step (BASELINE.json configs[4]; the reference wraps Net3 in
torch DDP, src/training/coach.py:46-85).  One process per GPU; after backward every rank holds the full gradient of its
shard of the batch, and the ranks average them with bucketed all-reduces (RCCL over xGMI with backend "nccl"; gloo in the
CPU tests).  Buckets are flat fp32 buffers of ~64 MB: the 164 M-parameter Net3 (644 MB of gradients) goes out as ~10
collectives whose ring time is bound by one xGMI link each, issued asynchronously in reverse parameter
order -- the order backward produces them -- and waited for together."""
import torch
import torch.distributed as dist


class GradAverager:
    def __init__(self, params, group=None, bucket_mb=64):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        cap = int(bucket_mb * (1 << 20) // 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(self.params):                  # backward finishes the LAST layers first
            if cur and n + p.numel() > cap:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def average(self):
        """grad <- mean over ranks of grad, for every parameter (a missing grad counts as zeros on that rank)."""
        if self.world == 1:
            return
        works = []
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            dev = bucket[0].device
            if self._flat[i] is None or self._flat[i].device != dev:
                self._flat[i] = torch.empty(n, device=dev, dtype=torch.float32)
            flat = self._flat[i]
            o = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[o:o + k].zero_()
                else:
                    flat[o:o + k].copy_(p.grad.reshape(-1))
                o += k
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for i, bucket in enumerate(self.buckets):
            works[i].wait()
            flat = self._flat[i]
            flat.div_(self.world)
            o = 0
            for p in bucket:
                k = p.numel()
                g = flat[o:o + k].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                o += k
