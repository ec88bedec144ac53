"""Image-parallel sharding of batched face swaps across the GPUs of one node (BASELINE.json configs[3]).

The path shards by independent images: every rank holds the full (replicated) Net3, computes a
contiguous slice of the batch with no exchange, and the only collective is one all-gather of the
[B/N, 3, H, W] outputs (RCCL over xGMI; `torch.distributed` backend "nccl" on ROCm).  The reference has
no counterpart (its only collective is DDP's gradient all-reduce, src/training/coach.py:46-85).
One process per GPU; on CPU the same code runs over gloo (tests/test_shard_gloo.py).

E4S_FORCE_COLLECTIVES=1 (or `force_collectives(True)`): a world of ONE rank normally skips every collective; with the switch on
and a process group initialised, the all-gathers (and ddp.GradAverager's all-reduces) are issued anyway -- numerically no-ops, but
the whole RCCL path (communicator init with device_id, all_gather_into_tensor, asynchronous work on RCCL's stream, stream capture
in thread_local mode) then executes on the one GPU that is reachable here: tests/test_gpu_nccl_world1.py,
`E4S_FORCE_COLLECTIVES=1 torchrun --nproc-per-node 1 bench.py --gpus 1`.
"""
import os

import torch
import torch.distributed as dist

_FORCE = os.environ.get("E4S_FORCE_COLLECTIVES", "0") == "1"


def force_collectives(flag=True):
    global _FORCE
    _FORCE = bool(flag)


def collectives_active(group=None):
    """True when collectives must be issued: more than one rank, or one rank with the force switch on."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _FORCE


def quiesce_before_capture(device=None):
    """Call right before a stream capture while a process group on backend "nccl" is alive.  The group's watchdog thread polls the end
    events of the EAGER collectives still in its list (every ~100 ms) and drops the completed ones; the capture that follows puts RCCL's
    stream into capture mode, and an event query that reaches the runtime at that moment is one more thing that can go wrong in a region
    where nothing may (a hipErrorCapturedEvent in the watchdog terminates the process).  Completing everything and giving the watchdog two
    of its periods leaves its list empty before the capture starts.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        if dist.get_backend() != "nccl":
            return
    except Exception:       # noqa: BLE001
        return
    import time
    torch.cuda.synchronize(device)
    time.sleep(0.25)


def shard_range(n, world, rank):
    """Contiguous balanced split of n items: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, world, rank):
    """Slice every [N, ...] tensor (or list of them) to this rank's shard."""
    n = None
    for t in tensors:
        if torch.is_tensor(t):
            n = t.shape[0]
            break
    lo, hi = shard_range(n, world, rank)

    def cut(t):
        if torch.is_tensor(t):
            return t[lo:hi].contiguous() if t.shape[0] == n else t
        if isinstance(t, (list, tuple)):
            return [cut(u) for u in t]
        return t
    return [cut(t) for t in tensors], (lo, hi)


def gather_outputs(local, n_total, group=None):
    """All-gather ragged shards [b_r, ...] into [n_total, ...] in rank order (every rank gets the result)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not collectives_active(group):
        return local
    rank = dist.get_rank(group)
    base, extra = divmod(n_total, world)
    if extra == 0:
        out = local.new_empty((n_total,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    cap = base + 1                                   # pad ragged shards to a common size, trim after
    padded = local.new_zeros((cap,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    buf = local.new_empty((world * cap,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        parts.append(buf[r * cap: r * cap + (hi - lo)])
    return torch.cat(parts, 0)


class OverlappedGather:
    """Asynchronous, double-buffered all-gather of per-step outputs (even shards).

    `submit(local)` copies this rank's [b, ...] shard into a staging slot and starts the all-gather on the collective's
    own stream (RCCL: its internal stream waits for the producer stream at enqueue), so step i's gather runs while step
    i+1 computes; only the gather that used the same slot `depth` steps ago is waited for first, because its staging
    and result buffers are about to be overwritten.  `drain()` waits for everything still in flight and returns the
    newest gathered [n_total, ...] tensor.  The producer may overwrite `local` as soon as `submit` returns on its
    stream (that is what a replayed HIP graph does with its static output buffer)."""

    def __init__(self, n_total, group=None, depth=2, pack=None):
        """pack: optional callable (local, out=None) -> packed tensor written straight into the staging slot -- e.g.
        e4s_amd.postproc.tensor2im, which turns the fp32 [b,3,H,W] shard into the uint8 [b,H,W,3] image the pipeline
        ends with (scripts/face_swap.py:276, torch_utils.tensor2im): the collective then moves a quarter of the bytes
        and the fp32 -> staging copy disappears (the pack kernel IS the copy)."""
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = collectives_active(group)
        if n_total % self.world:
            raise ValueError("OverlappedGather needs equal shards (use gather_outputs for ragged batches)")
        self.n_total, self.group, self.depth, self.pack = n_total, group, depth, pack
        self.stage, self.out, self.work = [None] * depth, [None] * depth, [None] * depth
        self.i = 0

    def submit(self, local):
        if not self.active:
            self.out[0] = self.pack(local) if self.pack is not None else local
            return
        k = self.i % self.depth
        if self.work[k] is not None:
            self.work[k].wait()                       # slot k is reused: its previous gather must be complete
        if self.pack is not None:
            if self.stage[k] is None:
                self.stage[k] = self.pack(local)      # first use of the slot allocates it
                self.out[k] = self.stage[k].new_empty((self.n_total,) + tuple(self.stage[k].shape[1:]))
            else:
                self.pack(local, out=self.stage[k])
        else:
            if self.stage[k] is None:
                self.stage[k] = torch.empty_like(local, memory_format=torch.contiguous_format)
                self.out[k] = local.new_empty((self.n_total,) + tuple(local.shape[1:]))
            self.stage[k].copy_(local)
        self.work[k] = dist.all_gather_into_tensor(self.out[k], self.stage[k], group=self.group, async_op=True)
        self.i += 1

    def drain(self):
        if not self.active:
            return self.out[0]
        for w in self.work:
            if w is not None:
                w.wait()
        self.work = [None] * self.depth
        return self.out[(self.i - 1) % self.depth] if self.i else None


def run_sharded(fn, tensors, group=None):
    """out = gather(fn(*shard(tensors))).  `fn` maps a shard of the inputs to [b_local, ...] outputs
    (e.g. functools.partial(e4s_amd.networks.face_swap_core, net))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = next(t.shape[0] for t in tensors if torch.is_tensor(t))
    shard, _ = shard_batch(tensors, world, rank)
    return gather_outputs(fn(*shard), n, group)
