"""Data-parallel gradient averaging for the joint train step (BASELINE.json configs[4]; the reference wraps Net3 in
torch DDP, src/training/coach.py:46-85).  One process per GPU; every rank computes the gradient of its shard of the batch and
the ranks average them with bucketed all-reduces (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).

Buckets are flat fp32 buffers of ~64 MB in REVERSE parameter order -- the order backward produces gradients: the 134 M
trainable parameters of Net3 (536 MB of gradients; 644 MB with the generator) go out as ~9 collectives whose ring time is bound
by one xGMI link each (~7 ms for the lot), far below the ~100 ms backward they hide under.

Overlap with the backward (VERDICT r2 #1d): a bucket's all-reduce is launched the moment its last gradient exists --
  * from `register_post_accumulate_grad_hook` for parameters whose gradient autograd accumulates node by node (LocalMLPs,
    generator), and
  * from INSIDE the monolithic encoder backward (encoder_autograd.EncoderFn calls `notify_grad` per parameter as it walks the
    24 units in reverse), so the encoder's buckets leave while the earlier units are still being differentiated.
`arm()` before `loss.backward()`, `finish()` after it: finish() launches whatever has not fired (parameters without a gradient
count as zeros), waits, divides by the world size and writes the averaged gradients back.  `average()` is the non-overlapped
form (everything after backward); both give bit-identical results (same buffers, same collectives, same order).

Collectives are issued in BUCKET-INDEX order on every rank, whatever order the gradients arrive in: a bucket that fills early
waits (host side) for its predecessors, exactly as torch DDP orders its buckets -- ranks whose autograd engines finish parameters
in different orders, or where one rank gets no gradient for a parameter, still issue the same sequence of same-sized collectives.

Everything here is capturable: inside a HIP graph capture (train.TrainIteration.graphed_g_step) the staging copies, the RCCL
all-reduces (on RCCL's stream, forked from and joined to the capturing stream by the events `Work.wait()` records), the division and
the write-back become graph nodes, and a replay runs the whole data-parallel G step without touching the host.

Streams.  The staging copies of one bucket need not all be enqueued on the same HIP stream: autograd runs every backward node on
the stream its forward ran on, but an AccumulateGrad node (whose post-accumulate hook announces the LocalMLP / generator gradients)
keeps the stream it was CREATED on -- a node that outlives an iteration (torch keeps it alive as long as anything references last
step's graph) stays on that stream when a later step runs on a side stream or under a capture (torch warns: "The AccumulateGrad
node's stream does not match ...") -- while `notify_grad` from inside the monolithic encoder backward runs on the backward's own
stream.  A collective only waits for the stream it is issued from, so `_fire` first makes that stream wait for every OTHER stream
that staged into the bucket (an event recorded behind the last staging copy on it).  With one rank the in-place all-reduce moves
nothing and the omission cannot be seen; with N > 1 a peer would have received a half-staged bucket.

`force=True` keeps the collectives on a world of ONE rank (they are no-ops numerically): the RCCL path -- communicator
init, all_reduce, stream fork/join, capture -- then runs on a single GPU (tests/test_gpu_nccl_world1.py)."""
import warnings

import torch
import torch.distributed as dist

_ACTIVE = None        # the armed averager; a plain global: autograd runs backward nodes on its own device thread


def notify_grad(param, grad):
    """Called by autograd Functions that produce FINAL parameter gradients inside one big backward node."""
    a = _ACTIVE
    if a is not None:
        a.notify(param, grad)


class GradAverager:
    def __init__(self, params, group=None, bucket_mb=64, force=None, payload_dtype=torch.float32):
        """payload_dtype: what travels over xGMI.  torch.float32 (default; what torch DDP sends for the reference, which trains fp32) or
        torch.bfloat16 -- BASELINE.json configs[4] names bf16: the staging copies round the fp32 gradients to bf16 (RNE), the all-reduce
        sums bf16 (322 MB instead of 644 MB per G step with the generator), the averages are widened back into the fp32 .grad tensors;
        master weights, Adam moments and the local gradient accumulation stay fp32."""
        if payload_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("GradAverager: payload_dtype is torch.float32 or torch.bfloat16")
        self.payload_dtype = payload_dtype
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force is None:
            from . import shard
            force = shard._FORCE                         # E4S_FORCE_COLLECTIVES=1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        cap = int(bucket_mb * (1 << 20) // (4 if payload_dtype == torch.float32 else 2))
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
        self._where = {}
        for i, bucket in enumerate(self.buckets):
            o = 0
            for p in bucket:
                self._where[id(p)] = (i, o)
                o += p.numel()
        self._armed = False
        self._sent, self._ready, self._works = set(), [0] * len(self.buckets), [None] * len(self.buckets)
        self._next = 0                                   # buckets [0, _next) have been launched (index order on every rank)
        self.fired_during_backward = 0                   # diagnostics: buckets launched before finish()
        self._streams = [dict() for _ in self.buckets]   # per bucket: {stream id: stream} its staging copies were enqueued on
        self._fire_stream = None                         # the stream arm() ran on: every bucket of this backward is issued from it
        self.fire_log = []                               # diagnostics: (bucket, filled from another stream?, that stream capturing?)
        self.staging_streams_seen = [0] * len(self.buckets)   # diagnostics: most streams any one backward staged a bucket from
        if self.active:
            if hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
                for p in self.params:
                    p.register_post_accumulate_grad_hook(self._hook)
            else:
                warnings.warn("GradAverager: this torch has no register_post_accumulate_grad_hook; only gradients announced through "
                              "ddp.notify_grad overlap with the backward, the rest are reduced in finish()")

    # ---- plumbing ---------------------------------------------------------------------------------------------------
    def _buffer(self, i):
        bucket = self.buckets[i]
        dev = bucket[0].device
        if self._flat[i] is None or self._flat[i].device != dev:
            self._flat[i] = torch.empty(sum(p.numel() for p in bucket), device=dev, dtype=self.payload_dtype)
        return self._flat[i]

    def _staged_on_current_stream(self, i, dev):
        if dev.type == "cuda":
            st = torch.cuda.current_stream(dev)
            self._streams[i][st.cuda_stream] = st

    def _fire(self, i):
        flat = self._buffer(i)
        if flat.is_cuda:
            # Every bucket is issued from ONE stream -- the stream arm() was called on (the caller's; the capturing stream inside a
            # capture) -- whichever stream the hook that filled it happens to run on:
            #  * the collective is ordered behind the stream it is issued from only, so that stream first joins every stream that staged
            #    into the bucket (an event recorded behind the last staging copy on each);
            #  * torch decides per collective, from the CURRENT stream's capture status, whether the Work goes to the process group's
            #    watchdog thread; a Work whose end event was recorded inside a capture must never get there (the watchdog's hipEventQuery
            #    then fails with hipErrorCapturedEvent and takes the process down -- the failure round 4 saw once in ~8 runs and retried).
            #    Issued from the capturing stream, the status is always "active".
            cur = torch.cuda.current_stream(flat.device)
            fs = self._fire_stream if self._fire_stream is not None else cur
            self.fire_log.append((i, cur.cuda_stream != fs.cuda_stream, bool(torch.cuda.is_current_stream_capturing())))
            self.staging_streams_seen[i] = max(self.staging_streams_seen[i], len(self._streams[i] | {fs.cuda_stream: fs}))
            for sid, st in self._streams[i].items():
                if sid != fs.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    fs.wait_event(ev)
            self._streams[i] = {}
            with torch.cuda.stream(fs):
                self._works[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        self._works[i] = dist.all_reduce(self._buffer(i), op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p):
        if self._armed and id(p) not in self._sent and p.grad is not None:
            self.notify(p, p.grad)

    def notify(self, p, grad):
        """`grad` is the complete gradient of `p` for this backward: stage it, and launch the bucket when it is full."""
        if not self._armed or not self.active:
            return
        key = id(p)
        if key not in self._where:
            return
        if key in self._sent:
            raise RuntimeError("GradAverager: a parameter's gradient was announced twice in one backward")
        i, o = self._where[key]
        with torch.no_grad():
            self._buffer(i)[o:o + p.numel()].copy_(grad.detach().reshape(-1))
        self._staged_on_current_stream(i, grad.device)
        self._sent.add(key)
        self._ready[i] += 1
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._fire(self._next)                       # strictly in index order: a later bucket that filled first waits here
            self._next += 1
            self.fired_during_backward += 1

    # ---- API --------------------------------------------------------------------------------------------------------
    def arm(self):
        global _ACTIVE
        if not self.active:
            return
        self._armed = True
        self._sent, self._ready, self._works = set(), [0] * len(self.buckets), [None] * len(self.buckets)
        self._next = 0
        self.fired_during_backward = 0
        self._streams = [dict() for _ in self.buckets]
        dev = self.params[0].device if self.params else None
        self._fire_stream = torch.cuda.current_stream(dev) if (dev is not None and dev.type == "cuda") else None
        self.fire_log = []
        _ACTIVE = self

    def finish(self):
        """grad <- mean over ranks of grad, for every parameter (a missing grad counts as zeros on that rank)."""
        global _ACTIVE
        if not self.active:
            return
        if _ACTIVE is self:
            _ACTIVE = None
        self._armed = False
        with torch.no_grad():
            for i, bucket in enumerate(self.buckets):
                if self._works[i] is not None:
                    continue
                flat = self._buffer(i)
                for p in bucket:
                    if id(p) in self._sent:
                        continue
                    _, o = self._where[id(p)]
                    if p.grad is None:
                        flat[o:o + p.numel()].zero_()
                    else:
                        flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
                # these copies ran on the CURRENT stream, which need not be the stream arm() ran on (the one every bucket is issued
                # from): announce it like a hook's staging copy, so that _fire joins it first
                self._staged_on_current_stream(i, flat.device)
                self._fire(i)
            self._next = len(self.buckets)
            for i, bucket in enumerate(self.buckets):
                self._works[i].wait()
                flat = self._flat[i]
                flat.div_(self.world)
                for p in bucket:
                    _, o = self._where[id(p)]
                    g = flat[o:o + p.numel()].view_as(p)
                    if p.grad is None:
                        p.grad = g.to(p.dtype) if g.dtype != p.dtype else g.clone()     # (a bf16 payload widens back into an fp32 .grad)
                    else:
                        p.grad.copy_(g)
        self._sent, self._works = set(), [None] * len(self.buckets)

    def average(self):
        """Non-overlapped form: everything after backward() has finished."""
        if not self.active:
            return
        self._armed = False
        self._sent, self._ready, self._works = set(), [0] * len(self.buckets), [None] * len(self.buckets)
        self._next = 0
        self._streams = [dict() for _ in self.buckets]
        self._fire_stream = None
        self.finish()
