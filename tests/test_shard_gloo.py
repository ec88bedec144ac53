"""CPU, world_size 2 over gloo: the N>1 path of bench.py / e4s_amd.shard (shard -> compute -> all-gather).
The compute callable is a stand-in (the HIP path needs a GPU); what is checked is the sharding
arithmetic, ragged shards, ordering of the gathered outputs and that every rank gets the same result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from e4s_amd import shard


def test_shard_range_partitions():
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(n, 3, 8, 8, generator=g)
    masks = torch.randint(0, 12, (n, 1, 4, 4), generator=g).float()
    noise = [torch.randn(n, 1, 4, 4, generator=g), torch.randn(1, 1, 4, 4, generator=g)]   # per-sample + shared

    def fake_swap(img, mask, nz):
        return img * 2.0 + mask.mean((1, 2, 3), keepdim=True) + nz[0].mean((1, 2, 3), keepdim=True) + nz[1].mean()

    out = shard.run_sharded(fake_swap, [imgs, masks, noise])
    want = fake_swap(imgs, masks, noise)
    q.put((rank, bool(torch.allclose(out, want)), tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n, 3, 8, 8) for _, _, shape in res)


def test_gloo_world2_even_batch():
    _run(8)


def test_gloo_world2_ragged_batch():
    _run(5)


def _worker_overlap(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = 3
    og = shard.OverlappedGather(world * b)
    static = torch.empty(b, 2, 4)                       # stands in for a replayed graph's static output buffer
    ok = True
    for step in range(5):
        static.copy_(torch.full((b, 2, 4), float(10 * step + rank)))
        og.submit(static)
        static.fill_(-1.0)                              # the producer overwrites its buffer right after submit
    out = og.drain()
    want = torch.cat([torch.full((b, 2, 4), float(40 + r)) for r in range(world)], 0)
    ok = ok and tuple(out.shape) == (world * b, 2, 4) and bool(torch.equal(out, want))
    prev = og.out[(og.i - 2) % og.depth]               # the step before is still intact in the other slot
    ok = ok and bool(torch.equal(prev, torch.cat([torch.full((b, 2, 4), float(30 + r)) for r in range(world)], 0)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_overlapped_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _pack_u8(local, out=None):
    """CPU stand-in with the contract of e4s_amd.postproc.tensor2im (the oracle's restatement of torch_utils.tensor2im)."""
    from oracle import e4s_oracle as orc
    packed = orc.tensor2im_u8(local)
    if out is None:
        return packed
    out.copy_(packed)
    return out


def _worker_overlap_u8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = 2
    og = shard.OverlappedGather(world * b, pack=_pack_u8)
    static = torch.empty(b, 3, 4, 6)
    imgs = []
    for step in range(4):
        g = torch.Generator().manual_seed(100 * step + rank)
        img = torch.randn(b, 3, 4, 6, generator=g)
        imgs.append(img)
        static.copy_(img)
        og.submit(static)
        static.fill_(7.0)                               # overwritten right after submit, like a replayed graph's output
    out = og.drain()
    want = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 * 3 + r)
        want.append(_pack_u8(torch.randn(b, 3, 4, 6, generator=g)))
    want = torch.cat(want, 0)
    ok = out.dtype == torch.uint8 and tuple(out.shape) == (world * b, 4, 6, 3) and bool(torch.equal(out, want))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_overlapped_gather_of_uint8_images():
    """N4: the shard is packed to the uint8 HWC image (tensor2im) straight into the staging slot; the collective moves a
    quarter of the bytes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap_u8, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _worker_ddp(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e4s_amd.ddp import GradAverager
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7), torch.nn.Linear(7, 3))
    model[3].weight.requires_grad = False                       # a frozen parameter is skipped
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 40, generator=g)
    y = torch.randn(8, 3, generator=g)
    lo, hi = shard.shard_range(8, world, rank)
    loss = torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi])
    loss.backward()
    if rank == 1:
        model[2].bias.grad = None                               # a rank without a gradient contributes zeros
    avg = GradAverager(model.parameters(), bucket_mb=0.02)      # tiny buckets: several collectives
    nb = len(avg.buckets)
    avg.average()
    # single-process reference: mean of the two shard losses' gradients
    ref = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7), torch.nn.Linear(7, 3))
    ref.load_state_dict(model.state_dict())
    want = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        ref.zero_grad()
        a, b = shard.shard_range(8, world, r)
        torch.nn.functional.mse_loss(ref(x[a:b]), y[a:b]).backward()
        for i, p in enumerate(ref.parameters()):
            if p.grad is not None and not (r == 1 and i == 3):  # parameter index 3 = model[2].bias, dropped on rank 1
                want[i] += p.grad / world
    ok = nb > 1
    for i, (p, w) in enumerate(zip(model.parameters(), want)):
        if not p.requires_grad:
            continue
        ok = ok and p.grad is not None and bool(torch.allclose(p.grad, w, atol=1e-6, rtol=1e-5))
    # configs[4] "bf16": the same averaging with a bf16 payload (half the bytes on the wire; fp32 gradients on both ends): every rank
    # rounds its gradient to bf16 (2^-9 relative), the sum is rounded once more -> within 2^-7 of the tensor's scale, and identical on both ranks
    model.zero_grad()
    torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi]).backward()
    if rank == 1:
        model[2].bias.grad = None
    avg16 = GradAverager(model.parameters(), bucket_mb=0.02, payload_dtype=torch.bfloat16)
    ok = ok and len(avg16.buckets) >= 2 and avg16._buffer(0).dtype == torch.bfloat16
    avg16.average()
    flat16 = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad])
    for p, w in zip(model.parameters(), want):
        if p.requires_grad:
            ok = ok and p.grad.dtype == torch.float32 and float((p.grad - w).abs().max()) <= 2 ** -7 * float(w.abs().max())
    gathered = [torch.empty_like(flat16) for _ in range(world)]
    dist.all_gather(gathered, flat16)
    ok = ok and all(torch.equal(gathered[0], g_) for g_ in gathered)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_averaging_equals_single_process():
    """config 5's collective: bucketed all-reduce averaging of the gradients == the gradient of the mean of the shard
    losses computed in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ddp, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


class _MonolithFn(torch.autograd.Function):
    """Stand-in for encoder_autograd.EncoderFn: ONE backward node that produces the gradients of several parameters, last
    layer first, and announces each as soon as it exists (ddp.notify_grad) -- long before autograd accumulates them."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3):
        h1 = x @ w1.t()
        h2 = torch.relu(h1) @ w2.t()
        ctx.save_for_backward(x, w1, w2, w3, h1, h2)
        return torch.relu(h2) @ w3.t()

    @staticmethod
    def backward(ctx, gy):
        from e4s_amd.ddp import notify_grad
        x, w1, w2, w3, h1, h2 = ctx.saved_tensors
        fired = []
        g3 = gy.t() @ torch.relu(h2)
        notify_grad(ctx.params[2], g3)
        fired.append(ctx.avg.fired_during_backward)
        gh2 = (gy @ w3) * (h2 > 0)
        g2 = gh2.t() @ torch.relu(h1)
        notify_grad(ctx.params[1], g2)
        fired.append(ctx.avg.fired_during_backward)
        gh1 = (gh2 @ w2) * (h1 > 0)
        g1 = gh1.t() @ x
        notify_grad(ctx.params[0], g1)
        fired.append(ctx.avg.fired_during_backward)
        ctx.log.extend(fired)
        return None, g1, g2, g3


def _worker_ddp_overlap(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e4s_amd.ddp import GradAverager
    torch.manual_seed(0)
    # "encoder" (monolithic node) followed by an ordinary torch head (hook-driven), as Net3 = EncoderFn -> LocalMLPs -> G
    enc = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(64, 40) * 0.1), torch.nn.Parameter(torch.randn(64, 64) * 0.1),
                                  torch.nn.Parameter(torch.randn(32, 64) * 0.1)])
    head = torch.nn.Sequential(torch.nn.Linear(32, 300), torch.nn.ReLU(), torch.nn.Linear(300, 3))
    params = list(enc) + list(head.parameters())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 40, generator=g)
    y = torch.randn(8, 3, generator=g)
    lo, hi = shard.shard_range(8, world, rank)
    avg = GradAverager(params, bucket_mb=0.02)                  # tiny buckets: several collectives
    log = []

    def backward():
        for p in params:
            p.grad = None

        class Fn(_MonolithFn):
            pass
        # the Function needs the parameter OBJECTS to announce them; ctx attributes are set through a thin subclass call
        orig = _MonolithFn.forward

        def fwd(ctx, xx, a, b, c):
            ctx.params, ctx.avg, ctx.log = list(enc), avg, log
            return orig(ctx, xx, a, b, c)
        Fn.forward = staticmethod(fwd)
        out = head(Fn.apply(x[lo:hi], *enc))
        return torch.nn.functional.mse_loss(out, y[lo:hi])

    # (a) overlapped: buckets fire from hooks / from inside the monolithic node
    loss = backward()
    avg.arm()
    loss.backward()
    fired = avg.fired_during_backward
    avg.finish()
    overl = [p.grad.clone() for p in params]
    # (b) post-hoc: the same gradients averaged after backward
    loss = backward()
    loss.backward()
    avg.average()
    post = [p.grad.clone() for p in params]
    ok = all(torch.equal(a, b) for a, b in zip(overl, post))
    ok = ok and fired >= 2 and len(avg.buckets) >= 3
    # head buckets had left before the monolithic node started, and more left while it was still running
    ok = ok and log[0] >= 1 and log[-1] >= log[0]
    q.put((rank, ok, fired, len(avg.buckets), log[:3]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_overlapped_bucket_allreduce_equals_post_hoc():
    """bucket all-reduces launched from gradient hooks and from inside a monolithic backward node (the
    encoder's) give bit-identical averages to the post-backward form, and really leave before backward() returns."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ddp_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _worker_train_iteration(rank, world, port, q):
    """TrainIteration.g_step / d_step with GradAverager armed (bucket all-reduces fired from hooks during backward) on two
    gloo ranks with different shards == the single-process step on the whole batch (mean of shard losses), CPU stand-in nets."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e4s_amd.ddp import GradAverager
    from e4s_amd.train import LossOpts, TrainIteration

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 3, 3, padding=1))

        def forward(self, img, onehot, **kw):
            return self.body(img), None

    class Disc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, stride=2), torch.nn.LeakyReLU(0.2), torch.nn.Flatten(),
                                            torch.nn.Linear(4 * 3 * 3, 1))

        def forward(self, x):
            return self.body(x)

    def make():
        torch.manual_seed(0)
        return Net(), Disc()
    g = torch.Generator().manual_seed(3)
    img = torch.rand(4, 3, 8, 8, generator=g)
    lo_, hi_ = shard.shard_range(4, world, rank)
    lo = LossOpts(face_parsing_lambda=0, id_lambda=0, lpips_lambda=0, d_every=1)

    def run(distributed):
        net, disc = make()
        opt, opt_d = torch.optim.SGD(net.parameters(), lr=0.1), torch.optim.SGD(disc.parameters(), lr=0.1)
        if distributed:
            it = TrainIteration(net, disc, {}, opt, opt_d, lo=lo, averager=GradAverager(net.parameters(), bucket_mb=1e-4),
                                averager_d=GradAverager(disc.parameters(), bucket_mb=1e-4))
            x = img[lo_:hi_]
        else:
            it = TrainIteration(net, disc, {}, opt, opt_d, lo=lo)
            x = img
        for step in range(2):
            it.iteration(x, None, batch_idx=1)
        fired = it.averager.fired_during_backward if distributed else None
        return [p.detach().clone() for p in list(net.parameters()) + list(disc.parameters())], fired
    dist_params, fired = run(True)
    ref_params, _ = run(False)
    ok = all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(dist_params, ref_params))
    q.put((rank, ok, fired))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_train_iteration_equals_single_process():
    """config 5 under data parallelism: two ranks, each on its shard of the batch, overlapped bucket all-reduces inside
    TrainIteration's G and D steps -> the same parameters after two iterations as one process on the whole batch (equal shard
    sizes: the mean of shard-mean losses is the batch-mean loss)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_train_iteration, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] is not None and r[2] >= 1 for r in res), res


def _worker_ddp_order(rank, world, port, q):
    """Ranks announce their gradients in DIFFERENT orders (rank 0 first parameter first, rank 1 last first) and rank 1 never announces
    one parameter: the collectives must still go out in bucket-index order on both ranks (torch DDP's rule), or the two
    ranks would pair all-reduces of different buckets -- a hang or silent corruption on RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e4s_amd.ddp import GradAverager
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(50 + 7 * i)) for i in range(9)]
    avg = GradAverager(params, bucket_mb=1e-4)                    # every parameter its own bucket (different sizes)
    fired = []
    real_fire = avg._fire
    avg._fire = lambda i: (fired.append(i), real_fire(i))[1]
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(p.shape, generator=g) for p in params]
    order = list(range(9)) if rank == 0 else list(range(8, -1, -1))
    avg.arm()
    for i in order:
        if rank == 1 and i == 4:
            continue                                              # never announced on rank 1: finish() sends zeros
        params[i].grad = grads[i].clone()
        avg.notify(params[i], params[i].grad)
    during = avg.fired_during_backward
    avg.finish()
    all_g = []
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        all_g.append([torch.randn(p.shape, generator=gr) for p in params])
    ok = fired == sorted(fired) and len(fired) == len(avg.buckets)
    for i, p in enumerate(params):
        want = (all_g[0][i] + (0 if i == 4 else all_g[1][i])) / world
        ok = ok and bool(torch.allclose(p.grad, want, atol=1e-6))
    q.put((rank, ok, during, fired))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_buckets_are_reduced_in_index_order_whatever_the_arrival_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ddp_order, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    by_rank = {r: during for r, _, during, _ in res}
    # bucket 0 holds the LAST parameter (reverse order): rank 1 announces it first and fires buckets as they complete a prefix; rank 0
    # announces it last and can only fire everything then
    assert by_rank[0] >= 1 and by_rank[1] >= 1, res
