"""Worker of tests/test_gpu_nccl_world1.py and tools/nccl_world1_stress.py (own process: a process group must not leak into the
pytest process).  A world of ONE rank on backend "nccl" (= RCCL) with e4s_amd.shard.force_collectives(True): every collective of the
N>1 paths really executes -- communicator init with device_id, all_gather_into_tensor (blocking, asynchronous + double buffered),
the bucketed gradient all-reduces of ddp.GradAverager fired from autograd hooks during the backward, both of them INSIDE HIP-graph
captures (GraphedFaceSwap under a live process group; TrainIteration.graphed_g_step with an averager), and torch's own
DistributedDataParallel around Net3 exactly as the reference wraps it (src/training/coach.py:74-85).  With one rank every collective
is the identity, so each result is compared bit for bit with the same computation without collectives.

Every comparison that fails says WHAT differed (tensor name, max-abs difference, how many elements) in `diffs`; `--repeat N` runs
the sections N times in this process (the stress harness), `--sections a,b` restricts them.  Prints one JSON line."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("E4S_ALLOW_UNINITIALIZED_LOSS_NETS", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from e4s_amd import kernels as K, postproc, shard, synth  # noqa: E402
from e4s_amd.ddp import GradAverager  # noqa: E402
from e4s_amd.networks import GraphedFaceSwap, Net3, face_swap_core  # noqa: E402
from e4s_amd.optim import FusedAdam  # noqa: E402
from e4s_amd.options import make_opts  # noqa: E402
from e4s_amd.train import LossOpts, TrainIteration  # noqa: E402

ALL = ("gather", "swap", "ddp_eager", "ddp_graphed", "torch_ddp")
ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--sections", default=",".join(ALL))
ap.add_argument("--train-g", type=int, default=1, help="1: Net3(train_G=True) in the data-parallel sections (the reference's default, "
                                                       "train_options.py:32-33); 0: encoder + LocalMLPs only")
args = ap.parse_args()
sections = [s for s in args.sections.split(",") if s]

res = {"diffs": [], "repeat": args.repeat, "sections": sections, "train_G": bool(args.train_g)}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t0 = time.time()
dist.init_process_group("nccl", device_id=dev)
res["backend"] = dist.get_backend()
res["init_s"] = round(time.time() - t0, 2)
shard.force_collectives(True)
assert shard.collectives_active()


def same(key, a, b, name="", it=0):
    """bitwise equality of two tensors, AND-ed into res[key]; a mismatch is described in res['diffs']"""
    ok = a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b))
    if not ok:
        d = (a.double() - b.double()).abs() if a.shape == b.shape else None
        res["diffs"].append({"key": key, "iter": it, "tensor": name, "max_abs": None if d is None else float(d.max()),
                             "n_diff": None if d is None else int((d != 0).sum()), "numel": a.numel(),
                             "scale": float(b.double().abs().max())})
    res[key] = bool(res.get(key, True) and ok)
    return ok


def flag(key, ok):
    res[key] = bool(res.get(key, True) and ok)


size, b = 256, 2
K.PRECISION = "f32"
template = {}


def fresh_net(train_g, mode):
    """a Net3 on the GPU with the seeded synthetic weights (a template per configuration is built once and deep-copied: no host work
    in the loop)"""
    if train_g not in template:
        n3 = Net3(make_opts(out_size=size, train_G=bool(train_g)))
        n3.load_state_dict(synth.synth_state_dict(size, 13), strict=True)
        n3.latent_avg = synth.synth_latent_avg(size).to(dev)
        template[train_g] = n3.to(dev)
    n3 = copy.deepcopy(template[train_g])
    return n3.train() if mode == "train" else n3.eval()


masks = [synth.onehot(synth.synth_labels_face(b, 512, seed=50 + i)).to(dev) for i in range(3)]

def mark(what):
    print(f"W1 iter {it} {what}", flush=True)          # a crash (watchdog abort) is then located by the last line printed


for it in range(args.repeat):
    mark("gather")
    # ---- shard.gather_outputs / OverlappedGather on RCCL -----------------------------------------------------------------------
    if "gather" in sections:
        g = torch.Generator().manual_seed(1 + it)
        x = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dev)
        same("gather_outputs_equal", shard.gather_outputs(x, 4), x, "gather_outputs", it)
        og = shard.OverlappedGather(4, pack=postproc.tensor2im)
        for i in range(5):                                   # slots are reused: the wait-before-overwrite path runs
            og.submit(x if i != 4 else -x)
        out = og.drain()
        same("overlapped_gather_equal", out, postproc.tensor2im(-x), "overlapped_gather", it)
        flag("overlapped_gather_equal", out.dtype == torch.uint8 and og.active)
        flag("overlapped_gather_works_were_real", og.i == 5)

    # ---- GraphedFaceSwap captured while a process group (and its watchdog thread) is alive, gather submitted per step -------------
    if "swap" in sections:
        mark("swap")
        net = fresh_net(0, "eval")
        drv = synth.synth_image(b, size, seed=3, tag="w1_d").to(dev)
        tgt = synth.synth_image(b, size, seed=3, tag="w1_t").to(dev)
        noise = [n.to(dev) for n in synth.synth_noise(size, seed=3, batch=b)]
        with torch.no_grad():
            eager = face_swap_core(net, drv, masks[0], tgt, masks[1], masks[2], noise=noise)
            gf = GraphedFaceSwap(net, b, img_size=size)
            og2 = shard.OverlappedGather(b, pack=postproc.tensor2im)
            for _ in range(3):
                img = gf(drv, masks[0], tgt, masks[1], masks[2], noise)
                og2.submit(img)
            gathered = og2.drain()
            gf.validate()
        torch.cuda.synchronize()
        same("graphed_swap_equal_eager", img, eager, "graphed swap image", it)
        same("graphed_swap_gather_equal", gathered, postproc.tensor2im(eager), "gathered uint8 image", it)
        del gf, og2, net

    # ---- data-parallel G steps: ddp.GradAverager on RCCL (eager: hooks fire buckets during the backward; captured inside
    # graphed_g_step) and torch's DistributedDataParallel around Net3 as coach.py:74-85 wraps it --------------------------------------
    ddp_sections = [s for s in ("ddp_eager", "ddp_graphed", "torch_ddp") if s in sections]
    if not ddp_sections:
        continue

    def build(with_averager, wrap=None):
        n3 = fresh_net(args.train_g, "train")
        params = [p for p in n3.parameters() if p.requires_grad]
        opt = FusedAdam(params, lr=1e-4, capturable=True)
        lo = LossOpts(face_parsing_lambda=0.0, id_lambda=0.0, lpips_lambda=0.0)      # l2 only: the collectives are what is under test
        avg = GradAverager(params, bucket_mb=16) if with_averager else None
        ema = copy.deepcopy(n3).eval()
        mod = wrap(n3) if wrap is not None else n3
        return TrainIteration(mod, None, {}, opt, None, lo=lo, averager=avg, net_ema=ema), n3, avg, ema

    def same_nets(key, na, nb, what):
        for (name, p), (_, q) in zip(na.named_parameters(), nb.named_parameters()):
            same(key, q, p, f"{what}: {name}", it)

    img_t = synth.synth_image(b, size, seed=9, tag="w1_img").to(dev)
    mask_t = masks[0]
    it0, net0, _, ema0 = build(False)
    for _ in range(4):
        loss0, _ = it0.g_step(img_t, mask_t, randomize_noise=False)
    res["trainable_tensors"] = len([p for p in net0.parameters() if p.requires_grad])

    if "ddp_eager" in sections:
        mark("ddp_eager")
        it1, net1, avg1, ema1 = build(True)
        flag("averager_active", bool(avg1.active) and len(avg1.buckets) > 2)
        loss1, _ = it1.g_step(img_t, mask_t, randomize_noise=False)
        res["buckets"] = len(avg1.buckets)
        res["buckets_fired_during_backward"] = int(avg1.fired_during_backward)
        res["bucket_staging_streams_eager"] = max(avg1.staging_streams_seen)
        for _ in range(3):
            loss1, _ = it1.g_step(img_t, mask_t, randomize_noise=False)
        torch.cuda.synchronize()
        same("eager_averaged_step_equal", loss1, loss0, "loss", it)
        same_nets("eager_averaged_step_equal", net0, net1, "net")
        same_nets("eager_averaged_step_equal", ema0, ema1, "ema")
        del it1, net1, avg1, ema1

    if "ddp_graphed" in sections:
        mark("ddp_graphed")
        it2, net2, avg2, ema2 = build(True)
        gs = it2.graphed_g_step(img_t, mask_t, warmup=2, randomize_noise=False)  # steps 1-2 eager, 3-4 replayed: all-reduces inside the graph
        mark("ddp_graphed captured")
        # (fire_log of the CAPTURE cycle: was any bucket filled from a stream other than the one it is issued from, and was that stream
        # part of the capture?)
        res["capture_fills_from_other_stream"] = sum(1 for _, other, _ in avg2.fire_log if other)
        res["capture_fills_from_non_capturing_stream"] = sum(1 for _, _, cap in avg2.fire_log if not cap)
        gs.step()
        loss2 = gs.step()
        gs.validate()
        torch.cuda.synchronize()
        res["bucket_staging_streams_graphed"] = max(avg2.staging_streams_seen)
        same("graphed_averaged_step_equal", loss2, loss0, "loss", it)
        same_nets("graphed_averaged_step_equal", net0, net2, "net")
        same_nets("graphed_averaged_step_equal", ema0, ema2, "ema")
        if it == args.repeat - 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gs.step()
            e1.record()
            torch.cuda.synchronize()
            res["graphed_averaged_g_step_ms_256"] = round(e0.elapsed_time(e1) / 5, 3)
        del gs, it2, net2, avg2, ema2

    if "torch_ddp" in sections:
        mark("torch_ddp")
        # coach.py:74-85: nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], output_device=local_rank,
        # broadcast_buffers=False, find_unused_parameters=True) -- torch's reducer over the monolithic EncoderFn / GeneratorFn nodes
        from torch.nn.parallel import DistributedDataParallel as DDP
        it3, net3, _, ema3 = build(False, wrap=lambda m: DDP(m, device_ids=[0], output_device=0, broadcast_buffers=False,
                                                            find_unused_parameters=True))
        for _ in range(4):
            loss3, _ = it3.g_step(img_t, mask_t, randomize_noise=False)
        torch.cuda.synchronize()
        same("torch_ddp_step_equal", loss3, loss0, "loss", it)
        same_nets("torch_ddp_step_equal", net0, net3, "net")
        same_nets("torch_ddp_step_equal", ema0, ema3, "ema")
        del it3, net3, ema3
    del it0, net0, ema0

mark("done")
res["diffs"] = res["diffs"][:40]
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1 " + json.dumps(res))
