"""Worker of tests/test_gpu_nccl_world1.py (own process: a process group must not leak into the pytest process).  A world of ONE rank on
backend "nccl" (= RCCL) with e4s_amd.shard.force_collectives(True): every collective of the N>1 paths really executes --
communicator init with device_id, all_gather_into_tensor (blocking, asynchronous + double buffered, ragged), the bucketed gradient
all-reduces of ddp.GradAverager fired from autograd hooks during the backward, and both of them INSIDE HIP-graph captures
(GraphedFaceSwap under a live process group; TrainIteration.graphed_g_step with an averager).  With one rank every collective is the
identity, so each result is compared bit for bit with the same computation without collectives.  Prints one JSON line."""
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

res = {}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t0 = time.time()
dist.init_process_group("nccl", device_id=dev)
res["backend"] = dist.get_backend()
res["init_s"] = round(time.time() - t0, 2)
shard.force_collectives(True)
assert shard.collectives_active()

# ---- shard.gather_outputs / OverlappedGather on RCCL ---------------------------------------------------------------------------
g = torch.Generator().manual_seed(1)
x = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dev)
res["gather_outputs_equal"] = bool(torch.equal(shard.gather_outputs(x, 4), x))
og = shard.OverlappedGather(4, pack=postproc.tensor2im)
want = postproc.tensor2im(x)
for i in range(5):                                   # slots are reused: the wait-before-overwrite path runs
    og.submit(x if i != 4 else -x)
out = og.drain()
res["overlapped_gather_equal"] = bool(torch.equal(out, postproc.tensor2im(-x))) and out.dtype == torch.uint8 and og.active
res["overlapped_gather_works_were_real"] = og.i == 5

# ---- GraphedFaceSwap captured while a process group (and its watchdog thread) is alive, gather submitted per step -----------------
size, b = 256, 2
K.PRECISION = "f32"
net = Net3(make_opts(out_size=size))
net.load_state_dict(synth.synth_state_dict(size, 13), strict=True)
net.latent_avg = synth.synth_latent_avg(size).to(dev)
net = net.to(dev).eval()
drv = synth.synth_image(b, size, seed=3, tag="w1_d").to(dev)
tgt = synth.synth_image(b, size, seed=3, tag="w1_t").to(dev)
masks = [synth.onehot(synth.synth_labels_face(b, 512, seed=50 + i)).to(dev) for i in range(3)]
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
res["graphed_swap_equal_eager"] = bool(torch.equal(img, eager))
res["graphed_swap_gather_equal"] = bool(torch.equal(gathered, postproc.tensor2im(eager)))

# ---- ddp.GradAverager on RCCL: eager (hooks fire buckets during the backward) and captured inside graphed_g_step -------------------


def build(with_averager):
    n3 = Net3(make_opts(out_size=size))
    n3.load_state_dict(synth.synth_state_dict(size, 13), strict=True)
    n3.latent_avg = synth.synth_latent_avg(size).to(dev)
    n3 = n3.to(dev).train()
    params = [p for p in n3.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, capturable=True)
    lo = LossOpts(face_parsing_lambda=0.0, id_lambda=0.0, lpips_lambda=0.0)          # l2 only: the collectives are what is under test
    avg = GradAverager(params, bucket_mb=16) if with_averager else None
    ema = copy.deepcopy(n3).eval()
    return TrainIteration(n3, None, {}, opt, None, lo=lo, averager=avg, net_ema=ema), n3, avg, ema


img_t = synth.synth_image(b, size, seed=9, tag="w1_img").to(dev)
mask_t = masks[0]
it0, net0, _, ema0 = build(False)
for _ in range(4):
    loss0, _ = it0.g_step(img_t, mask_t, randomize_noise=False)
it1, net1, avg1, ema1 = build(True)
res["averager_active"] = bool(avg1.active) and len(avg1.buckets) > 2
loss1, _ = it1.g_step(img_t, mask_t, randomize_noise=False)
res["buckets"] = len(avg1.buckets)
res["buckets_fired_during_backward"] = int(avg1.fired_during_backward)
for _ in range(3):
    loss1, _ = it1.g_step(img_t, mask_t, randomize_noise=False)
torch.cuda.synchronize()
res["eager_averaged_step_equal"] = bool(torch.equal(loss0, loss1)) and all(
    torch.equal(p, q) for p, q in zip(net0.parameters(), net1.parameters()))
it2, net2, avg2, ema2 = build(True)
gs = it2.graphed_g_step(img_t, mask_t, warmup=2, randomize_noise=False)      # steps 1-2 eager, 3-4 replayed: all-reduces inside the graph
gs.step()
loss2 = gs.step()
gs.validate()
torch.cuda.synchronize()
res["graphed_averaged_step_equal"] = bool(torch.equal(loss0, loss2)) and all(
    torch.equal(p, q) for p, q in zip(net0.parameters(), net2.parameters())) and all(
    torch.equal(p, q) for p, q in zip(ema0.parameters(), ema2.parameters()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    gs.step()
e1.record()
torch.cuda.synchronize()
res["graphed_averaged_g_step_ms_256"] = round(e0.elapsed_time(e1) / 5, 3)
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1 " + json.dumps(res))
