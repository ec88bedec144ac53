"""bench.py -- 1024^2 face-swap images/s on N MI355X (BASELINE.json metric) + headline-kernel roofline.

A "step" is one pass of the E4S-core hot path (scripts/face_swap.py:237-273) over one
batch of synthetic swaps resident in HBM: 2 encoder passes, regional style swap, 12 LocalMLPs and the
mask-guided 1024^2 generator, per swap.  Workload = BASELINE.json configs[3] sharded the way it
names it (batch 64 over 8 GPUs = 8 swaps per GPU; weak scaling: per-GPU batch fixed) -- at N=1 that
is 8 swaps on one GPU; the single-swap latency of configs[1] is reported alongside (`latency_b1_ms`).
For N>1 each rank runs its shard and the ranks all-gather the [B/N,3,1024,1024] outputs over RCCL
(the only collective on the path; asynchronous and double-buffered, so step i's gather overlaps step i+1, and every
gather is drained inside the timed region); timing is barrier+sync bracketed, max over ranks.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py --gpus 8

Extra legs (rank 0, N=1): `roofline` = the headline kernel ModulatedConv2d(512,512,3)@64x64 (masked,
region-select) timed with HIP events on its launch stream: the kernel the timed step really runs for that
layer -- the split-bf16 kernel against the dense bf16 MFMA peak (2500 TF; algorithmic FLOPs, so <= 1/3 by
construction) under E4S_PRECISION=auto/bf16x3, with the exact-fp32 kernel against the fp32-MFMA peak
(157.3 TF) nested as `exact_fp32_variant`, or that one alone under E4S_PRECISION=f32;
`cpu_baseline` = the CPU oracle (oracle/e4s_oracle.py, a port of the reference's pure-PyTorch path)
timed on the host cores for ONE swap of the same workload, which doubles as an end-to-end parity check.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from e4s_amd import kernels as K  # noqa: E402
from e4s_amd import synth  # noqa: E402
from e4s_amd.networks import GraphedFaceSwap, Net3, face_swap_core  # noqa: E402
from e4s_amd.options import make_opts  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table: BF16 dense (the 2:1-sparsity headline figure is not used)
DTYPE = {"f32": "f32",
         "bf16x3": "f32 operands as hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate (bf16x3)",
         "auto": "f32 operands as hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate (bf16x3) on the 3x3 convs whose "
                 "launch fills the chip; exact f32 elsewhere"}
SIZE, KREM = 1024, 13


def build_inputs(batch, dev, seed_base):
    driven = synth.synth_image(batch, SIZE, seed=seed_base, tag="bench_d").to(dev)
    target = synth.synth_image(batch, SIZE, seed=seed_base, tag="bench_t").to(dev)
    dm = synth.onehot(synth.synth_labels_face(batch, 512, seed=seed_base * 3 + 1)).to(dev)
    tm = synth.onehot(synth.synth_labels_face(batch, 512, seed=seed_base * 3 + 2)).to(dev)
    sm = synth.onehot(synth.synth_labels_face(batch, 512, seed=seed_base * 3 + 3)).to(dev)
    noise = [n.to(dev) for n in synth.synth_noise(SIZE, seed=seed_base, batch=batch)]
    return driven, dm, target, tm, sm, noise


def headline_probe(net, batch, mask, reps):
    """ModulatedConv2d(512,512,3) at 64x64 (G.convs[7], the north-star's '512x512 modulated conv'), one
    region-select pass over `batch` images = 2*512*512*9*64*64 FLOP per image."""
    dev = mask.device
    layer = net.G.convs[7]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(batch, 64, 64, 512, generator=g).to(dev)
    lat = (torch.randn(batch, 12, 18, 512, generator=g) * 0.5).to(dev)
    labels, _ = K.mask_labels(mask)
    mod = layer.conv.modulation
    s = K.modulate(lat, 8, True, mod.weight, mod.bias)
    pk = layer.conv.packed()
    d = K.demod_coefs(s, pk["wsq"], layer.conv.scale)
    nz = torch.randn(batch, 1, 64, 64, generator=g).to(dev)

    kw = dict(labels=labels, num_regions=12, in_scale=s, out_scale=d, noise=nz, noise_w=layer.noise.weight,
              bias=layer.activate.bias, act=1)
    flops = 2.0 * 512 * 512 * 9 * 64 * 64 * batch

    def timed(extra):
        for _ in range(3):
            K.conv_mfma(x, pk["w"], 512, **kw, **extra)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            K.conv_mfma(x, pk["w"], 512, **kw, **extra)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        return ms, flops / (ms * 1e-3) / 1e12

    traffic = {}
    traffic_stale = False
    tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.isfile(tp):
        try:
            traffic = json.load(open(tp))
            # the PMC figure belongs to ONE version of the kernel: keyed on a hash of its source, so that it cannot go stale silently
            import hashlib
            rows = traffic.get("bf16x3_rows", {})
            src = os.path.join(ROOT, rows.get("kernel_source", "e4s_amd/csrc/conv_region1w.hip"))
            if hashlib.sha256(open(src, "rb").read()).hexdigest() != rows.get("kernel_source_sha256"):
                traffic_stale = True
        except Exception:
            traffic = {}
    ms32, ach32 = timed({})
    exact = {"bound": "mfma", "kernel": "conv_mfma_kernel<128,128,2,2,spatial> (v_mfma_f32_32x32x2_f32, exact fp32)",
             "achieved": round(ach32, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach32 / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic.get("hbm_bytes_per_launch"),
             "avg_launch_ms": round(ms32, 4), "flop_per_launch": flops}
    what = "ModulatedConv2d(512,512,3)@64x64 masked, x%d img" % batch
    if not K.want_bf16x3(batch, 64, 64, 512, 512):
        exact["kernel"] += " " + what
        return exact
    # the kernel the timed step runs for this layer under E4S_PRECISION=auto/bf16x3.  `achieved` stays ALGORITHMIC
    # (one multiply-add per fp32 product); the kernel issues 3 bf16 MFMAs per product, so frac <= 1/3 by construction.
    from e4s_amd import stylegan2 as sg2
    extra = {"w_split": layer.conv.split_weights()}
    name = "conv_bf16x3_region_kernel"
    if sg2.REGION_ROWS:                   # what StyledConv.run_nhwc launches for this layer
        extra["w_split16"] = layer.conv.split_weights16()
        name = "conv_region_rows_kernel"
    ms, ach = timed(extra)
    if sg2.REGION_ROWS and K.LAST_REGION_PATH == 2:
        name = "conv_region_rows1w_kernel<1 x 4 waves, one per SIMD, 256 x 256 tile, B fragments from global>"
    return {"bound": "mfma", "kernel": name + " (3x v_mfma_f32_32x32x16_bf16 per product) " + what,
            "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "frac_ceiling": round(1.0 / 3.0, 4),
            "mfma_executed_tflops": round(3 * ach, 1),
            # power-limited rate of a register-resident v_mfma_f32_32x32x16_bf16 stream on random operands (tools/mfma_peak.sh,
            # profiles/r04a_mfma_peak.jsonl: 1 774-1 782 TF at 1.69 GHz / 1 350 W; zero operands reach 2 471-2 480 TF at 2.36 GHz):
            # what `frac` could reach for a 3-MFMA-per-product kernel is 1780 / 3 / 2500 = 0.237
            "frac_of_power_limited_mfma_rate": round(3 * ach / 1780.0, 4),
            "traffic": None if traffic_stale else traffic.get("hbm_bytes_per_launch_bf16x3"), "traffic_stale": traffic_stale,
            "traffic_source": "profiles/roofline_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of this kernel ("
                              + str(traffic.get("bf16x3_rows", {}).get("source", "separate passes")) + "), valid for the kernel source whose "
                              "sha256 the file records; a PMC pass cannot run inside this timed process",
            "avg_launch_ms": round(ms, 4), "flop_per_launch": flops, "exact_fp32_variant": exact}


PEAK_HBM_TBPS = 8.0               # same guide: HBM3E spec (6.3 TB/s is what a float4 copy reaches)
_TIMED = {"e4s_conv_wino_bf16x3_f32", "e4s_conv_bf16x3_f32", "e4s_conv_region_bf16x3_f32", "e4s_upconv_bf16x3_f32",
          "e4s_conv_c32_bf16x3_f32"}


class LaunchTimer:
    """HIP events on the launch stream around every native conv launch of an EAGER step (the graph replays exactly these launches):
    kernels.call is the one door every C-ABI entry point of e4s_amd.kernels goes through.  A record = (entry point, dims of its
    e4s_conv_params, start event, end event); an entry point that issues a second launch (the region-select fallback behind the
    variant-rows kernel, a split-K epilogue) is timed with it."""

    def __init__(self):
        self.recs = []

    def __enter__(self):
        self._orig = K.call
        recs = self.recs

        def timed_call(name, *args):
            if name not in _TIMED:
                return self._orig(name, *args)
            q = args[0]._obj
            dims = {k: int(getattr(q, k)) for k in ("B", "Hi", "Wi", "Ho", "Wo", "Cin", "Cout", "ncls", "istride", "ntaps")}
            dims["masked"] = bool(q.labels)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig(name, *args)
            e1.record()
            recs.append((name, dims, e0, e1))
            return r
        K.call = timed_call
        return self

    def __exit__(self, *exc):
        K.call = self._orig
        return False

    def rows(self):
        torch.cuda.synchronize()
        return [(n, d, e0.elapsed_time(e1)) for n, d, e0, e1 in self.recs]


def step_rooflines(net, inputs, reps=3):
    """`roofline_dominant`: the kernel that takes the largest share of the timed step -- the encoder's stride-1 3x3 convs on
    conv_wino_kernel (44 launches per step of 2 x B images) -- as sum of algorithmic FLOP (2 * B * H * W * Cin * Cout * 9 per launch; the
    kernel executes 2/3 of the direct form's products, three bf16 MFMAs each) / sum of HIP-event time.  `roofline_hbm`: the HBM-bound tail of
    the generator (the two exact up-convs, 64->64 @512^2, 32->32 @1024^2 with the ToRGB partial) as algorithmic bytes (input once + output
    once, fp32) / time against the 8 TB/s spec.  Events around each launch of `reps` eager steps after one warm-up step."""
    def run():
        face_swap_core(net, *inputs[:5], noise=inputs[5])
    run()
    with LaunchTimer() as lt:
        for _ in range(reps):
            run()
    rows = lt.rows()
    wino = [(d, ms) for n, d, ms in rows if n == "e4s_conv_wino_bf16x3_f32"]
    out = {}
    if wino:
        fl = sum(2.0 * d["B"] * d["Hi"] * d["Wi"] * d["Cin"] * d["Cout"] * 9 for d, _ in wino) / reps
        ms = sum(m for _, m in wino) / reps
        shapes = {}
        for d, m in wino:
            k = "%d->%d@%d" % (d["Cin"], d["Cout"], d["Hi"])
            a = shapes.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += m
            a[2] += 2.0 * d["B"] * d["Hi"] * d["Wi"] * d["Cin"] * d["Cout"] * 9
        ach = fl / (ms * 1e-3) / 1e12
        out["roofline_dominant"] = {
            "bound": "mfma", "kernel": "conv_wino_kernel (Winograd F(2,3) along the rows, 3x v_mfma_f32_32x32x16_bf16 per product): the "
                                       "encoder's stride-1 3x3 convs of one step (2 x %d images)" % inputs[0].shape[0],
            "launches_per_step": len(wino) // reps, "flop_per_step": fl, "ms_per_step": round(ms, 4),
            "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
            "frac_ceiling": 0.5, "mfma_executed_tflops": round(2 * ach, 1),
            "frac_of_power_limited_mfma_rate": round(2 * ach / 1780.0, 4),
            "by_shape": {k: {"launches": a[0] // reps, "ms": round(a[1] / reps, 4), "tflops": round(a[2] / a[1] / 1e9, 1)}
                         for k, a in shapes.items()}}
    tail = []
    for n, d, ms in rows:
        if n == "e4s_upconv_bf16x3_f32" or n == "e4s_conv_c32_bf16x3_f32" or \
                (n == "e4s_conv_bf16x3_f32" and not d["masked"] and d["Cin"] <= 64 and d["Hi"] >= 512 and d["istride"] == 1):
            by = 4.0 * d["B"] * (d["Hi"] * d["Wi"] * d["Cin"] + d["Ho"] * d["Wo"] * d["Cout"])
            tail.append(("%s %d->%d@%d" % (n.replace("e4s_", "").replace("_bf16x3_f32", ""), d["Cin"], d["Cout"], d["Ho"]), by, ms))
    if tail:
        per = {}
        for k, by, ms in tail:
            a = per.setdefault(k, [0.0, 0.0])
            a[0] += by / reps
            a[1] += ms / reps
        tb = sum(a[0] for a in per.values())
        tms = sum(a[1] for a in per.values())
        out["roofline_hbm"] = {
            "bound": "hbm", "kernels": {k: {"bytes": a[0], "ms": round(a[1], 4), "TBps": round(a[0] / a[1] / 1e9, 3)} for k, a in per.items()},
            "achieved": round(tb / tms / 1e9 * 1e3, 1), "peak": PEAK_HBM_TBPS * 1e3, "unit": "GB/s",
            "frac": round(tb / tms / 1e9 / PEAK_HBM_TBPS, 4), "ms_per_step": round(tms, 4), "bytes_per_step": tb,
            "traffic": None, "note": "algorithmic bytes = fp32 input once + fp32 output once per launch (weights are < 1 % of either)"}
    return out


def gather_contention_leg(step, steps, world_equiv=8, payload_per_rank=8 * 1024 * 1024 * 3):
    """What the all-gather of N = world_equiv ranks can cost a rank's step in CU / HBM contention, bounded on ONE GPU: beside every timed
    step a side stream lands (world_equiv - 1) x payload_per_rank bytes (the peers' uint8 shards: 7 x 25 MB per step at N = 8) in a
    gather-sized buffer with RCCL-shaped copy kernels -- a few persistent workgroups (e4s_stream_copy_u8: `blocks` of 512 threads), not
    the hundreds a torch copy would launch.  This over-counts RCCL's receive side (peers' writes over xGMI occupy no CU of the receiver
    for the direct-write protocols; the ring / LL protocols copy through local workgroups as modelled here) and adds the HBM write
    traffic of the real gather, so the ratio is an upper bound on the efficiency loss from contention; wire time and barrier skew are
    not in it."""
    from e4s_amd.lib import call, ptr
    dev = torch.device("cuda", torch.cuda.current_device())
    nbytes = (world_equiv - 1) * payload_per_rank
    src = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    dst = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    side = torch.cuda.Stream()

    def timed(blocks):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if blocks:
                side.wait_stream(torch.cuda.current_stream())     # the gather of step i - 1 starts when step i starts (OverlappedGather)
                call("e4s_stream_copy_u8", ptr(src), ptr(dst), nbytes, blocks, ctypes_stream(side))
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    base = timed(0)
    out = {"world_equiv": world_equiv, "bytes_per_step": nbytes, "ms_per_step_alone": round(base, 3)}
    for blocks in (8, 32, 64):
        ms = timed(blocks)
        out["ms_per_step_with_%d_copy_workgroups" % blocks] = round(ms, 3)
        out["efficiency_bound_%d" % blocks] = round(base / ms, 4)
    return out


def ctypes_stream(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


def optimisation_leg(net, one, steps, losses="full", graphed=False):
    """BASELINE.json configs[2] (scripts/optimization.py:209-232): Adam(lr=1e-2) on the [1,12,1280] regional style
    vectors through cal_style_codes -> gen_img (fresh noise every step, as the script does).  losses = "full": the
    script's default objective -- l2 * 1.0 + LPIPS-AlexNet at 1024/512/256 * 0.8 + IR-SE50 identity * 0.1 + parsing-UNet
    features * 0.1 (optim_options.py:44-48, optimization.py:88-122) on the native loss networks (e4s_amd.criteria, synthetic
    weights: the real ones are downloads); "mse": the l2 term alone.  graphed: the whole step (forward, losses, backward, Adam with
    its step count on the device) replayed as one HIP graph (e4s_amd.optim.GraphedStep)."""
    import types
    from e4s_amd.train import mse_loss
    driven, dm, target, tm, sm, _noise = one
    torch.manual_seed(1234)                                 # the per-step noise draws: same sequence in every run of this leg
    for p in net.parameters():
        p.requires_grad = False
    with torch.no_grad():
        sv, _ = net.get_style_vectors(target, tm)
    latent = sv.clone().requires_grad_(True)
    from e4s_amd.optim import FusedAdam
    opt = FusedAdam([latent], lr=1e-2, capturable=graphed)      # torch.optim.Adam's update as one kernel
    lpips = idl = fpl = None
    if losses == "full":
        from e4s_amd import criteria
        from e4s_amd.criteria import FaceParsingLoss, IDLoss, LPIPS
        criteria.ALLOW_UNINITIALIZED = True                  # seeded synthetic state dicts are loaded right below
        lpips = LPIPS()
        lpips.load_state_dict(synth.synth_module_state_dict(lpips, 0, "lp."))
        idl = IDLoss(types.SimpleNamespace(id_loss_multiscale=True))
        idl.load_state_dict(synth.synth_module_state_dict(idl, 0, "id."))
        fpl = FaceParsingLoss(types.SimpleNamespace())
        fpl.load_state_dict(synth.synth_module_state_dict(fpl, 0, "fp."))
        lpips, idl, fpl = lpips.to(target.device).eval(), idl.to(target.device).eval(), fpl.to(target.device).eval()

    # the three loss networks are independent chains of batch-1 launches: forked from and joined to the step's stream inside the capture
    # (e4s_amd.optim.forked_sum; E4S_BENCH_LOSS_STREAMS=0: one stream, for A/B runs -- same loss, same latent)
    fork = os.environ.get("E4S_BENCH_LOSS_STREAMS", "1") != "0"
    last = {}
    from e4s_amd.optim import forked_sum

    def body():
        codes = net.cal_style_codes(latent)
        img, _, _ = net.gen_img(None, codes, tm, randomize_noise=True)
        loss = mse_loss(img, target)            # (F.mse_loss's semaphore memset must not sit in the captured step: e4s_amd.train.mse_loss)
        if lpips is not None:
            terms = os.environ.get("E4S_BENCH_TERMS", "lpips,id,parsing").split(",")      # debugging aid: subset of the terms
            fns = []
            if "lpips" in terms:
                fns.append(lambda: 0.8 * lpips.forward_pooled(img, target, (1024, 512, 256)))
            if "id" in terms:
                fns.append(lambda: 0.1 * idl(img, target)[0])
            if "parsing" in terms:
                fns.append(lambda: 0.1 * fpl(img, target)[0])
            if fork:
                # ONE side branch -- the identity network, the longest chain -- and the rest on the capturing stream under it: every side branch of
                # a replayed hipGraph costs cross-queue edges.  Measured, two alternations on one box, ms per step: all three forked
                # 12.87-12.90, LPIPS alone 12.35-12.39, ID alone 11.98-12.00, parsing alone 14.43, ID + LPIPS 12.60, none 14.5.
                # (E4S_FORK_SIDE=0,1,2 restores the three branches.)  Same terms, same order of addition.
                side = [i for i, t in enumerate(t for t in ("lpips", "id", "parsing") if t in terms) if t == "id"] or None
                loss = forked_sum(loss, fns, inputs=(img, target), side_terms=None if os.environ.get("E4S_FORK_SIDE") else side)
            else:
                for fn in fns:
                    loss = loss + fn()
        loss.backward()
        opt.step()
        last["loss"] = loss.detach()
        return last["loss"]
    if graphed:
        from e4s_amd.optim import GraphedStep
        gs = GraphedStep(opt, body, warmup=2)       # 2 eager steps, then the step as ONE HIP graph
        gs.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gs.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gs.validate()                               # one-hot masks, finite loss (one sync, untimed)
        if os.environ.get("E4S_BENCH_PRINT_LOSS") == "1":
            print("opt leg %s: final loss %.9g, latent checksum %.9g" % (losses, float(last["loss"]), float(latent.detach().double().sum())),
                  file=sys.stderr, flush=True)
        return round(dt / steps * 1e3, 3)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        body()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e3, 3)


def config3_legs(net, one, args):
    modes = args.opt_modes.split(",")
    out = {}
    if "full" in modes:
        out["config3_opt_step_ms"] = optimisation_leg(net, one, args.opt_steps, "full", graphed=args.opt_graph)
    if "mse" in modes:
        out["config3_mse_only_step_ms"] = optimisation_leg(net, one, args.opt_steps, "mse", graphed=args.opt_graph)
    out["config3_graphed"] = bool(args.opt_graph)
    return out


def gpen_leg(dev, reps=10):
    """GPEN-BFR-512's FullGenerator (the restoration pass that runs once per swap before the parser,
    scripts/face_swap.py:206-210) on the same kernels: single-image latency and 8-image throughput, synthetic weights."""
    from e4s_amd.gpen import FullGenerator
    net = FullGenerator(512, 512, 8, channel_multiplier=2, narrow=1)
    net.load_state_dict(synth.synth_gpen_state_dict(512, n_mlp=8), strict=True)
    net = net.to(dev).eval()
    out = {}
    for b in (1, 8):
        x = synth.synth_image(b, 512, tag="gpen_bench").to(dev)
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net(x)
        torch.cuda.synchronize()
        out[f"b{b}_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    out["images_per_s_b8"] = round(8e3 / out["b8_ms"], 1)
    return out


def stitch_leg(net, inputs, reps=10):
    """SURVEY.md 8(f) N4 on the device, after the swap (scripts/face_swap.py:253-310): head-mask swap of the parsing maps ->
    (the generator output of the timed step) -> tensor2im -> foreground / create_masks -> default stitch (mask image, 11x11
    erode, fixed-point Gaussian, PIL-style alpha composite) or --lap_bld (paste + 10-level Laplacian pyramid blend) -> uint8
    HWC.  Batch of 8 at 1024^2; ms per batch, generator excluded (it is the timed step)."""
    from e4s_amd import postproc as PP
    driven, dm, target, tm, sm, noise = inputs
    b = driven.shape[0]
    src_lab = dm.argmax(1).to(torch.uint8).contiguous()
    tgt_lab = tm.argmax(1).to(torch.uint8).contiguous()
    tgt_u8 = PP.tensor2im(target)
    with torch.no_grad():
        img = face_swap_core(net, driven, dm, target, tm, sm, noise=noise)
    out = {}
    for name, lap in (("default_ms", False), ("lap_bld_ms", True)):
        def run():
            lab, hole = PP.swap_head_mask_revisit_considerGlass(src_lab, tgt_lab)
            return PP.stitch(img, tgt_u8, lab, hole, lap_bld=lap)
        for _ in range(2):
            res = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    out["batch"] = b
    out["out"] = f"uint8 {list(res.shape)}"
    # cv2 / skimage exist neither in the build container nor on the GPU box: the OpenCV halves of the stitch (erode, fixed-point
    # GaussianBlur, pyrDown / pyrUp, the Laplacian blend's schedule) are restatements of OpenCV's algorithms, cross-checked against
    # independent third-party implementations (scipy.ndimage, PIL) on the reference's example images (tests/golden/cv2free.pt) -- not cv2
    out["parity"] = ("mask swap / create_masks / tensor2im / alpha composite: pinned to the reference's own outputs; cv2 parts: cross-checked "
                     "against scipy.ndimage / PIL (erode, uint8 pyrDown exact; Gaussian, blend +-1 LSB; float pyramids 2 ulp), not pinned to cv2")
    return out


def train_leg(dev, lat, steps=10, batch=2, losses="full", warmup=3, train_G=True, time_d=True, bf16=False):
    """BASELINE.json configs[4] on ONE GPU -- the body of Coach.train() (coach.py:280-398) through e4s_amd.train.TrainIteration:
    G step = Net3.forward (train_G=True, the reference's default, train_options.py:32-33 / coach.py:324-331: encoder + LocalMLPs +
    the generator's convs[:K] / ToRGBs / constant input trainable, the mapping network and the layers past K frozen; train_G=False:
    encoder + LocalMLPs only, SURVEY.md 8(d)'s opts and coach.py:333-334's "only training Encoder" branch) on a batch of 2
    (train_options.py:24) at 1024^2 -> calc_loss's default terms (coach.py:403-453, train_options.py:47-54: parsing * 0.1 + ID * 0.1
    + l2 + LPIPS x3 * 0.8 on the native loss networks) + g_adv_lambda * AdvGLoss through the native Discriminator graph ->
    backward through the HIP loss-network / generator / MLP / encoder kernels -> fused Adam -> EMA of the weights;
    D step (every d_every = 15th iteration) = Net3 forward without a graph + D(real) + D(fake) + AdvDLoss backward + fused
    Adam on D; R1 step (d_reg_every; off by default in the reference, timed here at 16) = the second-order pass.
    `ms_per_step` is the G step (every iteration runs one); `ms_per_iteration_amortised` adds d_step / 15.  The D and R1 steps are
    replayed as HIP graphs too (graphed_d_step / graphed_r1_step; the eager timings beside them).
    bf16=True: BASELINE.json configs[4] as it NAMES the step -- the activations parked for the backward stored as bf16 (e4s_amd/tape.py);
    arithmetic, weights, moments and gradients fp32 as in the default leg (the reference trains fp32: coach.py has no autocast).
    losses = "mse": the l2 term alone, no Discriminator.  Synthetic weights everywhere; `warmup` untimed + `steps` timed."""
    import copy
    import types
    from e4s_amd.optim import FusedAdam
    from e4s_amd.train import LossOpts, TrainIteration
    net = Net3(make_opts(out_size=SIZE, train_G=train_G))
    net.load_state_dict(synth.synth_state_dict(SIZE, KREM), strict=True)
    net.latent_avg = lat.to(dev)
    net = net.to(dev).train()
    img = synth.synth_image(batch, SIZE, seed=7, tag="train_img").to(dev)
    mask = synth.onehot(synth.synth_labels_face(batch, 512, seed=21)).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, capturable=True)       # (step count on the device: the same object serves the graphed G step)
    crit, disc, opt_d = {}, None, None
    lo = LossOpts(d_reg_every=16)
    if losses == "full":
        from e4s_amd import criteria
        from e4s_amd.criteria import FaceParsingLoss, IDLoss, LPIPS
        from e4s_amd.stylegan2 import Discriminator
        criteria.ALLOW_UNINITIALIZED = True                  # seeded synthetic state dicts are loaded right below
        lp, idl, fpl = LPIPS(), IDLoss(types.SimpleNamespace(id_loss_multiscale=True)), FaceParsingLoss(types.SimpleNamespace())
        for m, tag in ((lp, "lp."), (idl, "id."), (fpl, "fp.")):
            m.load_state_dict(synth.synth_module_state_dict(m, 0, tag))
        crit = {"lpips": lp.to(dev).eval(), "id": idl.to(dev).eval(), "parsing": fpl.to(dev).eval()}
        disc = Discriminator(SIZE)
        disc.load_state_dict(synth.synth_disc_state_dict(SIZE), strict=True)
        disc = disc.to(dev).train()
        opt_d = FusedAdam(disc.parameters(), lr=1e-4, capturable=True)
    else:
        lo.face_parsing_lambda = lo.id_lambda = lo.lpips_lambda = 0.0
    net_ema = copy.deepcopy(net).eval() if losses == "full" else None      # coach.py:60-67
    it = TrainIteration(net, disc, crit, opt, opt_d, lo=lo, net_ema=net_ema, bf16_storage=bf16)

    def timed(fn, n, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    def g_eager():
        it.forget_targets()          # every iteration is a new batch: the target features are recomputed, as calc_loss does
        it.g_step(img, mask)
    # the step replayed as ONE HIP graph (e4s_amd.train.graphed_g_step; target features, pack rebuilds, Adam and EMA inside).  Captured
    # BEFORE any eager step: autograd's AccumulateGrad nodes remember the stream they were created on, and nodes left over from eager
    # steps on the default stream break a later capture on the side stream
    gs = it.graphed_g_step(img, mask, warmup=warmup)
    gd = gr = None
    if disc is not None and time_d:
        try:                                     # (captured before any eager step, like the G step)
            gd = it.graphed_d_step(img, mask, warmup=1)
            gr = it.graphed_r1_step(img, warmup=1)
        except Exception as e:      # noqa: BLE001
            gd = gr = None
            d_capture_error = f"{type(e).__name__}: {e}"[:200]
    g_ms = timed(gs.step, steps, 1)
    gs.validate()
    g_eager_ms = timed(g_eager, steps, 1)
    out = {"ms_per_step": round(g_ms, 2), "g_step_ms": round(g_ms, 2), "g_step_graphed": True, "g_step_eager_ms": round(g_eager_ms, 2),
           "batch": batch, "steps": steps, "warmup": warmup,
           "images_per_s": round(batch * 1e3 / g_ms, 2), "trainable_parameters": int(sum(p.numel() for p in params)),
           "losses": losses, "ema": net_ema is not None, "train_G": bool(train_G), "train_D": disc is not None,
           "activation_storage": "bf16" if bf16 else "fp32", "loss_networks_forked": {"id": "identity network on a side stream, the rest on the step's stream (eager and captured)"}.get(it.fork_losses, str(it.fork_losses)),
           # (ONE side branch: side branches of a replayed hipGraph are spread over hardware queues and cost more than they hide beyond the first)
           "g_step_best_ms": round(min(g_ms, g_eager_ms), 2), "g_step_best_mode": "graph" if g_ms <= g_eager_ms else "eager",
           "trainable_generator_parameters": int(sum(p.numel() for p in net.G.parameters() if p.requires_grad))}
    if disc is not None and time_d:
        d_eager = timed(lambda: it.d_step(img, mask), max(2, steps // 2), 1)
        r1_eager = timed(lambda: it.r1_step(img), max(2, steps // 3), 1)
        d_ms, r1_ms = d_eager, r1_eager
        if gd is not None:
            d_ms = timed(gd.step, max(2, steps // 2), 1)
            r1_ms = timed(gr.step, max(2, steps // 3), 1)
            gd.validate()
        else:
            out["d_capture_error"] = d_capture_error
        out.update(d_step_ms=round(d_ms, 2), r1_step_ms=round(r1_ms, 2), d_step_graphed=gd is not None, d_step_eager_ms=round(d_eager, 2),
                   r1_step_eager_ms=round(r1_eager, 2), d_every=lo.d_every, d_reg_every=lo.d_reg_every,
                   ms_per_iteration_amortised=round(g_ms + d_ms / lo.d_every, 2),
                   discriminator_parameters=int(sum(p.numel() for p in disc.parameters())))
    return out


def cpu_baseline(sd, lat, inputs, hip_img0, hip_img0_b1):
    """One swap (sample 0 of the bench batch) on the host cores with the CPU oracle; returns the baseline record and the
    max-abs differences of (sample 0 of the timed batch, the batch-1 run) against it."""
    from oracle import e4s_oracle as orc
    cores = min(os.cpu_count() or 1, 16)      # oneDNN at 1024^2 B=1 stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    driven, dm, target, tm, sm, noise = [t[:1].cpu() if torch.is_tensor(t) else [n[:1].cpu() for n in t] for t in inputs]
    reps = 3                                   # 1 warm-up + 3 timed: ~20 s of CPU work on the host cores
    with torch.no_grad():
        orc.face_swap_core(sd, driven, dm, target, tm, sm, lat, noise, SIZE, KREM)
        t0 = time.perf_counter()
        for _ in range(reps):
            img = orc.face_swap_core(sd, driven, dm, target, tm, sm, lat, noise, SIZE, KREM)
        dt = (time.perf_counter() - t0) / reps
    err = (float((img - hip_img0.cpu()).abs().max()), float((img - hip_img0_b1.cpu()).abs().max()))
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "1 swap (2 encoder passes + 12 LocalMLPs + 1024^2 generator, B=1), same seeded inputs as sample 0 "
                      "of the GPU batch, 1 warm-up + mean of %d runs, torch %d threads" % (reps, cores),
            "seconds": round(dt * reps, 2)}, err


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="swaps per GPU per step (configs[3]: 64 over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the ~700 launches per step eagerly instead of "
                                                            "replaying one captured HIP graph")
    ap.add_argument("--train-only", action="store_true", help="run only the configs[4] train-step legs")
    ap.add_argument("--train-legs", default="full,frozen,mse", help="with --train-only: which legs (full = train_G + train_D, the reference's "
                                                                    "default; frozen = G frozen; mse = l2 only, no D)")
    ap.add_argument("--train-g-only", action="store_true", help="with --train-only: skip the D / R1 step timings (clean G-step profiles)")
    ap.add_argument("--opt-modes", default="full,mse",
                    help="config-3 legs to run: full = l2 + LPIPS x3 + ID + parsing, mse = l2 only")
    ap.add_argument("--opt-eager", dest="opt_graph", action="store_false",
                    help="enqueue the config-3 steps eagerly; the default replays each step -- generator forward, the four loss "
                         "terms, backward, capturable Adam -- as ONE HIP graph (e4s_amd.optim.GraphedStep; bit-identical to the "
                         "eager loop, tests/test_gpu_optim.py)")
    ap.add_argument("--opt-graph", dest="opt_graph", action="store_true", help="(default) see --opt-eager")
    ap.set_defaults(opt_graph=True)
    ap.add_argument("--opt-steps", type=int, default=200,
                    help="configs[2] leg: run this many W+ optimisation steps (scripts/optimization.py runs 200: "
                         "cal_style_codes + 1024^2 generator fwd + MSE + bwd + Adam each) and report the measured total")
    ap.add_argument("--f32-steps", type=int, default=5, help="also time this many steps with E4S_PRECISION=f32 (exact "
                                                             "fp32 MFMA everywhere) and report value_f32 / ms_per_step_f32")
    ap.add_argument("--train-steps", type=int, default=10, help="configs[4] leg on one GPU: time this many joint train steps "
                                                                "after 3 warm-ups (batch 2 at 1024^2, encoder + LocalMLPs "
                                                                "trainable; + D step and R1 step timed separately)")
    ap.add_argument("--gather-fp32", action="store_true",
                    help="N>1: all-gather the fp32 [B,3,H,W] images (100 MB per 8 swaps) instead of the default uint8 HWC "
                         "images the pipeline ends with (torch_utils.tensor2im, packed on the device: 25 MB)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N>1: one blocking all-gather per step instead of the default double-buffered asynchronous "
                         "all-gather (e4s_amd.shard.OverlappedGather: step i's gather runs under step i+1's compute; "
                         "every gather completes inside the timed region)")
    ap.add_argument("--steps-only", action="store_true", help="only the timed steps (clean rocprofv3 kernel traces)")
    ap.add_argument("--probe-only", action="store_true", help="run only the headline-kernel probe (for rocprofv3)")
    ap.add_argument("--probe-reps", type=int, default=20)
    ap.add_argument("--opt-only", action="store_true", help="run only the configs[2] optimisation leg (for rocprofv3)")
    ap.add_argument("--stub-swap", action="store_true",
                    help="TEST ONLY (tests/test_bench_gloo.py): replace the HIP face swap by a trivial CPU function of the "
                         "inputs so that the N>1 plumbing of main() -- shard seeds, overlapped uint8 gather, drain, barrier, "
                         "MAX-reduced timing, the JSON line -- runs on CPU ranks over gloo.  The line says data: stub; it is "
                         "not a measurement")
    return ap


def main():
    args = build_parser().parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("E4S_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; gloo only to exercise the N>1 code path
    stub = bool(args.stub_swap)
    if stub and backend != "gloo":
        raise SystemExit("--stub-swap is the CPU/gloo plumbing test (E4S_DIST_BACKEND=gloo); it never runs beside the HIP path")
    if not torch.cuda.is_available() and not stub:
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        dev_index = local_rank % torch.cuda.device_count()      # == local_rank on a real N-GPU node
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        sync = torch.cuda.synchronize
    dist = None
    from e4s_amd import shard
    # E4S_FORCE_COLLECTIVES=1 with WORLD_SIZE=1 (torchrun --nproc-per-node 1): the N>1 code path -- RCCL init, the overlapped uint8
    # all-gather, barrier, MAX all-reduce of the time -- runs on one GPU (collectives of one rank; the line says so in `config`)
    multi = world > 1 or (shard._FORCE and "RANK" in os.environ)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)

    B = args.batch
    if stub:
        net = sd = lat = None
        g = torch.Generator().manual_seed(100 + rank)           # per-rank shard seed, as build_inputs(seed_base=100 + rank)
        inputs = [torch.rand(B, 3, 8, 8, generator=g) * 2 - 1 for _ in range(5)] + [[torch.zeros(B, 1, 4, 4)]]
    else:
        sd = synth.synth_state_dict(SIZE, KREM)
        lat = synth.synth_latent_avg(SIZE)
        net = Net3(make_opts(out_size=SIZE))
        net.load_state_dict(sd, strict=True)
        net.latent_avg = lat.to(dev)
        net = net.to(dev).eval()
        inputs = build_inputs(B, dev, seed_base=100 + rank)

    if args.probe_only:
        print(json.dumps(headline_probe(net, B, inputs[4], args.probe_reps)))
        return
    if args.train_only:
        legs = {"full": ("config5_train_step_1gpu", lambda: train_leg(dev, lat, args.train_steps, losses="full", time_d=not args.train_g_only)),
                "bf16": ("config5_train_step_1gpu_bf16", lambda: train_leg(dev, lat, args.train_steps, losses="full", time_d=False, bf16=True)),
                "frozen": ("config5_train_step_1gpu_G_frozen", lambda: train_leg(dev, lat, args.train_steps, losses="full", train_G=False,
                                                                                 time_d=False)),
                "mse": ("config5_train_step_1gpu_mse_only", lambda: train_leg(dev, lat, args.train_steps, losses="mse", train_G=False))}
        print(json.dumps({legs[k][0]: legs[k][1]() for k in args.train_legs.split(",") if k in legs}))
        return
    if args.opt_only:
        one = [t[:1].contiguous() if torch.is_tensor(t) else [n[:1].contiguous() for n in t] for t in inputs]
        print(json.dumps(dict(config3_legs(net, one, args), steps=args.opt_steps)))
        return

    graphed = None
    if stub:
        swap = lambda d, dm, t, tm, sm, noise: 0.5 * d + 0.25 * t          # any deterministic function of the rank's shard
    elif args.no_graph:
        swap = lambda *a, noise: face_swap_core(net, *a, noise=noise)
    else:
        # the batch lives in the graph's own input buffers (inputs are resident in HBM when the timed region starts: these buffers ARE
        # that residence -- a loader writes there directly); a step is one replay, not 22 device-to-device copies + a replay
        graphed = GraphedFaceSwap(net, B)
        graphed.load(*inputs[:5], inputs[5])
        swap = lambda *a, noise: graphed.replay()

    if stub:
        def pack(local, out=None):                           # the contract of postproc.tensor2im (torch_utils.tensor2im)
            u8 = ((local.clamp(-1, 1) + 1) * 127.5).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            if out is None:
                return u8
            out.copy_(u8)
            return out
    else:
        from e4s_amd import postproc
        pack = postproc.tensor2im
    if args.gather_fp32:
        pack = None
    overlap = shard.OverlappedGather(world * B, pack=pack) if (multi and not args.sync_gather) else None
    packed = [None]

    def step():
        img = swap(*inputs[:5], noise=inputs[5])
        if overlap is not None:
            overlap.submit(img)                       # step i's gather runs under step i+1's compute
        elif multi:                                   # RCCL all_gather_into_tensor of the rank's shard
            shard.gather_outputs(img if pack is None else pack(img), world * B)
        elif pack is not None:
            # N = 1 does the SAME per-step work as a rank of N > 1 minus the collective: the uint8 HWC pack of the outputs (the image
            # the pipeline ends with) runs inside the timed region at every N, so N = 1 of a scaling curve equals this line
            packed[0] = pack(img) if packed[0] is None else pack(img, out=packed[0])
        return img

    def fence():
        if overlap is not None:
            overlap.drain()                           # every submitted gather completes inside the timed region
        if multi:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        img = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step()
    fence()
    dt = time.perf_counter() - t0
    if graphed is not None:
        graphed.validate()                            # the one-hot precondition of the replayed graph (one sync, untimed)
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * B * args.steps / dt

    out = {"metric": "1024^2 face-swap images/sec", "value": round(value, 3), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[K.PRECISION],
           "data": "stub (CPU plumbing test, not a measurement)" if stub else "synthetic",
           "config": {"workload": "E4S-core face swap at 1024^2 (2x Net3 encoder @256^2, style swap, 12 LocalMLPs, "
                                  "mask-guided StyleGAN2 generator K=13), BASELINE.json configs[3] shard: "
                                  f"{B} swaps per GPU per step", "per_gpu_batch": B, "global_batch": B * world,
                      "out_size": SIZE, "hip_graph": not args.no_graph,
                      "per_step": "swap of the batch resident in the graph's input buffers + uint8 [B,H,W,3] pack of the outputs "
                                  "(tensor2im) at every N; N > 1 adds the all-gather",
                      "precision": {"f32": "exact fp32 MFMA everywhere",
                                    "bf16x3": "encoder stride-1 3x3 convs: 3 bf16 MFMAs per product on hi/lo-split "
                                              "fp32 operands, fp32 accumulate; everything else exact fp32",
                                    "auto": "as bf16x3 where the launch fills the chip (this batch), else exact fp32"
                                    }[K.PRECISION],
                      "encoder_convs": ("Winograd F(2,3) along the rows on the same split-bf16 MFMAs (e4s_conv_wino_bf16x3_f32: 1.5x fewer "
                                        "MFMAs; fp32 transforms with +-1, 1/2 coefficients) where a launch has >= 128 tiles or >= 64 blocks "
                                        "after split-K (kernels.wino_eligible), direct split-bf16 kernel otherwise"
                                        if (K.WINO and K.PRECISION != "f32") else "direct kernels"),
                      "parallelism": f"image-parallel x{world}" + (
                          (", RCCL all_gather of the " + ("fp32 [B,3,H,W]" if args.gather_fp32 else "uint8 [B,H,W,3]")
                           + " outputs" + ("" if args.sync_gather else " overlapped with the next step"))
                          if multi else "") + (" [forced collectives on a world of one rank]" if multi and world == 1 else "")}}
    if stub:
        # what the plumbing test asserts: the last step's gathered uint8 batch holds rank r's shard at rows [r*B, (r+1)*B)
        gathered = overlap.drain() if overlap is not None else (
            shard.gather_outputs(pack(img) if pack is not None else img, world * B) if world > 1 else (pack(img) if pack else img))
        want = []
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            ins = [torch.rand(B, 3, 8, 8, generator=g) * 2 - 1 for _ in range(5)]
            o = 0.5 * ins[0] + 0.25 * ins[2]
            want.append(pack(o) if pack is not None else o)
        out["stub_gather_ok"] = bool(torch.equal(gathered, torch.cat(want, 0)))
        out["stub_gather_shape"] = list(gathered.shape)
    if rank == 0 and world == 1 and not args.steps_only and not stub:
        # configs[1]: single-swap latency
        one = [t[:1].contiguous() if torch.is_tensor(t) else [n[:1].contiguous() for n in t] for t in inputs]
        swap1 = (lambda *a, noise: face_swap_core(net, *a, noise=noise)) if args.no_graph else None
        if swap1 is None:
            graphed1 = GraphedFaceSwap(net, 1)
            graphed1.load(*one[:5], one[5])
            swap1 = lambda *a, noise: graphed1.replay()
        for _ in range(2):
            img1 = swap1(*one[:5], noise=one[5])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            img1 = swap1(*one[:5], noise=one[5])
        torch.cuda.synchronize()
        out["latency_b1_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        if graphed is not None:
            # the same step FED every time: 22 device-to-device copies of a fresh batch into the graph's input buffers + the replay +
            # the uint8 pack -- what a pipeline pays whose previous stage does not write into `graphed.static` itself
            def fed():
                im = graphed(*inputs[:5], inputs[5])
                return im if pack is None else (pack(im) if packed[0] is None else pack(im, out=packed[0]))
            for _ in range(2):
                fed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fed()
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / args.steps
            out["value_with_input_copies"] = round(B / dtf, 3)
            out["ms_per_step_with_input_copies"] = round(dtf * 1e3, 3)
        out["roofline"] = headline_probe(net, B, inputs[4], args.probe_reps)
        try:
            out.update(step_rooflines(net, inputs))
        except Exception as e:      # noqa: BLE001
            out["roofline_dominant_error"] = f"{type(e).__name__}: {e}"[:300]
        if args.f32_steps > 0 and K.PRECISION != "f32" and not args.no_graph:
            # the same step in exact fp32 (v_mfma_f32_32x32x2_f32 everywhere): a second captured graph under the f32 policy
            saved = K.PRECISION
            K.PRECISION = "f32"
            try:
                g32 = GraphedFaceSwap(net, B)
                g32.load(*inputs[:5], inputs[5])
                for _ in range(2):
                    g32.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.f32_steps):
                    img32 = g32.replay()
                torch.cuda.synchronize()
                dt32 = (time.perf_counter() - t0) / args.f32_steps
            finally:
                K.PRECISION = saved
            out["value_f32"] = round(B / dt32, 3)
            out["ms_per_step_f32"] = round(dt32 * 1e3, 3)
            out["max_abs_default_vs_f32"] = float((img32 - img).abs().max())
            del g32
        # side legs: a failure in one of them is reported in the line, it must not cost the headline measurement above
        def side(key, fn):
            try:
                res = fn()
                out.update(res if key is None else {key: res})
            except Exception as e:      # noqa: BLE001
                out[(key or "config3") + "_error"] = f"{type(e).__name__}: {e}"[:300]
        if graphed is not None:
            side("gather_contention", lambda: gather_contention_leg(step, args.steps))
        side("gpen512", lambda: gpen_leg(dev))
        side("stitch_b8", lambda: stitch_leg(net, inputs))
        if args.train_steps > 0:
            # the reference's default: train_G = train_D = True (train_options.py:32-33); the round-4 configuration (G frozen while D
            # trains, which coach.py:324 says must not be combined) stays beside it for continuity, G step only
            side("config5_train_step_1gpu", lambda: train_leg(dev, lat, args.train_steps, losses="full"))
            side("config5_train_step_1gpu_bf16", lambda: train_leg(dev, lat, args.train_steps, losses="full", time_d=False, bf16=True))
            side("config5_train_step_1gpu_G_frozen", lambda: train_leg(dev, lat, args.train_steps, losses="full", train_G=False,
                                                                       time_d=False))
            side("config5_train_step_1gpu_mse_only", lambda: train_leg(dev, lat, args.train_steps, losses="mse", train_G=False))
        if args.opt_steps > 0:
            side(None, lambda: config3_legs(net, one, args))
            out["config3_steps_run"] = args.opt_steps
            if "config3_opt_step_ms" in out:
                out["config3_total_s"] = round(out["config3_opt_step_ms"] * args.opt_steps / 1e3, 3)
        if not args.no_cpu_baseline:
            cb, err = cpu_baseline(sd, lat, inputs, img[0:1], img1[0:1])
            out["cpu_baseline"] = cb
            out["parity_max_abs_vs_oracle"] = err[0]          # sample 0 of the TIMED batch
            out["parity_b1_max_abs_vs_oracle"] = err[1]       # the batch-1 latency run
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
