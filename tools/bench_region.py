"""Micro-benchmark of the region-select split-bf16 kernel on the masked StyledConvs of the generator (8 images): the same-resolution
layers and the polyphase up-convs, each timed and checked against the exact fp32 kernel on the same operands.
E4S_BENCH_ROWS=1: time e4s_conv_region_bf16x3_f32 (variant-rows kernel, what StyledConv.run_nhwc launches); unset: the region-select
kernel of e4s_conv_bf16x3_f32, whose variants (wave layout / split form, csrc/conv_bf16x3.hip:launch_region) profiling builds
(E4S_BUILD_ABLATIONS=1) select with env E4S_REGION_VAR = 0..2.  `plain_kernel_ms`: the same contraction with ONE style per sample on the
plain persistent kernel.  Prints one JSON line per layer and appends them to gpurun_out/bench_region.jsonl
(profiles/r03f_microbench_region.json collects the round-3 runs)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K, synth  # noqa: E402

dev = "cuda"
R = 12


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# (tag, batch, res_in, cin, cout, up)
CASES = [("same 512->512@64 (headline)", 8, 64, 512, 512, False), ("same 512->512@32", 8, 32, 512, 512, False),
         ("same 512->512@16", 8, 16, 512, 512, False), ("same 256->256@128", 8, 128, 256, 256, False),
         ("same 128->128@256", 8, 256, 128, 128, False), ("same 512->512@64 b1", 1, 64, 512, 512, False),
         ("mup 512->512 ->64", 8, 32, 512, 512, True), ("mup 512->256 ->128", 8, 64, 512, 256, True),
         ("mup 256->128 ->256", 8, 128, 256, 128, True), ("same 256->128@64 noise-mask", 2, 64, 256, 128, False)]
only = sys.argv[1:]
var = os.environ.get("E4S_REGION_VAR", "") + ("rows" if os.environ.get("E4S_BENCH_ROWS") == "1" else "")
os.makedirs("gpurun_out", exist_ok=True)
out = open("gpurun_out/bench_region.jsonl", "a")
for tag, b, res, cin, cout, up in CASES:
    if only and not any(o in tag for o in only):
        continue
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, res, res, cin, generator=g).to(dev)
    lab = torch.cat([synth.synth_labels_face(1, 512, seed=40 + i) for i in range(b)], 0)
    if "noise-mask" in tag:             # every tile overflows the variant rows: the whole launch falls back
        lab = torch.randint(0, R, lab.shape, generator=g)
    labels, _ = K.mask_labels(synth.onehot(lab).to(dev))
    ncls = 4 if up else 1
    ro = 2 * res if up else res
    w = (torch.randn(ncls, 9, cout, cin, generator=g) / (3 * cin ** 0.5)).to(dev)
    ws = K.split_bf16x2(w)
    s = (torch.rand(b * R, cin, generator=g) + 0.5).to(dev)
    d = (torch.rand(b * R, cout, generator=g) + 0.5).to(dev)
    nz = torch.randn(b, 1, ro, ro, generator=g).to(dev)
    kw = dict(in_scale=s, out_scale=d, noise=nz, noise_w=torch.tensor([0.1], device=dev), bias=torch.randn(cout, generator=g).to(dev),
              act=1, labels=labels, num_regions=R)
    if up:
        kw.update(ncls=4, ostride=2)
    ref = K.conv_mfma(x, w, cout, **kw)
    old = K.conv_mfma(x, w, cout, w_split=ws, **kw)
    if os.environ.get("E4S_BENCH_ROWS") == "1":                   # variant-rows kernel (conv_region.hip); its fallback is the region-select kernel
        kw["w_split16"] = K.split16_bf16x2(w)
    got = K.conv_mfma(x, w, cout, w_split=ws, **kw)
    err = float((got - ref).abs().max() / ref.abs().max())
    err_old = float((got - old).abs().max() / ref.abs().max())
    ms = timeit(lambda: K.conv_mfma(x, w, cout, w_split=ws, **kw))
    gf = 2.0 * b * res * res * cin * cout * 9 * ncls / 1e9
    # the same contraction with ONE style per sample on the plain (persistent, 32-channel chunks) kernel: the rate to compare with
    kwp = dict(kw, labels=None, num_regions=1, in_scale=s[::R].contiguous(), out_scale=d[::R].contiguous())
    kwp.pop("w_split16", None)
    ms_plain = timeit(lambda: K.conv_mfma(x, w, cout, w_split=ws, **kwp))
    row = {"layer": tag, "var": var, "ms": round(ms, 4), "tflops_executed_products": round(gf / ms, 1), "plain_kernel_ms": round(ms_plain, 4), "max_err_vs_f32": err, "max_diff_vs_region_select": err_old}
    print(json.dumps(row), flush=True)
    out.write(json.dumps(row) + "\n")
