"""Micro-benchmark: masked generator StyledConv contractions (8 images), exact fp32-MFMA vs region-select split-bf16."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K, synth

B, R = 8, 12
labels = synth.synth_labels_blocks(B, 512, 64, seed=3).to(torch.uint8).cuda().view(B, 512, 512)
# (input res, cin, cout, up)
SHAPES = [(64, 512, 512, False), (32, 512, 512, True), (128, 256, 256, False), (64, 512, 256, True),
          (256, 128, 128, False), (128, 256, 128, True), (32, 512, 512, False), (16, 512, 512, False)]
out = []
for res, cin, cout, up in SHAPES:
    ncls = 4 if up else 1
    x = torch.randn(B, res, res, cin, device="cuda")
    w = torch.randn(ncls, 9, cout, cin, device="cuda") / (3 * cin ** 0.5)
    ws = K.split_bf16x2(w)
    s = torch.rand(B * R, cin, device="cuda") + 0.5
    d = torch.rand(B * R, cout, device="cuda") + 0.5
    kw = dict(labels=labels, num_regions=R, ncls=ncls, ostride=2 if up else 1, in_scale=s, out_scale=d, act=1)
    flop = 2.0 * B * res * res * cin * cout * 9 * ncls
    row = {"shape": [B, res, cin, cout, "up" if up else "same"], "gflop_executed": flop / 1e9}
    for name, extra in (("f32", {}), ("bf16x3", {"w_split": ws})):
        for _ in range(3):
            y = K.conv_mfma(x, w, cout, **kw, **extra)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            y = K.conv_mfma(x, w, cout, **kw, **extra)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        row[name + "_ms"] = round(ms, 4); row[name + "_tflops"] = round(flop / ms / 1e9, 1)
    y32 = K.conv_mfma(x, w, cout, **kw); yb = K.conv_mfma(x, w, cout, **kw, w_split=ws)
    row["maxdiff_rel"] = float((y32 - yb).abs().max() / y32.abs().max())
    out.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_gen_conv.json", "w"), indent=1)
