"""Micro-benchmark: exact fp32-MFMA conv vs split-bf16 conv on the encoder's stride-1 shapes (16 images = 8 swaps)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K

SHAPES = [(16, 32, 512, 512), (16, 64, 256, 256), (16, 128, 128, 128), (16, 256, 64, 128), (16, 16, 512, 512),
          (2, 32, 512, 512)]
out = []
for b, res, cin, cout in SHAPES:
    x = torch.randn(b, res, res, cin, device="cuda")
    w = torch.randn(1, 9, cout, cin, device="cuda") / (3 * cin ** 0.5)
    ws = K.split_bf16x2(w)
    flop = 2.0 * b * res * res * cin * cout * 9
    row = {"shape": [b, res, cin, cout], "gflop": flop / 1e9}
    for name, kw in (("f32", {}), ("bf16x3", {"w_split": ws})):
        for _ in range(3):
            y = K.conv_mfma(x, w, cout, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            y = K.conv_mfma(x, w, cout, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        row[name + "_ms"] = round(ms, 4); row[name + "_tflops"] = round(flop / ms / 1e9, 1)
    y32 = K.conv_mfma(x, w, cout); yb = K.conv_mfma(x, w, cout, w_split=ws)
    row["maxdiff_rel"] = float((y32 - yb).abs().max() / y32.abs().max())
    out.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_conv.json", "w"), indent=1)
