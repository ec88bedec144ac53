"""Micro-benchmark: the encoder's stride-1 3x3 convs (16 images = 8 swaps) on the Winograd F(2,3) kernel (csrc/conv_wino.hip) against the direct
split-bf16 kernel, each checked against the other.  One JSON line per layer -> gpurun_out/bench_wino.jsonl."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K  # noqa: E402

dev = "cuda"


def timeit(fn, n=30):
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


CASES = [(16, 32, 512, 512), (16, 64, 256, 256), (16, 64, 256, 512), (16, 128, 128, 128), (16, 128, 128, 256), (16, 256, 64, 128), (16, 16, 512, 512),
         (2, 32, 512, 512)]
os.makedirs("gpurun_out", exist_ok=True)
out = open("gpurun_out/bench_wino.jsonl", "a")
for b, res, cin, cout in CASES:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, res, res, cin, generator=g).to(dev)
    w = (torch.randn(1, 9, cout, cin, generator=g) / (3 * cin ** 0.5)).to(dev)
    ws = K.split_bf16x2(w)
    u = K.wino_weights(w)
    st, _ = K.instnorm_stats(x)
    slope = torch.rand(cout, generator=g).to(dev)
    row = {"layer": f"{cin}->{cout}@{res}^2 x{b}", "gflop": round(2.0 * b * res * res * cin * cout * 9 / 1e9, 1)}
    for tag, kw in (("in_prelu", dict(in_stats=st, act=2, slope=slope)), ("stats", dict(want_stats=True))):
        yd = K.conv_mfma(x, w, cout, w_split=ws, **kw)
        yw = K.conv_wino(x, u, cout, **kw)
        if tag == "stats":
            yd, yw = yd[0], yw[0]
        row[tag + "_maxdiff_rel"] = float((yd - yw).abs().max() / yd.abs().max())
        row[tag + "_direct_ms"] = round(timeit(lambda: K.conv_mfma(x, w, cout, w_split=ws, **kw)), 4)
        row[tag + "_wino_ms"] = round(timeit(lambda: K.conv_wino(x, u, cout, **kw)), 4)
    row["algorithmic_tflops_wino"] = round(row["gflop"] / row["in_prelu_wino_ms"], 1)
    print(json.dumps(row), flush=True)
    out.write(json.dumps(row) + "\n")
