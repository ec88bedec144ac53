"""Micro-benchmark of the distinct 3x3 contractions of one timed step (8 swaps: encoder on 16 images, generator on 8):
exact fp32 kernels vs the split-bf16 kernels, with the HBM floor of each layer.  Writes gpurun_out/bench_layers.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K  # noqa: E402

dev = "cuda"
k4 = (torch.tensor([1., 3., 3., 1.])[:, None] * torch.tensor([1., 3., 3., 1.])[None, :] / 64 * 4).to(dev)


def timeit(fn, n=10):
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


rows = []
# (tag, batch, res_in, cin, cout, kind)   kind: enc (plain +PReLU, with IN), same (unmasked styled), up (unmasked up-conv)
CASES = [("enc 64->128@256", 16, 256, 64, 128, "enc"), ("enc 128->128@128", 16, 128, 128, 128, "enc"),
         ("enc 256->256@64", 16, 64, 256, 256, "enc"), ("enc 512->512@32", 16, 32, 512, 512, "enc"),
         ("enc 512->512@16", 16, 16, 512, 512, "enc"),
         ("gen 64->64@512", 8, 512, 64, 64, "same"), ("gen 32->32@1024", 8, 1024, 32, 32, "same"),
         ("gen up 128->64 ->512", 8, 256, 128, 64, "up"), ("gen up 64->32 ->1024", 8, 512, 64, 32, "up")]
CASES += [("s2 128->128 @256->128", 16, 256, 128, 128, "s2"), ("s2 256->256 @128->64", 16, 128, 256, 256, "s2"),
          ("s2 512->512 @64->32", 16, 64, 512, 512, "s2"), ("s2 512->512 @32->16", 16, 32, 512, 512, "s2"),
          ("sc 64->128 @256->128", 16, 256, 64, 128, "sc"), ("sc 128->256 @128->64", 16, 128, 128, 256, "sc"),
          ("sc 256->512 @64->32", 16, 64, 256, 512, "sc")]
# masked up-convs (region-select): exact fp32 (one pass per region present in a tile) vs polyphase split-bf16
CASES += [("mup 512->512 ->64", 8, 32, 512, 512, "mup"), ("mup 512->256 ->128", 8, 64, 512, 256, "mup"),
          ("mup 256->128 ->256", 8, 128, 256, 128, "mup")]
CASES += [("inapply 512@32", 16, 32, 512, 512, "inapply"), ("inapply 256@64", 16, 64, 256, 256, "inapply"),
          ("inapply 128@128", 16, 128, 128, 128, "inapply")]
CASES += [("stem 3->64@256", 16, 256, 32, 64, "stem")]
only = sys.argv[1:]
for tag, b, res, cin, cout, kind in CASES:
    if only and not any(o in tag for o in only):
        continue
    x = torch.randn(b, res, res, cin, device=dev)
    if kind == "stem":
        xs = torch.randn(b, res, res, 3, device=dev)
        ws_ = torch.randn(cout, 3, 3, 3, device=dev)
        ms = timeit(lambda: K.conv3x3_small(xs, ws_), 20)
        row = {"layer": tag, "ms": round(ms, 4), "tflops": round(2 * 27 * cout * b * res * res / ms / 1e9, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        continue
    if kind == "inapply":
        st, _ = K.instnorm_stats(x)
        gate = torch.rand(b, cin, device=dev)
        res = torch.randn_like(x)
        ms = timeit(lambda: K.instnorm_apply(x, st, gate=gate, res=res, want_stats=True), 20)
        row = {"layer": tag, "apply_stats_ms": round(ms, 4), "tb_per_s": round(3 * x.numel() * 4 / ms / 1e9, 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        continue
    if kind == "mup":
        from e4s_amd import synth
        R = 12
        lab = torch.cat([synth.synth_labels_face(1, 512, seed=40 + i) for i in range(b)], 0)
        labels, _ = K.mask_labels(synth.onehot(lab).to(dev))
        w = torch.randn(4, 9, cout, cin, device=dev) / (3 * cin ** 0.5)
        w3 = torch.randn(1, 9, cout, cin, device=dev) / (3 * cin ** 0.5)
        ws = K.split_bf16x2(w)
        s = torch.rand(b * R, cin, device=dev) + 0.5
        d = torch.rand(b * R, cout, device=dev) + 0.5
        nz = torch.randn(b, 1, 2 * res, 2 * res, device=dev)
        kw = dict(in_scale=s, out_scale=d, noise=nz, noise_w=torch.tensor([0.1], device=dev), bias=torch.randn(cout, device=dev),
                  act=1, labels=labels, num_regions=R)
        row = {"layer": tag, "gflop_alg": 2.0 * b * res * res * cin * cout * 9 / 1e9}
        row["f32_exact_ms"] = timeit(lambda: K.upconv_mfma(x, w3, cout, k4, **kw))
        row["bf16x3_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, ncls=4, ostride=2, w_split=ws, **kw))
        row = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}
        rows.append(row)
        print(json.dumps(row), flush=True)
        continue
    if kind in ("s2", "sc"):
        nt = 9 if kind == "s2" else 1
        w = torch.randn(1, nt, cout, cin, device=dev) / (nt * cin) ** 0.5
        ws = K.split_bf16x2(w)
        row = {"layer": tag, "gflop_alg": 2.0 * b * (res // 2) ** 2 * cin * cout * nt / 1e9,
               "hbm_floor_ms": (x.numel() + b * (res // 2) ** 2 * cout) * 4 / 6.0e12 * 1e3}
        row["f32_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, istride=2, ntaps=nt))
        row["bf16x3_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, istride=2, ntaps=nt, w_split=ws))
        row = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}
        row["bf16x3_tflops_alg"] = round(row["gflop_alg"] / row["bf16x3_ms"], 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
        continue
    ro = res * 2 if kind == "up" else res
    ncls = 4 if kind == "up" else 1
    w = torch.randn(ncls, 9, cout, cin, device=dev) / (3 * cin ** 0.5)
    ws = K.split_bf16x2(w)
    row = {"layer": tag, "gflop_alg": 2.0 * b * res * res * cin * cout * 9 / 1e9,
           "hbm_floor_ms": (x.numel() + b * ro * ro * cout) * 4 / 6.0e12 * 1e3}
    if kind == "enc":
        slope = torch.rand(cout, device=dev)
        st, _ = K.instnorm_stats(x)
        row["f32_ms"] = timeit(lambda: K.conv_mfma(K.instnorm_apply(x, st), w, cout, act=2, slope=slope))
        row["bf16x3_unfused_ms"] = timeit(lambda: K.conv_mfma(K.instnorm_apply(x, st), w, cout, w_split=ws, act=2, slope=slope))
        row["bf16x3_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, w_split=ws, in_stats=st, act=2, slope=slope))
    else:
        s = torch.rand(b, cin, device=dev) + 0.5
        d = torch.rand(b, cout, device=dev) + 0.5
        nz = torch.randn(b, 1, ro, ro, device=dev)
        nw = torch.tensor([0.1], device=dev)
        bias = torch.randn(cout, device=dev)
        kw = dict(in_scale=s, out_scale=d, noise=nz, noise_w=nw, bias=bias, act=1)
        if kind == "same":
            row["f32_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, **kw))
            row["bf16x3_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, w_split=ws, **kw))
            if cin == 32:
                row["bf16x3_generic_ms"] = row["bf16x3_ms"]
                row["bf16x3_ms"] = timeit(lambda: K.conv_c32(x, ws, cout, **kw))             # resident weights
                wrgb = torch.randn(b, 3, 32, device=dev)
                row["bf16x3_with_torgb_ms"] = timeit(lambda: K.conv_c32(x, ws, cout, rgb_ws=wrgb, **kw))
        else:
            w3 = torch.randn(1, 9, cout, cin, device=dev) / (3 * cin ** 0.5)
            row["f32_ms"] = timeit(lambda: K.upconv_mfma(x, w3, cout, k4, **kw))
            row["f32_poly_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, ncls=4, ostride=2, **kw))
            row["bf16x3_poly_ms"] = timeit(lambda: K.conv_mfma(x, w, cout, ncls=4, ostride=2, w_split=ws, **kw))
            wraw = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
            wsub = K.subpixel_weights(wraw)
            row["bf16x3_ms"] = timeit(lambda: K.upconv_bf16x3(x, wsub, cout, k4, **kw))     # exact: fused sub-pixel GEMM + FIR epilogue
    row = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}
    row["bf16x3_tflops_alg"] = round(row["gflop_alg"] / row["bf16x3_ms"], 1)
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/bench_layers.json", "w"), indent=1)
