"""Phase costs of upconv_fused_kernel (the two-blocks-per-CU exact up-conv): E4S_UPCONV3_ABL variants of a profiling build, one process.

    E4S_BUILD_ABLATIONS=1 E4S_BUILD_OUT=$PWD/gpurun_out/libabl.so python -m e4s_amd.build
    E4S_LIB_PATH=$PWD/gpurun_out/libabl.so python tools/upconv3_ablate.py > gpurun_out/upconv3_ablations.json

Results of the ablated variants are WRONG by construction; only their times mean anything."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K  # noqa: E402

NAMES = {0: "full", 1: "no MFMA stages", 2: "no FIR / output stores", 3: "no epilogue", 4: "no global loads", 5: "no LDS staging",
         6: "FIR without output stores", 7: "FIR without the I-tile write"}


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    out = {}
    g = torch.Generator().manual_seed(0)
    k4 = (torch.tensor([1., 3, 3, 1])[None] * torch.tensor([1., 3, 3, 1])[:, None] / 16).cuda()
    for tag, b, h, cin, cout in (("128->64 ->512^2", 8, 256, 128, 64), ("64->32 ->1024^2", 8, 512, 64, 32)):
        x = torch.randn(b, h, h, cin, generator=g).cuda()
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda() / (3 * cin ** 0.5)
        ws = K.subpixel_weights(w)
        s = (torch.rand(b, cin, generator=g) + 0.5).cuda()
        d = (torch.rand(b, cout, generator=g) + 0.5).cuda()
        nz = torch.randn(b, 1, 2 * h, 2 * h, generator=g).cuda()
        nw = torch.tensor([0.1]).cuda()
        bias = torch.randn(cout, generator=g).cuda()
        run = lambda: K.upconv_bf16x3(x, ws, cout, k4, in_scale=s, out_scale=d, noise=nz, noise_w=nw, bias=bias, act=1)
        row = {}
        for a in sorted(NAMES):
            os.environ["E4S_UPCONV3_ABL"] = str(a)
            row[NAMES[a]] = round(timeit(run), 4)
        os.environ["E4S_UPCONV3_ABL"] = "0"
        out[tag] = row
        print(tag, json.dumps(row), file=sys.stderr, flush=True)
    json.dump({"what": "upconv_fused_kernel, ms per launch (8 images); ablated variants compute wrong results", "ms": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
