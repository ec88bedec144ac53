"""Ablation timing of the plain split-bf16 kernel on the encoder's 512->512 @32x32 layer (16 images).  Needs a profiling
build: E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build --force; run once per variant: E4S_BF16X3_ABL=<n> python tools/bench_abl.py
(variants: conv_bf16x3.hip, `VAR`)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K
zero = os.environ.get("E4S_ABL_ZERO") == "1"      # zero operands: same instruction stream, minimal switching power (DVFS probe)
torch.manual_seed(0)
x = torch.zeros(16, 32, 32, 512, device="cuda") if zero else torch.randn(16, 32, 32, 512, device="cuda")
w = torch.zeros(1, 9, 512, 512, device="cuda") if zero else torch.randn(1, 9, 512, 512, device="cuda") / (3 * 512 ** 0.5)
ws = K.split_bf16x2(w)
for _ in range(5):
    K.conv_mfma(x, w, 512, w_split=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    K.conv_mfma(x, w, 512, w_split=ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
y = K.conv_mfma(x, w, 512, w_split=ws).double()
print(json.dumps({"zero": zero, "abl": os.environ.get("E4S_BF16X3_ABL", "0"), "ms": round(ms, 4), "sum": float(y.sum()), "abs": float(y.abs().sum()), "tflops_alg": round(2 * 16 * 1024 * 512 * 512 * 9 / ms / 1e9, 1)}))
