"""Ablation timings of the split-bf16 conv kernel.  Needs a profiling build:
    E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build --force   (rebuild without the variable afterwards)"""
import os, sys, subprocess, json
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from e4s_amd import kernels as K
b, res, cin, cout = [int(v) for v in sys.argv[1:5]]
x = torch.randn(b, res, res, cin, device="cuda"); w = torch.randn(1, 9, cout, cin, device="cuda") / (3 * cin ** 0.5)
ws = K.split_bf16x2(w)
for _ in range(3): K.conv_mfma(x, w, cout, w_split=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.conv_mfma(x, w, cout, w_split=ws)
e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1) / 20)
'''
for shape in ((16, 32, 512, 512), (2, 32, 512, 512)):
    for abl in (0, 1, 2, 3, 4):
        env = dict(os.environ, E4S_BF16X3_ABL=str(abl))
        out = subprocess.run([sys.executable, "-c", code] + [str(v) for v in shape], env=env, capture_output=True, text=True)
        print(shape, "ABL", abl, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
