"""Ablation timings of conv_wino_kernel (needs a library built with E4S_BUILD_ABLATIONS=1): E4S_WINO_VAR = 0..6 on 512->512@32^2 x16 (stats
epilogue, XF = 0).  Prints one JSON line."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from e4s_amd import kernels as K
    b, res, cin, cout = 16, int(sys.argv[2]) if len(sys.argv) > 2 else 32, 512, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, res, res, cin, generator=g).cuda()
    w = (torch.randn(1, 9, cout, cin, generator=g) / (3 * cin ** 0.5)).cuda()
    u = K.wino_weights(w)
    f = lambda: K.conv_wino(x, u, cout, want_stats=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record(); torch.cuda.synchronize()
    print("MS %.4f" % (e0.elapsed_time(e1) / 30))
else:
    names = {0: "full", 1: "no input-transform staging", 2: "no weight staging", 3: "no staging", 4: "MFMAs + barriers", 5: "MFMAs only", 6: "no MFMAs", 7: "items loaded, not transformed", 8: "items transformed, not loaded",
             9: "A fragments of odd positions not re-read (16 of 24 reads: a 64x64x2-position wave tile's traffic)",
             10: "A and B fragments of odd positions not re-read (12 of 24 reads)"}
    res = {}
    for v in range(11):
        env = dict(os.environ, E4S_WINO_VAR=str(v))
        out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout
        ms = [l for l in out.splitlines() if l.startswith("MS ")]
        res[names[v]] = float(ms[-1][3:]) if ms else out[-200:]
    print(json.dumps({"layer": "512->512@32^2 x16 stats epilogue", "ms": res}))
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/wino_ablate.json", "w").write(json.dumps(res))
