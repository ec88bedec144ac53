"""Which Python call sites of e4s_amd make ATen COPY a tensor in one eager face swap (Tensor.to / float / contiguous / clone / copy_ / torch.cat / zeros ...):
the methods are wrapped and every call that returns new storage is counted by its innermost e4s_amd frame.  usage: python tools/aten_sites.py [batch]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from e4s_amd import synth  # noqa: E402
from e4s_amd.networks import face_swap_core  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
net = bench.Net3(bench.make_opts(out_size=bench.SIZE))
net.load_state_dict(synth.synth_state_dict(bench.SIZE, bench.KREM), strict=True)
net.latent_avg = synth.synth_latent_avg(bench.SIZE).to(dev)
net = net.to(dev).eval()
inputs = bench.build_inputs(B, dev, seed_base=100)
agg = collections.Counter()
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "e4s_amd/" in fr.filename and "tools/" not in fr.filename:
            return f"{fr.filename.split('e4s_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:110]}"
    return "?"


def wrap_method(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if on[0] and torch.is_tensor(out) and self.is_cuda and (name == "copy_" or out.data_ptr() != self.data_ptr()) and out.numel() > 0:
            agg[(name, tuple(out.shape), site())] += 1
        return out
    setattr(torch.Tensor, name, f)


for m in ("to", "float", "contiguous", "clone", "copy_", "repeat", "expand_as", "flip", "half", "double", "long", "int", "bool"):
    wrap_method(m)
for fn in ("cat", "stack", "zeros", "ones", "full", "zeros_like", "ones_like", "where", "gather"):
    orig = getattr(torch, fn)

    def g(*a, _orig=orig, _fn=fn, **k):
        out = _orig(*a, **k)
        if on[0] and torch.is_tensor(out) and out.is_cuda:
            agg[("torch." + _fn, tuple(out.shape), site())] += 1
        return out
    setattr(torch, fn, g)

with torch.no_grad():
    for _ in range(2):
        face_swap_core(net, *inputs[:5], noise=inputs[5])
    on[0] = True
    face_swap_core(net, *inputs[:5], noise=inputs[5])
    on[0] = False
    torch.cuda.synchronize()
for (name, shape, st), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} {name:12s} {str(shape):28s} {st}")
print("total:", sum(agg.values()))
