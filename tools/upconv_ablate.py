"""Time e4s_upconv_mfma_f32 on one layer shape; E4S_UPCONV_ABL selects an ablated kernel (profiling builds only:
E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build --force)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import kernels as K
B, H, cin, cout = [int(a) for a in sys.argv[1:5]]
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, H, cin, generator=g).cuda()
w = torch.randn(1, 9, cout, cin, generator=g).cuda()
s = torch.rand(B, cin, generator=g).cuda() + 0.5
d = torch.rand(B, cout, generator=g).cuda()
k4 = (torch.tensor([1., 3, 3, 1])[None] * torch.tensor([1., 3, 3, 1])[:, None] / 16).cuda()
nz = torch.randn(B, 1, 2 * H, 2 * H, generator=g).cuda()
nw = torch.tensor([0.1]).cuda(); bias = torch.zeros(cout).cuda()
run = lambda: K.upconv_mfma(x, w, cout, k4, in_scale=s, out_scale=d, noise=nz, noise_w=nw, bias=bias, act=1)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("ABL=%s B=%d H=%d %d->%d: %.1f us" % (os.environ.get("E4S_UPCONV_ABL", "0"), B, H, cin, cout, e0.elapsed_time(e1) * 100))
