import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import synth, kernels as K
from e4s_amd import autograd as AG
from e4s_amd.stylegan2 import Generator
from oracle import e4s_oracle as orc
DEV = "cuda"
size, Kk, cells = 32, 5, 8
full = synth.synth_state_dict(size, 13)
sd = {k[2:]: v for k, v in full.items() if k.startswith("G.")}
gen = Generator(size, 512, 8, split_layer_idx=5, remaining_layer_idx=Kk)
gen.load_state_dict(sd, strict=True); gen = gen.to(DEV).eval()
for n_, p in gen.named_parameters(): p.requires_grad = not n_.startswith("style.")
g = torch.Generator().manual_seed(80)
b = 2
lat = torch.randn(b, 12, gen.n_latent, 512, generator=g) * 0.5
mask = synth.onehot(synth.synth_labels_blocks(b, 512, cells, seed=4))
noise = synth.synth_noise(size, seed=5, batch=b)
w_img = torch.randn(b, 3, size, size, generator=g)
cap = {}
orig_bwd = AG.styled_conv_backward
def spy(rec, dy, r, extras=None):
    if rec["layer"] is gen.convs[5]:
        cap["dy"] = dy.clone(); cap["rec"] = rec
    return orig_bwd(rec, dy, r, extras)
AG.styled_conv_backward = spy
img, _, _ = gen([lat.to(DEV)], None, mask.to(DEV), input_is_latent=True, noise=[n.to(DEV) for n in noise])
(img * w_img.to(DEV)).sum().backward()
mine = gen.convs[5].conv.weight.grad.cpu().double()
rec = cap["rec"]
# layer-level fp64 reference on the SAME inputs
f64 = torch.float64
pfx = "convs.5."
sdl = {k[len(pfx):]: v.to(f64) for k, v in sd.items() if k.startswith(pfx)}
sdl["conv.weight"].requires_grad_(True)
x = K.nhwc_to_nchw(rec["x"]).cpu().to(f64)
style = lat[:, 0, 6].to(f64)
y = orc.styled_conv(sdl, "", x, style, None, noise[6].to(f64), False, False)
dy = K.nhwc_to_nchw(cap["dy"]).cpu().to(f64)
(y * dy).sum().backward()
ref_layer = sdl["conv.weight"].grad
print("mine vs layer-level f64 on same inputs:", float((mine - ref_layer).abs().max()), "scale", float(ref_layer.abs().max()))
print("fwd y parity:", float((K.nhwc_to_nchw(rec["y"]).cpu().double() - y).abs().max()))
# full-generator fp64 reference
sd64 = {"G." + k: v.to(f64).requires_grad_(k == "convs.5.conv.weight") for k, v in sd.items()}
img_r, _ = orc.generator_forward(sd64, lat.to(f64), mask.to(f64), [n.to(f64) for n in noise], size, Kk)
(img_r * w_img.to(f64)).sum().backward()
ref_full = sd64["G.convs.5.conv.weight"].grad
print("layer-level ref vs full ref:", float((ref_layer - ref_full).abs().max()))
print("mine vs full ref:", float((mine - ref_full).abs().max()), "img parity", float((img.cpu().double() - img_r).abs().max()))
# ---- step-by-step oracle to expose x5 / y5 / dy5
sdd = {"G." + k: v.to(f64) for k, v in sd.items()}
L = lat.to(f64); M = mask.to(f64); N = [n.to(f64) for n in noise]
p = "G."
x = sdd[p + "input.input"].repeat(b, 1, 1, 1)
x = orc.styled_conv(sdd, p + "conv1.", x, L[:, :, 0], M, N[0], False, True)
skip = orc.to_rgb(sdd, p + "to_rgb1.", x, L[:, :, 1], M, None, True)
x = orc.styled_conv(sdd, p + "convs.0.", x, L[:, :, 1], M, N[1], True, True)
x = orc.styled_conv(sdd, p + "convs.1.", x, L[:, :, 2], M, N[2], False, True)
skip = orc.to_rgb(sdd, p + "to_rgbs.0.", x, L[:, :, 3], M, skip, True)
x = orc.styled_conv(sdd, p + "convs.2.", x, L[:, :, 3], M, N[3], True, True)
x = orc.styled_conv(sdd, p + "convs.3.", x, L[:, :, 4], M, N[4], False, True)
skip = orc.to_rgb(sdd, p + "to_rgbs.1.", x, L[:, 0, 5], M, skip, False)
x4 = orc.styled_conv(sdd, p + "convs.4.", x, L[:, 0, 5], M, N[5], True, False)
x4 = x4.detach().requires_grad_(True)
y5 = orc.styled_conv(sdd, p + "convs.5.", x4, L[:, 0, 6], M, N[6], False, False)
y5.retain_grad()
img2 = orc.to_rgb(sdd, p + "to_rgbs.2.", y5, L[:, 0, 7], M, skip.detach(), False)
(img2 * w_img.to(f64)).sum().backward()
print("img2 vs img_r", float((img2 - img_r).abs().max()))
print("x5: mine vs oracle", float((K.nhwc_to_nchw(rec["x"]).cpu().double() - x4).abs().max()), float(x4.abs().max()))
print("dy5: mine vs oracle", float((dy - y5.grad).abs().max()), float(y5.grad.abs().max()))
