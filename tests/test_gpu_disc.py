"""GPU parity of the Discriminator under autograd (config 5: src/criteria/adv_loss.py:8-60): logits,
first-order gradients (image + every parameter) and the R1 penalty's second-order gradients against the oracle's fp64
autograd, plus the closed Function families at layer level."""
import math

import pytest
import torch
import torch.nn.functional as F

from e4s_amd import synth
from oracle import e4s_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _f32(monkeypatch):
    from e4s_amd import kernels as K
    monkeypatch.setattr(K, "PRECISION", "f32")


def _ref_conv(x, w, kind):
    if kind == "s1":
        return F.conv2d(x, w, padding=1)
    return F.conv2d(x, w, stride=2)


@pytest.mark.parametrize("kind,cin,cout,h", [("s1", 64, 96, 12), ("s2k3", 64, 64, 17), ("s2k1", 96, 64, 15), ("s1", 544, 512, 4),
                                             ("s2k3", 32, 64, 33)])
def test_conv_family_first_and_second_order(kind, cin, cout, h):
    """Conv / ConvDgrad / ConvWgrad (disc_autograd.py): y, dx, dW, and the gradients of a function of (dx, dW) -- i.e. every
    backward of the three Functions -- against fp64 autograd of F.conv2d."""
    from e4s_amd import kernels as K
    from e4s_amd.disc_autograd import Conv
    k = 1 if kind == "s2k1" else 3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, cin, h, h + 1, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = _ref_conv(x64, w64, kind)
    gy = torch.randn(y64.shape, generator=g)
    px, pw = torch.randn(x.shape, generator=g), torch.randn(w.shape, generator=g)
    dx64, dw64 = torch.autograd.grad(y64, (x64, w64), gy.double(), create_graph=True)
    s64 = (dx64 * px.double()).sum() + 0.5 * (dx64 ** 2).sum() + (dw64 * pw.double()).sum()
    ggx64, ggw64 = torch.autograd.grad(s64, (x64, w64))

    xd = K.nchw_to_nhwc(x.to(DEV)).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = Conv.apply(xd, wd, kind)
    assert rel_l2(y.permute(0, 3, 1, 2), y64) < 1e-5
    gyd = K.nchw_to_nhwc(gy.to(DEV))
    dx, dw = torch.autograd.grad(y, (xd, wd), gyd, create_graph=True)
    assert rel_l2(dx.permute(0, 3, 1, 2), dx64) < 1e-5 and rel_l2(dw, dw64) < 1e-5
    pxd = K.nchw_to_nhwc(px.to(DEV))
    s = (dx * pxd).sum() + 0.5 * (dx ** 2).sum() + (dw * pw.to(DEV)).sum()
    ggx, ggw = torch.autograd.grad(s, (xd, wd))
    assert rel_l2(ggx.permute(0, 3, 1, 2), ggx64) < 2e-5
    assert rel_l2(ggw, ggw64) < 2e-5


def test_stem_linear_blur_biasact_families():
    from e4s_amd import kernels as K
    from e4s_amd.disc_autograd import BiasAct, BlurNHWC, Lin, Stem
    g = torch.Generator().manual_seed(4)
    # stem: 1x1 conv of the NCHW image
    img = torch.randn(2, 3, 16, 16, generator=g)
    w = torch.randn(64, 3, generator=g)
    i64, w64 = img.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(i64, w64[:, :, None, None])
    gy = torch.randn(y64.shape, generator=g)
    di64, dw64 = torch.autograd.grad(y64, (i64, w64), gy.double(), create_graph=True)
    gg64 = torch.autograd.grad((di64 ** 2).sum() + (dw64 ** 3).sum(), (i64, w64))
    imd, wd = img.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = Stem.apply(imd, wd)
    assert rel_l2(y.permute(0, 3, 1, 2), y64) < 1e-5
    di, dw = torch.autograd.grad(y, (imd, wd), K.nchw_to_nhwc(gy.to(DEV)), create_graph=True)
    assert rel_l2(di, di64) < 1e-5 and rel_l2(dw, dw64) < 1e-5
    gg = torch.autograd.grad((di ** 2).sum() + (dw ** 3).sum(), (imd, wd))
    assert rel_l2(gg[0], gg64[0]) < 2e-5 and rel_l2(gg[1], gg64[1]) < 2e-5
    # linear (incl. a single output row)
    for o in (512, 1):
        x = torch.randn(4, 1024, generator=g)
        wl = torch.randn(o, 1024, generator=g) / 32
        x64, wl64 = x.double().requires_grad_(True), wl.double().requires_grad_(True)
        y64 = F.linear(x64, wl64)
        gy = torch.randn(y64.shape, generator=g)
        dx64, dwl64 = torch.autograd.grad(y64, (x64, wl64), gy.double(), create_graph=True)
        gg64 = torch.autograd.grad((dx64 ** 2).sum() + (dwl64 ** 2).sum(), (x64, wl64))
        xd, wld = x.to(DEV).requires_grad_(True), wl.to(DEV).requires_grad_(True)
        y = Lin.apply(xd, wld)
        assert rel_l2(y, y64) < 1e-5
        dx, dwl = torch.autograd.grad(y, (xd, wld), gy.to(DEV), create_graph=True)
        assert rel_l2(dx, dx64) < 1e-5 and rel_l2(dwl, dwl64) < 1e-5
        gg = torch.autograd.grad((dx ** 2).sum() + (dwl ** 2).sum(), (xd, wld))
        assert rel_l2(gg[0], gg64[0]) < 2e-5 and rel_l2(gg[1], gg64[1]) < 2e-5
    # blur (pad modes of model.py:683-689) and bias + leaky ReLU, twice
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = (k1[None] * k1[:, None]) / 64.0
    for pad in ((2, 2), (1, 1)):
        x = torch.randn(2, 32, 12, 12, generator=g)
        x64 = x.double().requires_grad_(True)
        y64 = orc.upfirdn2d(x64, k2.double(), pad=pad)
        gy = torch.randn(y64.shape, generator=g)
        (dx64,) = torch.autograd.grad(y64, x64, gy.double())
        xd = K.nchw_to_nhwc(x.to(DEV)).requires_grad_(True)
        y = BlurNHWC.apply(xd, k2.to(DEV), pad)
        assert rel_l2(y.permute(0, 3, 1, 2), y64) < 1e-5
        (dx,) = torch.autograd.grad(y, xd, K.nchw_to_nhwc(gy.to(DEV)))
        assert rel_l2(dx.permute(0, 3, 1, 2), dx64) < 1e-5
    x = torch.randn(2, 6, 5, 64, generator=g)
    b = torch.randn(64, generator=g)
    x64, b64 = x.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = F.leaky_relu(x64 + b64, 0.2) * math.sqrt(2)
    gy = torch.randn(y64.shape, generator=g)
    dx64, db64 = torch.autograd.grad(y64, (x64, b64), gy.double(), create_graph=True)
    xd, bd = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = BiasAct.apply(xd, bd, 0.2, math.sqrt(2))
    dx, db = torch.autograd.grad(y, (xd, bd), gy.to(DEV), create_graph=True)
    assert rel_l2(y, y64) < 1e-6 and rel_l2(dx, dx64) < 1e-6 and rel_l2(db, db64) < 1e-5
    # d/d(gy) of <dx, p> is the same piecewise-linear map applied to p
    gyd = gy.to(DEV).requires_grad_(True)
    (dx2,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
    p = torch.randn(x.shape, generator=g)
    (dgy,) = torch.autograd.grad((dx2 * p.to(DEV)).sum(), gyd)
    ref = p.double() * torch.where(y64 > 0, 1.0, 0.2) * math.sqrt(2)
    assert rel_l2(dgy, ref) < 1e-6


def _disc(size):
    from e4s_amd.stylegan2 import Discriminator
    sd = synth.synth_disc_state_dict(size)
    d = Discriminator(size)
    d.load_state_dict(sd, strict=True)
    return d.to(DEV), sd


def test_discriminator_autograd_first_order_vs_oracle_f64(golden):
    d, sd = _disc(64)
    x = synth.synth_image(4, 64, tag="disc")
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and "kernel" not in k) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    out64 = orc.discriminator_forward(sd64, x64, 64)
    loss64 = F.softplus(-out64).mean()                                   # g_nonsaturating_loss / d_logistic's fake term
    loss64.backward()
    xd = x.to(DEV).requires_grad_(True)
    out = d(xd)
    assert float((out.detach().cpu() - golden("disc64.pt")["logits"]).abs().max()) < 1e-4     # the reference's own logits
    assert rel_l2(out, out64) < 1e-5
    F.softplus(-out).mean().backward()
    # fp32 vs fp64: a handful of leaky-ReLU inputs within 1e-7 of zero take the other slope (measured 5.7e-4 on x.grad)
    assert rel_l2(xd.grad, x64.grad) < 2e-3
    for name, p in d.named_parameters():
        assert p.grad is not None, name
        assert rel_l2(p.grad, sd64[name].grad) < 2e-3, name


def test_discriminator_r1_penalty_second_order_vs_oracle_f64():
    """d_r1_loss (adv_loss.py:48-60): grad of D(real).sum() w.r.t. the image with create_graph, squared norm, then the
    gradient of that penalty w.r.t. every Discriminator parameter."""
    d, sd = _disc(32)
    x = synth.synth_image(4, 32, tag="r1")
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and "kernel" not in k) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    (g64,) = torch.autograd.grad(orc.discriminator_forward(sd64, x64, 32).sum(), x64, create_graph=True)
    pen64 = g64.pow(2).reshape(4, -1).sum(1).mean()
    pen64.backward()
    xd = x.to(DEV).requires_grad_(True)
    (gr,) = torch.autograd.grad(d(xd).sum(), xd, create_graph=True)
    assert rel_l2(gr, g64) < 1e-4
    pen = gr.pow(2).reshape(4, -1).sum(1).mean()
    assert abs(float(pen) / float(pen64) - 1.0) < 1e-4
    pen.backward()
    checked = 0
    for name, p in d.named_parameters():
        ref = sd64[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:                # biases do not enter dD/dx's derivative
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, name
            continue
        assert rel_l2(p.grad, ref) < 5e-4, name
        checked += 1
    assert checked >= 10
