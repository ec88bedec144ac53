"""GPU parity of the GPEN FullGenerator and of the Discriminator's native forward (8(a) a14) on the
HIP kernels, against the REAL reference's outputs (tests/golden/gpen64.pt, disc64.pt) and the CPU oracle."""
import pytest
import torch

from e4s_amd import synth
from oracle import e4s_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def maxabs(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@torch.no_grad()
def test_gpen_full_generator_vs_reference_golden(golden, monkeypatch, precision):
    from e4s_amd import kernels as K
    from e4s_amd.gpen import FullGenerator
    monkeypatch.setattr(K, "PRECISION", precision)
    g = golden("gpen64.pt")
    c = g["cfg"]
    sd = synth.synth_gpen_state_dict(c["size"], n_mlp=c["n_mlp"], narrow=c["narrow"])
    net = FullGenerator(c["size"], 512, c["n_mlp"], channel_multiplier=2, narrow=c["narrow"])
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    x = synth.synth_image(2, c["size"], tag="gpen")
    img, lat = net(x.to(DEV))
    assert lat is None and tuple(img.shape) == tuple(g["img"].shape)
    scale = float(g["img"].abs().max())
    err = maxabs(img, g["img"])
    print(f"gpen64 {precision}: max-abs {err:.3e} (output scale {scale:.2f})")
    assert err < (1e-4 if precision == "f32" else 1e-3) * max(1.0, scale)
    img2, lat2 = net(x.to(DEV), return_latents=True)
    assert torch.equal(img2, img) and tuple(lat2.shape) == (2, net.generator.n_latent, 512)


@torch.no_grad()
def test_gpen_512_vs_oracle(monkeypatch):
    """The shipped configuration (GPEN-BFR-512: size 512, channel_multiplier 2, narrow 1, face_enhancement.py:34-37) on
    one image, default precision policy, against the CPU oracle at the north-star tolerance."""
    from e4s_amd import kernels as K
    from e4s_amd.gpen import FullGenerator
    monkeypatch.setattr(K, "PRECISION", "auto")
    sd = synth.synth_gpen_state_dict(512, n_mlp=8)
    net = FullGenerator(512, 512, 8, channel_multiplier=2, narrow=1)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    x = synth.synth_image(1, 512, tag="gpen512")
    img, _ = net(x.to(DEV))
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    want, _ = orc.gpen_full_generator(sd, x, 512, 8)
    scale = float(want.abs().max())
    err = maxabs(img, want)
    print(f"gpen512: max-abs {err:.3e} (output scale {scale:.2f})")
    assert tuple(img.shape) == (1, 3, 512, 512) and err < 1e-3 * max(1.0, scale)


@torch.no_grad()
def test_discriminator_native_forward_vs_golden_and_aten(golden, monkeypatch):
    from e4s_amd import kernels as K
    from e4s_amd.stylegan2 import Discriminator
    monkeypatch.setattr(K, "PRECISION", "f32")
    g = golden("disc64.pt")
    d = Discriminator(64)
    d.load_state_dict(synth.synth_disc_state_dict(64), strict=True)
    d = d.to(DEV).eval()
    x = synth.synth_image(4, 64, tag="disc").to(DEV)
    logits = d(x)                                          # no grad -> the native schedule
    assert maxabs(logits, g["logits"]) < 1e-4 * max(1.0, float(g["logits"].abs().max()))
    # a bigger instance (256^2: 128..512 channels, the stride-2 gather kernels incl. split-bf16) vs the ATen path
    d2 = Discriminator(256)
    d2.load_state_dict(synth.synth_disc_state_dict(256), strict=True)
    d2 = d2.to(DEV).eval()
    x2 = synth.synth_image(8, 256, tag="disc256").to(DEV)
    with torch.enable_grad():
        ref = d2(x2.clone().requires_grad_(True)).detach()  # autograd requested -> conv2d_gradfix / ATen
    nat = d2.forward_native(x2)
    assert maxabs(nat, ref) < 1e-4 * max(1.0, float(ref.abs().max()))
    monkeypatch.setattr(K, "PRECISION", "bf16x3")
    nat3 = d2.forward_native(x2)
    assert 0 < maxabs(nat3, nat) < 1e-3 * max(1.0, float(ref.abs().max()))
