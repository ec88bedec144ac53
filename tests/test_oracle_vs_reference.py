"""CPU, build container only (needs /root/reference): layer-level checks of the oracle restatement against
the REAL reference modules on shapes the golden fixtures do not cover."""
import pytest
import torch

from e4s_amd import synth
from oracle import e4s_oracle as orc
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not mounted")


@pytest.mark.parametrize("cin,cout,up", [(16, 24, False), (16, 8, True)])
@torch.no_grad()
def test_styled_conv_masked_matches_reference_module(cin, cout, up):
    ns = ref_shim.reference_modules()
    m = ns.StyledConv(cin, cout, 3, 512, upsample=up, mask_op=True)
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.5 + (1.0 if p.ndim == 1 and p.numel() == cin else 0.0))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, cin, 8, 8, generator=g)
    style = torch.randn(2, 12, 512, generator=g)
    mask = synth.onehot(synth.synth_labels_blocks(2, 64, 8, seed=1))
    res = 16 if up else 8
    noise = torch.randn(2, 1, res, res, generator=g)
    want = m(x, style, mask, noise=noise)
    got = orc.styled_conv(sd, "", x, style, mask, noise, up, True)
    assert float((got - want).abs().max()) < 1e-5


@torch.no_grad()
def test_torgb_matches_reference_module():
    ns = ref_shim.reference_modules()
    m = ns.ToRGB(16, 512, mask_op=True)
    g = torch.Generator().manual_seed(1)
    for p in m.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 16, 8, 8, generator=g)
    style = torch.randn(2, 12, 512, generator=g)
    mask = synth.onehot(synth.synth_labels_blocks(2, 64, 8, seed=2))
    skip = torch.randn(2, 3, 4, 4, generator=g)
    want = m(x, style, mask, skip)
    got = orc.to_rgb(sd, "", x, style, mask, skip, True)
    assert float((got - want).abs().max()) < 1e-5


def test_w_norm_loss_matches_reference_class():
    """train.w_norm_loss == src/criteria/w_norm.py:WNormLoss (loaded by file path), value and gradient."""
    import importlib.util
    import os
    from e4s_amd.train import w_norm_loss
    sp = importlib.util.spec_from_file_location("ref_w_norm", os.path.join(ref_shim.REF if hasattr(ref_shim, "REF") else "/root/reference",
                                                                          "src", "criteria", "w_norm.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2, 12, 18, 512, generator=g, requires_grad=True)
    lat2 = lat.detach().clone().requires_grad_(True)
    avg = torch.randn(18, 512, generator=g)
    for flag in (True, False):
        a = w_norm_loss(lat, avg, flag)
        b = mod.WNormLoss(start_from_latent_avg=flag)(lat2, avg)
        assert float((a - b).abs()) < 1e-5 * float(b.abs())
    a.backward()
    b.backward()
    assert torch.allclose(lat.grad, lat2.grad)


def test_reference_itself_fails_on_the_branches_net3_refuses():
    """e4s_amd.networks.Net3.cal_style_codes raises NotImplementedError for start_from_latent_avg=False and learn_in_w=True; so does the
    reference, less politely (networks.py:145-158: `style_codes` is never assigned; a [bs,R,512] tensor is added to [bs,R,K,512] codes)."""
    sd = synth.synth_state_dict(256, 13)
    lat = synth.synth_latent_avg(256)
    net = ref_shim.build_reference_net3(sd, lat, 256, 13)
    sv = torch.randn(1, 12, 1280)
    net.opts.start_from_latent_avg = False
    with pytest.raises(UnboundLocalError):
        net.cal_style_codes(sv)
    net.opts.start_from_latent_avg, net.opts.learn_in_w = True, True
    net.latent_avg = lat[:1].repeat(1, 1)                 # coach.py:118-119
    with pytest.raises(RuntimeError):
        net.cal_style_codes(sv)
