"""CPU: the oracle restatement (oracle/e4s_oracle.py) against the golden vectors produced by the
REAL reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import torch

from e4s_amd import synth
from oracle import e4s_oracle as orc


def test_ops_kat_and_random(golden):
    g = golden("ops.pt")
    assert len(g["kat"]) == 5
    for case in g["kat"] + g["rand"]:
        if case["op"] == "upfirdn2d":
            y = orc.upfirdn2d(case["x"], case["k"], up=case["up"], down=case["down"], pad=case["pad"])
        else:
            y = orc.fused_leaky_relu(case["x"], case["b"])
            y2 = orc.fused_bias_act(case["x"], case["b"], None, 3, 0, 0.2, 2 ** 0.5)
            assert torch.allclose(y, y2, atol=1e-6)
        assert y.shape == case["y"].shape
        assert torch.allclose(y, case["y"], atol=1e-5, rtol=1e-5), case["op"]


def test_kat_literals(golden):
    """The five SURVEY.md 8(c) vectors, spelled out (guards the fixture file itself)."""
    k = golden("ops.pt")["kat"]
    assert torch.allclose(k[0]["y"].flatten()[:4], torch.tensor([0.5625, 0.9375, 1.3125, 1.125]))
    assert torch.allclose(k[1]["y"].flatten(), torch.tensor([11.8125, 13.5625, 17.0625, 18.8125]))
    assert torch.allclose(k[2]["y"].flatten(), torch.tensor([3.5, 4.703125, 8.3125, 9.515625]))
    assert torch.allclose(k[3]["y"].flatten(), torch.tensor([1.0, 4.0, 6.0, 20.0]))
    assert torch.allclose(k[4]["y"].flatten(), torch.tensor([-0.141421, -0.141421, 3.535534]), atol=1e-5)


@torch.no_grad()
def _run_oracle_swap(out_size, mask_kind):
    K = 13
    sd = synth.synth_state_dict(out_size, K)
    lat = synth.synth_latent_avg(out_size)
    driven = synth.synth_image(1, 1024, tag="driven")
    target = synth.synth_image(1, 1024, tag="target")
    mk = synth.synth_labels_face if mask_kind == "face" else (lambda b, s, seed: synth.synth_labels_blocks(b, s, 64, seed=seed))
    dm, tm, sm = (synth.onehot(mk(1, 512, seed=s)) for s in (1, 2, 3))
    noise = synth.synth_noise(out_size)
    d_sv, _ = orc.get_style_vectors(sd, driven, dm)
    t_sv, _ = orc.get_style_vectors(sd, target, tm)
    sv = orc.swap_style_vectors(t_sv, d_sv)
    codes = orc.cal_style_codes(sd, sv, lat, K)
    img, feats = orc.gen_img(sd, codes, sm, noise, out_size, K)
    return d_sv, t_sv, sv, codes, img, feats


def test_net256_swap_matches_reference(golden):
    g = golden("net256.pt")
    d_sv, t_sv, sv, codes, img, feats = _run_oracle_swap(256, "blocks")
    assert torch.allclose(d_sv, g["driven_sv"], atol=2e-5)
    assert torch.allclose(t_sv, g["target_sv"], atol=2e-5)
    assert torch.equal(sv == 0, g["swapped_sv"] == 0)          # exact zeros for empty regions
    assert torch.allclose(codes[:, :, :, ::8], g["codes_stride"], atol=1e-4, rtol=1e-5)
    assert torch.allclose(feats[:, ::4], g["feats_stride"], atol=1e-4)
    assert float((img - g["img"]).abs().max()) < 1e-4           # fp32 vs fp32, same ATen kernels


def test_net1024_swap_matches_reference(golden):
    g = golden("net1024.pt")
    d_sv, t_sv, sv, codes, img, feats = _run_oracle_swap(1024, "face")
    assert torch.allclose(d_sv, g["driven_sv"], atol=2e-5)
    assert torch.allclose(sv, g["swapped_sv"], atol=2e-5)
    assert torch.allclose(codes[:, :, :, ::8], g["codes_stride"], atol=1e-4, rtol=1e-5)
    assert float((img[:, :, ::8, ::8] - g["img_stride8"]).abs().max()) < 1e-4
    c0 = 1024 // 2 - 64
    assert float((img[:, :, c0:c0 + 128, c0:c0 + 128] - g["img_crop"]).abs().max()) < 1e-4
    assert torch.allclose(img.mean((2, 3)), g["img_mean"], atol=1e-5)


def test_discriminator_matches_reference(golden):
    """Discriminator(64) of the real reference (model.py:740-799) on a seeded batch of 4."""
    g = golden("disc64.pt")
    sd = synth.synth_disc_state_dict(64)
    x = synth.synth_image(4, 64, tag="disc")
    with torch.no_grad():
        y = orc.discriminator_forward(sd, x, 64)
    assert tuple(y.shape) == (4, 1)
    assert float((y - g["logits"]).abs().max()) < 1e-5


def test_gpen_full_generator_matches_reference(golden):
    """SURVEY.md 8(f) N2 (next row): GPEN FullGenerator(64, narrow=0.25) of the real reference
    (src/pretrained/gpen/face_model/gpen_model.py:628-690) -- the oracle for the widening step is pinned before any
    HIP code for it exists."""
    g = golden("gpen64.pt")
    c = g["cfg"]
    sd = synth.synth_gpen_state_dict(c["size"], n_mlp=c["n_mlp"], narrow=c["narrow"])
    x = synth.synth_image(2, c["size"], tag="gpen")
    with torch.no_grad():
        y, w = orc.gpen_full_generator(sd, x, c["size"], c["n_mlp"])
    assert tuple(y.shape) == (2, 3, c["size"], c["size"]) and tuple(w.shape) == (2, 512)
    assert float((y - g["img"]).abs().max()) < 1e-5


def test_loss_networks_match_reference(golden):
    """the oracle's IDLoss / LPIPS restatement against the reference's own classes run on the same
    seeded weights (tests/golden/make_golden.py:criteria_case): loss values, feature heads, image gradients."""
    import types
    from e4s_amd import criteria as C
    gold = golden("criteria.pt")
    sd = synth.synth_module_state_dict(C.IDLoss(types.SimpleNamespace()), 0, "id.")
    yh, y = synth.synth_image_pair(2, 256, seed=3)
    yh.requires_grad_(True)
    loss, imp = orc.id_loss(sd, yh, y)
    loss.backward()
    g = gold["id256"]
    assert abs(float(loss) - float(g["loss"])) < 1e-5 and abs(float(imp) - g["improvement"]) < 1e-5
    with torch.no_grad():
        x112 = torch.nn.functional.adaptive_avg_pool2d(y[:, :, 35:223, 32:220], (112, 112))
        for f, ref in zip(orc.irse50_features(sd, x112), g["feat_heads"]):
            assert float((f[:, :64] - ref).abs().max()) < 1e-5
    d = yh.grad[:, :, ::4, ::4] - g["grad_strided"]
    assert float(d.norm() / g["grad_strided"].norm()) < 5e-3             # fp32 vs fp32: PReLU sides flip on ~1e-7 inputs
    sdl = synth.synth_module_state_dict(C.LPIPS(), 0, "lp.")
    yh, y = synth.synth_image_pair(2, 256, seed=4)
    yh.requires_grad_(True)
    loss = orc.lpips(sdl, yh, y)
    loss.backward()
    g = gold["lpips256"]
    assert abs(float(loss) / float(g["loss"]) - 1.0) < 1e-5
    with torch.no_grad():
        for f, ref in zip(orc.alexnet_features(sdl, y), g["feat_heads"]):
            assert float((f[:, :8, :4, :4] - ref).abs().max()) < 1e-6
    d = yh.grad[:, :, ::4, ::4] - g["grad_strided"]
    assert float(d.norm() / g["grad_strided"].norm()) < 1e-4
    sdp = synth.synth_module_state_dict(C.FaceParsingLoss(types.SimpleNamespace()), 0, "fp.")
    yh, y = synth.synth_image_pair(1, 512, seed=6)
    yh.requires_grad_(True)
    loss, imp = orc.face_parsing_loss(sdp, yh, y)
    loss.backward()
    g = gold["parsing512"]
    assert abs(float(loss) - float(g["loss"])) < 1e-5 and abs(float(imp) - g["improvement"]) < 1e-5
    with torch.no_grad():
        for f, ref in zip(orc.unet_encoder_features(sdp, y), g["feat_heads"]):
            assert float((f[:, :64] - ref).abs().max()) < 1e-6
    d = yh.grad[:, :, ::8, ::8] - g["grad_strided"]
    assert float(d.norm() / g["grad_strided"].norm()) < 1e-3


def test_opencv_restatements_vs_scipy_pil_golden(golden):
    """The oracle's restatements of cv2.erode / GaussianBlur (CV_8U fixed point) / pyrDown / pyrUp and of multi_band_blending's schedule
    (oracle/e4s_oracle.py; scripts/face_swap.py:81-97, src/utils/multi_band_blending.py:4-75) against INDEPENDENT
    third-party implementations -- scipy.ndimage and PIL, tests/golden/make_cv2_free_golden.py -- on crops of the reference's example
    images and parsing maps.  cv2 itself exists neither here nor on the GPU box: this is "cross-checked against scipy / PIL", not
    "pinned to cv2".  Bounds: erode, uint8 pyrDown exact; fixed-point Gaussian within 1 LSB of the float Gaussian; float pyramids within
    2 ulp of 255 (3.1e-5: the float32 summation order vs float64); the 6-level Laplacian blend within 1 LSB."""
    import numpy as np
    g = golden("cv2free.pt")
    inp = {k: v.numpy() for k, v in g["inputs"].items()}
    want = {k: v.numpy() for k, v in g["scipy"].items()}
    try:                                               # scipy / PIL are importable: recompute and require the committed file to be current
        import importlib.util
        import os
        spec = importlib.util.spec_from_file_location("make_cv2_free_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                                         "make_cv2_free_golden.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        live = mod.compute(inp)
        for k, v in want.items():
            assert np.allclose(live[k].astype(np.float64), v.astype(np.float64), atol=1e-4), k
    except ImportError:
        pass
    for name in ("mask_a", "mask_b"):
        e = orc.cv2_erode_u8(torch.from_numpy(inp[name])[None], 5, 255)
        assert np.array_equal(e[0].numpy(), want["erode_" + name]), name
        assert 0.02 < float((e[0] != torch.from_numpy(inp[name])).float().mean())           # the erosion really moved the boundary
        gb = orc.cv2_gaussian_blur_u8(e, 11)[0].numpy().astype(np.float64)
        assert np.abs(gb - want["gauss_" + name]).max() < 1.0, name
    for name in ("img_a", "img_b"):
        gb = orc.cv2_gaussian_blur_u8(torch.from_numpy(inp[name][..., 0].copy())[None], 11)[0].numpy().astype(np.float64)
        assert np.abs(gb - want["gauss_" + name]).max() < 1.0, name
        assert np.array_equal(orc.cv2_pyrdown(inp[name]), want["pyrdown_u8_" + name]), name
        f = inp[name].astype(np.float32) * np.float32(0.731) + np.float32(3.3)
        assert np.abs(orc.cv2_pyrdown(f) - want["pyrdown_f_" + name]).max() <= 3.1e-5, name
        assert np.abs(orc.cv2_pyrup(f[: f.shape[0] // 2, : f.shape[1] // 2]) - want["pyrup_f_" + name]).max() <= 3.1e-5, name
    m3 = np.repeat(inp["blend_mask"][:, :, None], 3, axis=2)
    bl = orc.laplacian_blend_u8(inp["blend_full"], inp["blend_ori"], m3, num_levels=6)
    d = np.abs(bl.astype(np.int64) - want["blend"].astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
