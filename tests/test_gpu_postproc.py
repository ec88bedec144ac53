"""GPU parity of the device pre/post-processing kernels (e4s_amd/postproc.py) against outputs of the
REAL reference on its own example parsing maps (tests/golden/realmask.pt: swap_face_mask.py, face_swap.py:create_masks,
morphology.py, torch_utils.tensor2im) and against the CPU restatements in oracle/e4s_oracle.py.  Integer / comparison /
single-rounding arithmetic: everything here is bit-exact (bar the bilinear resize inside `paste`, see there)."""
import pytest
import torch

from conftest import unz
from e4s_amd import synth
from oracle import e4s_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_mask_swap_onehot_and_blend_masks_vs_reference_outputs(golden):
    from e4s_amd import postproc as PP
    g = golden("realmask.pt")
    d, t = unz(g["D_mask"]).to(DEV), unz(g["T_mask"]).to(DEV)
    swapped, hole = PP.swap_head_mask_revisit_considerGlass(d, t)
    assert torch.equal(swapped.cpu(), unz(g["swapped_mask"]))
    assert torch.equal((hole.cpu() != 0).to(torch.uint8), unz(g["hole_map"])) and set(hole.unique().tolist()) <= {0, 255}
    oh = PP.labelMap2OneHot(swapped[None, None], 12)
    assert torch.equal(oh.cpu(), synth.onehot(unz(g["swapped_mask"]).long()[None, None]))
    oh64 = PP.labelMap2OneHot(swapped[None, None].long(), 12)                   # int64 ids, as the scripts pass them
    assert torch.equal(oh64, oh)
    fg = PP.foreground_mask(swapped, hole)
    assert torch.equal(fg.cpu(), unz(g["foreground"]).float())
    for op in ("dilation", "expansion", "erosion"):
        content, border, full = PP.create_masks(fg[None, None], outer_dilation=5, operation=op)
        assert torch.equal(content[0, 0], fg)
        assert torch.equal(border[0, 0].cpu(), unz(g["masks_" + op]["border"]).float()), op
        assert torch.equal(full[0, 0].cpu(), unz(g["masks_" + op]["full"]).float()), op


@pytest.mark.parametrize("b,c,h,w,r", [(2, 3, 40, 70, 2), (1, 1, 33, 9, 5), (1, 2, 8, 8, 0), (1, 1, 64, 64, 16)])
def test_morphology_on_grey_images_vs_oracle(b, c, h, w, r):
    from e4s_amd import postproc as PP
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, c, h, w, generator=g)
    k = torch.ones(2 * r + 1, 2 * r + 1)
    assert torch.equal(PP.dilation(x.to(DEV), k.to(DEV)).cpu(), orc.morph_flat(x, r, "dilation"))
    assert torch.equal(PP.erosion(x.to(DEV), k.to(DEV)).cpu(), orc.morph_flat(x, r, "erosion"))
    rnd = (torch.rand(b, c, h, w, generator=g) > 0.6).float()
    for op in ("dilation", "erosion", "expansion"):
        got = PP.create_masks(rnd.to(DEV), r, op)
        want = orc.create_masks(rnd, r, op)
        assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[2].cpu(), want[2])
    with pytest.raises(NotImplementedError):
        PP.dilation(x.to(DEV), torch.tensor([[0., 1., 0.], [1., 1., 1.], [0., 1., 0.]], device=DEV))


def test_mask_swap_all_label_pairs_vs_oracle():
    from e4s_amd import postproc as PP
    s, t = torch.meshgrid(torch.arange(12), torch.arange(12), indexing="ij")
    s, t = s.reshape(1, -1).to(torch.uint8), t.reshape(1, -1).to(torch.uint8)
    got, hole = PP.swap_head_mask_revisit_considerGlass(s.to(DEV), t.to(DEV))
    want, whole = orc.swap_head_mask(s.long(), t.long())
    assert torch.equal(got.cpu().long(), want) and torch.equal(hole.cpu().long(), whole)


def test_tensor2im_and_paste_bit_exact():
    from e4s_amd import postproc as PP
    g = torch.Generator().manual_seed(4)
    img = torch.randn(2, 3, 96, 128, generator=g) * 0.8
    img[0, :, :2] = torch.tensor([-1.0, 1.0, 0.99999994, -0.99999994, 1.5, -2.0, 0.0, 1e-8]).repeat(16)[None, None, :128]
    u8 = PP.tensor2im(img.to(DEV))
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (2, 96, 128, 3)
    assert torch.equal(u8.cpu(), orc.tensor2im_u8(img))
    tgt = torch.randint(0, 256, (2, 96, 128, 3), generator=g, dtype=torch.uint8)
    mask = torch.rand(2, 1, 48, 64, generator=g)
    mask[:, :, :8] = 1.0
    mask[:, :, -8:] = 0.0
    got = PP.paste(u8, tgt.to(DEV), mask.to(DEV)).cpu()
    want = orc.paste_u8(u8.cpu(), tgt, mask)
    # exact wherever the resized mask is exactly 0 or 1; elsewhere ATen's own bilinear association order / FMA use is
    # implementation-defined, so a value that lands within an ulp of an integer may truncate one step apart
    diff = (got.int() - want.int()).abs()
    assert int(diff.max()) <= 1 and float((diff == 0).float().mean()) > 0.999
    assert torch.equal(got[:, :12], u8.cpu()[:, :12]) and torch.equal(got[:, -12:], tgt[:, -12:])


@torch.no_grad()
def test_swap_image_to_uint8_vs_reference_tensor2im(golden):
    """The real-mask swap (tests/test_gpu_timed_path.py) converted on the device vs the reference's tensor2im of ITS
    image: the fp32 images differ by <= 1e-4, so a pixel may fall on the other side of a 1/255 step -- never by more."""
    from e4s_amd import kernels as K, postproc as PP
    from e4s_amd.networks import Net3, face_swap_core
    from e4s_amd.options import make_opts
    g = golden("realmask.pt")
    net = Net3(make_opts(out_size=1024))
    net.load_state_dict(synth.synth_state_dict(1024, 13), strict=True)
    net.latent_avg = synth.synth_latent_avg(1024).to(DEV)
    net = net.to(DEV).eval()
    lab = lambda z: unz(z).to(DEV)[None, None]
    dm, tm, sm = (PP.labelMap2OneHot(lab(g[k]), 12) for k in ("D_mask", "T_mask", "swapped_mask"))
    driven, target = synth.synth_image(1, 1024, tag="driven").to(DEV), synth.synth_image(1, 1024, tag="target").to(DEV)
    img = face_swap_core(net, driven, dm, target, tm, sm, noise=[x.to(DEV) for x in synth.synth_noise(1024)])
    u8 = PP.tensor2im(img)[0].cpu()
    y0, x0, want = g["img_u8_crop"]
    diff = (u8[y0:y0 + 128, x0:x0 + 128].int() - want.int()).abs()
    assert int(diff.max()) <= 1 and float((diff == 0).float().mean()) > 0.995
    assert abs(int(u8.long().sum()) - g["img_u8_sum"]) < 3e-5 * g["img_u8_sum"]


# ---- stitching: the default smooth_face_boundry path and the Laplacian blend ---------------------------------
def _blob_mask(b, h, w, seed):
    """binary {0,255} mask: a few random ellipses (face-like blobs touching the border in some samples)"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    out = torch.zeros(b, h, w, dtype=torch.uint8)
    for i in range(b):
        for _ in range(3):
            cy, cx = float(torch.rand(1, generator=g)) * h, float(torch.rand(1, generator=g)) * w
            ry, rx = (0.1 + 0.3 * float(torch.rand(1, generator=g))) * h, (0.1 + 0.3 * float(torch.rand(1, generator=g))) * w
            out[i][((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 < 1] = 255
    return out


@pytest.mark.parametrize("b,h,w,r", [(2, 64, 96, 5), (1, 37, 23, 5), (1, 16, 16, 2), (1, 1024, 1024, 5)])
def test_erode_gaussian_and_composite_bit_exact_vs_oracle(b, h, w, r):
    """cv2.erode / cv2.GaussianBlur (CV_8U fixed point) restated in the oracle, PIL's own alpha_composite: bit-exact."""
    from e4s_amd import postproc as PP
    m = _blob_mask(b, h, w, 5)
    g = torch.Generator().manual_seed(9)
    grey = torch.randint(0, 256, (b, h, w), generator=g, dtype=torch.uint8)                 # any uint8 image, not just masks
    er = PP.erode_u8(m.to(DEV), r, 255)
    assert torch.equal(er.cpu(), orc.cv2_erode_u8(m, r, 255))
    assert torch.equal(PP.erode_u8(grey.to(DEV), r, 0).cpu(), orc.cv2_erode_u8(grey, r, 0))
    for src in (er.cpu(), grey):
        assert torch.equal(PP.gaussian_blur_u8(src.to(DEV), 2 * r + 1).cpu(), orc.cv2_gaussian_blur_u8(src, 2 * r + 1))
    face = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8)
    tgt = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8)
    got = PP.smooth_face_boundry(face.to(DEV), tgt.to(DEV), m.to(DEV), radius=r)
    assert torch.equal(got.cpu(), orc.smooth_face_boundry(face, tgt, m, radius=r))            # PIL does the composite there
    got0 = PP.smooth_face_boundry(face.to(DEV), tgt.to(DEV), grey.to(DEV), radius=0)          # every alpha value 0..255
    assert torch.equal(got0.cpu(), orc.smooth_face_boundry(face, tgt, grey, radius=0))


def test_pyramids_vs_oracle():
    """cv2.pyrDown (uint8 and fp32) / cv2.pyrUp (fp32) restatements: bit-exact; odd sizes, 2x2."""
    from e4s_amd import postproc as PP
    g = torch.Generator().manual_seed(4)
    for (h, w) in ((32, 48), (17, 9), (2, 2), (4, 2)):
        u = torch.randint(0, 256, (2, h, w, 3), generator=g, dtype=torch.uint8)
        f = torch.rand(2, h, w, 3, generator=g) * 255
        du = PP.pyr_down(u.to(DEV)).cpu()
        df = PP.pyr_down(f.to(DEV)).cpu()
        upf = PP.pyr_up(f.to(DEV)).cpu()
        for i in range(2):
            assert torch.equal(du[i], torch.from_numpy(orc.cv2_pyrdown(u[i].numpy()))), (h, w)
            assert torch.equal(df[i], torch.from_numpy(orc.cv2_pyrdown(f[i].numpy()))), (h, w)
            # pyrUp: bit-exact since round 5 (round 4 saw <= 4 ulp: `__fadd_rn(a, __fmul_rn(b, 6.f))` had been contracted to
            # v_fmamk_f32 -- the header's inline intrinsics are outside the translation unit's `fp contract(off)`; csrc/stitch.hip)
            assert torch.equal(upf[i], torch.from_numpy(orc.cv2_pyrup(f[i].numpy()))), (h, w)
    const = torch.full((1, 8, 8, 3), 100, dtype=torch.uint8)
    assert bool((PP.pyr_up(PP.pyr_down(const.to(DEV)).float()) == 100).all())


def test_stitch_pipeline_both_modes_vs_oracle(golden):
    """scripts/face_swap.py:276-310 end to end on the reference's example parsing maps: foreground -> create_masks -> (default)
    mask image -> erode -> Gaussian -> alpha composite, bit-exact; (--lap_bld) paste -> 10-level Laplacian blend, +-1 LSB."""
    from e4s_amd import postproc as PP
    gold = golden("realmask.pt")
    swapped, hole = unz(gold["swapped_mask"]), unz(gold["hole_map"]) * 255
    b = 2
    lab = torch.stack([swapped, swapped.flip(-1)]).contiguous()
    hol = torch.stack([hole, hole.flip(-1)]).contiguous()
    face = synth.synth_image(b, 1024, tag="stitch_face")
    g = torch.Generator().manual_seed(12)
    tgt = torch.randint(0, 256, (b, 1024, 1024, 3), generator=g, dtype=torch.uint8)
    want = orc.stitch(face, tgt, lab, hol, lap_bld=False)
    got = PP.stitch(face.to(DEV), tgt.to(DEV), lab.to(DEV), hol.to(DEV), lap_bld=False)
    assert torch.equal(got.cpu(), want)
    changed = float((want != tgt).float().mean())
    assert 0.05 < changed < 0.95                                       # the face really was pasted, the background really kept
    want_l = orc.stitch(face[:1], tgt[:1], lab[:1], hol[:1], lap_bld=True)
    got_l = PP.stitch(face[:1].to(DEV), tgt[:1].to(DEV), lab[:1].to(DEV), hol[:1].to(DEV), lap_bld=True)
    diff = (got_l.cpu().int() - want_l.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3, (int(diff.max()), float((diff > 0).float().mean()))


def test_hip_stitch_kernels_vs_scipy_pil_golden(golden):
    """The HIP erode / fixed-point Gaussian / pyrDown / pyrUp kernels and the Laplacian blend against the THIRD-PARTY results of
    tests/golden/make_cv2_free_golden.py (scipy.ndimage + PIL on crops of the reference's example images; cv2 is in neither container):
    erode and uint8 pyrDown exact, Gaussian within 1 LSB of the float Gaussian, float pyramids within 2 ulp of 255, blend within 1 LSB."""
    import numpy as np
    from e4s_amd import postproc as PP
    g = golden("cv2free.pt")
    inp, want = g["inputs"], {k: v.numpy() for k, v in g["scipy"].items()}
    for name in ("mask_a", "mask_b"):
        e = PP.erode_u8(inp[name][None].to(DEV), 5, 255)
        assert np.array_equal(e[0].cpu().numpy(), want["erode_" + name]), name
        gb = PP.gaussian_blur_u8(e, 11)[0].cpu().numpy().astype(np.float64)
        assert np.abs(gb - want["gauss_" + name]).max() < 1.0, name
    for name in ("img_a", "img_b"):
        ch0 = inp[name][..., 0].contiguous()
        gb = PP.gaussian_blur_u8(ch0[None].to(DEV), 11)[0].cpu().numpy().astype(np.float64)
        assert np.abs(gb - want["gauss_" + name]).max() < 1.0, name
        assert np.array_equal(PP.pyr_down(inp[name][None].to(DEV))[0].cpu().numpy(), want["pyrdown_u8_" + name]), name
        f = inp[name].float() * 0.731 + 3.3
        assert np.abs(PP.pyr_down(f[None].to(DEV))[0].cpu().numpy() - want["pyrdown_f_" + name]).max() <= 3.1e-5, name
        half = f[: f.shape[0] // 2, : f.shape[1] // 2].contiguous()
        assert np.abs(PP.pyr_up(half[None].to(DEV))[0].cpu().numpy() - want["pyrup_f_" + name]).max() <= 3.1e-5, name
    m3 = inp["blend_mask"][:, :, None].expand(-1, -1, 3).contiguous()
    ls = PP.Laplacian_Pyramid_Blending_with_mask(inp["blend_full"][None].to(DEV), inp["blend_ori"][None].to(DEV), m3[None].to(DEV), 6)
    got = ls[0].cpu().numpy().clip(0, 255).astype(np.uint8)
    d = np.abs(got.astype(np.int64) - want["blend"].astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (int(d.max()), float((d > 0).mean()))
