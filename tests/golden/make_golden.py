"""Generate tests/golden/*.pt by running the REAL reference (from /root/reference) on CPU.

Run in the build container only:  python tests/golden/make_golden.py
The fixtures are small (strided / cropped outputs) and committed; the GPU box
has no /root/reference and only reads the .pt files.

What is pinned:
  ops.pt      known-answer vectors for upfirdn2d / fused_leaky_relu
              + seeded random cases over the (up, down, pad, kernel) modes the path uses
  net256.pt   Net3(out_size=256, K=13): style vectors, style codes, full 256^2 image, feats16
              (block-random mask: every region at every tile)
  net1024.pt  Net3(out_size=1024, K=13): the E4S-core swap (2x encoder, swap, MLPs, generator)
              on a face-like mask; image stored strided (::8) plus a full-res centre crop
  disc64.pt   Discriminator(64) (config 5): logits and the 4x4 feature map of a seeded batch of 4
  gpen64.pt   GPEN FullGenerator(64, narrow=0.25): restored image of a seeded batch of 2
  realmask.pt the 1024^2 swap on the reference's REAL example parsing maps (example/input/faceswap/{source,target}_mask.png,
              19 CelebAMask-HQ ids remapped to the 12 classes by src/datasets/dataset.py:153-209): thin brow / eye / lip /
              teeth regions.  Holds the label maps (zlib), the swapped mask + hole map of src/utils/swap_face_mask.py:33-82,
              the blending masks of scripts/face_swap.py:30-48,278-292 (src/utils/morphology.py dilation / erosion), the
              swap's style vectors and image (strided + crops), its uint8 image (tensor2im, torch_utils.py:63-69), and the
              gradient of a crop-sized MSE loss w.r.t. the [1,12,1280] style vectors through cal_style_codes -> gen_img
              (scripts/optimization.py:209-232) computed by the reference's own autograd.
  criteria.pt the loss networks of the optimisation loop run by the reference's own classes on seeded weights
              (e4s_amd.synth.synth_module_state_dict): IDLoss (src/criteria/id_loss.py, multi-scale, 256^2 and 1024^2 inputs)
              and LPIPS-AlexNet (src/criteria/lpips/lpips.py; torchvision's AlexNet topology restated, see
              oracle/ref_shim.reference_criteria) at 256^2 and as the three-scale sum of scripts/optimization.py:100-108 on a
              1024^2 pair: loss values, feature heads, and the gradient w.r.t. the generated image (strided); FaceParsingLoss
              (src/criteria/face_parsing/face_parsing_loss.py: UNet encoder features) at 512^2 and 1024^2.
"""
import os
import sys

os.environ.setdefault("E4S_ALLOW_UNINITIALIZED_LOSS_NETS", "1")     # synthetic state dicts are loaded after construction

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from e4s_amd import synth  # noqa: E402
from oracle import ref_shim  # noqa: E402


def ops_cases():
    ns = ref_shim.reference_modules()
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = (k1[None] * k1[:, None]) / 64.0
    out = {"kat": [], "rand": []}

    def up(x, k, **kw):
        return ns.upfirdn2d(x, k, **kw)

    a22 = torch.tensor([[1.0, 2.0], [3.0, 4.0]]).view(1, 1, 2, 2)
    out["kat"].append(dict(op="upfirdn2d", x=a22, k=4 * k2, up=2, down=1, pad=(2, 1), y=up(a22, 4 * k2, up=2, pad=(2, 1))))
    a33 = torch.arange(1.0, 10.0).view(1, 1, 3, 3)
    out["kat"].append(dict(op="upfirdn2d", x=a33, k=4 * k2, up=1, down=1, pad=(1, 1), y=up(a33, 4 * k2, pad=(1, 1))))
    a44 = torch.arange(1.0, 17.0).view(1, 1, 4, 4)
    out["kat"].append(dict(op="upfirdn2d", x=a44, k=k2, up=1, down=2, pad=(1, 1), y=up(a44, k2, down=2, pad=(1, 1))))
    out["kat"].append(dict(op="upfirdn2d", x=a22, k=a22[0, 0].clone(), up=1, down=1, pad=(1, 0),
                           y=up(a22, a22[0, 0].clone(), pad=(1, 0))))
    xb = torch.tensor([[-1.0, 0.0, 2.0]])
    bb = torch.tensor([0.5, -0.5, 0.5])
    out["kat"].append(dict(op="fused_leaky_relu", x=xb, b=bb, y=ns.fused_leaky_relu(xb, bb)))

    g = torch.Generator().manual_seed(1234)
    modes = [  # (N, C, H, W, kernel, up, down, pad)
        (2, 3, 9, 9, 4 * k2, 1, 1, (1, 1)),      # Blur after transposed conv (model.py:206-213)
        (1, 3, 8, 8, 4 * k2, 2, 1, (2, 1)),      # Upsample of the RGB skip (model.py:34-53)
        (2, 5, 16, 16, k2, 1, 1, (2, 2)),        # D ConvLayer blur k=3 (model.py:683-689)
        (2, 5, 16, 16, k2, 1, 1, (1, 1)),        # D skip blur k=1
        (1, 4, 12, 10, k2, 1, 2, (1, 1)),        # Downsample (model.py:56-75)
        (1, 2, 7, 5, torch.randn(3, 3, generator=g), 1, 1, (1, 1)),   # asymmetric kernel: flip convention
        (1, 2, 6, 6, torch.randn(2, 2, generator=g), 2, 1, (1, 0)),
        (1, 2, 10, 10, k2, 1, 1, (-1, -1)),      # negative pad = crop
        (1, 2, 9, 9, 4 * k2, 2, 2, (2, 1)),
    ]
    for n, c, h, w, k, u, d, p in modes:
        x = torch.randn(n, c, h, w, generator=g)
        out["rand"].append(dict(op="upfirdn2d", x=x, k=k.clone(), up=u, down=d, pad=p, y=up(x, k, up=u, down=d, pad=p)))
    for shape in [(2, 7, 5, 3), (3, 16), (1, 4, 8, 8)]:
        x = torch.randn(*shape, generator=g)
        b = torch.randn(shape[1], generator=g)
        out["rand"].append(dict(op="fused_leaky_relu", x=x, b=b, y=ns.fused_leaky_relu(x, b)))
    return out


@torch.no_grad()
def net_case(out_size, mask_kind):
    K = 13
    sd = synth.synth_state_dict(out_size, K)
    lat = synth.synth_latent_avg(out_size)
    net = ref_shim.build_reference_net3(sd, lat, out_size, K)
    driven = synth.synth_image(1, 1024, tag="driven")
    target = synth.synth_image(1, 1024, tag="target")
    if mask_kind == "face":
        dl = synth.synth_labels_face(1, 512, seed=1)
        tl = synth.synth_labels_face(1, 512, seed=2)
        sl = synth.synth_labels_face(1, 512, seed=3)
    else:
        dl = synth.synth_labels_blocks(1, 512, 64, seed=1)
        tl = synth.synth_labels_blocks(1, 512, 64, seed=2)
        sl = synth.synth_labels_blocks(1, 512, 64, seed=3)
    dm, tm, sm = synth.onehot(dl), synth.onehot(tl), synth.onehot(sl)
    noise = synth.synth_noise(out_size)
    d_sv, d_struct = net.get_style_vectors(driven, dm)
    t_sv, _ = net.get_style_vectors(target, tm)
    assert float(d_struct.abs().max()) == 0.0 and tuple(d_struct.shape) == (1, 512, 16, 16)
    # scripts/face_swap.py:117-146,261-262
    sv = t_sv.clone()
    for c in sorted(set(range(12)) - {0, 4, 11, 10}):
        sv[:, c] = d_sv[:, c]
    if torch.sum(d_sv[:, 7]) == 0:
        sv[:, 7] = (t_sv[:, 7] + d_sv[:, 7]) / 2
    if torch.sum(d_sv[:, 9]) == 0:
        sv[:, 9] = t_sv[:, 9]
    codes = net.cal_style_codes(sv)
    img, _minus1, feats = net.gen_img(torch.zeros(1, 512, 32, 32), codes, sm, noise=noise)
    assert _minus1 == -1
    rec = dict(out_size=out_size, K=K, mask_kind=mask_kind, driven_sv=d_sv, target_sv=t_sv, swapped_sv=sv,
               codes_stride=codes[:, :, :, ::8].clone(), codes_absmean=codes.abs().mean(),
               feats_stride=feats[:, ::4].clone(), img_mean=img.mean((2, 3)), img_absmean=img.abs().mean((2, 3)))
    if out_size <= 256:
        rec["img"] = img.clone()
    else:
        rec["img_stride8"] = img[:, :, ::8, ::8].clone()
        c0 = out_size // 2 - 64
        rec["img_crop"] = img[:, :, c0:c0 + 128, c0:c0 + 128].clone()
    return rec


@torch.no_grad()
def disc_case(size=64):
    """Discriminator(size) of the real reference on a seeded batch of 4 (the minibatch-stddev group)."""
    ns = ref_shim.reference_modules()
    sd = synth.synth_disc_state_dict(size)
    d = ns.Discriminator(size)
    d.load_state_dict(sd, strict=True)
    x = synth.synth_image(4, size, tag="disc")
    feats = d.convs(x)
    return dict(size=size, logits=d(x), feats=feats.clone())


GPEN_CFG = dict(size=64, n_mlp=8, narrow=0.25)


@torch.no_grad()
def gpen_case():
    """GPEN FullGenerator of the real reference at a CPU-sized configuration."""
    m = ref_shim.reference_gpen()
    c = GPEN_CFG
    sd = synth.synth_gpen_state_dict(c["size"], n_mlp=c["n_mlp"], narrow=c["narrow"])
    net = m.FullGenerator(c["size"], 512, c["n_mlp"], channel_multiplier=2, narrow=c["narrow"], device="cpu")
    net.load_state_dict(sd, strict=True)
    x = synth.synth_image(2, c["size"], tag="gpen")
    img, _ = net(x)
    return dict(cfg=c, img=img.clone())


def _z(t):
    """uint8 tensor -> (zlib bytes, shape): label maps and binary masks compress ~100x."""
    import zlib
    t = t.contiguous().to(torch.uint8)
    return zlib.compress(t.numpy().tobytes(), 9), tuple(t.shape)


def realmask_case():
    import numpy as np
    from PIL import Image
    fs = ref_shim.import_reference_script("face_swap")         # the reference's script, imported as shipped
    import importlib
    ds = importlib.import_module("src.datasets.dataset")
    to12 = getattr(ds, "__celebAHQ_masks_to_faceParser_mask_detailed")
    ex = os.path.join(ref_shim.REF_ROOT, "example", "input", "faceswap")
    D_mask = to12(np.array(Image.open(os.path.join(ex, "source_mask.png"))))
    T_mask = to12(np.array(Image.open(os.path.join(ex, "target_mask.png"))))
    swapped_msk, hole_map = fs.swap_head_mask_revisit_considerGlass(D_mask, T_mask)
    K, out_size = 13, 1024
    sd = synth.synth_state_dict(out_size, K)
    lat = synth.synth_latent_avg(out_size)
    net = ref_shim.build_reference_net3(sd, lat, out_size, K)
    driven = synth.synth_image(1, 1024, tag="driven")
    target = synth.synth_image(1, 1024, tag="target")
    lab = lambda a: torch.from_numpy(np.ascontiguousarray(a)).long()[None, None]
    dm, tm, sm = (synth.onehot(lab(a)) for a in (D_mask, T_mask, swapped_msk))
    noise = synth.synth_noise(out_size)
    rec = dict(out_size=out_size, K=K, D_mask=_z(lab(D_mask)[0, 0]), T_mask=_z(lab(T_mask)[0, 0]),
               swapped_mask=_z(lab(swapped_msk)[0, 0]), hole_map=_z(torch.from_numpy((hole_map != 0).astype(np.uint8))))
    with torch.no_grad():
        d_sv, _ = net.get_style_vectors(driven, dm)
        t_sv, _ = net.get_style_vectors(target, tm)
        comp = sorted(set(range(12)) - {0, 4, 11, 10})
        sv = fs.swap_comp_style_vector(t_sv, d_sv, comp)
        codes = net.cal_style_codes(sv)
        img, _, _ = net.gen_img(torch.zeros(1, 512, 32, 32), codes, sm, noise=noise)
    rec.update(driven_sv=d_sv, target_sv=t_sv, swapped_sv=sv, img_stride8=img[:, :, ::8, ::8].clone(),
               img_mean=img.mean((2, 3)), img_absmean=img.abs().mean((2, 3)))
    # crops over the thin regions: eyes/brows (rows ~400-530) and mouth/teeth (rows ~620-750) of the 1024^2 image
    rec["crops"] = {}
    for name, (y0, x0) in dict(eyes=(400, 448), mouth=(620, 448), edge=(0, 0)).items():
        rec["crops"][name] = (y0, x0, img[:, :, y0:y0 + 128, x0:x0 + 128].clone())
    # --- post-processing of scripts/face_swap.py:278-292 by the reference's own code (N4) ---
    sw = lab(swapped_msk)
    mask_bg = fs.logical_or_reduce(*[sw == clz for clz in [0, 11, 4]])
    is_fg = torch.logical_not(mask_bg)
    is_fg[torch.from_numpy(hole_map == 255)[None][None]] = True
    fg = is_fg.float()
    for op in ("dilation", "expansion", "erosion"):
        content, border, full = fs.create_masks(fg, outer_dilation=5, operation=op)
        rec["masks_" + op] = dict(border=_z(border[0, 0]), full=_z(full[0, 0]))
    rec["foreground"] = _z(fg[0, 0])
    im = fs.torch_utils.tensor2im(img[0])                                    # PIL image, uint8 HWC
    rec["img_u8_crop"] = (448, 448, torch.from_numpy(np.array(im))[448:576, 448:576].clone())
    rec["img_u8_sum"] = int(np.array(im).astype(np.int64).sum())
    # --- gradient of scripts/optimization.py's objective (MSE term on a crop) w.r.t. the style vectors ---
    latent = t_sv.clone().requires_grad_(True)
    codes = net.cal_style_codes(latent)
    recon, _, _ = net.gen_img(torch.zeros(1, 512, 32, 32), codes, tm, noise=noise)
    y0, x0, hw = 384, 384, 256
    loss = torch.nn.functional.mse_loss(recon[:, :, y0:y0 + hw, x0:x0 + hw], target[:, :, y0:y0 + hw, x0:x0 + hw])
    loss.backward()
    rec["opt"] = dict(crop=(y0, x0, hw), loss=float(loss), grad=latent.grad.clone(),
                      recon_stride8=recon.detach()[:, :, ::8, ::8].clone())
    return rec


def criteria_case():
    import tempfile
    import types
    import torch.nn.functional as F
    from e4s_amd import criteria as C
    ns = ref_shim.reference_criteria(C.alexnet_features)
    rec = {}
    # ---- IDLoss ----
    sd = synth.synth_module_state_dict(C.IDLoss(types.SimpleNamespace()), 0, "id.")
    tmp = tempfile.mkdtemp()
    torch.save({k[len("facenet."):]: v for k, v in sd.items() if k.startswith("facenet.")}, os.path.join(tmp, "irse50.pth"))
    ref = ns.IDLoss(types.SimpleNamespace(ir_se50_path=os.path.join(tmp, "irse50.pth"), id_loss_multiscale=True)).eval()
    for size, stride in ((256, 4), (1024, 16)):
        yh, y = synth.synth_image_pair(2, size, seed=3)
        yh.requires_grad_(True)
        loss, imp, _ = ref(yh, y)
        loss.backward()
        with torch.no_grad():
            feats = ref.extract_feats(y)
        rec[f"id{size}"] = dict(loss=loss.detach().clone(), improvement=float(imp),
                                feat_heads=[f[:, :64].clone() for f in feats],
                                grad_strided=yh.grad[:, :, ::stride, ::stride].clone(), grad_absmax=yh.grad.abs().max().clone(),
                                grad_l2=yh.grad.norm().clone(), stride=stride)
    # ---- LPIPS ----
    sdl = synth.synth_module_state_dict(C.LPIPS(), 0, "lp.")
    refl = ns.LPIPS(net_type="alex").eval()
    refl.load_state_dict(sdl)
    yh, y = synth.synth_image_pair(2, 256, seed=4)
    yh.requires_grad_(True)
    loss = refl(yh, y)
    loss.backward()
    with torch.no_grad():
        feats = refl.net(y)
    rec["lpips256"] = dict(loss=loss.detach().clone(), feat_heads=[f[:, :8, :4, :4].clone() for f in feats],
                           grad_strided=yh.grad[:, :, ::4, ::4].clone(), grad_l2=yh.grad.norm().clone(), stride=4)
    yh, y = synth.synth_image_pair(1, 1024, seed=5)
    yh.requires_grad_(True)
    loss = sum(refl(F.adaptive_avg_pool2d(yh, (1024 // 2 ** i,) * 2), F.adaptive_avg_pool2d(y, (1024 // 2 ** i,) * 2))
               for i in range(3))                                        # scripts/optimization.py:100-108
    loss.backward()
    rec["lpips1024x3"] = dict(loss=loss.detach().clone(), grad_strided=yh.grad[:, :, ::16, ::16].clone(),
                              grad_l2=yh.grad.norm().clone(), stride=16)
    # ---- FaceParsingLoss (UNet encoder features) ----
    sdp = synth.synth_module_state_dict(C.FaceParsingLoss(types.SimpleNamespace()), 0, "fp.")
    torch.save({k[len("G."):]: v for k, v in sdp.items() if k.startswith("G.")}, os.path.join(tmp, "unet.pth"))
    refp = ns.FaceParsingLoss(types.SimpleNamespace(face_parsing_model_path=os.path.join(tmp, "unet.pth"))).eval()
    for size, stride in ((512, 8), (1024, 16)):
        yh, y = synth.synth_image_pair(1, size, seed=6)
        yh.requires_grad_(True)
        loss, imp = refp(yh, y)
        loss.backward()
        with torch.no_grad():
            feats = refp.extract_feats(y)
        rec[f"parsing{size}"] = dict(loss=loss.detach().clone(), improvement=float(imp),
                                     feat_heads=[f[:, :64].clone() for f in feats],
                                     grad_strided=yh.grad[:, :, ::stride, ::stride].clone(), grad_l2=yh.grad.norm().clone(),
                                     stride=stride)
    return rec


def main():
    torch.manual_seed(0)
    if "--criteria-only" in sys.argv:
        torch.save(criteria_case(), os.path.join(HERE, "criteria.pt"))
        print("criteria.pt done")
        return
    if "--realmask-only" in sys.argv:
        torch.save(realmask_case(), os.path.join(HERE, "realmask.pt"))
        print("realmask.pt done")
        return
    if "--gpen-only" in sys.argv:
        torch.save(gpen_case(), os.path.join(HERE, "gpen64.pt"))
        print("gpen64.pt done")
        return
    if "--disc-only" in sys.argv:
        torch.save(disc_case(64), os.path.join(HERE, "disc64.pt"))
        print("disc64.pt done")
        return
    torch.save(ops_cases(), os.path.join(HERE, "ops.pt"))
    print("ops.pt done")
    torch.save(net_case(256, "blocks"), os.path.join(HERE, "net256.pt"))
    print("net256.pt done")
    torch.save(net_case(1024, "face"), os.path.join(HERE, "net1024.pt"))
    print("net1024.pt done")
    torch.save(disc_case(64), os.path.join(HERE, "disc64.pt"))
    print("disc64.pt done")
    torch.save(gpen_case(), os.path.join(HERE, "gpen64.pt"))
    print("gpen64.pt done")
    torch.save(realmask_case(), os.path.join(HERE, "realmask.pt"))
    print("realmask.pt done")
    torch.save(criteria_case(), os.path.join(HERE, "criteria.pt"))
    print("criteria.pt done")


if __name__ == "__main__":
    main()
