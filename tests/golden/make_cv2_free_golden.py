"""Third-party cross-check of the OpenCV half of the stitch (scripts/face_swap.py:81-97,278-310,
src/utils/multi_band_blending.py:4-75) with what this container HAS: cv2 (opencv-python 4.7.0.72, e4s_env.yaml:96) is absent here and
on the GPU box, so `oracle/e4s_oracle.py`'s cv2_erode_u8 / cv2_gaussian_blur_u8 / cv2_pyrdown / cv2_pyrup / laplacian_blend_u8 are
restatements of OpenCV's published algorithms by the same author as the HIP kernels.  This script computes the same five operations
with INDEPENDENT implementations -- scipy.ndimage (1.15) and PIL (12.2) -- on crops of the reference's own example images and parsing
maps (example/input/faceswap/*), and writes inputs + results to tests/golden/cv2free.pt:

    erode 11x11, border 255     scipy.ndimage.grey_erosion(size=11, mode='constant', cval=255)  ==  PIL.ImageFilter.MinFilter(11)   [exact]
    GaussianBlur 11x11, sigma 0 float Gaussian, OpenCV's sigma rule 0.3*((k-1)/2-1)+0.8, scipy correlate1d mode='mirror' (= BORDER_REFLECT_101)
                                -> the CV_8U fixed-point path must agree within +-1 LSB
    pyrDown (uint8)             scipy correlate1d with INTEGER taps [1,4,6,4,1] on int64, mode='mirror', decimate, (acc+128)>>8   [exact]
    pyrDown / pyrUp (float32)   float64 correlate1d ([1,4,6,4,1]/16, decimate; zero-insert + [1,4,6,4,1]/8, mode='mirror' -- which
                                reproduces pyramids.cpp's reflected near edge and replicated far edge)   [<= 1e-6 of 255]
    Laplacian blend             multi_band_blending.blending's schedule on those scipy pyramids in float64   [+-1 LSB]

That pins the oracle's OpenCV restatements to third-party code ("cross-checked against scipy / PIL"); it is still not cv2 itself
(OpenCV's SIMD paths are specified to equal its scalar paths, which is what the restatements follow).

Run in the build container:  python tests/golden/make_cv2_free_golden.py   (reads /root/reference, writes tests/golden/cv2free.pt)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("E4S_REFERENCE", "/root/reference")


def sp_erode(mask_u8, radius):
    from scipy import ndimage
    return ndimage.grey_erosion(mask_u8, size=(2 * radius + 1, 2 * radius + 1), mode="constant", cval=255)


def pil_erode(mask_u8, radius):
    from PIL import Image, ImageFilter
    return np.array(Image.fromarray(mask_u8).filter(ImageFilter.MinFilter(2 * radius + 1)))


def sp_gaussian_float(img_u8, ksize, sigma=0.0):
    """cv2.getGaussianKernel's float kernel (sigma <= 0: 0.3*((ksize-1)*0.5 - 1) + 0.8), separable, BORDER_REFLECT_101 = 'mirror'."""
    from scipy import ndimage
    sg = sigma if sigma > 0 else 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    k = np.exp(-((np.arange(ksize) - (ksize - 1) / 2) ** 2) / (2 * sg * sg))
    k /= k.sum()
    x = img_u8.astype(np.float64)
    x = ndimage.correlate1d(x, k, axis=0, mode="mirror")
    return ndimage.correlate1d(x, k, axis=1, mode="mirror")


TAPS = np.array([1, 4, 6, 4, 1])


def sp_pyrdown(x):
    """HWC; uint8 -> exact integer arithmetic and OpenCV's (acc + 128) >> 8; float -> float64 taps / 16 per axis."""
    from scipy import ndimage
    if x.dtype == np.uint8:
        v = x.astype(np.int64)
        v = ndimage.correlate1d(v, TAPS, axis=0, mode="mirror")
        v = ndimage.correlate1d(v, TAPS, axis=1, mode="mirror")
        return ((v[::2, ::2] + 128) >> 8).astype(np.uint8)
    v = x.astype(np.float64)
    v = ndimage.correlate1d(v, TAPS / 16.0, axis=0, mode="mirror")
    v = ndimage.correlate1d(v, TAPS / 16.0, axis=1, mode="mirror")
    return v[::2, ::2]


def sp_pyrup(x):
    """HWC float: zero insertion, then [1,4,6,4,1]/8 per axis.  On the zero-inserted signal u (u[2i] = s[i]) scipy's 'mirror' gives
    u[-2] = u[2] = s[1] (pyramids.cpp's first column: s0*6 + s1*2) and u[2w] = u[2w-2] = s[w-1] (its last: s[w-2] + s[w-1]*7, s[w-1]*8)."""
    from scipy import ndimage
    h, w = x.shape[:2]
    u = np.zeros((2 * h, 2 * w) + x.shape[2:], dtype=np.float64)
    u[::2, ::2] = x
    u = ndimage.correlate1d(u, TAPS / 8.0, axis=0, mode="mirror")
    return ndimage.correlate1d(u, TAPS / 8.0, axis=1, mode="mirror")


def sp_laplacian_blend(full_img, ori_img, mask, num_levels):
    """multi_band_blending.py:4-75 on the scipy pyramids (uint8 Gaussian levels rounded as cv2.pyrDown rounds them, everything else
    float64); returns the float image before np.uint8(np.clip(.))."""
    GA, GB, GM = full_img.copy(), ori_img.copy(), mask.astype(np.float64)
    gpA, gpB, gpM = [GA], [GB], [GM]
    for _ in range(num_levels):
        GA, GB, GM = sp_pyrdown(GA), sp_pyrdown(GB), sp_pyrdown(GM)
        gpA.append(GA.astype(np.float64)); gpB.append(GB.astype(np.float64)); gpM.append(GM)
    lpA, lpB, gpMr = [gpA[num_levels - 1]], [gpB[num_levels - 1]], [gpM[num_levels - 1]]
    for i in range(num_levels - 1, 0, -1):
        lpA.append(gpA[i - 1].astype(np.float64) - sp_pyrup(gpA[i]))
        lpB.append(gpB[i - 1].astype(np.float64) - sp_pyrup(gpB[i]))
        gpMr.append(gpM[i - 1])
    LS = [la * gm + lb * (1.0 - gm) for la, lb, gm in zip(lpA, lpB, gpMr)]
    ls_ = LS[0]
    for i in range(1, num_levels):
        ls_ = sp_pyrup(ls_) + LS[i]
    return ls_


def compute(inputs):
    """inputs: dict of numpy arrays (see main) -> dict of third-party results."""
    out = {}
    r = 5                                                              # face_swap.py:286 outer_dilation = 5 -> radius 5, 11x11
    for name in ("mask_a", "mask_b"):
        e = sp_erode(inputs[name], r)
        assert np.array_equal(e, pil_erode(inputs[name], r)), "scipy grey_erosion and PIL MinFilter disagree"
        out["erode_" + name] = e
        out["gauss_" + name] = sp_gaussian_float(e, 2 * r + 1).astype(np.float32)
    for name in ("img_a", "img_b"):
        out["gauss_" + name] = sp_gaussian_float(inputs[name][..., 0], 11).astype(np.float32)
        out["pyrdown_u8_" + name] = sp_pyrdown(inputs[name])
        f = inputs[name].astype(np.float32) * np.float32(0.731) + np.float32(3.3)      # a float image that is not integer-valued
        out["pyrdown_f_" + name] = sp_pyrdown(f).astype(np.float32)
        out["pyrup_f_" + name] = sp_pyrup(f[: f.shape[0] // 2, : f.shape[1] // 2]).astype(np.float32)
    m3 = np.repeat(inputs["blend_mask"][:, :, None], 3, axis=2)           # face_swap.py:305: the border mask, repeated over RGB
    out["blend"] = np.uint8(np.clip(sp_laplacian_blend(inputs["blend_full"], inputs["blend_ori"], m3, 6), 0, 255))
    return out


def main():
    from PIL import Image
    ex = os.path.join(REF, "example", "input", "faceswap")
    src = np.array(Image.open(os.path.join(ex, "source.jpg")).convert("RGB"))
    tgt = np.array(Image.open(os.path.join(ex, "target.jpg")).convert("RGB"))
    lab_s = np.array(Image.open(os.path.join(ex, "source_mask.png")))
    lab_t = np.array(Image.open(os.path.join(ex, "target_mask.png")))
    if lab_s.ndim == 3:
        lab_s, lab_t = lab_s[..., 0], lab_t[..., 0]
    # foreground = everything but background / hair-like ids of the 19-id CelebAMask-HQ map: a face-shaped binary mask with holes
    fg_s = ((lab_s != 0) & (lab_s != 13) & (lab_s != 16)).astype(np.uint8) * 255
    fg_t = ((lab_t != 0) & (lab_t != 13) & (lab_t != 16)).astype(np.uint8) * 255
    inputs = {
        "img_a": np.ascontiguousarray(src[300:460, 380:572]),               # 160 x 192 around the eyes
        "img_b": np.ascontiguousarray(tgt[411:508, 300:431]),               # 97 x 131: odd sizes
        "mask_a": np.ascontiguousarray(fg_s[96:352, 128:384]),              # 256 x 256 of the 512^2 parsing map
        "mask_b": np.ascontiguousarray(fg_t[31:200, 77:300]),               # 169 x 223, touches the map's upper part
        "blend_full": np.ascontiguousarray(src[256:512, 384:640]),          # 256 x 256, 6 pyramid levels
        "blend_ori": np.ascontiguousarray(tgt[256:512, 384:640]),
    }
    m = np.array(Image.fromarray(fg_s).resize((1024, 1024), Image.BILINEAR))[256:512, 384:640].astype(np.float32) / 255.0
    inputs["blend_mask"] = np.ascontiguousarray(m)                          # [256,256] float32 in [0,1]; consumers repeat it over RGB
    out = compute(inputs)
    blob = {"inputs": {k: torch.from_numpy(v) for k, v in inputs.items()}, "scipy": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()},
            "versions": {"scipy": __import__("scipy").__version__, "PIL": __import__("PIL").__version__, "numpy": np.__version__}}
    path = os.path.join(HERE, "cv2free.pt")
    torch.save(blob, path)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
