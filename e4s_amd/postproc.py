"""Device-side pre/post-processing of the face-swap pipeline: the numpy / OpenCV / PIL glue of
scripts/face_swap.py:226-312 as HIP kernels (e4s_amd/csrc/postproc.hip) on device tensors -- no host round trips at
batch 64, and the output that is all-gathered is the uint8 image (25 MB per 8 swaps instead of 100 MB of fp32).
Function names and argument meaning follow the reference's."""
import torch

from .lib import call, fptr, ptr, stream


def _u8(t):
    if t.dtype != torch.uint8:
        raise RuntimeError(f"expected a uint8 label map, got {t.dtype}")
    return t.contiguous()


def labelMap2OneHot(label, num_cls):
    """src/utils/torch_utils.py:166-172: label [B,1,H,W] (uint8 or int64 ids < num_cls) -> one-hot fp32 [B,num_cls,H,W]."""
    b, _, h, w = label.shape
    lab = label.to(torch.uint8).contiguous() if label.dtype != torch.uint8 else _u8(label)
    out = torch.empty(b, num_cls, h, w, device=label.device, dtype=torch.float32)
    call("e4s_onehot_u8_f32", ptr(lab), fptr(out), b, num_cls, h, w, stream())
    return out


def swap_head_mask_revisit_considerGlass(source, target):
    """src/utils/swap_face_mask.py:33-82 (hair_first=True) on uint8 12-class label maps of any shape.
    Returns (swapped labels uint8, hole map uint8 {0,255})."""
    source, target = _u8(source), _u8(target)
    if source.shape != target.shape:
        raise RuntimeError("source / target label maps must have the same shape")
    out, hole = torch.empty_like(target), torch.empty_like(target)
    call("e4s_swap_head_mask_u8", ptr(source), ptr(target), ptr(out), ptr(hole), target.numel(), stream())
    return out, hole


def foreground_mask(swapped, hole):
    """scripts/face_swap.py:280-284: everything but background / ear-rings / hair, plus the filled holes -> fp32 {0,1}."""
    swapped, hole = _u8(swapped), _u8(hole)
    fg = torch.empty(swapped.shape, device=swapped.device, dtype=torch.float32)
    call("e4s_foreground_mask_f32", ptr(swapped), ptr(hole), fptr(fg), swapped.numel(), stream())
    return fg


def _flat_radius(kernel):
    kh, kw = kernel.shape
    if kh != kw or kh % 2 == 0 or not bool((kernel != 0).all()):
        raise NotImplementedError("only flat, square, odd structuring elements (torch.ones(2r+1, 2r+1)), as the scripts use")
    return kh // 2


def dilation(tensor, kernel, engine="convolution"):
    """src/utils/morphology.py:23-108 for a flat structuring element and the geodesic border: tensor [B,C,H,W]."""
    r = _flat_radius(kernel)
    b, c, h, w = tensor.shape
    x = tensor.contiguous()
    out = torch.empty_like(x)
    call("e4s_morph_f32", fptr(x), fptr(out), None, b * c, h, w, r, stream())
    return out


def erosion(tensor, kernel, engine="convolution"):
    """src/utils/morphology.py:111-198, same restrictions."""
    r = _flat_radius(kernel)
    b, c, h, w = tensor.shape
    x = tensor.contiguous()
    out = torch.empty_like(x)
    call("e4s_morph_f32", fptr(x), None, fptr(out), b * c, h, w, r, stream())
    return out


def create_masks(mask, outer_dilation=0, operation="dilation"):
    """scripts/face_swap.py:30-48 -> (content_mask, border_mask, full_mask); mask [B,1,H,W] fp32."""
    op = {"dilation": 0, "erosion": 1, "expansion": 2}[operation]
    b, c, h, w = mask.shape
    m = mask.contiguous()
    border, full = torch.empty_like(m), torch.empty_like(m)
    ws = torch.empty(2 * m.numel(), device=m.device, dtype=torch.float32)
    call("e4s_create_masks_f32", fptr(m), fptr(border), fptr(full), fptr(ws), b * c, h, w, int(outer_dilation), op, stream())
    return mask, border, full


def tensor2im(var, out=None):
    """src/utils/torch_utils.py:63-69 on a batch: [B,3,H,W] fp32 in [-1,1] -> uint8 [B,H,W,3] (HWC, device).
    `out`: write into this uint8 [B,H,W,3] buffer (e.g. the staging slot of the all-gather)."""
    if var.dim() == 3:
        var = var.unsqueeze(0)
    b, c, h, w = var.shape
    if c != 3:
        raise RuntimeError("tensor2im expects RGB images [B,3,H,W]")
    x = var.contiguous()
    if out is None:
        out = torch.empty(b, h, w, 3, device=x.device, dtype=torch.uint8)
    elif out.dtype != torch.uint8 or tuple(out.shape) != (b, h, w, 3) or not out.is_contiguous():
        raise RuntimeError("tensor2im: out must be a contiguous uint8 [B,H,W,3] tensor")
    call("e4s_tensor2im_u8", fptr(x), ptr(out), b, h, w, stream())
    return out


def paste(face_u8, target_u8, content_mask):
    """scripts/face_swap.py:291-292,301-303: swapped * content + T * (1 - content) with the [B,1,Hm,Wm] content mask
    bilinearly resized to the image size; uint8 HWC in and out."""
    b, h, w, _ = face_u8.shape
    hm, wm = content_mask.shape[-2:]
    out = torch.empty_like(face_u8)
    call("e4s_paste_u8", ptr(_u8(face_u8)), ptr(_u8(target_u8)), fptr(content_mask.contiguous()), ptr(out), b, h, w,
         hm, wm, stream())
    return out


# ---- stitching the swapped face back onto the target (scripts/face_swap.py:278-310; csrc/stitch.hip) -------------------------
_SMALL_GAUSSIAN_TAB = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                       7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_kernel_fixed8(ksize, sigma=0.0):
    """The 8.8 fixed-point taps cv2.GaussianBlur uses on CV_8U images (OpenCV 4.x smooth.dispatch.cpp, restated: the library
    is absent here).  getGaussianKernelBitExact: sigma <= 0 -> 0.3*((ksize-1)*0.5 - 1) + 0.8; odd ksize <= 7 with sigma <= 0
    take the fixed table; otherwise exp(-x^2 / (2 sigma^2)) normalised.  getGaussianKernelFixedPoint_ED: taps * 256 rounded
    (half to even) from the edge inwards with the rounding error carried to the next tap, the centre tap takes what is left of
    256.  ksize 11, sigma 0 (scripts/face_swap.py:91): [2, 7, 17, 31, 45, 52, 45, 31, 17, 7, 2]."""
    import math
    if ksize < 1 or ksize % 2 == 0:
        raise ValueError("odd ksize")
    if sigma <= 0 and ksize <= 7:
        k = list(_SMALL_GAUSSIAN_TAB[ksize])
    else:
        s = sigma if sigma > 0 else 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        k = [math.exp(-((i - (ksize - 1) * 0.5) ** 2) / (2.0 * s * s)) for i in range(ksize)]
        tot = sum(k)
        k = [v / tot for v in k]
    n2 = ksize // 2
    out, err, acc = [0] * ksize, 0.0, 0
    for i in range(n2):
        adj = k[i] * 256.0 + err
        v0 = int(round(adj))                     # cvRound: half to even, as Python's round()
        err = adj - v0
        out[i] = out[ksize - 1 - i] = v0
        acc += v0
    out[n2] = 256 - 2 * acc
    return out


def mask_to_u8(mask, size):
    """`255 * F.interpolate(mask, size, mode='bilinear')[0, 0].numpy().astype(np.uint8)` (scripts/face_swap.py:291-294) for a
    batch: mask [B,1,Hm,Wm] fp32 -> uint8 [B,H,W] (255 only where the resized mask reaches 1.0: the cast truncates first)."""
    b, c, hm, wm = mask.shape
    if c != 1:
        raise RuntimeError("mask_to_u8 expects [B,1,Hm,Wm]")
    h, w = size
    out = torch.empty(b, h, w, device=mask.device, dtype=torch.uint8)
    call("e4s_mask_to_u8", fptr(mask.contiguous()), ptr(out), b, h, w, hm, wm, stream())
    return out


def erode_u8(mask_u8, radius, border_value=255):
    """cv2.erode(mask, np.ones((2r+1, 2r+1)), borderType=cv2.BORDER_CONSTANT, borderValue=border_value); [B,H,W] uint8."""
    m = _u8(mask_u8)
    b, h, w = m.shape
    out = torch.empty_like(m)
    call("e4s_erode_u8", ptr(m), ptr(out), b, h, w, int(radius), int(border_value), stream())
    return out


def gaussian_blur_u8(img_u8, ksize, sigma=0.0):
    """cv2.GaussianBlur(img, (ksize, ksize), sigmaX=sigma) on a single-channel uint8 image batch [B,H,W]."""
    import ctypes
    m = _u8(img_u8)
    b, h, w = m.shape
    taps = gaussian_kernel_fixed8(ksize, sigma)
    out = torch.empty_like(m)
    call("e4s_gaussian_blur_u8", ptr(m), ptr(out), b, h, w, ksize, (ctypes.c_int * ksize)(*taps), stream())
    return out


def alpha_composite(image_u8, dst_image_u8, alpha_u8):
    """PIL: dst.convert('RGBA').alpha_composite(image.convert('RGBA') with putalpha(alpha)) -> RGB uint8 [B,H,W,3]."""
    f, t, a = _u8(image_u8), _u8(dst_image_u8), _u8(alpha_u8)
    b, h, w, _ = f.shape
    if t.shape != f.shape or tuple(a.shape) != (b, h, w):
        raise RuntimeError("alpha_composite: image / dst [B,H,W,3] and alpha [B,H,W]")
    out = torch.empty_like(f)
    call("e4s_alpha_composite_u8", ptr(f), ptr(t), ptr(a), ptr(out), b, h, w, stream())
    return out


def smooth_face_boundry(image, dst_image, mask, radius=0, sigma=0.0):
    """scripts/face_swap.py:81-97 on device batches: image / dst_image uint8 [B,H,W,3], mask uint8 [B,H,W] -> RGB uint8 of the
    pasted RGBA image (its alpha channel is 255 everywhere)."""
    if radius != 0:
        k = 2 * radius + 1
        mask = gaussian_blur_u8(erode_u8(mask, radius, 255), k, sigma)
    return alpha_composite(image, dst_image, mask)


def pyr_down(x):
    """cv2.pyrDown on an HWC image batch [B,H,W,C], uint8 or fp32."""
    b, h, w, c = x.shape
    out = torch.empty(b, (h + 1) // 2, (w + 1) // 2, c, device=x.device, dtype=x.dtype)
    if x.dtype == torch.uint8:
        call("e4s_pyrdown_u8", ptr(x.contiguous()), ptr(out), b, h, w, c, stream())
    else:
        call("e4s_pyrdown_f32", fptr(x.contiguous()), fptr(out), b, h, w, c, stream())
    return out


def pyr_up(x):
    """cv2.pyrUp on an fp32 HWC image batch [B,h,w,C] -> [B,2h,2w,C]."""
    b, h, w, c = x.shape
    out = torch.empty(b, 2 * h, 2 * w, c, device=x.device, dtype=torch.float32)
    call("e4s_pyrup_f32", fptr(x.contiguous()), fptr(out), b, h, w, c, stream())
    return out


def _to_f32(x_u8):
    out = torch.empty(x_u8.shape, device=x_u8.device, dtype=torch.float32)
    call("e4s_u8_to_f32", ptr(x_u8), fptr(out), x_u8.numel(), stream())
    return out


def Laplacian_Pyramid_Blending_with_mask(A, B, m, num_levels=6):
    """src/utils/multi_band_blending.py:4-50: A, B uint8 [Bn,H,W,3], m fp32 [Bn,H,W,3] in [0,1] -> fp32 [Bn,H,W,3].
    The Gaussian pyramids of A and B stay uint8 all the way down (the reference keeps feeding pyrDown its uint8 output)."""
    gA, gB, gM = [_u8(A)], [_u8(B)], [m.contiguous()]
    for _ in range(num_levels):
        gA.append(pyr_down(gA[-1]))
        gB.append(pyr_down(gB[-1]))
        gM.append(pyr_down(gM[-1]))
    top = num_levels - 1

    def level(a, ua, bb, ub, mm, acc):
        out = torch.empty(a.shape, device=a.device, dtype=torch.float32)
        call("e4s_lap_level_f32", ptr(a), fptr(ua), ptr(bb), fptr(ub), fptr(mm), fptr(acc), fptr(out), a.numel(), 1, stream())
        return out
    ls = level(gA[top], None, gB[top], None, gM[top], None)
    for i in range(top, 0, -1):
        ls = level(gA[i - 1], pyr_up(_to_f32(gA[i])), gB[i - 1], pyr_up(_to_f32(gB[i])), gM[i - 1], pyr_up(ls))
    return ls


def blending(full_img, ori_img, mask):
    """src/utils/multi_band_blending.py:52-75 for 1024^2 inputs (its cv2.resize calls are then identities): uint8 [B,H,W,3]."""
    b, h, w, _ = full_img.shape
    if (h, w) != (1024, 1024):
        raise NotImplementedError("blending resizes everything to 1024^2 first; only 1024^2 inputs (the pipeline's) are accepted")
    img = Laplacian_Pyramid_Blending_with_mask(full_img, ori_img, mask.to(torch.float32), 10)
    out = torch.empty(b, h, w, 3, device=img.device, dtype=torch.uint8)
    call("e4s_clip_u8", fptr(img), ptr(out), img.numel(), stream())
    return out


def stitch(swapped_face, target_u8, swapped_labels, hole, lap_bld=False, outer_dilation=5):
    """scripts/face_swap.py:276-310 on the device for a batch: swapped_face fp32 [B,3,1024,1024] (the generator's output),
    target_u8 uint8 [B,1024,1024,3] (T), swapped_labels / hole uint8 [B,512,512] (swap_head_mask_revisit_considerGlass's
    outputs) -> the stitched uint8 [B,1024,1024,3] image."""
    b, _, h, w = swapped_face.shape
    face_u8 = tensor2im(swapped_face)
    fg = foreground_mask(swapped_labels, hole).view(b, 1, *swapped_labels.shape[-2:])
    content, border, full = create_masks(fg, outer_dilation=outer_dilation, operation="expansion" if lap_bld else "dilation")
    if lap_bld:
        pasted = paste(face_u8, target_u8, content)
        from . import kernels as K
        bm = K.resize_bilinear_to_nhwc(border.contiguous(), h, w)                 # F.interpolate(..., 'bilinear') -> [B,H,W,1]
        return blending(target_u8, pasted, bm.expand(b, h, w, 3).contiguous())
    mask_img = mask_to_u8(content if outer_dilation == 0 else full, (h, w))
    return smooth_face_boundry(face_u8, target_u8, mask_img, radius=outer_dilation)
