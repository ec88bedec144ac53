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
