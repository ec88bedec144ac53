"""Tensor-level wrappers over the C-ABI of libe4s_hip.so.

Each function allocates its outputs with torch (device memory + caching allocator are the
plumbing PyTorch provides) and enqueues one native call on the current HIP stream.  No function
here computes anything in torch; a missing library or a CPU tensor raises.
"""
import contextlib
import ctypes
import math
import os
import threading

import torch

from . import lib
from .lib import ConvBwdParams, ConvParams, ConvWgradParams, c_p, call, fptr, ptr, stream

BM = 128          # GEMM row tile of e4s_conv_mfma_f32
# Arithmetic of the contractions that have a split-bf16 kernel (the encoder's stride-1 3x3 convs):
#   "f32"    exact fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32)
#   "bf16x3" three bf16 MFMAs per product on hi/lo-split fp32 operands, fp32 accumulate (~2^-16 per product; measured
#            1.6e-4 max-abs on the 1024^2 image against the reference, bound 1e-3), wherever the kernel applies
#   "auto"   (default) bf16x3 where it applies AND the launch fills the chip (its 256x128 tiles need >= BF16X3_MIN_BLOCKS
#            blocks to beat the fp32 kernel's 128-row tiles: batch-1 latency runs stay on fp32)
PRECISION = os.environ.get("E4S_PRECISION", "auto")
if PRECISION not in ("f32", "bf16x3", "auto"):
    raise RuntimeError(f"E4S_PRECISION must be f32, bf16x3 or auto, got {PRECISION!r}")
BF16X3_MIN_BLOCKS = int(os.environ.get("E4S_BF16X3_MIN_BLOCKS", "128"))          # (env: policy experiments, tools/; the default is what is tested)
LRELU_GAIN = math.sqrt(2.0)


def _f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


# ---- 1:1 ops ---------------------------------------------------------------------------------
def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale) -- fused_bias_act.cpp:11-21."""
    x = _f32(x)
    y = torch.empty_like(x)
    has_b = bias is not None and bias.numel() > 0
    has_r = ref is not None and ref.numel() > 0
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    call("e4s_fused_bias_act_f32", fptr(x), fptr(_f32(bias)) if has_b else None,
         fptr(_f32(ref)) if has_r else None, fptr(y), x.numel(), step_b, bias.numel() if has_b else 1,
         int(act), int(grad), float(alpha), float(scale), stream())
    return y


def channel_sum(g):
    """sum over every dim but 1 (grad_bias, op/fused_act.py:33-38)."""
    g = _f32(g)
    c = g.shape[1]
    step_b = 1
    for d in g.shape[2:]:
        step_b *= d
    out = torch.empty(c, device=g.device, dtype=torch.float32)
    call("e4s_channel_sum_f32", fptr(g), fptr(out), g.numel(), step_b, c, stream())
    return out


def upfirdn2d_raw(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel, ...) -- upfirdn2d.cpp:12-22."""
    x = _f32(x)
    kernel = _f32(kernel)
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    y = torch.empty(major, out_h, out_w, minor, device=x.device, dtype=torch.float32)
    call("e4s_upfirdn2d_f32", fptr(x), fptr(kernel), fptr(y), major, in_h, in_w, minor, kh, kw, up_x, up_y,
         down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream())
    return y


# ---- layout ----------------------------------------------------------------------------------
def nchw_to_nhwc(x):
    x = _f32(x)
    b, c, h, w = x.shape
    y = torch.empty(b, h, w, c, device=x.device, dtype=torch.float32)
    call("e4s_nchw_to_nhwc_f32", fptr(x), fptr(y), b, c, h, w, stream())
    return y


def nhwc_to_nchw(x):
    x = _f32(x)
    b, h, w, c = x.shape
    y = torch.empty(b, c, h, w, device=x.device, dtype=torch.float32)
    call("e4s_nhwc_to_nchw_f32", fptr(x), fptr(y), b, c, h, w, stream())
    return y


def const_input(inp, batch):
    """ConstantInput.forward (model.py:345-349): [1,C,H,W] parameter -> NHWC [B,H,W,C]."""
    inp = _f32(inp)
    _, c, h, w = inp.shape
    y = torch.empty(batch, h, w, c, device=inp.device, dtype=torch.float32)
    call("e4s_const_input_f32", fptr(inp), fptr(y), batch, c, h, w, stream())
    return y


# ---- style prologue --------------------------------------------------------------------------
def modulate(latent, layer_idx, masked, mod_weight, mod_bias):
    """s = EqualLinear(512 -> Cin, bias_init=1)(style) for every (sample[, region]) group.
    latent [B,R,L,512] contiguous.  Returns [G, Cin] with G = B*R (masked) or B (region 0)."""
    b, r, nl, d = latent.shape
    cin = mod_weight.shape[0]
    g = b * r if masked else b
    stride = nl * d if masked else r * nl * d
    out = torch.empty(g, cin, device=latent.device, dtype=torch.float32)
    base = c_p(latent.data_ptr() + layer_idx * d * 4)
    call("e4s_rowdot_f32", base, stride, fptr(mod_weight), fptr(mod_bias), fptr(out), g, cin, d, 0,
         1.0 / math.sqrt(d), stream())
    return out


def modulate_vec(style, mod_weight, mod_bias):
    """Same for an explicit [G, 512] style matrix (module-level forward)."""
    style = _f32(style)
    g, d = style.shape
    cin = mod_weight.shape[0]
    out = torch.empty(g, cin, device=style.device, dtype=torch.float32)
    call("e4s_rowdot_f32", fptr(style), d, fptr(mod_weight), fptr(mod_bias), fptr(out), g, cin, d, 0,
         1.0 / math.sqrt(d), stream())
    return out


_JOB_FMT = "qqqQQiiif"            # e4s_rowdot_job (include/e4s_hip.h): 3 x int64, 2 x pointer, 3 x int, float = 56 bytes


def rowdot_jobs(jobs, device):
    """jobs: list of dict(in_off, in_stride, out_off, M, bias, G, O, K, scale) -> (device table uint8, njobs, max_O, max_G).
    M / bias are tensors (kept alive by the caller: the table holds their raw pointers).  One H2D copy: build OUTSIDE graph
    capture (the generator caches the table per weight version)."""
    import struct
    buf = b"".join(struct.pack(_JOB_FMT, int(j["in_off"]), int(j["in_stride"]), int(j["out_off"]), j["M"].data_ptr(),
                               j["bias"].data_ptr() if j["bias"] is not None else 0, int(j["G"]), int(j["O"]), int(j["K"]),
                               float(j["scale"])) for j in jobs)
    assert struct.calcsize(_JOB_FMT) == 56
    for j in jobs:
        if j["M"].dtype != torch.float32 or not j["M"].is_contiguous() or j["K"] % 4:
            raise RuntimeError("rowdot_jobs: contiguous fp32 matrices with K % 4 == 0")
    table = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(device)
    return table, len(jobs), max(j["O"] for j in jobs), max(j["G"] for j in jobs)


def rowdot_multi(table, njobs, max_o, max_g, in_base, out_base, mode):
    call("e4s_rowdot_multi_f32", ptr(table), njobs, fptr(in_base), fptr(out_base), max_o, max_g, mode, stream())


def demod_coefs(s, wsq, conv_scale):
    """conv_scale * rsqrt(conv_scale^2 * sum_ci s^2 Wsq[co,ci] + 1e-8): [G, Cout]."""
    g, cin = s.shape
    cout = wsq.shape[0]
    out = torch.empty(g, cout, device=s.device, dtype=torch.float32)
    call("e4s_rowdot_f32", fptr(s), cin, fptr(wsq), None, fptr(out), g, cout, cin, 1, float(conv_scale), stream())
    return out


def weight_sqsum(w, out=None):
    """w [Cout,Cin,kh,kw] -> [Cout,Cin] sum of squares over taps (into `out` when given: the generator keeps ONE such matrix per
    layer for its lifetime, so the style prologue's job tables -- which hold its address -- survive weight updates)."""
    w = _f32(w)
    cout, cin = w.shape[:2]
    taps = w.shape[2] * w.shape[3]
    if out is None:
        out = torch.empty(cout, cin, device=w.device, dtype=torch.float32)
    elif out.shape != (cout, cin) or out.dtype != torch.float32 or out.device != w.device or not out.is_contiguous():
        raise RuntimeError("weight_sqsum: out must be a contiguous fp32 [Cout,Cin] tensor on the weight's device")
    call("e4s_weight_sqsum_f32", fptr(w), fptr(out), cout, cin, taps, stream())
    return out


def _into(out, shape, device):
    """`out` if given (checked: contiguous fp32 of exactly `shape` on `device`), else a fresh tensor.  The weight-pack builders take `out`
    so that a module can keep ONE buffer per pack for its lifetime: a re-pack after a weight update -- eager, or replayed inside a captured
    train step -- then writes in place and allocates nothing (stylegan2.ModulatedConv2d._buf)."""
    if out is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or out.device != device or not out.is_contiguous():
        raise RuntimeError(f"pack buffer mismatch: need contiguous fp32 {tuple(shape)} on {device}, got {tuple(out.shape)} {out.dtype} {out.device}")
    return out


def pack_taps(w, out=None):
    """w [Cout,Cin,kh,kw] -> [1, kh*kw, Cout, Cin] (the conv kernel's B operand layout)."""
    w = _f32(w)
    cout, cin = w.shape[:2]
    taps = w.shape[2] * w.shape[3]
    out = _into(out, (1, taps, cout, cin), w.device)
    call("e4s_pack_taps_f32", fptr(w), fptr(out), cout, cin, taps, stream())
    return out


def polyphase_weights(w, blur_kernel, out=None):
    """w [Cout,Cin,3,3] + 4x4 blur -> [4, 9, Cout, Cin] phase kernels of the fused transposed conv + blur."""
    w = _f32(w)
    cout, cin = w.shape[:2]
    out = _into(out, (4, 9, cout, cin), w.device)
    call("e4s_polyphase_weights_f32", fptr(w), fptr(_f32(blur_kernel)), fptr(out), cout, cin, stream())
    return out


def polyphase_fold(deff, blur_kernel, cout, cin):
    """Gradient of the 4 x 9 polyphase kernels [36, Cout, Cin] (contiguous) -> gradient of the 3x3 weight [Cout, Cin, 3, 3]: the transpose of
    polyphase_weights."""
    deff = _f32(deff)
    if deff.numel() != 36 * cout * cin:
        raise RuntimeError("polyphase_fold: deff must hold 4 * 9 * Cout * Cin values")
    dw = torch.empty(cout, cin, 3, 3, device=deff.device, dtype=torch.float32)
    call("e4s_polyphase_fold_f32", fptr(deff), fptr(_f32(blur_kernel)), fptr(dw), cout, cin, stream())
    return dw


def rgb_weights(w, s, scale):
    """ws[g,c,ci] = scale*w[c,ci]*s[g,ci]; w [3,Cin]."""
    g, cin = s.shape
    out = torch.empty(g, 3, cin, device=s.device, dtype=torch.float32)
    call("e4s_rgb_weights_f32", fptr(w), fptr(s), fptr(out), g, cin, float(scale), stream())
    return out


# ---- mask plan -------------------------------------------------------------------------------
_tls = threading.local()


@contextlib.contextmanager
def flag_sink(flags):
    """While active (HIP-graph capture), every mask_labels call OF THIS THREAD ORs its not-one-hot flag into `flags`
    (int32[1] device tensor) instead of a fresh tensor, so that a replayed graph can still be validated afterwards.
    Thread-local: another thread's strict-mask check keeps reading its own flag tensor."""
    prev = getattr(_tls, "sink", None)
    _tls.sink = flags
    try:
        yield flags
    finally:
        _tls.sink = prev


def mask_labels(mask):
    """one-hot [B,R,Hm,Wm] fp32 -> (labels uint8 [B,Hm,Wm], flags int32[1]); flags!=0 => not one-hot."""
    mask = _f32(mask)
    b, r, hm, wm = mask.shape
    labels = torch.empty(b, hm, wm, device=mask.device, dtype=torch.uint8)
    sink = getattr(_tls, "sink", None)
    flags = sink if sink is not None else torch.zeros(1, device=mask.device, dtype=torch.int32)
    call("e4s_mask_labels", fptr(mask), ptr(labels), ptr(flags), b, r, hm, wm, stream())
    return labels, flags


def swap_styles(tgt, src, comp_indices, below_face=False):
    """scripts/face_swap.py:117-146 per sample, one launch (e4s_swap_styles_f32): tgt / src [B,R,C] -> [B,R,C]."""
    tgt, src = _f32(tgt), _f32(src)
    b, r, c = tgt.shape
    out = torch.empty_like(tgt)
    sel = 0
    for i in comp_indices:
        sel |= 1 << int(i)
    call("e4s_swap_styles_f32", fptr(tgt), fptr(src), fptr(out), b, r, c, sel, 7, 9, 8 if below_face else -1, stream())
    return out


class RowPlan:
    __slots__ = ("rows", "tiles", "meta", "tiles_cap", "Ha", "Wa", "nphase", "R")


def region_plan(labels, num_regions, ha, wa, nphase):
    b, hm, wm = labels.shape
    nkeys = b * num_regions * nphase
    rows_cap = b * ha * wa * nphase + nkeys * BM
    tiles_cap = (rows_cap + BM - 1) // BM
    dev = labels.device
    pl = RowPlan()
    pl.rows = torch.empty(rows_cap, device=dev, dtype=torch.int32)
    pl.tiles = torch.empty(tiles_cap * 4, device=dev, dtype=torch.int32)
    pl.meta = torch.empty(4, device=dev, dtype=torch.int32)
    work = torch.empty(3 * nkeys, device=dev, dtype=torch.int32)
    call("e4s_region_plan", ptr(labels), b, num_regions, hm, wm, ha, wa, nphase, BM, ptr(pl.rows), ptr(pl.tiles),
         ptr(pl.meta), ptr(work), rows_cap, tiles_cap, stream())
    pl.tiles_cap, pl.Ha, pl.Wa, pl.nphase, pl.R = tiles_cap, ha, wa, nphase, num_regions
    return pl


# ---- the conv --------------------------------------------------------------------------------
_split_tl = threading.local()


@contextlib.contextmanager
def f32_plain_split(on=True):
    """Inside this context e4s_conv_mfma_f32 may split K on plain maps with < 256 blocks per sample (e4s_conv_params.split_hint): the frozen
    loss networks opt in (criteria.py, forward AND backward -- autograd runs the backward on its own thread, so each Function sets it again)."""
    prev = getattr(_split_tl, "on", False)
    _split_tl.on = bool(on)
    try:
        yield
    finally:
        _split_tl.on = prev


def conv_mfma(x, w, cout, *, plan=None, istride=1, ostride=1, ntaps=9, ncls=1, in_scale=None, out_scale=None,
              noise=None, noise_w=None, noise_per_channel=False, bias=None, slope=None, act=0, alpha=0.2,
              gain=LRELU_GAIN, spatial=None, anchors=None, labels=None, num_regions=1, w_split=None, in_stats=None,
              out=None, tap_shift=0, want_stats=False, w_split16=None, se=None):
    """x NHWC [B,Hi,Wi,Cin]; w [ncls, ntaps, Cout, Cin] -> y NHWC [B,Ho,Wo,Cout].
    anchors = (Ha, Wa); defaults: up-conv (ncls=4) anchors = input grid, output 2x;
    strided conv anchors = output grid.
    w_split: the split-bf16 image of w (split_bf16x2); when given the contraction runs on e4s_conv_bf16x3_f32
    (callers check bf16x3_eligible first -- an ineligible shape is an error, not a silent fp32 run).
    w_split16: (masked layers, with w_split) the 16-channel-chunk split image (split16_bf16x2): the contraction runs on
    e4s_conv_region_bf16x3_f32 (variant-rows kernel; the region-select kernel for tiles / launches it does not take).
    in_stats: [B,Cin,2] InstanceNorm statistics of x; the normalisation is applied while the input is staged (split-bf16
    kernel only).
    out: write the Cout channels into the FIRST channels of this wider NHWC tensor [B,Ho,Wo,Cy >= Cout] (returned).
    tap_shift: gather kernels only: 1 = padding-0 strided conv (input coord = anchor*istride + tap).
    se = (fc1 [Cr,C], fc2 [C,Cr]) with want_stats: the second element of the returned pair is the SE gate [B,Cout] of the
    normalised output instead of `pooled` (one launch with the statistics' second stage where the epilogue emitted them).
    want_stats: also return the InstanceNorm statistics of the OUTPUT, (y, (stats [B,Cout,2], pooled [B,Cout])): emitted by
    the split-bf16 kernels' epilogue (no extra pass over y) where that applies, by e4s_instnorm_stats_f32 otherwise."""
    b, hi, wi, cin = x.shape
    if anchors is None:
        anchors = (hi // istride, wi // istride)
    ha, wa = anchors
    ho, wo = ha * ostride, wa * ostride
    if spatial is None:
        spatial = plan is None and istride == 1 and ntaps == 9
    if labels is not None and not spatial:
        raise RuntimeError("per-pixel region labels need the spatial (halo-tiled) mode")
    if plan is None and not spatial and (ha * wa) % BM != 0 and w_split is None and (in_scale is not None or out_scale is not None):
        # with a per-sample style / demodulation row the tiles must not straddle samples: a trivial one-region plan
        plan = region_plan(torch.zeros(b, 1, 1, device=x.device, dtype=torch.uint8), 1, ha, wa, ncls)
    if out is None:
        y = torch.empty(b, ho, wo, cout, device=x.device, dtype=torch.float32)
    else:
        y = out
        if tuple(y.shape[:3]) != (b, ho, wo) or y.shape[3] < cout or labels is not None or plan is not None:
            raise RuntimeError("conv_mfma(out=...): need an unlabelled conv and an NHWC buffer [B,Ho,Wo,>=Cout]")
    p = ConvParams()
    p.x, p.w, p.y = fptr(x), fptr(w), fptr(y)
    p.y_cstride = y.shape[3] if out is not None else 0
    p.tap_shift = int(tap_shift)
    if plan is not None:
        p.rows, p.tiles, p.meta, p.tiles_cap = ptr(plan.rows), ptr(plan.tiles), ptr(plan.meta), plan.tiles_cap
        p.groups_per_batch = plan.R
    else:
        p.rows = p.tiles = p.meta = None
        p.tiles_cap = 0
        p.groups_per_batch = num_regions if labels is not None else 1
    if labels is not None:
        p.labels, p.Hm, p.Wm = ptr(labels), labels.shape[1], labels.shape[2]
    else:
        p.labels, p.Hm, p.Wm = None, 0, 0
    p.B, p.Ha, p.Wa = b, ha, wa
    p.Hi, p.Wi, p.Ho, p.Wo, p.Cin, p.Cout = hi, wi, ho, wo, cin, cout
    p.istride, p.ostride, p.ntaps, p.ncls = istride, ostride, ntaps, ncls
    p.in_scale, p.out_scale = fptr(in_scale), fptr(out_scale)
    if noise is not None:
        p.noise, p.noise_w = fptr(noise), fptr(noise_w)
        p.noise_bstride = ho * wo if noise.shape[0] > 1 else 0
        p.noise_per_channel = 1 if noise_per_channel else 0
    else:
        p.noise = p.noise_w = None
        p.noise_bstride = 0
        p.noise_per_channel = 0
    p.bias, p.slope = fptr(bias), fptr(slope)
    p.act, p.alpha, p.gain = act, alpha, gain
    p.in_stats = fptr(in_stats)
    p.split_hint = 1 if getattr(_split_tl, "on", False) else 0
    if w_split is not None:
        gather_ok = (istride == 2 or ntaps == 1) and cin % 32 == 0 and (cout % 128 == 0 or cout == 64) and ncls == 1 and ostride == 1 \
            and plan is None and labels is None and in_scale is None and out_scale is None and noise is None \
            and in_stats is None
        if not gather_ok and (
                not bf16x3_eligible(cin, cout, istride=istride, ostride=ostride, ntaps=ntaps, ncls=ncls,
                                    masked=labels is not None)
                or not spatial or noise_per_channel or (labels is not None and in_scale is None)
                or (in_stats is not None and (in_scale is not None or labels is not None))):
            raise RuntimeError("e4s_conv_bf16x3_f32 does not cover this contraction")
        p.w = fptr(w_split)
        if w_split16 is not None:
            if labels is None or out is not None:
                raise RuntimeError("e4s_conv_region_bf16x3_f32 is the masked-layer kernel")
            nws = lib.load().e4s_conv_region_ws_floats(ctypes.byref(p))    # tile flags, or split-K slabs
            if nws <= 0:
                raise RuntimeError("e4s_conv_region_bf16x3_f32 does not cover this contraction")
            skws = torch.empty(nws, device=x.device, dtype=torch.float32)
            p.splitk_ws = fptr(skws)
            global LAST_REGION_PATH
            LAST_REGION_PATH = lib.load().e4s_conv_region_path(ctypes.byref(p))     # 1: 8-wave kernel, 2: one wave per SIMD (tests / bench)
            call("e4s_conv_region_bf16x3_f32", ctypes.byref(p), ptr(w_split16), stream())
            return y
        nws = lib.load().e4s_conv_bf16x3_ws_floats(ctypes.byref(p))        # split-K partial sums (few-tile launches)
        skws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
        p.splitk_ws = fptr(skws)
        fused = None
        if want_stats and nws == 0 and act == 0 and noise is None and out_scale is None and labels is None and ncls == 1 \
                and cout % 64 == 0 and out is None:
            if gather_ok:
                slots = (ha * wa) // 256 if (ha * wa) % 256 == 0 else 0
            else:
                slots = ((ha + 15) // 16) * ((wa + 15) // 16)
            if slots:
                fused = torch.empty(b * cout * slots * 2, device=x.device, dtype=torch.float64)
                p.stats_ws, p.stats_slots = ptr(fused), slots
        call("e4s_conv_bf16x3_f32", ctypes.byref(p), stream())
        if fused is not None:
            stats = torch.empty(b, cout, 2, device=x.device, dtype=torch.float32)
            pooled = torch.empty(b, cout, device=x.device, dtype=torch.float32)
            if se is not None:
                call("e4s_instnorm_finalize_se_f32", ptr(fused), fptr(stats), fptr(se[0]), fptr(se[1]), fptr(pooled), b, ho * wo,
                     cout, se[0].shape[0], p.stats_slots, 1e-5, stream())
            else:
                call("e4s_instnorm_finalize_f32", ptr(fused), fptr(stats), fptr(pooled), b, ho * wo, cout, p.stats_slots, 1e-5,
                     stream())
            return y, (stats, pooled)
    elif in_stats is not None:
        raise RuntimeError("fused InstanceNorm staging exists only in e4s_conv_bf16x3_f32")
    else:
        nws = lib.load().e4s_conv_mfma_ws_floats(ctypes.byref(p), 1 if spatial else 0)     # split-K slabs (few-block launches)
        skws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
        p.splitk_ws = fptr(skws)
        call("e4s_conv_mfma_f32", ctypes.byref(p), 1 if spatial else 0, stream())
    if want_stats:
        stats, pooled = instnorm_stats(y, want_pooled=True)
        return y, (stats, se_gate(pooled, se[0], se[1]) if se is not None else pooled)
    return y


def bf16x3_eligible(cin, cout, *, istride=1, ostride=1, ntaps=9, ncls=1, masked=False):
    """Shapes e4s_conv_bf16x3_f32 covers (include/e4s_hip.h): natural-order 3x3, plain or polyphase up-conv; column
    tiles of 128 / 64 / 32 without a label map, 128 with one (region-select kernel)."""
    return (cin % 32 == 0 and cout % (128 if masked else 32) == 0 and istride == 1 and ntaps == 9
            and (ncls, ostride) in ((1, 1), (4, 2)))


def want_bf16x3(b, h, w, cin, cout, ncls=1, masked=False):
    """Policy of PRECISION for a natural-order 3x3 conv (ncls = 4: polyphase up-conv) on [b,h,w,cin] -> cout."""
    if PRECISION == "f32" or not bf16x3_eligible(cin, cout, ncls=ncls, ostride=2 if ncls == 4 else 1, masked=masked):
        return False
    if PRECISION == "bf16x3":
        return True
    # 256-pixel x 128/64/32-column tiles (unmasked polyphase up-conv: ONE GEMM with 4*Cout columns)
    n = 4 * cout if (ncls == 4 and not masked) else cout
    bn = 128 if n % 128 == 0 else 64 if n % 64 == 0 else 32
    tiles = b * ((h + 15) // 16) * ((w + 15) // 16) * (n // bn) * (ncls if masked else 1)
    if tiles >= BF16X3_MIN_BLOCKS:
        return True       # (at exactly 128 tiles the kernels also split K two ways: csrc few_tiles_split)
    # few tiles (batch-1 latency runs): the kernels split the input channels over blocks (csrc: few_tiles_split)
    nchunk = cin // 32
    split = min(nchunk // 2, -(-256 // tiles)) if nchunk >= 4 else 1
    return tiles * max(split, 1) >= 64


def split_bf16x2(w, out=None):
    """fp32 [..., Cin] -> its split-bf16 image (hi|lo per 32-channel chunk), returned as an opaque fp32-typed tensor of
    the same shape/byte size (only e4s_conv_bf16x3_f32 reads it)."""
    w = _f32(w)
    out = _into(out, tuple(w.shape), w.device)
    cin = w.shape[-1]
    call("e4s_split_bf16x2_f32", fptr(w), ptr(out), w.numel() // cin, cin, stream())
    return out


def split16_shape(w_shape):
    """Shape of the opaque fp32-typed tensor split16_bf16x2 fills for tap-packed weights of shape w_shape: (k,) + w_shape, k = 2 when Cout % 32 == 0
    (plane-major image + fragment-major image, e4s_split16_bytes), else 1."""
    return (2 if w_shape[-2] % 32 == 0 else 1,) + tuple(w_shape)


def split16_bf16x2(w, out=None):
    """fp32 tap-packed weights [ncls, 9, Cout, Cin] -> the split images e4s_conv_region_bf16x3_f32 reads (opaque): [ncls*9][Cin/16][Cout][16 hi | 16 lo]
    and, for Cout % 32 == 0, behind it the same values fragment-major (the one-wave-per-SIMD kernel's B fragments come straight from it)."""
    w = _f32(w)
    out = _into(out, split16_shape(w.shape), w.device)
    cout, cin = w.shape[-2], w.shape[-1]
    rows = w.numel() // (cout * cin)
    if lib.load().e4s_split16_bytes(rows, cout, cin) != out.numel() * 4:
        raise RuntimeError("split16_bf16x2: buffer size")
    call("e4s_split16_bf16x2_f32", fptr(w), ptr(out), rows, cout, cin, stream())
    return out


REGION_1W = os.environ.get("E4S_REGION_1W", "1") != "0"        # mirrors the library's switch (csrc/conv_region.hip)
LAST_REGION_PATH = 0           # which kernel the last masked launch took (e4s_conv_region_path)
WINO = os.environ.get("E4S_WINO", "1") == "1"        # policy switch of the Winograd F(2,3) kernel (encoders._conv3x3)


def wino_eligible(b, h, w, cin, cout, in_stats=False):
    """Shapes e4s_conv_wino_bf16x3_f32 covers, and whether the launch fills the chip (one 16x16-pixel x 128-column tile per block; launches
    of <= 128 tiles split the input-channel chunks over blocks, csrc/conv_wino.hip:wino_split).  in_stats: the launch folds an
    InstanceNorm into its input transform, whose {mean, rstd} tables live in LDS: Cin <= 1024 (the launcher refuses more)."""
    if not WINO or PRECISION == "f32":
        return False
    if h % 16 or w % 16 or cin % 16 or cin < 32 or cout % 128:
        return False
    if in_stats and cin > 1024:
        return False
    if PRECISION == "bf16x3":
        return True
    tiles = b * (h // 16) * (w // 16) * (cout // 128)
    if tiles >= BF16X3_MIN_BLOCKS:
        return True
    nchunk = cin // 16
    split = min(nchunk // 2, -(-256 // tiles)) if nchunk >= 4 else 1
    return tiles * max(split, 1) >= 64


def wino_weights(w9):
    """tap-packed fp32 weights [1, 9, Cout, Cin] (pack_taps) -> the transformed, hi/lo-split operand of conv_wino (opaque uint8)."""
    w9 = _f32(w9)
    cout, cin = w9.shape[-2], w9.shape[-1]
    if w9.numel() != 9 * cout * cin:
        raise RuntimeError("wino_weights: one set of nine taps [9, Cout, Cin]")
    n = lib.load().e4s_wino_weights_bytes(cout, cin)
    if n <= 0:
        raise RuntimeError("wino_weights: Cin must be a multiple of 16")
    out = torch.empty(n, device=w9.device, dtype=torch.uint8)
    call("e4s_wino_weights_f32", fptr(w9), ptr(out), cout, cin, stream())
    return out


def conv_wino(x, u, cout, *, in_stats=None, bias=None, slope=None, act=0, alpha=0.2, gain=LRELU_GAIN, want_stats=False, se=None):
    """Stride-1 3x3 conv of x NHWC [B,H,W,Cin] as Winograd F(2,3) along the rows (e4s_conv_wino_bf16x3_f32); u = wino_weights(...).
    in_stats / act / slope / want_stats / se as conv_mfma."""
    b, h, w, cin = x.shape
    y = torch.empty(b, h, w, cout, device=x.device, dtype=torch.float32)
    p = ConvParams()
    p.x, p.w, p.y = fptr(x), ptr(u), fptr(y)
    p.rows = p.tiles = p.meta = None
    p.B, p.Ha, p.Wa, p.Hi, p.Wi, p.Ho, p.Wo, p.Cin, p.Cout = b, h, w, h, w, h, w, cin, cout
    p.istride, p.ostride, p.ntaps, p.ncls, p.groups_per_batch = 1, 1, 9, 1, 1
    p.bias, p.slope = fptr(bias), fptr(slope)
    p.act, p.alpha, p.gain = act, alpha, gain
    p.in_stats = fptr(in_stats)
    fused = None
    if want_stats and act != 0:
        raise RuntimeError("conv_wino: output statistics are those of the raw conv output (act = 0)")
    nws = lib.load().e4s_conv_wino_ws_floats(ctypes.byref(p))            # split-K slabs (few-tile launches)
    skws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
    p.splitk_ws = fptr(skws)
    if want_stats and not nws:
        slots = (h // 16) * (w // 16)
        fused = torch.empty(b * cout * slots * 2, device=x.device, dtype=torch.float64)
        p.stats_ws, p.stats_slots = ptr(fused), slots
    call("e4s_conv_wino_bf16x3_f32", ctypes.byref(p), stream())
    if fused is None:
        if not want_stats:
            return y
        stats, pooled = instnorm_stats(y, want_pooled=True)              # split launch: the separate statistics pass
        return y, (stats, se_gate(pooled, se[0], se[1]) if se is not None else pooled)
    stats = torch.empty(b, cout, 2, device=x.device, dtype=torch.float32)
    pooled = torch.empty(b, cout, device=x.device, dtype=torch.float32)
    if se is not None:
        call("e4s_instnorm_finalize_se_f32", ptr(fused), fptr(stats), fptr(se[0]), fptr(se[1]), fptr(pooled), b, h * w, cout,
             se[0].shape[0], p.stats_slots, 1e-5, stream())
    else:
        call("e4s_instnorm_finalize_f32", ptr(fused), fptr(stats), fptr(pooled), b, h * w, cout, p.stats_slots, 1e-5, stream())
    return y, (stats, pooled)


def upconv_mfma(x, w3, cout, k4, *, in_scale=None, out_scale=None, labels=None, num_regions=1, noise=None,
                noise_w=None, noise_per_channel=False, bias=None, act=0, alpha=0.2, gain=LRELU_GAIN, out=None):
    """Exact transposed-conv + blur up-sampling conv.  x NHWC [B,H,W,Cin]; w3 [1,9,Cout,Cin] (plain taps);
    k4 the 4x4 blur kernel -> y NHWC [B,2H,2W,Cout]."""
    b, hi, wi, cin = x.shape
    ho, wo = 2 * hi, 2 * wi
    y = torch.empty(b, ho, wo, cout, device=x.device, dtype=torch.float32) if out is None else out
    p = ConvParams()
    p.x, p.w, p.y = fptr(x), fptr(w3), fptr(y)
    p.y_cstride = y.shape[3] if out is not None else 0
    p.rows = p.tiles = p.meta = None
    p.tiles_cap = 0
    p.B, p.Ha, p.Wa = b, hi, wi
    p.Hi, p.Wi, p.Ho, p.Wo, p.Cin, p.Cout = hi, wi, ho, wo, cin, cout
    p.istride, p.ostride, p.ntaps, p.ncls = 1, 2, 9, 1
    p.in_scale, p.out_scale = fptr(in_scale), fptr(out_scale)
    p.groups_per_batch = num_regions if labels is not None else 1
    if labels is not None:
        p.labels, p.Hm, p.Wm = ptr(labels), labels.shape[1], labels.shape[2]
    else:
        p.labels, p.Hm, p.Wm = None, 0, 0
    if noise is not None:
        p.noise, p.noise_w = fptr(noise), fptr(noise_w)
        p.noise_bstride = ho * wo if noise.shape[0] > 1 else 0
        p.noise_per_channel = 1 if noise_per_channel else 0
    else:
        p.noise = p.noise_w = None
        p.noise_bstride = 0
        p.noise_per_channel = 0
    p.bias, p.slope = fptr(bias), None
    p.act, p.alpha, p.gain = act, alpha, gain
    call("e4s_upconv_mfma_f32", ctypes.byref(p), fptr(_f32(k4)), stream())
    return y


def subpixel_weights(w, out=None):
    """w [Cout,Cin,3,3] -> the packed, hi/lo-split sub-pixel GEMM operand of e4s_upconv_bf16x3_f32 (csrc/upconv_bf16x3.hip):
    [Cin/32, Cout/32, 9 blocks, 32 co, 32 hi | 32 lo bf16], returned as an opaque fp32-typed tensor of the same byte size."""
    w = _f32(w)
    cout, cin = w.shape[:2]
    out = _into(out, (cin // 32, cout // 32, 9, 32, 32), w.device)
    call("e4s_subpixel_weights_f32", fptr(w), ptr(out), cout, cin, stream())
    return out


def upconv_bf16x3_eligible(cin, cout):
    """e4s_upconv_bf16x3_f32: 32-channel output tiles, 32-channel chunks of packed weights, the sample's style row (<= 512 channels) in LDS."""
    return cin % 64 == 0 and cin <= 512 and cout % 32 == 0


def upconv_bf16x3(x, w_sub, cout, k4, *, in_scale=None, out_scale=None, noise=None, noise_w=None, bias=None, act=0,
                  alpha=0.2, gain=LRELU_GAIN, out=None):
    """Exact transposed conv + blur + noise + bias + act on the split-bf16 matrix-core path, one style per sample.
    x NHWC [B,H,W,Cin]; w_sub = subpixel_weights(w); k4 the 4x4 blur kernel (device tensor) -> NHWC [B,2H,2W,Cout]."""
    b, hi, wi, cin = x.shape
    ho, wo = 2 * hi, 2 * wi
    y = torch.empty(b, ho, wo, cout, device=x.device, dtype=torch.float32) if out is None else out
    p = ConvParams()
    p.x, p.w, p.y = fptr(x), fptr(w_sub), fptr(y)
    p.y_cstride = y.shape[3] if out is not None else 0
    p.rows = p.tiles = p.meta = None
    p.tiles_cap = 0
    p.B, p.Ha, p.Wa = b, hi, wi
    p.Hi, p.Wi, p.Ho, p.Wo, p.Cin, p.Cout = hi, wi, ho, wo, cin, cout
    p.istride, p.ostride, p.ntaps, p.ncls = 1, 2, 9, 1
    p.in_scale, p.out_scale = fptr(in_scale), fptr(out_scale)
    p.groups_per_batch = 1
    p.labels, p.Hm, p.Wm = None, 0, 0
    if noise is not None:
        p.noise, p.noise_w = fptr(noise), fptr(noise_w)
        p.noise_bstride = ho * wo if noise.shape[0] > 1 else 0
    else:
        p.noise = p.noise_w = None
        p.noise_bstride = 0
    p.noise_per_channel = 0
    p.bias, p.slope = fptr(bias), None
    p.act, p.alpha, p.gain = act, alpha, gain
    call("e4s_upconv_bf16x3_f32", ctypes.byref(p), fptr(_f32(k4)), stream())
    return y


def conv_c32(x, w_split, cout, *, in_scale=None, out_scale=None, noise=None, noise_w=None, bias=None, act=0, alpha=0.2,
             gain=LRELU_GAIN, rgb_ws=None):
    """3x3 stride-1 conv with 32 input channels on e4s_conv_c32_bf16x3_f32 (weights resident in LDS).  x NHWC [B,H,W,32];
    w_split = split_bf16x2(pack_taps(w)); rgb_ws [B,3,32]: also return the ToRGB partial [B,3,H,W] of the OUTPUT (Cout == 32).
    Returns y NHWC [B,H,W,Cout] or (y, rgb_partial)."""
    b, h, w, cin = x.shape
    if cin != 32 or cout % 32:
        raise RuntimeError("conv_c32: Cin == 32 and Cout % 32 == 0")
    y = torch.empty(b, h, w, cout, device=x.device, dtype=torch.float32)
    p = ConvParams()
    p.x, p.w, p.y = fptr(x), fptr(w_split), fptr(y)
    p.y_cstride = 0
    p.rows = p.tiles = p.meta = None
    p.tiles_cap = 0
    p.B, p.Ha, p.Wa = b, h, w
    p.Hi, p.Wi, p.Ho, p.Wo, p.Cin, p.Cout = h, w, h, w, cin, cout
    p.istride, p.ostride, p.ntaps, p.ncls = 1, 1, 9, 1
    p.in_scale, p.out_scale = fptr(in_scale), fptr(out_scale)
    p.groups_per_batch = 1
    p.labels, p.Hm, p.Wm = None, 0, 0
    if noise is not None:
        p.noise, p.noise_w = fptr(noise), fptr(noise_w)
        p.noise_bstride = h * w if noise.shape[0] > 1 else 0
    else:
        p.noise = p.noise_w = None
        p.noise_bstride = 0
    p.noise_per_channel = 0
    p.bias, p.slope = fptr(bias), None
    p.act, p.alpha, p.gain = act, alpha, gain
    partial = None
    if rgb_ws is not None:
        if cout != 32 or tuple(rgb_ws.shape) != (b, 3, 32):
            raise RuntimeError("conv_c32: the fused ToRGB partial needs Cout == 32 and rgb_ws [B,3,32]")
        partial = torch.empty(b, 3, h, w, device=x.device, dtype=torch.float32)
    call("e4s_conv_c32_bf16x3_f32", ctypes.byref(p), fptr(rgb_ws), fptr(partial), stream())
    return y if partial is None else (y, partial)


def torgb_finish(partial, bias, skip, k4):
    """partial [B,3,H,W] (the fused ToRGB contraction) + bias + FIR-upsampled skip [B,3,H/2,W/2] -> [B,3,H,W]."""
    b, _, h, w = partial.shape
    out = torch.empty_like(partial)
    call("e4s_torgb_finish_f32", fptr(partial), fptr(_f32(bias)), fptr(skip), fptr(_f32(k4)) if skip is not None else None,
         fptr(out), b, h, w, stream())
    return out


def torgb(x, ws, bias, skip, k4, labels, num_regions):
    """x NHWC [B,H,W,Cin]; ws [G,3,Cin]; skip NCHW [B,3,H/2,W/2] or None -> NCHW [B,3,H,W]."""
    b, h, w, cin = x.shape
    out = torch.empty(b, 3, h, w, device=x.device, dtype=torch.float32)
    hm = wm = 0
    if labels is not None:
        hm, wm = labels.shape[1:]
    call("e4s_torgb_f32", fptr(x), fptr(ws), fptr(bias), fptr(skip), fptr(k4), ptr(labels), hm, wm, num_regions,
         fptr(out), b, h, w, cin, stream())
    return out


def mask_mul_add(y, mask, r, out, channels_last):
    """out (+)= y * nearest(mask)[:, r]  (soft-mask fallback); out=None allocates (overwrite)."""
    if channels_last:
        b, h, w, c = y.shape
    else:
        b, c, h, w = y.shape
    acc = 1 if out is not None else 0
    if out is None:
        out = torch.empty_like(y)
    mask = _f32(mask)
    call("e4s_mask_mul_add_f32", fptr(y), fptr(mask), fptr(out), r, b, h, w, c, mask.shape[1], mask.shape[2],
         mask.shape[3], 1 if channels_last else 0, acc, stream())
    return out


def noise_bias_act_nhwc(x, noise, noise_w, bias, alpha, gain):
    b, h, w, c = x.shape
    y = torch.empty_like(x)
    nb = 0 if noise is None or noise.shape[0] == 1 else h * w
    call("e4s_noise_bias_act_nhwc_f32", fptr(x), fptr(noise), fptr(noise_w) if noise is not None else None, nb,
         fptr(bias), fptr(y), b, h * w, c, float(alpha), float(gain), stream())
    return y


# ---- encoder ---------------------------------------------------------------------------------
def resize_bilinear_to_nhwc(x, ho, wo):
    x = _f32(x)
    b, c, hi, wi = x.shape
    y = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.float32)
    call("e4s_resize_bilinear_f32", fptr(x), fptr(y), b, c, hi, wi, ho, wo, stream())
    return y


def conv3x3_small(x, w, want_stats=False):
    """Stem conv (3x3, pad 1, tiny Cin): x NHWC [B,H,W,Cin], w [Cout,Cin,3,3] -> NHWC [B,H,W,Cout].  Cin = 3, Cout = 64 on maps that are
    multiples of 16 run on the tiled kernel (e4s_conv3x3_stem_f32); want_stats: (y, InstanceNorm statistics of y [B,Cout,2]), emitted by
    that kernel's epilogue where it applies, by e4s_instnorm_stats_f32 otherwise."""
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    y = torch.empty(b, h, wd, cout, device=x.device, dtype=torch.float32)
    if cin == 3 and cout == 64 and h % 16 == 0 and wd % 16 == 0:
        slots = (h // 16) * (wd // 16)
        fused = torch.empty(b * cout * slots * 2, device=x.device, dtype=torch.float64) if want_stats else None
        call("e4s_conv3x3_stem_f32", fptr(x), fptr(_f32(w)), fptr(y), ptr(fused), b, h, wd, cin, cout, stream())
        if not want_stats:
            return y
        stats = torch.empty(b, cout, 2, device=x.device, dtype=torch.float32)
        call("e4s_instnorm_finalize_f32", ptr(fused), fptr(stats), None, b, h * wd, cout, slots, 1e-5, stream())
        return y, stats
    call("e4s_conv3x3_small_f32", fptr(x), fptr(w), fptr(y), b, h, wd, cin, cout, stream())
    if want_stats:
        return y, instnorm_stats(y)[0]
    return y


def instnorm_stats(x, want_pooled=False, eps=1e-5):
    b, h, w, c = x.shape
    stats = torch.empty(b, c, 2, device=x.device, dtype=torch.float32)
    pooled = torch.empty(b, c, device=x.device, dtype=torch.float32) if want_pooled else None
    ws = torch.empty(lib.load().e4s_instnorm_ws_doubles(b, h * w, c), device=x.device, dtype=torch.float64)
    call("e4s_instnorm_stats_f32", fptr(x), fptr(stats), fptr(pooled), ptr(ws), b, h * w, c, float(eps), stream())
    return stats, pooled


def instnorm_apply(x, stats, gate=None, res=None, res_stats=None, slope=None, rs=1, want_stats=False):
    """want_stats: also return the InstanceNorm statistics of the OUTPUT ((y, stats [B,C,2])), accumulated by the same
    pass (the next encoder unit normalises this tensor first thing, helpers.py:128)."""
    b, h, w, c = x.shape
    y = torch.empty_like(x)
    if want_stats and c % 64 == 0:
        ws = torch.empty(lib.load().e4s_instnorm_ws_doubles(b, h * w, c), device=x.device, dtype=torch.float64)
        nslots = ctypes.c_int(0)
        call("e4s_instnorm_apply_stats_f32", fptr(x), fptr(stats), fptr(gate), fptr(res), fptr(res_stats), fptr(slope),
             fptr(y), ptr(ws), ctypes.byref(nslots), b, h, w, c, rs, stream())
        out_stats = torch.empty(b, c, 2, device=x.device, dtype=torch.float32)
        call("e4s_instnorm_finalize_f32", ptr(ws), fptr(out_stats), None, b, h * w, c, nslots.value, 1e-5, stream())
        return y, out_stats
    call("e4s_instnorm_apply_f32", fptr(x), fptr(stats), fptr(gate), fptr(res), fptr(res_stats), fptr(slope), fptr(y),
         b, h, w, c, rs, stream())
    if want_stats:
        return y, instnorm_stats(y)[0]
    return y


def se_gate(pooled, fc1, fc2):
    b, c = pooled.shape
    cr = fc1.shape[0]
    gate = torch.empty(b, c, device=pooled.device, dtype=torch.float32)
    call("e4s_se_gate_f32", fptr(pooled), fptr(fc1), fptr(fc2), fptr(gate), b, c, cr, stream())
    return gate


def region_mean_into(feats, labels, out, num_regions, out_off):
    b, h, w, c = feats.shape
    hm, wm = labels.shape[1:]
    call("e4s_region_mean_f32", fptr(feats), ptr(labels), hm, wm, fptr(out), b, h, w, c, num_regions,
         out.shape[2], out_off, stream())


def grouped_linear(x, w, bias, add, scale, act=0, alpha=0.01):
    """x [B,R,K]; w [R,O,K]; bias [R,O]; add [O] or None -> [B,R,O]."""
    b, r, k = x.shape
    o = w.shape[1]
    y = torch.empty(b, r, o, device=x.device, dtype=torch.float32)
    call("e4s_grouped_linear_f32", fptr(x), fptr(w), fptr(bias), fptr(add), fptr(y), b, r, k, o, float(scale), act,
         float(alpha), stream())                    # any batch: each weight row is read once and kept in registers
    return y


def grouped_linear_t(g, w, scale, base=None, mul=None, ref=None, alpha=1.0):
    """out[b,r,k] = base + mul * scale * (ref > 0 ? 1 : alpha) * sum_o g[b,r,o] * w[r,o,k]   (g [B,R,O]; w [R,O,K]):
    the transposed contraction of the LocalMLP backward and of the style prologue's chain rule, weights streamed once
    per 16 samples, split partial sums added in order (bit-reproducible)."""
    b, r, o = g.shape
    k = w.shape[2]
    out = torch.empty(b, r, k, device=g.device, dtype=torch.float32)
    for lo in range(0, b, 16):
        n = min(16, b - lo)
        ws = torch.empty(lib.load().e4s_grouped_linear_t_ws_floats(n, r, o, k), device=g.device, dtype=torch.float32)
        sl = slice(lo, lo + n)
        call("e4s_grouped_linear_t_f32", fptr(g[sl]), fptr(w), fptr(out[sl]), fptr(ws), n, r, o, k, float(scale),
             fptr(base[sl]) if base is not None else None, fptr(mul[sl]) if mul is not None else None,
             fptr(ref[sl]) if ref is not None else None, float(alpha), stream())
    return out


# the style-gradient tails of the generator backward batched over the layers (E4S_STYLE_GRAD_MULTI=0: one chain per layer, for A/B runs)
STYLE_GRAD_MULTI = os.environ.get("E4S_STYLE_GRAD_MULTI", "1") != "0"


def style_grad_multi(jobs, b, r, nl, sdim, device):
    """The style-gradient tail of the generator backward for every layer in two launches (e4s_style_grad_multi_f32).
    jobs: dicts in the order their layers' gradients arrive, each
        StyledConv: ds_raw [G,Cin], dd_d [G,Cout], d [G,Cout], s [G,Cin], wsq [Cout,Cin]
        ToRGB:      dws [G,3,Cin], w3 [3,Cin], conv_scale
        both:       wmod [Cin,S], mod_scale, slot, masked
    Returns (dlat [B,R,NL,S], [ds_total [G,Cin] per job]): dL/ds per layer including the demodulation path, dL/dlatent with every
    layer's contribution in its slot (jobs of one slot added in job order)."""
    if len(jobs) > lib.STYLE_GRAD_MAX_JOBS:
        raise RuntimeError(f"{len(jobs)} style-gradient jobs: the ABI carries at most {lib.STYLE_GRAD_MAX_JOBS}")
    arr = (lib.StyleGradJob * max(len(jobs), 1))()
    ws = torch.empty(sum(j["G"] * j["Cin"] for j in jobs), device=device, dtype=torch.float32)
    outs, keep, off = [], [], 0
    for q, j in zip(arr, jobs):
        n = j["G"] * j["Cin"]
        out = ws[off:off + n].view(j["G"], j["Cin"])
        off += n
        outs.append(out)
        rgb = "dws" in j
        for name in (("dws", "w3") if rgb else ("ds_raw", "dd_d", "d", "s", "wsq")) + ("wmod",):
            t = _f32(j[name])
            keep.append(t)
            setattr(q, name, fptr(t).value)
        q.ds_total = fptr(out).value
        q.conv_scale = float(j.get("conv_scale", 0.0))
        q.mod_scale = float(j["mod_scale"])
        q.G, q.Cin, q.Cout, q.slot, q.masked = j["G"], j["Cin"], j.get("Cout", 0), j["slot"], int(j["masked"])
    dlat = torch.empty(b, r, nl, sdim, device=device, dtype=torch.float32)
    call("e4s_style_grad_multi_f32", ctypes.cast(arr, ctypes.c_void_p), len(jobs), fptr(dlat), b, r, nl, sdim, stream())
    return dlat, outs


def grouped_outer(g, h, scale):
    """dw[r,o,k] = scale * sum_b g[b,r,o] * h[b,r,k]."""
    b, r, o = g.shape
    k = h.shape[2]
    dw = torch.empty(r, o, k, device=g.device, dtype=torch.float32)
    call("e4s_grouped_outer_f32", fptr(g), fptr(h), fptr(dw), b, r, o, k, float(scale), stream())
    return dw


def batch_sum(x):
    """sum over dim 0 of a contiguous [B, ...] tensor."""
    x = _f32(x)
    out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
    call("e4s_batch_sum_f32", fptr(x), fptr(out), x.shape[0], out.numel(), stream())
    return out


def colsum(x):
    """[C] = sum over every dim but the last of a contiguous channels-last tensor (ordered two-level reduction)."""
    x = _f32(x)
    c = x.shape[-1]
    rows = x.numel() // c
    if rows <= 64 or c % 4 or c > 1024:
        return batch_sum(x.view(rows, c))
    out = torch.empty(c, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_colsum_ws_floats(rows, c), device=x.device, dtype=torch.float32)
    call("e4s_colsum_f32", fptr(x), fptr(out), fptr(ws), rows, c, stream())
    return out


def sum_all(x):
    """0-dim sum of every element on the native ordered two-level column sum (bit-reproducible).  Also the reason it exists: ATen's large
    reductions allocate a semaphore buffer and clear it with hipMemsetAsync -- captured into a HIP graph that is a MEMSET NODE, and memset
    nodes misbehave on replay on this ROCm (round 3: the region plan's memsets faulted, profiles/r03a_plan_fault_ab.json; round 5: a
    captured train step with `gz.sum((0, 1, 2))` in it came back with its loss scalar zeroed, profiles/r05_graphed_trainG_bisect.json).
    Nothing this library captures issues a memset."""
    x = _f32(x)
    n = x.numel()
    if n % 4 == 0 and n // 4 > 64:
        return colsum(x.view(-1, 4)).sum()                 # the last 4 values: a single-block ATen reduce (no semaphores)
    return x.sum()


def adam_step(p, grad, m, v, lr, beta1, beta2, eps, weight_decay, step):
    """torch.optim.Adam's update of one fp32 tensor, in place, as ONE kernel."""
    call("e4s_adam_step_f32", fptr(p), fptr(_f32(grad)), fptr(m), fptr(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), int(step), stream())


def _ptr_array(tensors, dtype=torch.float32):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError(f"multi-tensor call: contiguous {dtype} ROCm tensors expected")
        arr[i] = t.data_ptr()
    return arr


def advance_steps(steps):
    """steps (device int64, any shape) += 1 in one launch."""
    if steps.dtype != torch.int64 or not steps.is_cuda:
        raise RuntimeError("advance_steps: device int64 tensor")
    call("e4s_advance_i64", ptr(steps), steps.numel(), stream())


def adam_step_dev(p, grad, m, v, lr, beta1, beta2, eps, weight_decay, step, lr_dev=None):
    """adam_step with the step count (already advanced: the step being taken) in a device int64[1] tensor and, optionally, the
    learning rate in a device float64[1] tensor: capturable in a HIP graph."""
    if step.dtype != torch.int64 or not step.is_cuda:
        raise RuntimeError("adam_step_dev: step must be a device int64 tensor")
    if lr_dev is not None and (lr_dev.dtype != torch.float64 or not lr_dev.is_cuda):
        raise RuntimeError("adam_step_dev: lr_dev must be a device float64 tensor")
    call("e4s_adam_step_dev_f32", fptr(p), fptr(_f32(grad)), fptr(m), fptr(v), p.numel(), float(lr), ptr(lr_dev), float(beta1),
         float(beta2), float(eps), float(weight_decay), ptr(step), stream())


def adam_multi_dev(ps, grads, ms, vs, steps, lr, beta1, beta2, eps, weight_decay, lr_dev=None):
    """adam_step_dev for lists of tensors: ceil(len / 48) launches (pointers travel in the kernel arguments)."""
    n = len(ps)
    if not (n == len(grads) == len(ms) == len(vs) == len(steps)):
        raise RuntimeError("adam_multi_dev: list lengths differ")
    if lr_dev is not None and (lr_dev.dtype != torch.float64 or not lr_dev.is_cuda):
        raise RuntimeError("adam_multi_dev: lr_dev must be a device float64 tensor")
    for p, g in zip(ps, grads):
        if g.shape != p.shape:
            raise RuntimeError("adam_multi_dev: gradient shape differs from its parameter's")
    sizes = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
    call("e4s_adam_multi_dev_f32", n, _ptr_array(ps), _ptr_array(grads), _ptr_array(ms), _ptr_array(vs), sizes,
         _ptr_array(steps, torch.int64), float(lr), ptr(lr_dev), float(beta1), float(beta2), float(eps), float(weight_decay), stream())


def ema_multi_(dsts, srcs, decay):
    """dst <- dst * decay + src * (1 - decay) for lists of tensors: ceil(len / 48) launches; version counters advance."""
    n = len(dsts)
    if n != len(srcs):
        raise RuntimeError("ema_multi_: list lengths differ")
    for d, s_ in zip(dsts, srcs):
        if d.shape != s_.shape:
            raise RuntimeError("ema_multi_: shapes differ")
    sizes = (ctypes.c_int64 * n)(*[d.numel() for d in dsts])
    call("e4s_ema_multi_f32", n, _ptr_array(dsts), _ptr_array(srcs), sizes, float(decay), stream())
    torch.autograd.graph.increment_version(dsts)         # raw-pointer writes: keep (data_ptr, _version) pack keys honest


def ema_(dst, src, decay):
    """dst <- dst * decay + src * (1 - decay) in place (torch_utils.accumulate, one launch per tensor)."""
    if dst.dtype != torch.float32 or src.dtype != torch.float32 or dst.shape != src.shape or not dst.is_contiguous():
        raise RuntimeError("ema_: contiguous fp32 tensors of one shape")
    call("e4s_ema_f32", fptr(dst), fptr(src.contiguous()), dst.numel(), float(decay), stream())
    torch.autograd.graph.increment_version(dst)          # raw-pointer write: keep (data_ptr, _version) pack keys honest
    return dst


LAST_WGRAD_PATH = 0            # which kernel the last conv_wgrad launch took (e4s_conv_wgrad_path)


def conv_wgrad(gz, x, *, ntaps=9, istride=1, anchors=None, ostride=1, phase=(0, 0), s=None, d=None, labels=None,
               num_regions=1, tap_shift=0):
    """dw [ntaps, Cout, Cin] of a 3x3 / 1x1 conv (see e4s_conv_wgrad_f32): gz NHWC [B,Ho,Wo,Cout], x NHWC [B,Hi,Wi,Cin].
    anchors default to the output grid (ostride 1) / the input grid (polyphase phase, ostride 2)."""
    b, ho, wo, cout = gz.shape
    _, hi, wi, cin = x.shape
    if anchors is None:
        anchors = (ho // ostride, wo // ostride)
    dw = torch.empty(ntaps, cout, cin, device=x.device, dtype=torch.float32)
    p = ConvWgradParams()
    p.gz, p.x, p.dw, p.s, p.d = fptr(_f32(gz)), fptr(x), fptr(dw), fptr(s), fptr(d)
    if labels is not None:
        p.labels, p.Hm, p.Wm, p.R = ptr(labels), labels.shape[1], labels.shape[2], num_regions
    else:
        p.labels, p.Hm, p.Wm, p.R = None, 0, 0, 1
    p.B, p.Hi, p.Wi, p.Cin, p.Ha, p.Wa, p.Ho, p.Wo, p.Cout = b, hi, wi, cin, anchors[0], anchors[1], ho, wo, cout
    p.istride, p.ostride, p.py, p.px, p.ntaps = istride, ostride, phase[0], phase[1], ntaps
    p.tap_shift = int(tap_shift)
    p.ws = None
    global LAST_WGRAD_PATH
    LAST_WGRAD_PATH = lib.load().e4s_conv_wgrad_path(ctypes.byref(p))          # 1: split-bf16 kernel, 0: exact-fp32 kernel (tests)
    ws = torch.empty(lib.load().e4s_conv_wgrad_ws_floats(ctypes.byref(p)), device=x.device, dtype=torch.float32)
    p.ws = fptr(ws)
    call("e4s_conv_wgrad_f32", ctypes.byref(p), stream())
    return dw


# ---- encoder backward (config 5) -------------------------------------------------------------------
def instnorm_bwd(dy, x, stats, gate=None, dx_acc=None):
    """InstanceNorm2d backward (x NHWC [B,H,W,C], stats from instnorm_stats; gate [B,C] = the SE gate that multiplied the
    normalised tensor, or None).  Returns (dx (accumulated into dx_acc if given), sums [B,C,2] = {sum dy, sum dy*xhat})."""
    b, h, w, c = x.shape
    sums = torch.empty(b, c, 2, device=x.device, dtype=torch.float32)
    dx = dx_acc if dx_acc is not None else torch.empty_like(x)
    ws = torch.empty(lib.load().e4s_instnorm_bwd_ws_doubles(b, h * w, c), device=x.device, dtype=torch.float64)
    call("e4s_instnorm_bwd_f32", fptr(_f32(dy)), fptr(x), fptr(stats), fptr(gate), fptr(sums), fptr(dx), ptr(ws), b, h * w, c,
         1 if dx_acc is not None else 0, stream())
    return dx, sums


def prelu(u, slope):
    y = torch.empty_like(u)
    call("e4s_prelu_f32", fptr(u), fptr(slope), fptr(y), u.numel() // u.shape[-1], u.shape[-1], stream())
    return y


def prelu_bwd(dy, u, slope):
    """(du, dslope [C]) of y = PReLU(u), u NHWC [..., C]."""
    c = u.shape[-1]
    npix = u.numel() // c
    du = torch.empty_like(u)
    dslope = torch.empty(c, device=u.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_prelu_bwd_ws_floats(npix, c), device=u.device, dtype=torch.float32)
    call("e4s_prelu_bwd_f32", fptr(_f32(dy)), fptr(u), fptr(slope), fptr(du), fptr(dslope), fptr(ws), npix, c, stream())
    return du, dslope


def pixel_unshuffle2(x):
    """NHWC [B,2H,2W,C] -> [B,H,W,4C]: channel (py*2+px)*C + c of pixel (a, b) = x[2a+py, 2b+px, c]."""
    b, h2, w2, c = x.shape
    if h2 % 2 or w2 % 2 or c % 4:
        raise RuntimeError("pixel_unshuffle2: even H, W and C % 4 == 0")
    out = torch.empty(b, h2 // 2, w2 // 2, 4 * c, device=x.device, dtype=torch.float32)
    call("e4s_pixel_unshuffle2_f32", fptr(_f32(x)), fptr(out), b, h2 // 2, w2 // 2, c, stream())
    return out


def strided_scatter(src, s, out=None):
    """out[b, y*s, x*s] (+)= src[b, y, x]; src NHWC [B,H,W,C].  out=None: a zero-inserted [B,H*s,W*s,C] tensor."""
    b, h, w, c = src.shape
    acc = 1 if out is not None else 0
    if out is None:
        out = torch.empty(b, h * s, w * s, c, device=src.device, dtype=torch.float32)
    call("e4s_strided_scatter_f32", fptr(_f32(src)), fptr(out), b, h, w, c, int(s), acc, stream())
    return out


def strided_place(src, s, oy, ox, out_hw):
    """[B,Ho,Wo,C] with out[b, y*s+oy, x*s+ox] = src[b, y, x] and zeros elsewhere."""
    b, h, w, c = src.shape
    out = torch.empty(b, out_hw[0], out_hw[1], c, device=src.device, dtype=torch.float32)
    call("e4s_strided_place_f32", fptr(_f32(src)), fptr(out), b, h, w, c, int(s), int(oy), int(ox), out_hw[0], out_hw[1],
         stream())
    return out


def torgb_bwd_w(drgb, x):
    """dws[b,c,ci] = sum_p drgb[b,c,p] * x[b,p,ci] (drgb NCHW [B,3,H,W], x NHWC [B,H,W,C]) -> [B,3,C]."""
    b, h, w, c = x.shape
    dws = torch.empty(b, 3, c, device=x.device, dtype=torch.float32)
    L = lib.load()
    scratch = torch.empty(L.e4s_reduce_parts_ws_floats(L.e4s_seg_reduce_nsplit(b, h, w, c), dws.numel()), device=x.device,
                          dtype=torch.float32)
    call("e4s_torgb_bwd_w_f32", fptr(_f32(drgb)), fptr(x), None, 0, 0, 1, fptr(dws), fptr(scratch), b, h, w, c, stream())
    return dws


def region_mean_bwd(dcodes, labels, num_regions, shape, off, dfeat_acc=None):
    """dfeat[b,p,c] (+)= dcodes[b, label(p), off + c] / count[b, label(p)];  shape = (B,H,W,C) of the feature map."""
    b, h, w, c = shape
    dfeat = dfeat_acc if dfeat_acc is not None else torch.empty(b, h, w, c, device=dcodes.device, dtype=torch.float32)
    counts = torch.empty(b * num_regions, device=dcodes.device, dtype=torch.int32)
    dcodes = _f32(dcodes)
    call("e4s_region_mean_bwd_f32", fptr(dcodes), ptr(labels), labels.shape[1], labels.shape[2], ptr(counts), fptr(dfeat),
         b, h, w, c, num_regions, dcodes.shape[2], int(off), 1 if dfeat_acc is not None else 0, stream())
    return dfeat


# ---- GPEN FullGenerator / Discriminator support ------------------------------------------------
def conv1x1_small(x_nchw, w, bias, scale, act=0, alpha=0.2, gain=LRELU_GAIN, out=None):
    """x NCHW [B,Cin<=4,H,W]; w [Cout,Cin] -> NHWC [B,H,W,Cout] = act(x.w*scale + bias)."""
    x = _f32(x_nchw)
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    y = torch.empty(b, h, wd, cout, device=x.device, dtype=torch.float32) if out is None else out
    call("e4s_conv1x1_small_f32", fptr(x), fptr(_f32(w)), fptr(bias), fptr(y), b, h * wd, cin, cout, y.shape[3],
         float(scale), int(act), float(alpha), float(gain), stream())
    return y


def noise_half(feat, noise_w, bias, out, coff, alpha=0.2, gain=LRELU_GAIN):
    """out[..., coff:coff+C] = lrelu(noise_w*feat + bias)*gain; feat NHWC [B,H,W,C], out NHWC [B,H,W,Cy]."""
    b, h, w, c = feat.shape
    call("e4s_noise_half_f32", fptr(feat), fptr(noise_w), fptr(bias), fptr(out), b * h * w, c, out.shape[3], int(coff),
         float(alpha), float(gain), stream())
    return out


def pixelnorm(x):
    x = _f32(x)
    y = torch.empty_like(x)
    call("e4s_pixelnorm_f32", fptr(x), fptr(y), x.shape[0], x.shape[1], stream())
    return y


def add_scale(a, b, scale):
    out = torch.empty_like(a)
    call("e4s_add_scale_f32", fptr(a), fptr(b), fptr(out), float(scale), a.numel(), stream())
    return out


def minibatch_stddev(x, cy, group):
    """x NHWC [B,H,W,C] -> NHWC [B,H,W,cy] = [x | group stddev statistic | zeros]."""
    b, h, w, c = x.shape
    y = torch.empty(b, h, w, cy, device=x.device, dtype=torch.float32)
    call("e4s_minibatch_stddev_f32", fptr(x), fptr(y), b, h * w, c, cy, int(group), stream())
    return y


def upfirdn2d_nhwc(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d on an NHWC activation: the op's native [major, H, W, minor] view with major = B, minor = C."""
    return upfirdn2d_raw(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


# ---- backward (generator) --------------------------------------------------------------------
def pack_taps_bwd(w_fwd, out=None):
    """forward-packed [ncls,9,Cout,Cin] -> backward layout [ncls,9,Cin,Cout] (taps flipped)."""
    ncls, taps, cout, cin = w_fwd.shape
    assert taps == 9
    wt = _into(out, (ncls, 9, cin, cout), w_fwd.device)
    call("e4s_pack_taps_bwd_f32", fptr(w_fwd), fptr(wt), ncls, cout, cin, stream())
    return wt


def conv_bwd(gz, wt, x, s, d, labels, num_regions, ncls, want_ds=True):
    """gz NHWC [B,Hy,Wy,Cy]; wt [ncls,9,Cx,Cy]; x NHWC [B,Hx,Wx,Cx] -> (dx like x, ds [G,Cx] or None)."""
    b, hy, wy, cy = gz.shape
    _, hx, wx, cx = x.shape
    dx = torch.empty_like(x)
    want_ds = want_ds and s is not None
    ds = torch.empty(s.shape[0], cx, device=x.device, dtype=torch.float32) if want_ds else None
    p = ConvBwdParams()
    p.gz, p.wt, p.dx, p.x, p.ds, p.s, p.d = fptr(gz), fptr(wt), fptr(dx), fptr(x), fptr(ds), fptr(s), fptr(d)
    if labels is not None:
        p.labels, p.Hm, p.Wm, p.R = ptr(labels), labels.shape[1], labels.shape[2], num_regions
    else:
        p.labels, p.Hm, p.Wm, p.R = None, 0, 0, 1
    p.B, p.Hx, p.Wx, p.Cx, p.Hy, p.Wy, p.Cy, p.ncls = b, hx, wx, cx, hy, wy, cy, ncls
    # partial sums of ds (and of dx where the tap groups are split over blocks), added in order by the entry point
    nws = lib.load().e4s_conv_bwd_ws_floats(ctypes.byref(p))
    ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
    p.ds_ws = fptr(ws)
    call("e4s_conv_bwd_mfma_f32", ctypes.byref(p), stream())
    return dx, ds


def region_scale(gz, d, labels, num_regions, ncls=1):
    """u = gz * d[region of the output pixel] (e4s_region_scale_f32): gz NHWC [B,Ho,Wo,C], d [B*R,C], labels uint8 [B,Hm,Wm].
    ncls == 1: u like gz.  ncls == 4 (polyphase up-conv, Ho = 2H): u [4,B,H,W,C], one contiguous map per output phase."""
    gz = _f32(gz)
    b, ho, wo, c = gz.shape
    if ncls == 4:
        if ho % 2 or wo % 2:
            raise RuntimeError("region_scale: the polyphase form needs an even output grid")
        h, w = ho // 2, wo // 2
        u = torch.empty(4, b, h, w, c, device=gz.device, dtype=torch.float32)
    else:
        h, w = ho, wo
        u = torch.empty_like(gz)
    call("e4s_region_scale_f32", fptr(gz), fptr(_f32(d)), ptr(labels), labels.shape[1], labels.shape[2], int(num_regions), fptr(u),
         b, h, w, c, int(ncls), stream())
    return u


def col2im_region_ok(c):
    return c % 4 == 0 and 4 <= c <= 1024 and 256 % (c // 4) == 0


SCATTER_DGRAD = os.environ.get("E4S_SCATTER_DGRAD", "1") != "0"


SCATTER_DGRAD_MAX_BYTES = int(os.environ.get("E4S_SCATTER_DGRAD_MAX_BYTES", str(6 << 30)))      # 6 GiB: batch 2 of the 1024^2 generator's largest masked layer is 2.4 GB


def scatter_dgrad_wanted(b, h, w, cy, cx):
    """Policy of the scatter-form input gradient of a masked StyledConv (x [b,h,w,cx] -> cy channels; for an up-conv h, w are the INPUT
    grid and one launch runs per output phase): a 1x1 split-bf16 contraction [b h w, cy] x [cy, 9 cx] on the gather kernel (256-row x
    128-column tiles) + e4s_col2im_region_f32.  `auto`: where that launch fills the chip, `bf16x3`: wherever the kernels apply, `f32`: never
    (the exact-fp32 dx + ds kernel)."""
    if not SCATTER_DGRAD or PRECISION == "f32" or cy % 32 or (9 * cx) % 128 or not col2im_region_ok(cx):
        return False
    if 4 * b * h * w * 9 * cx * 4 > SCATTER_DGRAD_MAX_BYTES:          # the product tensor G [ncls <= 4, b, h, w, 9 cx] fp32 (a captured step's pool keeps it)
        return False                                                    # -> the exact-fp32 dx + ds kernel, which needs no scratch
    if PRECISION == "bf16x3":
        return True
    return (b * h * w + 255) // 256 * ((9 * cx) // 128) >= BF16X3_MIN_BLOCKS


def col2im_region(G, x, s, labels, num_regions, ncls=1):
    """(dx, ds) of a masked StyledConv from the scatter-form products (e4s_col2im_region_f32): G [ncls,B,H,W,9*C] (tap-major columns: the 1x1
    contraction of u with the tap-stacked weights), x NHWC [B,H,W,C] the layer's input, s [B*R,C] -> dx like x, ds like s."""
    b, h, w, c = x.shape
    if G.numel() != ncls * b * h * w * 9 * c or not G.is_contiguous() or not x.is_contiguous():
        raise RuntimeError("col2im_region: G must be contiguous [ncls,B,H,W,9*C] for x [B,H,W,C]")
    r = int(num_regions)
    dx = torch.empty_like(x)
    ds = torch.empty(b * r, c, device=x.device, dtype=torch.float32)
    L = lib.load()
    nsplit = L.e4s_col2im_region_nsplit(b, h, w, c)
    if nsplit <= 0:
        raise RuntimeError("col2im_region: unsupported channel count")
    ws = torch.empty(L.e4s_reduce_parts_ws_floats(nsplit, ds.numel()), device=x.device, dtype=torch.float32)
    call("e4s_col2im_region_f32", fptr(G), fptr(x), fptr(_f32(s)), ptr(labels), labels.shape[1], labels.shape[2], r, fptr(dx), fptr(ds),
         fptr(ws), b, h, w, c, int(ncls), stream())
    return dx, ds


def scale_dot(u, x, s):
    """u, x NHWC [B,H,W,C]; s [B,C]: u <- u * s[b] in place (returned) and ds[b,c] = sum_p x * u (the unscaled u), ordered sums."""
    b, h, w, c = u.shape
    assert u.is_contiguous() and x.is_contiguous() and x.shape == u.shape and s.shape == (b, c)
    ds = torch.empty(b, c, device=u.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_scale_dot_ws_floats(b, h * w, c), device=u.device, dtype=torch.float32)
    call("e4s_scale_dot_f32", fptr(u), fptr(x), fptr(s), fptr(ds), fptr(ws), b, h * w, c, stream())
    return u, ds


def act_bwd_demod(dy, y, noise, noise_w, bias, alpha, gain, labels, num_regions):
    """(gz, dd): gz = dy * lrelu'(y) * gain (fused_bias_act(dy, None, y, 3, 1, alpha, gain)) and dd = demod_grad(gz, y, ...) in ONE pass
    over dy and y (e4s_act_bwd_demod_f32); None when the channel count is outside the kernel's set."""
    b, h, w, c = dy.shape
    if c % 4 or c > 1024 or c < 4 or 256 % (c // 4) or not dy.is_contiguous() or not y.is_contiguous():
        return None
    r = num_regions if labels is not None else 1
    gz = torch.empty_like(dy)
    dd = torch.empty(b * r, c, device=dy.device, dtype=torch.float32)
    hm = wm = 0
    if labels is not None:
        hm, wm = labels.shape[1:]
    nb = 0
    if noise is not None:
        nb = h * w if noise.shape[0] > 1 else 0
    L = lib.load()
    ws = torch.empty(L.e4s_reduce_parts_ws_floats(L.e4s_act_bwd_demod_nsplit(b, h, w, c), dd.numel()), device=dy.device, dtype=torch.float32)
    call("e4s_act_bwd_demod_f32", fptr(_f32(dy)), fptr(y), fptr(gz), fptr(noise), fptr(noise_w) if noise is not None else None, nb,
         fptr(bias), float(alpha), float(gain), ptr(labels), hm, wm, r, fptr(dd), fptr(ws), b, h, w, c, stream())
    return gz, dd


def demod_grad(gz, y, noise, noise_w, bias, alpha, gain, labels, num_regions):
    """dL/dd [G, C] for out_pre = d * c (see e4s_demod_grad_f32); the caller divides by d."""
    b, h, w, c = gz.shape
    r = num_regions if labels is not None else 1
    dd = torch.empty(b * r, c, device=gz.device, dtype=torch.float32)
    hm = wm = 0
    if labels is not None:
        hm, wm = labels.shape[1:]
    nb = 0
    if noise is not None:
        nb = h * w if noise.shape[0] > 1 else 0
    L = lib.load()
    ws = torch.empty(L.e4s_reduce_parts_ws_floats(L.e4s_seg_reduce_nsplit(b, h, w, c), dd.numel()), device=gz.device,
                     dtype=torch.float32)
    call("e4s_demod_grad_f32", fptr(gz), fptr(y), fptr(noise), fptr(noise_w) if noise is not None else None, nb,
         fptr(bias), float(alpha), float(gain), ptr(labels), hm, wm, r, fptr(dd), fptr(ws), b, h, w, c, stream())
    return dd


def torgb_bwd(drgb, x, ws, labels, num_regions, dx_acc=None):
    """drgb NCHW [B,3,H,W]; x NHWC; ws [G,3,C] -> (dx NHWC (accumulated into dx_acc if given), dws [G,3,C])."""
    b, h, w, c = x.shape
    r = num_regions if labels is not None else 1
    hm = wm = 0
    if labels is not None:
        hm, wm = labels.shape[1:]
    dws = torch.empty(b * r, 3, c, device=x.device, dtype=torch.float32)
    drgb = _f32(drgb)
    L = lib.load()
    scratch = torch.empty(L.e4s_reduce_parts_ws_floats(L.e4s_seg_reduce_nsplit(b, h, w, c), dws.numel()), device=x.device,
                          dtype=torch.float32)
    call("e4s_torgb_bwd_w_f32", fptr(drgb), fptr(x), ptr(labels), hm, wm, r, fptr(dws), fptr(scratch), b, h, w, c, stream())
    acc = 1 if dx_acc is not None else 0
    dx = dx_acc if dx_acc is not None else torch.empty_like(x)
    call("e4s_torgb_bwd_x_f32", fptr(drgb), fptr(ws), ptr(labels), hm, wm, r, fptr(dx), b, h, w, c, acc, stream())
    return dx, dws


def shift_scale(x, tab, labels, num_regions, anchors, istride=1, dy=0, dx=0, os=1, py=0, px=0):
    """out[b,a,c] = tab[group(out pixel of a)][c] * x[b, a*istride + (dy,dx), c] (0 outside); x NHWC."""
    b, hi, wi, c = x.shape
    ha, wa = anchors
    out = torch.empty(b, ha, wa, c, device=x.device, dtype=torch.float32)
    hm = wm = 0
    if labels is not None:
        hm, wm = labels.shape[1:]
    call("e4s_shift_scale_f32", fptr(x), fptr(tab), ptr(labels), hm, wm, num_regions if labels is not None else 1,
         fptr(out), b, ha, wa, hi, wi, c, istride, dy, dx, os, py, px, stream())
    return out


# ---- loss networks (csrc/criteria.hip) -------------------------------------------------------------------------
def adaptive_pool(x, out_hw, crop=None, in_nchw=True, scale=None, shift=None):
    """F.adaptive_avg_pool2d of x[:, :, y0:y0+hc, x0:x0+wc] (+ per-channel affine) -> NHWC [B,Ho,Wo,C]."""
    x = _f32(x)
    if in_nchw:
        b, c, hi, wi = x.shape
    else:
        b, hi, wi, c = x.shape
    y0, x0, hc, wc = crop if crop is not None else (0, 0, hi, wi)
    ho, wo = out_hw
    y = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.float32)
    call("e4s_adaptive_pool_f32", fptr(x), fptr(y), b, c, hi, wi, y0, x0, hc, wc, ho, wo, 1 if in_nchw else 0,
         fptr(scale), fptr(shift), stream())
    return y


def adaptive_pool_bwd(dy, in_shape, crop=None, in_nchw=True, scale=None, dx_acc=None):
    """gradient of adaptive_pool w.r.t. its input (shape in_shape, the input's layout); accumulated into dx_acc if given."""
    if in_nchw:
        b, c, hi, wi = in_shape
    else:
        b, hi, wi, c = in_shape
    y0, x0, hc, wc = crop if crop is not None else (0, 0, hi, wi)
    ho, wo = dy.shape[1:3]
    dx = dx_acc if dx_acc is not None else torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    call("e4s_adaptive_pool_bwd_f32", fptr(_f32(dy)), fptr(dx), b, c, hi, wi, y0, x0, hc, wc, ho, wo, 1 if in_nchw else 0,
         fptr(scale), 1 if dx_acc is not None else 0, stream())
    return dx


def pack_smallcin(w):
    """[Cout,Cin,k,k] -> [k*k*Cin, Cout] (tap-major; e4s_conv_smallcin_f32's operand) -- a one-off layout change."""
    cout, cin, k, _ = w.shape
    return w.detach().float().permute(2, 3, 1, 0).reshape(k * k * cin, cout).contiguous()


def conv_smallcin(x, wp, bias, cout, k, stride, pad, relu=False):
    b, hi, wi, cin = x.shape
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    y = torch.empty(b, ho, wo, cout, device=x.device, dtype=torch.float32)
    call("e4s_conv_smallcin_f32", fptr(x), fptr(wp), fptr(bias), fptr(y), b, hi, wi, cin, ho, wo, cout, k, stride, pad,
         1 if relu else 0, stream())
    return y


def conv_smallcin_bwd(dy, wp, in_shape, k, stride, pad, cache=None):
    """Image gradient of conv_smallcin.  Large kernels (AlexNet's 11x11 / 4 stem: k*k*Cin >= 128 columns): GEMM form -- ONE 1x1
    contraction z = dy . wp^T on the fp32 MFMA conv kernel + a col2im pass (e4s_conv_smallcin_col2im_f32); small kernels: the per-pixel
    kernel (e4s_conv_smallcin_bwd_f32).  cache: a dict owned by whoever owns `wp` (the module's weight pack), where the zero-padded GEMM
    image of wp is kept -- per pack, not per process: two loss networks alternating no longer evict each other, and a pack that dies takes
    its image with it."""
    b, hi, wi, cin = in_shape
    _, ho, wo, cout = dy.shape
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    ncol = k * k * cin
    if ncol >= 128 and cout % 32 == 0 and dy.is_contiguous():
        zc = (ncol + 31) // 32 * 32
        key = (wp.data_ptr(), wp._version, zc)
        hit = cache.get("wz") if cache is not None else None
        if hit is None or hit[0] != key:
            wz = torch.zeros(1, 1, zc, cout, device=wp.device, dtype=torch.float32)
            wz[0, 0, :ncol] = wp                                                  # wp is [k*k*Cin][Cout]: the GEMM's [N][K] matrix
            hit = (key, wz)
            if cache is not None:
                cache["wz"] = hit
        wz = hit[1]
        z = conv_mfma(_f32(dy), wz, zc, ntaps=1, spatial=False)
        call("e4s_conv_smallcin_col2im_f32", fptr(z), fptr(dx), b, hi, wi, cin, ho, wo, zc, k, stride, pad, stream())
        return dx
    call("e4s_conv_smallcin_bwd_f32", fptr(_f32(dy)), fptr(wp), fptr(dx), b, hi, wi, cin, ho, wo, cout, k, stride, pad,
         stream())
    return dx


def maxpool3s2(x):
    b, hi, wi, c = x.shape
    ho, wo = (hi - 3) // 2 + 1, (wi - 3) // 2 + 1
    y = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.float32)
    idx = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.uint8)
    call("e4s_maxpool3s2_f32", fptr(x), fptr(y), ptr(idx), b, hi, wi, c, stream())
    return y, idx


def maxpool3s2_bwd(dy, idx, in_shape):
    b, hi, wi, c = in_shape
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    call("e4s_maxpool3s2_bwd_f32", fptr(_f32(dy)), ptr(idx), fptr(dx), b, hi, wi, c, stream())
    return dx


def maxpool2(x):
    b, hi, wi, c = x.shape
    y = torch.empty(b, hi // 2, wi // 2, c, device=x.device, dtype=torch.float32)
    idx = torch.empty(b, hi // 2, wi // 2, c, device=x.device, dtype=torch.uint8)
    call("e4s_maxpool2_f32", fptr(x), fptr(y), ptr(idx), b, hi, wi, c, stream())
    return y, idx


def maxpool2_bwd(dy, idx, in_shape):
    b, hi, wi, c = in_shape
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    call("e4s_maxpool2_bwd_f32", fptr(_f32(dy)), ptr(idx), fptr(dx), b, hi, wi, c, stream())
    return dx


def relu_bwd(dy, y, dx_acc=None):
    """dx (+)= dy * [y > 0], from the ReLU's output."""
    dx = dx_acc if dx_acc is not None else torch.empty_like(y)
    call("e4s_relu_bwd_f32", fptr(_f32(dy)), fptr(y), fptr(dx), y.numel(), 1 if dx_acc is not None else 0, stream())
    return dx


def lpips_layer(fx, fy, w):
    """[B] = spatial mean of the lin-weighted squared distance of the unit-normalised features (NHWC)."""
    b, h, wd, c = fx.shape
    out = torch.empty(b, device=fx.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_lpips_layer_ws_doubles(b, h * wd), device=fx.device, dtype=torch.float64)
    call("e4s_lpips_layer_f32", fptr(fx), fptr(fy), fptr(w), fptr(out), ptr(ws), b, h * wd, c, stream())
    return out


def lpips_layer_bwd(fx, fy, w, gout, gmul, dfx_acc=None):
    """dfx (+)= gout[0] * gmul * d(sum_b lpips_layer[b]) / d(fx)."""
    b, h, wd, c = fx.shape
    dfx = dfx_acc if dfx_acc is not None else torch.empty_like(fx)
    call("e4s_lpips_layer_bwd_f32", fptr(fx), fptr(fy), fptr(w), fptr(gout), float(gmul), fptr(dfx), b, h * wd, c,
         1 if dfx_acc is not None else 0, stream())
    return dfx


def instnorm_bwd_sums(dy, x, stats):
    """[B,C,2] = {sum_p dy, sum_p dy * (x - mean) * rstd} (ordered)."""
    b, h, w, c = x.shape
    sums = torch.empty(b, c, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_instnorm_bwd_ws_doubles(b, h * w, c), device=x.device, dtype=torch.float64)
    call("e4s_instnorm_bwd_sums_f32", fptr(_f32(dy)), fptr(x), fptr(stats), fptr(sums), ptr(ws), b, h * w, c, stream())
    return sums


def norm_bwd_frozen(dy, stats, gate=None, extra=None, dx_acc=None):
    """dx (+)= rstd * (gate * dy + extra): backward of a normalisation whose statistics are constants."""
    b, h, w, c = dy.shape
    dx = dx_acc if dx_acc is not None else torch.empty_like(dy)
    call("e4s_norm_bwd_frozen_f32", fptr(_f32(dy)), fptr(stats), fptr(gate), fptr(extra), fptr(dx), b, h * w, c,
         1 if dx_acc is not None else 0, stream())
    return dx


def cosine(a, b):
    """rows of a, b [B, D] -> [B,3] = {cos, alpha, beta}, d(cos)/da = alpha*b + beta*a."""
    n, d = a.shape[0], a.numel() // a.shape[0]
    out = torch.empty(n, 3, device=a.device, dtype=torch.float32)
    ws = torch.empty(lib.load().e4s_cosine_ws_doubles(n, d), device=a.device, dtype=torch.float64)
    call("e4s_cosine_f32", fptr(a), fptr(b), fptr(out), ptr(ws), n, d, stream())
    return out


def cosine_bwd(a, b, coef, gout, gmul, da_acc=None):
    n, d = a.shape[0], a.numel() // a.shape[0]
    da = da_acc if da_acc is not None else torch.empty_like(a)
    call("e4s_cosine_bwd_f32", fptr(a), fptr(b), fptr(coef), fptr(gout), float(gmul), fptr(da), n, d,
         1 if da_acc is not None else 0, stream())
    return da
