"""Discriminator under autograd -- first AND second order -- on the native kernels (config 5:
src/training/coach.py:340-401 with src/criteria/adv_loss.py:8-60, whose R1 penalty differentiates dD/d(image) again).

The reference gets its double backward from `conv2d_gradfix` (ATen) and the `*Backward` Functions of its two ops.  Here
every bilinear map is a closed family of three autograd Functions whose backward is written in terms of the other two, so
autograd can differentiate any number of times and each node is one native launch:

    conv      y  = C(x, W)        e4s_conv_mfma_f32 / e4s_conv_bf16x3_f32   backward: (Dgrad(gy, W), Wgrad(gy, x))
    dgrad     dx = D(gy, W)       the conv kernels on flipped, transposed taps (stride 2, padding 0: on gy placed at the
                                  odd positions of a zero grid, e4s_strided_place_f32)  backward: (C(ggx, W), Wgrad(gy, ggx))
    wgrad     dW = G(gy, x)       e4s_conv_wgrad_f32                         backward: (C(x, ggW), D(gy, ggW))

likewise for the 3-channel stem (e4s_conv1x1_small_f32 / e4s_torgb_f32 / e4s_torgb_bwd_w_f32), the final linears
(e4s_grouped_linear_f32 / _t / e4s_grouped_outer_f32), the FIR blur (e4s_upfirdn2d_f32, self-adjoint family) and the fused
bias + leaky ReLU (e4s_fused_bias_act_f32 with its `ref` form, as op/fused_act.py:18-47).  Activations are NHWC throughout.
The minibatch standard deviation acts on a [B,4,4,512] map and is left to torch (differentiable as is)."""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import kernels as K
from .lib import call, fptr, stream
from .packs import param_key


# ---- conv family ------------------------------------------------------------------------------------------------
class _PackCache:
    """Tap-packed / transposed / split-bf16 images of ONE ConvLayer weight, valid for one (storage, version) of the parameter
    (packs.param_key).  While D is frozen (the generator step, 14 of 15 iterations) nothing is re-packed; while it trains, the
    three or four uses inside one step (D(fake), D(real), their backward) share one pack."""
    __slots__ = ("key", "packs")

    def __init__(self):
        self.key, self.packs = None, {}

    def get(self, key, name, make):
        if self.key != key:
            self.key, self.packs = key, {}
        if name not in self.packs:
            self.packs[name] = make()
        return self.packs[name]


def invalidate_packs(disc):
    """Forget D's cached weight packs (a captured graph that must re-pack from the weights of the moment: train.graphed_g_step)."""
    for m in disc.modules():
        pc = getattr(m, "_e4s_dpacks", None)
        if pc is not None:
            pc.key, pc.packs = None, {}


def _pack(w, cache=None):
    make = lambda: K.pack_taps(w.detach().float().contiguous())
    return cache[0].get(cache[1], "fwd", make) if cache is not None else make()


def _pack_t(w, cache=None):
    if w.shape[2] == 3:          # one launch from the forward pack (was flip + transpose copy + pack)
        make = lambda: K.pack_taps_bwd(_pack(w, cache))
    else:
        make = lambda: K.pack_taps(w.detach().float().flip(2, 3).transpose(0, 1).contiguous())
    return cache[0].get(cache[1], "bwd", make) if cache is not None else make()


def _conv_s1(x, wp, cout, cache=None, name=None):
    b, h, w, cx = x.shape
    if K.want_bf16x3(b, h, w, cx, cout):
        make = lambda: K.split_bf16x2(wp)
        ws = cache[0].get(cache[1], name + "_split", make) if cache is not None else make()
        return K.conv_mfma(x, wp, cout, w_split=ws)
    return K.conv_mfma(x, wp, cout)


def _out_hw(kind, hw):
    k = 3 if kind == "s2k3" else 1
    return ((hw[0] - k) // 2 + 1, (hw[1] - k) // 2 + 1)


def _conv_forward(x, w, kind, cache=None):
    """x NHWC, w [Cout,Cin,k,k] (scale already applied).  kind: s1 = 3x3 stride 1 padding 1; s2k3 / s2k1 = 3x3 / 1x1 stride 2
    padding 0 (on the blurred map)."""
    x = x.contiguous()
    cout = w.shape[0]
    if kind == "s1":
        return _conv_s1(x, _pack(w, cache), cout, cache, "fwd")
    anchors = _out_hw(kind, x.shape[1:3])
    return K.conv_mfma(x, _pack(w, cache), cout, istride=2, ntaps=9 if kind == "s2k3" else 1, anchors=anchors,
                       tap_shift=1 if kind == "s2k3" else 0)


def _conv_dgrad(gy, w, kind, x_shape, cache=None):
    gy = gy.contiguous()
    cin = w.shape[1]
    if kind == "s1":
        return _conv_s1(gy, _pack_t(w, cache), cin, cache, "bwd")
    if kind == "s2k3":
        # y[o] = sum_k x[2o + k] w[k]  =>  dx = 'same' 3x3 conv of the grid holding gy[o] at (2o+1, 2o+1) with the flipped taps
        return _conv_s1(K.strided_place(gy, 2, 1, 1, x_shape[1:3]), _pack_t(w, cache), cin, cache, "bwd")
    t = K.conv_mfma(gy, _pack_t(w, cache), cin, ntaps=1, spatial=False)
    return K.strided_place(t, 2, 0, 0, x_shape[1:3])


def _conv_wgrad(gy, x, kind, w_shape):
    gy, x = gy.contiguous(), x.contiguous()
    cout, cin, k, _ = w_shape
    if kind == "s1":
        dw = K.conv_wgrad(gy, x, ntaps=9, istride=1)
    else:
        dw = K.conv_wgrad(gy, x, ntaps=k * k, istride=2, anchors=tuple(gy.shape[1:3]), tap_shift=1 if k == 3 else 0)
    return dw.permute(1, 2, 0).reshape(cout, cin, k, k)


class Conv(Function):
    """cache: (_PackCache, key) when `w` IS a ConvLayer's (scaled, padded) weight -- the packs are then shared between calls;
    None for the second-order nodes, whose `w` operand is a gradient."""

    @staticmethod
    def forward(ctx, x, w, kind, cache=None):
        ctx.kind, ctx.cache = kind, cache
        ctx.save_for_backward(x, w)
        return _conv_forward(x, w, kind, cache)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = ConvDgrad.apply(gy, w, ctx.kind, tuple(x.shape), ctx.cache) if ctx.needs_input_grad[0] else None
        gw = ConvWgrad.apply(gy, x, ctx.kind, tuple(w.shape)) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


class ConvDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w, kind, x_shape, cache=None):
        ctx.kind, ctx.x_shape, ctx.cache = kind, x_shape, cache
        ctx.save_for_backward(gy, w)
        return _conv_dgrad(gy, w, kind, x_shape, cache)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        d_gy = Conv.apply(ggx, w, ctx.kind, ctx.cache) if ctx.needs_input_grad[0] else None
        d_w = ConvWgrad.apply(gy, ggx, ctx.kind, tuple(w.shape)) if ctx.needs_input_grad[1] else None
        return d_gy, d_w, None, None, None


class ConvWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, kind, w_shape):
        ctx.kind, ctx.w_shape = kind, w_shape
        ctx.save_for_backward(gy, x)
        return _conv_wgrad(gy, x, kind, w_shape)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        d_gy = Conv.apply(x, ggw, ctx.kind) if ctx.needs_input_grad[0] else None
        d_x = ConvDgrad.apply(gy, ggw, ctx.kind, tuple(x.shape)) if ctx.needs_input_grad[1] else None
        return d_gy, d_x, None, None


# ---- 3-channel stem: ConvLayer(3, C, 1) (model.py:752) on the NCHW image ---------------------------------------------
def _stem_forward(img, w):
    return K.conv1x1_small(img.contiguous(), w.detach().contiguous(), None, 1.0, act=0)


def _stem_dgrad(gy, w):
    gy = gy.contiguous()
    b = gy.shape[0]
    ws = w.detach().t().contiguous()[None].expand(b, -1, -1).contiguous()          # [B,3,C]
    zero = torch.zeros(3, device=gy.device, dtype=torch.float32)
    return K.torgb(gy, ws, zero, None, None, None, 1)                               # NCHW [B,3,H,W]


def _stem_wgrad(gy, img):
    return K.batch_sum(K.torgb_bwd_w(img.contiguous(), gy.contiguous())).t()      # [C,3]


class Stem(Function):
    @staticmethod
    def forward(ctx, img, w):
        ctx.save_for_backward(img, w)
        return _stem_forward(img, w)

    @staticmethod
    def backward(ctx, gy):
        img, w = ctx.saved_tensors
        return (StemDgrad.apply(gy, w) if ctx.needs_input_grad[0] else None,
                StemWgrad.apply(gy, img) if ctx.needs_input_grad[1] else None)


class StemDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w):
        ctx.save_for_backward(gy, w)
        return _stem_dgrad(gy, w)

    @staticmethod
    def backward(ctx, ggimg):
        gy, w = ctx.saved_tensors
        return (Stem.apply(ggimg, w) if ctx.needs_input_grad[0] else None,
                StemWgrad.apply(gy, ggimg) if ctx.needs_input_grad[1] else None)


class StemWgrad(Function):
    @staticmethod
    def forward(ctx, gy, img):
        ctx.save_for_backward(gy, img)
        return _stem_wgrad(gy, img)

    @staticmethod
    def backward(ctx, ggw):
        gy, img = ctx.saved_tensors
        return (Stem.apply(img, ggw) if ctx.needs_input_grad[0] else None,
                StemDgrad.apply(gy, ggw) if ctx.needs_input_grad[1] else None)


# ---- linear family (EqualLinear, model.py:135-169) ----------------------------------------------------------------
class Lin(Function):
    """y [B,O] = x [B,K] w[O,K]^T"""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return K.grouped_linear(x.contiguous()[:, None], w.detach().contiguous()[None], None, None, 1.0)[:, 0]

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return (LinT.apply(g, w) if ctx.needs_input_grad[0] else None,
                Outer.apply(g, x) if ctx.needs_input_grad[1] else None)


class LinT(Function):
    """dx [B,K] = g [B,O] w[O,K]"""

    @staticmethod
    def forward(ctx, g, w):
        ctx.save_for_backward(g, w)
        return K.grouped_linear_t(g.contiguous()[:, None], w.detach().contiguous()[None], 1.0)[:, 0]

    @staticmethod
    def backward(ctx, gg):
        g, w = ctx.saved_tensors
        return (Lin.apply(gg, w) if ctx.needs_input_grad[0] else None,
                Outer.apply(g, gg) if ctx.needs_input_grad[1] else None)


class Outer(Function):
    """dw [O,K] = g [B,O]^T x [B,K]"""

    @staticmethod
    def forward(ctx, g, x):
        ctx.save_for_backward(g, x)
        return K.grouped_outer(g.contiguous()[:, None], x.contiguous()[:, None], 1.0)[0]

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        return (Lin.apply(x, ggw) if ctx.needs_input_grad[0] else None,
                LinT.apply(g, ggw) if ctx.needs_input_grad[1] else None)


# ---- FIR blur on NHWC (Blur, model.py:65-94) ------------------------------------------------------------------------
class BlurNHWC(Function):
    @staticmethod
    def forward(ctx, x, kernel, pad):
        ctx.pad = pad
        ctx.save_for_backward(kernel)
        return K.upfirdn2d_nhwc(x.contiguous(), kernel, pad=pad)

    @staticmethod
    def backward(ctx, g):
        (kernel,) = ctx.saved_tensors
        kh = kernel.shape[0]
        return BlurNHWC.apply(g, kernel.flip(0, 1).contiguous(), (kh - 1 - ctx.pad[0], kh - 1 - ctx.pad[1])), None, None


# ---- bias + leaky ReLU * gain on a channels-last tensor (FusedLeakyReLU / ScaledLeakyReLU, model.py:17-31, op/fused_act.py) --
def _bias_act_raw(x, bias, ref, grad, alpha, gain):
    x = x.contiguous()
    y = torch.empty_like(x)
    c = x.shape[-1]
    call("e4s_fused_bias_act_f32", fptr(x), fptr(bias), fptr(ref), fptr(y), x.numel(), 1, c if bias is not None else 1, 3,
         grad, float(alpha), float(gain), stream())
    return y


class ColSum(Function):
    """[C] = sum over every dim but the last"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return K.colsum(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape)


class BiasAct(Function):
    @staticmethod
    def forward(ctx, x, bias, alpha, gain):
        y = _bias_act_raw(x, bias.detach().contiguous() if bias is not None else None, None, 0, alpha, gain)
        ctx.alpha, ctx.gain, ctx.has_bias = alpha, gain, bias is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = BiasActBwd.apply(gy, y, ctx.alpha, ctx.gain)
        gb = ColSum.apply(gx) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gx, gb, None, None


class BiasActBwd(Function):
    """gx = gy * (y > 0 ? 1 : alpha) * gain; piecewise linear in gy, so its own backward is the same map."""

    @staticmethod
    def forward(ctx, gy, y, alpha, gain):
        ctx.alpha, ctx.gain = alpha, gain
        ctx.save_for_backward(y)
        return _bias_act_raw(gy, None, y, 1, alpha, gain)

    @staticmethod
    def backward(ctx, ggx):
        (y,) = ctx.saved_tensors
        return BiasActBwd.apply(ggx, y, ctx.alpha, ctx.gain), None, None, None


# ---- module-level schedule -------------------------------------------------------------------------------------------
def _conv_layer(layer, x):
    """ConvLayer (model.py:670-716) on an NHWC activation, differentiable twice."""
    from .stylegan2 import Blur, FusedLeakyReLU, ScaledLeakyReLU
    mods = list(layer)
    i = 0
    if isinstance(mods[0], Blur):
        x = BlurNHWC.apply(x, mods[0].kernel, tuple(mods[0].pad))
        i = 1
    conv = mods[i]
    tail = mods[i + 1] if len(mods) > i + 1 else None
    cout, cin, k, _ = conv.weight.shape
    w = conv.weight * conv.scale
    cx = x.shape[3]
    if cx > cin:                                      # the 513-channel map is zero-padded to the K step (544)
        w = F.pad(w, (0, 0, 0, 0, 0, cx - cin))
    kind = "s1" if conv.stride == 1 else ("s2k3" if k == 3 else "s2k1")
    if (kind == "s1" and (k != 3 or conv.padding != 1)) or (kind != "s1" and conv.padding != 0):
        raise NotImplementedError("ConvLayer geometry outside model.py:683-703")
    pc = getattr(conv, "_e4s_dpacks", None)
    if pc is None:
        pc = conv._e4s_dpacks = _PackCache()
    y = Conv.apply(x, w, kind, (pc, param_key(conv.weight) + (cx,)))
    if isinstance(tail, FusedLeakyReLU):
        return BiasAct.apply(y, tail.bias, tail.negative_slope, tail.scale)
    if isinstance(tail, ScaledLeakyReLU):
        return BiasAct.apply(y, None, tail.negative_slope, math.sqrt(2))
    if conv.bias is not None:
        y = y + conv.bias
    return y


def _equal_linear(lin, x):
    y = Lin.apply(x, lin.weight * lin.scale)
    if lin.activation:
        return BiasAct.apply(y, lin.bias * lin.lr_mul, 0.2, math.sqrt(2))
    return y + lin.bias * lin.lr_mul if lin.bias is not None else y


def discriminator_forward(disc, img):
    """Discriminator.forward (model.py:775-799) with a graph autograd can differentiate twice."""
    from .stylegan2 import FusedLeakyReLU
    convs = list(disc.convs)
    stem = list(convs[0])
    w0 = (stem[0].weight * stem[0].scale).reshape(stem[0].weight.shape[0], -1)
    x = Stem.apply(img, w0)
    if not isinstance(stem[1], FusedLeakyReLU):
        raise NotImplementedError("the stem is ConvLayer(3, C, 1) with a FusedLeakyReLU (model.py:752)")
    x = BiasAct.apply(x, stem[1].bias, stem[1].negative_slope, stem[1].scale)
    for rb in convs[1:]:
        r = _conv_layer(rb.conv2, _conv_layer(rb.conv1, x))
        x = (r + _conv_layer(rb.skip, x)) * (1.0 / math.sqrt(2))
    b, h, w, c = x.shape
    group = min(b, disc.stddev_group)
    if b % group:
        raise RuntimeError("minibatch stddev needs the batch to be a multiple of the group (as the reference's view)")
    sd = torch.sqrt(x.view(group, -1, h, w, c).var(0, unbiased=False) + 1e-8).mean((1, 2, 3), keepdim=True)   # [M,1,1,1]
    sd = sd.repeat(group, h, w, 1)
    pad = (c + 1 + 31) // 32 * 32 - (c + 1)
    x = torch.cat([x, sd, x.new_zeros(b, h, w, pad)], 3)
    x = _conv_layer(disc.final_conv, x)
    flat = x.permute(0, 3, 1, 2).reshape(b, -1)
    return _equal_linear(disc.final_linear[1], _equal_linear(disc.final_linear[0], flat))
