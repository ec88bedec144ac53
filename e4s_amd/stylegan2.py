"""StyleGAN2 generator with mask-guided regional style injection -- MI355X-native.

Host-side mirror of the reference module tree (src/models/stylegan2/model.py:34-667): the same
class names, constructor arguments, parameter/buffer names and shapes, so reference checkpoints
(`G.conv1.conv.weight`, `G.convs.N.conv.modulation.weight`, `G.to_rgbs.N.upsample.kernel`, ...)
load with strict=True.  What differs is the execution: ``Generator.forward`` does not call a
chain of per-region ATen ops; it drives the fused HIP kernels of libe4s_hip.so
(e4s_amd/csrc/*.hip) over NHWC activations:

    reference (model.py)                               here
    ---------------------------------------------      -----------------------------------------
    modulation + materialised [B,Cout,Cin,3,3]          e4s_rowdot_f32 x2 on cached sum_k W^2
      weights + demod (276-281)
    12 x grouped conv * one-hot mask (386-400)          ONE implicit-GEMM launch with region-select inside
                                                        (per-pixel style on the A fragment, d[region] in the
                                                        epilogue); split-bf16 or exact-fp32 MFMA (E4S_PRECISION)
    conv_transpose2d + Blur/upfirdn2d (287-300)         4-phase 3x3 polyphase weights on the same GEMM, or the
                                                        exact tile-fused up-conv kernel
    NoiseInjection + FusedLeakyReLU (329-335, 404)      GEMM epilogue
    ToRGB conv + bias + Upsample(skip) (422-448)        e4s_torgb_f32

There is no CPU path: tensors must live on a ROCm device and the library must be built.
"""
import math
import os
import random

import torch
from torch import nn
from torch.nn import functional as F

from . import kernels as K
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix


# Up-sampling StyledConvs: exact tile-fused transposed conv + blur (e4s_upconv_mfma_f32, 9*Cin*Cout MACs per input
# pixel) or the polyphase form on the generic conv kernel (36).  Same function, different rounding order.
UPCONV_EXACT = os.environ.get("E4S_UPCONV", "exact") != "polyphase"
UPCONV_EXACT_MIN_RES = int(os.environ.get("E4S_UPCONV_MIN_RES", "256"))     # masked layers below this stay polyphase
# unmasked up-convs under E4S_PRECISION=auto/bf16x3: exact sub-pixel GEMM (csrc/upconv_bf16x3.hip) unless "polyphase" is asked for
UPCONV_BF16X3_EXACT = os.environ.get("E4S_UPCONV_BF16X3", "exact") != "polyphase"
# the Cin == 32 StyledConv (32 -> 32 at 1024^2) on the resident-weights kernel with the ToRGB contraction in its epilogue
CONV_C32 = os.environ.get("E4S_CONV_C32", "1") != "0"
# masked layers on the variant-rows kernel (csrc/conv_region.hip); "0": the region-select kernel of conv_bf16x3.hip everywhere
REGION_ROWS = os.environ.get("E4S_REGION_ROWS", "1") != "0"


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


class Upsample(nn.Module):
    """model.py:34-53"""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """model.py:56-75"""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    """model.py:78-94"""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """model.py:97-132 (Discriminator only)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return conv2d_gradfix.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride,
                                     padding=self.padding)


class EqualLinear(nn.Module):
    """model.py:135-169"""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if self.activation:
            out = F.linear(input, self.weight * self.scale)
            return fused_leaky_relu(out, self.bias * self.lr_mul)
        return F.linear(input, self.weight * self.scale, bias=self.bias * self.lr_mul)


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


def polyphase_upconv_weights(w, blur_kernel):
    """Fold conv_transpose2d(stride=2, padding=0) followed by the 4x4 blur with pad (1,1)
    (model.py:287-300, 206-213) into four phase-specific 3x3 kernels over the INPUT grid.

    With I[q] = sum_{2u+k=q} x[u] W[k] (transposed conv) and out[p] = sum_j I[p+j-1] kflip[j] (upfirdn2d is a
    true convolution, upfirdn2d_kernel.cu:77), out[p] = sum_u x[u] E[p-2u] with
        E[t] = sum_j kflip[j] W[t+j-1],  t in [-2, 3].
    Output pixel p = 2a + py only sees u in {a-1, a, a+1}: tap e (u = a+e-1) uses E[py - 2(e-1)].
    w [Cout,Cin,3,3], blur_kernel [4,4] -> [4 (py*2+px), 9 (ey*3+ex), Cout, Cin]."""
    cout, cin = w.shape[:2]
    kf = torch.flip(blur_kernel, [0, 1])
    e = F.conv2d(F.pad(w.reshape(cout * cin, 1, 3, 3), (3, 3, 3, 3)), kf[None, None]).reshape(cout, cin, 6, 6)
    phases = []
    for py in range(2):
        for px in range(2):
            iy = [py + 4 - 2 * ey for ey in range(3)]          # E index = t + 2
            ix = [px + 4 - 2 * ex for ex in range(3)]
            sub = e[:, :, iy][:, :, :, ix]
            phases.append(sub.permute(2, 3, 0, 1).reshape(9, cout, cin))
    return torch.stack(phases, 0).contiguous()


from .packs import param_key as _param_key  # noqa: E402


class ModulatedConv2d(nn.Module):
    """model.py:184-320.  Parameters as in the reference; `packed()` lays the weight out for the
    implicit-GEMM kernel ([ncls][tap][Cout][Cin]) and caches sum_k W^2 for demodulation."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1], fused=True):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel, self.out_channel = in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if downsample:
            raise NotImplementedError("downsampling ModulatedConv2d is never instantiated by E4S")
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self.fused = fused
        self._pack = None

    def _buf(self, name, shape, device):
        """ONE buffer per weight pack for the module's lifetime: re-packs after a weight update write in place.  Nothing is allocated
        when the weights change -- not eagerly, and not inside a captured train step that re-packs in the graph (train.graphed_g_step with
        train_G: the capture then holds no pack allocations at all) -- and whatever holds a pack's address (the style prologue's job
        tables, Generator._style_plan) stays valid."""
        bufs = self.__dict__.setdefault("_e4s_bufs", {})
        t = bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device:
            t = torch.empty(shape, device=device, dtype=torch.float32)
            bufs[name] = t
        return t

    def packed(self):
        """Returns dict(w=[ncls,taps,Cout,Cin], wsq=[Cout,Cin] | None, w_rgb=[Cout,Cin] for 1x1)."""
        key = _param_key(self.weight) + ((_param_key(self.blur.kernel)) if self.upsample else ())
        if self._pack is not None and self._pack["key"] == key:
            return self._pack
        with torch.no_grad():
            w = self.weight.detach()[0].float()                                  # [Cout,Cin,k,k]
            cout, cin, k, _ = w.shape
            dev = w.device
            pack = {"key": key}
            if k == 1:
                pack["w"] = w.reshape(1, 1, cout, cin).contiguous()
            elif not self.upsample:
                pack["w"] = K.pack_taps(w.contiguous(), out=self._buf("w", (1, 9, cout, cin), dev))
            else:
                # same math as polyphase_upconv_weights() below (the CPU-tested statement of it)
                pack["w"] = K.polyphase_weights(w.contiguous(), self.blur.kernel.detach().float(), out=self._buf("w", (4, 9, cout, cin), dev))
                pack["w3"] = K.pack_taps(w.contiguous(), out=self._buf("w3", (1, 9, cout, cin), dev))   # plain taps: exact tile-fused up-conv
            pack["wsq"] = K.weight_sqsum(w.contiguous(), out=self._buf("wsq", (cout, cin), dev)) if self.demodulate else None
        self._pack = pack
        return pack

    def _derived(self, name, build):
        """A lazily built image of the current pack (split-bf16 operands, backward layouts), in its lifetime buffer."""
        pk = self.packed()
        if name not in pk:
            with torch.no_grad():
                pk[name] = build(pk)
        return pk[name]

    def subpixel_split_weights(self):
        """Split-bf16 image of the sub-pixel GEMM operand of e4s_upconv_bf16x3_f32 (exact up-conv; cached with the pack)."""
        cout, cin = self.out_channel, self.in_channel
        return self._derived("w_sub_split", lambda pk: K.subpixel_weights(
            self.weight.detach()[0].float().contiguous(), out=self._buf("w_sub_split", (cin // 32, cout // 32, 9, 32, 32), self.weight.device)))

    def split_weights(self):
        """Split-bf16 image of packed()["w"] for e4s_conv_bf16x3_f32 (built on first use, cached with the pack)."""
        return self._derived("w_split", lambda pk: K.split_bf16x2(pk["w"], out=self._buf("w_split", tuple(pk["w"].shape), pk["w"].device)))

    def split_weights16(self):
        """16-channel-chunk split image of packed()["w"] for e4s_conv_region_bf16x3_f32 (masked layers; cached with the pack)."""
        return self._derived("w_split16", lambda pk: K.split16_bf16x2(pk["w"], out=self._buf("w_split16", K.split16_shape(pk["w"].shape), pk["w"].device)))

    def scatter_taps(self):
        """packed()["w"] [ncls,9,Cout,Cin] as the operand of the scatter-form input gradient (autograd.styled_conv_backward, masked
        layers): per class a 1x1 contraction [Cout] -> [9*Cin] with the taps stacked in the COLUMNS, wg[cls][0][t*Cin + ci][co] =
        w[cls][t][co][ci]; returns (wg, its split-bf16 image), both in lifetime buffers."""
        def build(pk):
            ncls, _, cout, cin = pk["w"].shape
            wg = self._buf("wg", (ncls, 1, 9 * cin, cout), pk["w"].device)
            wg.view(ncls, 9, cin, cout).copy_(pk["w"].permute(0, 1, 3, 2))
            return wg, K.split_bf16x2(wg, out=self._buf("wg_split", (ncls, 1, 9 * cin, cout), pk["w"].device))
        return self._derived("wg", build)

    def bwd_taps(self):
        """packed()["w"] in the backward layout [ncls,9,Cin,Cout], taps flipped (e4s_conv_bwd_mfma_f32's operand)."""
        def build(pk):
            ncls, _, cout, cin = pk["w"].shape
            return K.pack_taps_bwd(pk["w"], out=self._buf("wt", (ncls, 9, cin, cout), pk["w"].device))
        return self._derived("wt", build)

    def forward(self, input, style):
        """Drop-in single-style forward (NCHW in/out), model.py:242-320."""
        b = input.shape[0]
        x = K.nchw_to_nhwc(input)
        pk = self.packed()
        s = K.modulate_vec(style, self.modulation.weight, self.modulation.bias)
        if self.kernel_size == 1:
            if self.out_channel != 3 or self.demodulate:
                raise NotImplementedError("1x1 ModulatedConv2d exists only as the ToRGB conv (model.py:417)")
            ws = K.rgb_weights(pk["w"].view(3, -1), s, self.scale)
            zero = torch.zeros(3, device=x.device)
            return K.torgb(x, ws, zero, None, None, None, 1)
        else:
            d = K.demod_coefs(s, pk["wsq"], self.scale) if self.demodulate else \
                torch.full((b, self.out_channel), self.scale, device=x.device)
            if self.upsample:
                y = K.conv_mfma(x, pk["w"], self.out_channel, ncls=4, ostride=2, in_scale=s, out_scale=d)
            else:
                y = K.conv_mfma(x, pk["w"], self.out_channel, in_scale=s, out_scale=d)
        return K.nhwc_to_nchw(y)


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


def _prep_noise(noise, b, h, w, device):
    """Returns (tensor, per_channel).  None -> fresh N(0,1) like NoiseInjection (model.py:329-333)."""
    if noise is None:
        return torch.randn(b, 1, h, w, device=device), False
    noise = noise.to(device=device, dtype=torch.float32)
    if noise.ndim != 4 or noise.shape[2] != h or noise.shape[3] != w or noise.shape[0] not in (1, b):
        raise RuntimeError(f"noise of shape {tuple(noise.shape)} is not broadcastable to [{b},C,{h},{w}]")
    if noise.shape[1] == 1:
        return noise.contiguous(), False
    return K.nchw_to_nhwc(noise), True           # [1,C,H,W] noise of scripts/face_edit.py:49-52


class StyledConv(nn.Module):
    """model.py:351-406"""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, mask_op=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)
        self.mask_op = mask_op

    def c32_eligible(self, b, h, w, labels=None, plan=None):
        """The Cin == 32 resident-weights kernel (e4s_conv_c32_bf16x3_f32) applies: unmasked 32 -> 32k, stride 1, split-bf16 on."""
        conv = self.conv
        return (CONV_C32 and not conv.upsample and labels is None and plan is None and conv.in_channel == 32
                and conv.out_channel % 32 == 0 and K.want_bf16x3(b, h, w, 32, conv.out_channel))

    def run_nhwc(self, x, s, noise, labels=None, num_regions=1, plan=None, rec=None, rgb_ws=None, d=None):
        """x NHWC, s [G,Cin] modulation (G = B*R when masked).  Masked layers pass the label map
        (region-select inside the GEMM) or, alternatively, a gathered RowPlan.  Returns NHWC output
        after noise + bias + leaky-ReLU*sqrt(2).  rgb_ws [B,3,32] (only where c32_eligible and Cout == 32): also return the
        ToRGB partial of the output, (y, partial [B,3,H,W])."""
        conv = self.conv
        pk = conv.packed()
        b, h, w, _ = x.shape
        ho, wo = (2 * h, 2 * w) if conv.upsample else (h, w)
        if d is None:                            # (the generator's fused forward hands in all layers' coefficients, computed up front)
            d = K.demod_coefs(s, pk["wsq"], conv.scale)
        nz, per_ch = _prep_noise(noise, b, ho, wo, x.device)
        if rec is not None:
            if per_ch:
                raise NotImplementedError("backward with per-channel noise maps")
            rec.update(d=d, noise=nz)
        ncls = 4 if conv.upsample else 1
        if not per_ch and self.c32_eligible(b, h, w, labels, plan):
            return K.conv_c32(x, conv.split_weights(), conv.out_channel, in_scale=s, out_scale=d, noise=nz,
                              noise_w=self.noise.weight, bias=self.activate.bias, act=1, alpha=self.activate.negative_slope,
                              gain=self.activate.scale, rgb_ws=rgb_ws)
        if rgb_ws is not None:
            raise RuntimeError("the fused ToRGB partial exists only on the Cin == 32 kernel")
        if (conv.upsample and labels is None and plan is None and not per_ch and UPCONV_BF16X3_EXACT
                and K.upconv_bf16x3_eligible(conv.in_channel, conv.out_channel)
                and K.want_bf16x3(b, h, w, conv.in_channel, conv.out_channel, ncls, masked=False)):
            # unmasked up-conv: the exact transposed conv as a sub-pixel GEMM on the split-bf16 path (9 Cin Cout MACs per
            # input pixel instead of the polyphase form's 36) + one FIR / noise / bias / activation pass
            return K.upconv_bf16x3(x, conv.subpixel_split_weights(), conv.out_channel, conv.blur.kernel, in_scale=s,
                                   out_scale=d, noise=nz, noise_w=self.noise.weight, bias=self.activate.bias, act=1,
                                   alpha=self.activate.negative_slope, gain=self.activate.scale)
        if plan is None and not per_ch and K.want_bf16x3(b, h, w, conv.in_channel, conv.out_channel, ncls,
                                                         masked=labels is not None):
            # split-bf16 matrix-core path (polyphase form for up-convs: 4x the MACs of the exact kernel at > 3x its rate)
            return K.conv_mfma(x, pk["w"], conv.out_channel, labels=labels, num_regions=num_regions, ncls=ncls,
                               ostride=2 if conv.upsample else 1, in_scale=s, out_scale=d, noise=nz,
                               noise_w=self.noise.weight, bias=self.activate.bias, act=1,
                               alpha=self.activate.negative_slope, gain=self.activate.scale,
                               w_split=conv.split_weights(),
                               w_split16=conv.split_weights16() if (labels is not None and REGION_ROWS) else None)
        # a masked tile runs one pass per region present: exact only where 12x28 output tiles are mostly uniform
        if conv.upsample and plan is None and UPCONV_EXACT and (labels is None or ho >= UPCONV_EXACT_MIN_RES):
            return K.upconv_mfma(x, pk["w3"], conv.out_channel, conv.blur.kernel, in_scale=s, out_scale=d,
                                 labels=labels, num_regions=num_regions, noise=nz, noise_w=self.noise.weight,
                                 noise_per_channel=per_ch, bias=self.activate.bias, act=1,
                                 alpha=self.activate.negative_slope, gain=self.activate.scale)
        return K.conv_mfma(x, pk["w"], conv.out_channel, plan=plan, labels=None if plan is not None else labels,
                           num_regions=num_regions, ncls=4 if conv.upsample else 1,
                           ostride=2 if conv.upsample else 1, in_scale=s, out_scale=d, noise=nz,
                           noise_w=self.noise.weight, noise_per_channel=per_ch, bias=self.activate.bias, act=1,
                           alpha=self.activate.negative_slope, gain=self.activate.scale)

    def run_nhwc_soft(self, x, s, noise, mask):
        """Soft (non one-hot) masks: the reference's own formulation, sum_r conv(x, style_r) * nearest(mask)_r
        (model.py:386-400), as R natural-order launches + a masked accumulate.  s: [B*R, Cin]."""
        conv = self.conv
        pk = conv.packed()
        b, h, w, _ = x.shape
        r = mask.shape[1]
        ho, wo = (2 * h, 2 * w) if conv.upsample else (h, w)
        d = K.demod_coefs(s, pk["wsq"], conv.scale)
        s3, d3 = s.view(b, r, -1), d.view(b, r, -1)
        acc = None
        for i in range(r):
            y = K.conv_mfma(x, pk["w"], conv.out_channel, ncls=4 if conv.upsample else 1,
                            ostride=2 if conv.upsample else 1, in_scale=s3[:, i].contiguous(),
                            out_scale=d3[:, i].contiguous())
            acc = K.mask_mul_add(y, mask, i, acc, True)
        nz, per_ch = _prep_noise(noise, b, ho, wo, x.device)
        if per_ch:
            raise NotImplementedError("per-channel noise with soft masks")
        return K.noise_bias_act_nhwc(acc, nz, self.noise.weight, self.activate.bias, self.activate.negative_slope,
                                     self.activate.scale)

    def forward(self, input, style, mask, noise=None, use_plan=False):
        """Drop-in NCHW forward.  style [B,R,512] + one-hot mask when mask_op else [B,512].
        use_plan=True selects the region-gathered row plan instead of in-GEMM region-select."""
        x = K.nchw_to_nhwc(input)
        b, h, w, _ = x.shape
        mod = self.conv.modulation
        if self.mask_op:
            r = style.shape[1]
            s = K.modulate_vec(style.reshape(b * r, -1), mod.weight, mod.bias)
            labels, _ = K.mask_labels(mask)
            plan = K.region_plan(labels, r, h, w, 4 if self.conv.upsample else 1) if use_plan else None
            return K.nhwc_to_nchw(self.run_nhwc(x, s, noise, labels, r, plan))
        s = K.modulate_vec(style, mod.weight, mod.bias)
        return K.nhwc_to_nchw(self.run_nhwc(x, s, noise))


class ToRGB(nn.Module):
    """model.py:409-448"""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1], mask_op=False):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))
        self.mask_op = mask_op

    def run_nhwc(self, x, s, labels, num_regions, skip, rec=None):
        pk = self.conv.packed()
        ws = K.rgb_weights(pk["w"].view(3, -1), s, self.conv.scale)
        if rec is not None:
            rec.update(ws=ws)
        k4 = self.upsample.kernel if skip is not None else None
        return K.torgb(x, ws, self.bias, skip, k4, labels, num_regions)

    def run_nhwc_soft(self, x, s, mask, skip):
        """Soft masks: sum_r torgb(x, style_r) * nearest(mask)_r + bias + upsample(skip) (model.py:426-448)."""
        pk = self.conv.packed()
        b = x.shape[0]
        r = mask.shape[1]
        ws = K.rgb_weights(pk["w"].view(3, -1), s, self.conv.scale).view(b, r, 3, -1)
        zero3 = torch.zeros(3, device=x.device)
        acc = None
        for i in range(r):
            y = K.torgb(x, ws[:, i].contiguous(), zero3, None, None, None, 1)
            acc = K.mask_mul_add(y, mask, i, acc, False)
        zw = torch.zeros(b, 3, x.shape[3], device=x.device)
        k4 = self.upsample.kernel if skip is not None else None
        tail = K.torgb(x, zw, self.bias, skip, k4, None, 1)          # bias + FIR-upsampled skip
        return acc + tail

    def forward(self, input, style, mask, skip=None):
        x = K.nchw_to_nhwc(input)
        b = x.shape[0]
        mod = self.conv.modulation
        if self.mask_op:
            r = style.shape[1]
            s = K.modulate_vec(style.reshape(b * r, -1), mod.weight, mod.bias)
            labels, _ = K.mask_labels(mask)
            return self.run_nhwc(x, s, labels, r, skip)
        s = K.modulate_vec(style, mod.weight, mod.bias)
        return self.run_nhwc(x, s, None, 1, skip)


class Generator(nn.Module):
    """model.py:451-667"""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 split_layer_idx=7, remaining_layer_idx=18):
        super().__init__()
        self.split_layer_idx = split_layer_idx
        self.remaining_layer_idx = remaining_layer_idx
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                         512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel,
                                mask_op=True)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False, mask_op=True)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        K_ = self.remaining_layer_idx
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            conv_masked = not (i > 2 + K_ // 2)                                   # model.py:537,545
            rgb_masked = not (K_ != 17 and i >= 2 + K_ // 2)                      # model.py:553
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True,
                                         blur_kernel=blur_kernel, mask_op=conv_masked))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel,
                                         mask_op=conv_masked))
            self.to_rgbs.append(ToRGB(out_channel, style_dim, mask_op=rgb_masked))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        # one-hot masks (labelMap2OneHot) take the region-select fast path; with strict_mask the mask is verified on
        # every eager call (one host sync) and soft masks fall back to the reference's R-pass formulation.
        self.strict_mask = True

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.get_latent(latent_in).mean(0, keepdim=True)

    def _style_is_canonical(self):
        """PixelNorm followed by EqualLinear(activation='fused_lrelu') layers only (model.py:489-497) -- checked once per module."""
        ok = getattr(self, "_e4s_style_ok", None)
        if ok is None:
            mods = list(self.style)
            ok = (len(mods) >= 2 and isinstance(mods[0], PixelNorm)
                  and all(isinstance(m, EqualLinear) and m.activation == "fused_lrelu" and m.bias is not None for m in mods[1:]))
            self._e4s_style_ok = ok
        return ok

    def get_latent(self, input):
        """The mapping network z -> w (model.py:489-497, 570-574: PixelNorm + n_mlp x EqualLinear(lr_mul, fused_lrelu)) on the native
        kernels (e4s_pixelnorm_f32 + one e4s_grouped_linear_f32 per layer, leaky-ReLU gain folded into scale and bias) for ROCm inputs
        that need no gradient (Net3 freezes G.style, networks.py:68-70); otherwise the module chain as written (differentiable ATen)."""
        if input.is_cuda and input.ndim == 2 and not (torch.is_grad_enabled() and (
                input.requires_grad or any(p.requires_grad for p in self.style.parameters()))):
            # the native chain restates exactly this structure; anything else (a swapped-in layer) takes the module chain below.
            # NOTE the result carries no autograd history here: callers that want gradients of G.style enable requires_grad BEFORE the
            # call (then the ATen chain runs; both paths agree to 1e-6, tests/test_gpu_parity.py::test_mapping_network_native_vs_module_chain)
            if not self._style_is_canonical():
                return self.style(input)
            with torch.no_grad():
                w = K.pixelnorm(input.to(torch.float32).contiguous())
                for lin in list(self.style)[1:]:
                    w = equal_linear_lrelu(lin, w)
            return w
        return self.style(input)

    # ------------------------------------------------------------------------------------------
    def _assemble_latent(self, styles, input_is_latent, inject_index, truncation, truncation_latent):
        if not input_is_latent:
            styles = [self.get_latent(s) for s in styles]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            if styles[0].ndim < 4:
                return styles[0].unsqueeze(1).repeat(1, self.n_latent, 1)
            return styles[0]
        if inject_index is None:
            inject_index = random.randint(1, self.n_latent - 1)
        l1 = styles[0].unsqueeze(1).repeat(1, inject_index, 1)
        l2 = styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)
        return torch.cat([l1, l2], 1)

    def forward(self, styles, structure_feats, mask, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True,
                use_structure_code=False):
        """model.py:576-667.  `styles` = [latent [B,R,n_latent,512]] with input_is_latent=True is the E4S call
        (networks.py:110-112,175-177).  Returns (image NCHW, latent|None, feats at 16x16 NCHW)."""
        if use_structure_code:
            raise NotImplementedError("use_structure_code=True is never used by E4S (networks.py:112,177)")
        latent = self._assemble_latent(styles, input_is_latent, inject_index, truncation, truncation_latent)
        if latent.ndim != 4:
            raise RuntimeError("E4S generator expects a regional latent [B, R, n_latent, 512]")
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else \
                [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if torch.is_grad_enabled() and any(n is not None and n.requires_grad for n in noise):
            raise NotImplementedError("gradients w.r.t. the injected noise maps are not implemented")
        params = [p for p in self.parameters() if p.requires_grad] if torch.is_grad_enabled() else []
        if any(p is q for q in self.style.parameters() for p in params) and not input_is_latent:
            raise NotImplementedError("gradients of the mapping network G.style (always frozen by Net3, networks.py:68-70)")
        if torch.is_grad_enabled() and (latent.requires_grad or params):
            from .autograd import GeneratorFn
            image, feats = GeneratorFn.apply(self, latent, mask, noise, *params)
        else:
            image, feats = self._fused_forward(latent, mask, noise)
        return (image, latent, feats) if return_latents else (image, None, feats)

    def _style_plan(self, b, r, nlat, dev):
        """Job tables of the batched style prologue (e4s_rowdot_multi_f32): every layer's modulation s = EqualLinear(style)
        and demodulation d in two launches instead of ~43.  Cached per (shapes, addresses): the tables hold raw pointers of the
        modulation parameters and of the per-layer sum_k W^2 buffers (ModulatedConv2d.packed keeps those in place across weight updates)."""
        layers = [(self.conv1, 0, "conv"), (self.to_rgb1, 1, "rgb")]
        i = 1
        for c1, c2, tr in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            layers += [(c1, i, "conv"), (c2, i + 1, "conv"), (tr, i + 2, "rgb")]
            i += 2
        # packed() first, for every layer: it refreshes sum_k W^2 IN PLACE when a weight changed, so the tables below depend on
        # ADDRESSES only (modulation parameters, the per-layer sum_k W^2 buffers), not on weight versions -- a trainable generator
        # (train_G) re-uses one table for the whole run, eagerly and inside a captured train step (no host-to-device copy per update)
        packs = [l.conv.packed() for l, _, _ in layers]
        key = (b, r, nlat, str(dev)) + tuple(
            (l.conv.modulation.weight.data_ptr(), l.conv.modulation.bias.data_ptr(), pk["wsq"].data_ptr() if pk["wsq"] is not None else 0)
            for (l, _, _), pk in zip(layers, packs))
        plan = getattr(self, "_e4s_style_plan", None)
        if plan is not None and plan["key"] == key:
            return plan
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("Generator._style_plan: the job tables of the style prologue must exist before a stream capture "
                               "(run the same shapes once eagerly first -- GraphedStep / GraphedFaceSwap warm up for that)")
        sjobs, djobs, meta, keep = [], [], {}, []
        s_off = d_off = 0
        for layer, idx, kind in layers:
            mod = layer.conv.modulation
            cin = mod.weight.shape[0]
            g = b * r if layer.mask_op else b
            stride = nlat * 512 if layer.mask_op else r * nlat * 512
            w, bs = mod.weight.detach(), mod.bias.detach()
            keep += [w, bs]
            sjobs.append(dict(in_off=idx * 512, in_stride=stride, out_off=s_off, M=w, bias=bs, G=g, O=cin, K=512,
                              scale=1.0 / math.sqrt(512)))
            m = {"s": (s_off, g, cin)}
            if kind == "conv":
                pk = layer.conv.packed()
                cout = layer.conv.out_channel
                keep.append(pk["wsq"])
                djobs.append(dict(in_off=s_off, in_stride=cin, out_off=d_off, M=pk["wsq"], bias=None, G=g, O=cout, K=cin,
                                  scale=layer.conv.scale))
                m["d"] = (d_off, g, cout)
                d_off += g * cout
            s_off += g * cin
            meta[id(layer)] = m
        plan = {"key": key, "s": K.rowdot_jobs(sjobs, dev), "d": K.rowdot_jobs(djobs, dev), "meta": meta, "s_floats": s_off,
                "d_floats": d_off, "keep": keep}
        self._e4s_style_plan = plan
        return plan

    @torch.no_grad()
    def _fused_forward(self, latent, mask, noise, tape=None):
        """`tape` (a list) records per layer what the backward needs (e4s_amd/autograd.py)."""
        lat = latent.detach().to(torch.float32).contiguous()
        b, r = lat.shape[:2]
        labels, flags = K.mask_labels(mask)
        soft = False
        if self.strict_mask and not torch.cuda.is_current_stream_capturing():
            soft = bool(flags.item())          # one host sync; skipped while a HIP graph is being captured
        if soft and tape is not None:
            raise NotImplementedError("backward with soft (non one-hot) masks")

        # style prologue of ALL layers up front (two launches): s = modulation, d = demodulation coefficients
        sp = None
        if not soft and lat.shape[3] == 512:
            sp = self._style_plan(b, r, lat.shape[2], lat.device)
            sbuf = torch.empty(sp["s_floats"], device=lat.device, dtype=torch.float32)
            dbuf = torch.empty(sp["d_floats"], device=lat.device, dtype=torch.float32)
            K.rowdot_multi(*sp["s"], lat, sbuf, 0)
            K.rowdot_multi(*sp["d"], sbuf, dbuf, 1)

        def pre(layer, which):
            off, g, c = sp["meta"][id(layer)][which]
            return (sbuf if which == "s" else dbuf)[off:off + g * c].view(g, c)

        def styled(layer, x, idx, nz, rgb_ws=None):
            mod = layer.conv.modulation
            s = pre(layer, "s") if sp is not None else K.modulate(lat, idx, layer.mask_op, mod.weight, mod.bias)
            if soft and layer.mask_op:
                return layer.run_nhwc_soft(x, s, nz, mask)
            rec = {} if tape is not None else None
            y = layer.run_nhwc(x, s, nz, labels if layer.mask_op else None, r, rec=rec, rgb_ws=rgb_ws,
                               d=pre(layer, "d") if sp is not None else None)
            partial = None
            if rgb_ws is not None:
                y, partial = y
            if tape is not None:
                rec.update(kind="conv", layer=layer, idx=idx, masked=layer.mask_op, x=x, y=y, s=s,
                           labels=labels if layer.mask_op else None)
                tape.append(rec)
            return y if rgb_ws is None else (y, partial)

        def styled_then_rgb(conv2, to_rgb, x, idx, nz, skip):
            """conv2 (Cin == Cout == 32, unmasked) with the ToRGB 1x1 modulated conv of ITS OUTPUT in the epilogue
            (model.py:422-440), then bias + FIR-upsampled skip (model.py:441-446): the 1024^2 activation is not read again."""
            mod = to_rgb.conv.modulation
            s_rgb = pre(to_rgb, "s") if sp is not None else K.modulate(lat, idx + 1, False, mod.weight, mod.bias)
            ws = K.rgb_weights(to_rgb.conv.packed()["w"].view(3, -1), s_rgb, to_rgb.conv.scale)
            y, partial = styled(conv2, x, idx, nz, rgb_ws=ws)
            out = K.torgb_finish(partial, to_rgb.bias, skip, to_rgb.upsample.kernel if skip is not None else None)
            if tape is not None:
                tape.append(dict(kind="rgb", layer=to_rgb, idx=idx + 1, masked=False, x=y, s=s_rgb, ws=ws,
                                 has_skip=skip is not None, out=out, labels=None))
            return y, out

        def rgb(layer, x, idx, skip):
            mod = layer.conv.modulation
            s = pre(layer, "s") if sp is not None else K.modulate(lat, idx, layer.mask_op, mod.weight, mod.bias)
            if soft and layer.mask_op:
                return layer.run_nhwc_soft(x, s, mask, skip)
            rec = {} if tape is not None else None
            out = layer.run_nhwc(x, s, labels if layer.mask_op else None, r, skip, rec=rec)
            if tape is not None:
                rec.update(kind="rgb", layer=layer, idx=idx, masked=layer.mask_op, x=x, s=s, has_skip=skip is not None, out=out,
                           labels=labels if layer.mask_op else None)
                tape.append(rec)
            return out

        x = K.const_input(self.input.input, b)
        x = styled(self.conv1, x, 0, noise[0])
        skip = rgb(self.to_rgb1, x, 1, None)
        feats = None
        i = 1
        for conv1, conv2, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            x = styled(conv1, x, i, noise[i])
            if i + 2 == self.split_layer_idx:
                feats = K.nhwc_to_nchw(x)                                          # model.py:642-647
                if tape is not None:
                    tape[-1]["is_feats"] = True
            nz2 = noise[i + 1]
            if (not soft and not conv2.mask_op and not to_rgb.mask_op and conv2.conv.out_channel == 32
                    and (nz2 is None or nz2.shape[1] == 1) and conv2.c32_eligible(b, x.shape[1], x.shape[2])):
                x, skip = styled_then_rgb(conv2, to_rgb, x, i + 1, nz2, skip)
            else:
                x = styled(conv2, x, i + 1, nz2)
                skip = rgb(to_rgb, x, i + 2, skip)
            i += 2
        return skip, feats


# ---- ConvLayer on the HIP kernels (Discriminator, GPEN encoder) ----------------------------------------------------------
def _packed_equal_conv(conv, cin_pad):
    """EqualConv2d weight * scale (model.py:117-123) tap-packed [1,k*k,Cout,cin_pad] (input channels zero-padded to the
    kernels' 32-channel K step), + its split-bf16 image; cached on the module."""
    key = _param_key(conv.weight) + (cin_pad,)
    pk = getattr(conv, "_e4s_pack", None)
    if pk is None or pk["key"] != key:
        with torch.no_grad():
            w = conv.weight.detach().float() * conv.scale
            cout, cin, k, _ = w.shape
            if cin_pad != cin:
                w = torch.cat([w, w.new_zeros(cout, cin_pad - cin, k, k)], 1)
            pk = {"key": key, "w": K.pack_taps(w.contiguous())}
        conv._e4s_pack = pk
    return pk


def conv_layer_nhwc(layer, x, x_is_nchw=False):
    """ConvLayer (model.py:670-716; GPEN's copy gpen_model.py:558-606) on NHWC activations: [Blur] -> EqualConv2d ->
    [FusedLeakyReLU | ScaledLeakyReLU] as [e4s_upfirdn2d_f32] + ONE conv launch with bias and activation in its epilogue.
    x NHWC [B,H,W,Cin] (or the NCHW image for the 3 -> C stem, x_is_nchw); returns NHWC."""
    mods = list(layer)
    i = 0
    blur = None
    if isinstance(mods[0], Blur):
        blur, i = mods[0], 1
    conv = mods[i]
    tail = mods[i + 1] if len(mods) > i + 1 else None
    cout, cin, k, _ = conv.weight.shape
    if isinstance(tail, FusedLeakyReLU):
        bias, act, alpha, gain = tail.bias, 1, tail.negative_slope, tail.scale
    elif isinstance(tail, ScaledLeakyReLU):
        bias, act, alpha, gain = None, 1, tail.negative_slope, math.sqrt(2)
    else:
        bias, act, alpha, gain = conv.bias, 0, 0.2, 1.0
    if x_is_nchw:
        if k != 1 or conv.stride != 1 or cin > 4:
            raise NotImplementedError("NCHW input is only accepted by the 1x1 stem ConvLayer")
        return K.conv1x1_small(x, conv.weight.detach().reshape(cout, cin), bias, conv.scale, act, alpha, gain)
    cx = x.shape[3]
    pk = _packed_equal_conv(conv, cx)                   # cx > cin: the caller padded the activation (513 -> 544)
    if cx < cin or cx % 32:
        raise RuntimeError(f"ConvLayer on {cx} channels: need a multiple of 32 >= {cin}")
    kw = dict(bias=bias, act=act, alpha=alpha, gain=gain)
    b, h, w, _ = x.shape

    def split():
        if "w_split" not in pk:
            pk["w_split"] = K.split_bf16x2(pk["w"])
        return pk["w_split"]
    if conv.stride == 1:
        if k != 3 or conv.padding != 1:
            raise NotImplementedError("stride-1 ConvLayers are 3x3 / padding 1 (model.py:701-703)")
        if K.want_bf16x3(b, h, w, cx, cout):
            return K.conv_mfma(x, pk["w"], cout, w_split=split(), **kw)
        return K.conv_mfma(x, pk["w"], cout, **kw)
    if blur is None or conv.stride != 2 or conv.padding != 0 or k not in (1, 3):
        raise NotImplementedError("down-sampling ConvLayers are Blur + stride-2 padding-0 convs (model.py:683-700)")
    xb = K.upfirdn2d_nhwc(x, blur.kernel, pad=blur.pad)
    hb, wb = xb.shape[1:3]
    anchors = ((hb - k) // 2 + 1, (wb - k) // 2 + 1)
    gk = dict(istride=2, ntaps=k * k, anchors=anchors, tap_shift=1 if k == 3 else 0, **kw)
    tiles = (b * anchors[0] * anchors[1] + 255) // 256 * (cout // 128 if cout % 128 == 0 else 0)
    if K.PRECISION != "f32" and cout % 128 == 0 and (K.PRECISION == "bf16x3" or tiles >= K.BF16X3_MIN_BLOCKS):
        return K.conv_mfma(xb, pk["w"], cout, w_split=split(), **gk)
    return K.conv_mfma(xb, pk["w"], cout, **gk)


def equal_linear_lrelu(lin, x):
    """EqualLinear(activation='fused_lrelu') (model.py:159-164): lrelu(x W^T scale + bias lr_mul, 0.2) * sqrt(2) on
    e4s_grouped_linear_f32; the positive gain is folded into scale and bias.  Plain EqualLinear when no activation."""
    w = lin.weight.detach()
    g = math.sqrt(2) if lin.activation else 1.0
    bias = (lin.bias.detach() * (lin.lr_mul * g)).unsqueeze(0) if lin.bias is not None else None
    y = K.grouped_linear(x.unsqueeze(1).contiguous(), w.unsqueeze(0), bias, None, lin.scale * g,
                         act=1 if lin.activation else 0, alpha=0.2)
    return y.squeeze(1)


# ---- Discriminator (config 5): native forward on the MFMA conv kernels; under autograd the same kernels as closed families
# of autograd Functions (disc_autograd.py) that can be differentiated twice (the R1 penalty).  CPU tensors are refused (no CPU
# path); the module tree (ConvLayer / ResBlock as nn.Modules) exists for state_dict / API parity with the reference ----
class ConvLayer(nn.Sequential):
    """model.py:670-716"""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)


class ResBlock(nn.Module):
    """model.py:719-737"""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        out = self.conv2(self.conv1(input))
        return (out + self.skip(input)) / math.sqrt(2)


class Discriminator(nn.Module):
    """model.py:740-799"""

    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                    256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        convs = [ConvLayer(3, channels[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"),
                                          EqualLinear(channels[4], 1))

    def _needs_autograd(self, input):
        return torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))

    @torch.no_grad()
    def forward_native(self, input):
        """model.py:740-799 as a schedule of HIP kernels on NHWC tensors (no autograd): ConvLayers on the MFMA convs,
        ResBlock combine, minibatch stddev (the 513-channel map is zero-padded to 544 for the 32-channel K step)."""
        convs = list(self.convs)
        x = conv_layer_nhwc(convs[0], input, x_is_nchw=True)
        for rb in convs[1:]:
            r = conv_layer_nhwc(rb.conv2, conv_layer_nhwc(rb.conv1, x))
            x = K.add_scale(r, conv_layer_nhwc(rb.skip, x), 1.0 / math.sqrt(2))
        b, h, w, c = x.shape
        group = min(b, self.stddev_group)
        if b % group:
            raise RuntimeError("minibatch stddev needs the batch to be a multiple of the group (as the reference's view)")
        x = K.minibatch_stddev(x, (c + 1 + 31) // 32 * 32, group)
        x = conv_layer_nhwc(self.final_conv, x)
        flat = K.nhwc_to_nchw(x).reshape(b, -1)
        return equal_linear_lrelu(self.final_linear[1], equal_linear_lrelu(self.final_linear[0], flat))

    def forward(self, input):
        if input.is_cuda:
            if not self._needs_autograd(input):
                return self.forward_native(input)
            from .disc_autograd import discriminator_forward      # native graph, differentiable twice (R1, adv_loss.py:48-60)
            return discriminator_forward(self, input)
        raise RuntimeError("Discriminator.forward needs a ROCm tensor: there is no CPU path (the module tree exists for "
                           "state_dict / API parity; its layers run on the HIP ops)")
