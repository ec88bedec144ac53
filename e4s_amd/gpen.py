"""GPEN FullGenerator (blind face restoration; the other consumer of modulated conv / upfirdn2d / fused_bias_act in the
face-swap wall clock, scripts/face_swap.py:206-210) -- MI355X-native.  SURVEY.md 8(f) N2.

Module tree / state_dict identical to the reference (src/pretrained/gpen/face_model/gpen_model.py:380-690):
`generator.{style.N, input, conv1, to_rgb1, convs.N, to_rgbs.N}`, `ecd{j}.0.{0,1,2}`, `final_linear.0`, so GPEN-BFR-512
checkpoints load with strict=True.  Execution is a schedule of the same HIP kernels as the E4S generator, on NHWC tensors:

    reference                                           here
    --------------------------------------------------  ---------------------------------------------------------------
    ecd0: EqualConv2d 1x1 (3 -> C) + FusedLeakyReLU      e4s_conv1x1_small_f32 (reads NCHW, writes NHWC)
    ecd j: Blur + EqualConv2d 3x3 s2 p0 + FusedLeakyReLU  e4s_upfirdn2d_f32 (NHWC = its native [major,H,W,minor] view)
      (gpen_model.py:558-606)                             + stride-2 gather conv (tap_shift = 1), bias + act in the epilogue
    final_linear / style MLP (8 x EqualLinear+lrelu)      e4s_grouped_linear_f32 (gain folded into scale and bias)
    StyledConv, isconcat (:318-357): modulated conv,      ONE conv launch writing the first C channels of the 2C-channel
      cat(out, w_noise * encoder_feature), lrelu over 2C    output (y_cstride) + e4s_noise_half_f32 for the other C
    ToRGB (:359-377)                                      e4s_torgb_f32
There is no CPU path: tensors must live on a ROCm device and the library must be built."""
import math

import torch
from torch import nn

from . import kernels as K
from .stylegan2 import (ConstantInput, ConvLayer, EqualLinear, FusedLeakyReLU, ModulatedConv2d, PixelNorm, ToRGB,
                        conv_layer_nhwc, equal_linear_lrelu)


class NoiseInjection(nn.Module):
    """gpen_model.py:279-300 (isconcat: the scaled noise is concatenated, not added)."""

    def __init__(self, isconcat=True):
        super().__init__()
        self.isconcat = isconcat
        self.weight = nn.Parameter(torch.zeros(1))


class StyledConv(nn.Module):
    """gpen_model.py:318-357."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, isconcat=True):
        super().__init__()
        if not isconcat:
            raise NotImplementedError("GPEN ships isconcat=True models only (face_enhancement.py:34-37)")
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection(isconcat)
        self.activate = FusedLeakyReLU(out_channel * 2)

    def run_nhwc(self, x, style, feat):
        """x NHWC [B,H,W,Cin]; style [B,512]; feat: encoder feature map NHWC [B,Ho,Wo,Cout] (the layer's 'noise').
        Returns NHWC [B,Ho,Wo,2*Cout] = lrelu(cat(modconv(x), w_noise*feat) + bias) * sqrt(2)."""
        conv, act = self.conv, self.activate
        c = conv.out_channel
        b, h, w, _ = x.shape
        ho, wo = (2 * h, 2 * w) if conv.upsample else (h, w)
        if tuple(feat.shape) != (b, ho, wo, c):
            raise RuntimeError(f"noise feature map {tuple(feat.shape)} does not match the layer output [{b},{ho},{wo},{c}]")
        mod = conv.modulation
        s = K.modulate_vec(style, mod.weight, mod.bias)
        pk = conv.packed()
        d = K.demod_coefs(s, pk["wsq"], conv.scale)
        y = torch.empty(b, ho, wo, 2 * c, device=x.device, dtype=torch.float32)
        bias = act.bias[:c]
        kw = dict(in_scale=s, out_scale=d, bias=bias, act=1, alpha=act.negative_slope, gain=act.scale, out=y)
        ncls = 4 if conv.upsample else 1
        if K.want_bf16x3(b, h, w, conv.in_channel, c, ncls):
            K.conv_mfma(x, pk["w"], c, ncls=ncls, ostride=2 if conv.upsample else 1, w_split=conv.split_weights(), **kw)
        elif conv.upsample:
            K.upconv_mfma(x, pk["w3"], c, conv.blur.kernel, **kw)
        else:
            K.conv_mfma(x, pk["w"], c, **kw)
        K.noise_half(feat, self.noise.weight, act.bias[c:], y, c, act.negative_slope, act.scale)
        return y


class Generator(nn.Module):
    """gpen_model.py:380-556 (the StyleGAN2 decoder of GPEN)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01, isconcat=True,
                 narrow=1, device="cpu"):
        super().__init__()
        self.size, self.n_mlp, self.style_dim = size, n_mlp, style_dim
        self.feat_multiplier = 2 if isconcat else 1
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: int(512 * narrow), 8: int(512 * narrow), 16: int(512 * narrow), 32: int(512 * narrow),
                         64: int(256 * channel_multiplier * narrow), 128: int(128 * channel_multiplier * narrow),
                         256: int(64 * channel_multiplier * narrow), 512: int(32 * channel_multiplier * narrow),
                         1024: int(16 * channel_multiplier * narrow), 2048: int(8 * channel_multiplier * narrow)}
        fm = self.feat_multiplier
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel, isconcat=isconcat)
        self.to_rgb1 = ToRGB(self.channels[4] * fm, style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel * fm, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel,
                                         isconcat=isconcat))
            self.convs.append(StyledConv(out_channel * fm, out_channel, 3, style_dim, blur_kernel=blur_kernel,
                                         isconcat=isconcat))
            self.to_rgbs.append(ToRGB(out_channel * fm, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2

    def get_latent(self, z):
        """style MLP: PixelNorm + n_mlp x (EqualLinear(lr_mul) + fused leaky-ReLU) on [B, style_dim]."""
        w = K.pixelnorm(z)
        for lin in list(self.style)[1:]:
            w = equal_linear_lrelu(lin, w)
        return w

    @torch.no_grad()
    def decode(self, w, noise):
        """w [B,512] (one style for every layer); noise: 2*(log_size-2)+1 NHWC encoder feature maps, coarse to fine."""
        b = w.shape[0]
        x = K.const_input(self.input.input, b)
        x = self.conv1.run_nhwc(x, w, noise[0])
        mod = self.to_rgb1.conv.modulation
        skip = self.to_rgb1.run_nhwc(x, K.modulate_vec(w, mod.weight, mod.bias), None, 1, None)
        i = 1
        for conv1, conv2, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            x = conv1.run_nhwc(x, w, noise[i])
            x = conv2.run_nhwc(x, w, noise[i + 1])
            mod = to_rgb.conv.modulation
            skip = to_rgb.run_nhwc(x, K.modulate_vec(w, mod.weight, mod.bias), None, 1, skip)
            i += 2
        return skip


class FullGenerator(nn.Module):
    """gpen_model.py:628-690.  forward(inputs [B,3,size,size] in [-1,1]) -> (restored image, latent | None)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 isconcat=True, narrow=1, device="cpu"):
        super().__init__()
        self.log_size = int(math.log(size, 2))
        self.generator = Generator(size, style_dim, n_mlp, channel_multiplier=channel_multiplier, blur_kernel=blur_kernel,
                                   lr_mlp=lr_mlp, isconcat=isconcat, narrow=narrow)
        channels = self.generator.channels
        self.ecd0 = nn.Sequential(ConvLayer(3, channels[size], 1))
        in_channel = channels[size]
        self.names = ["ecd%d" % i for i in range(self.log_size - 1)]
        for i in range(self.log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            setattr(self, self.names[self.log_size - i + 1],
                    nn.Sequential(ConvLayer(in_channel, out_channel, 3, downsample=True)))
            in_channel = out_channel
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, style_dim, activation="fused_lrelu"))

    @torch.no_grad()
    def forward(self, inputs, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False):
        if inject_index is not None or input_is_latent:
            raise NotImplementedError("style mixing / latent input are never used by the face-swap pipeline "
                                      "(face_enhancement.py:104-106 calls model(img) only)")
        feats = []
        x = inputs
        for i, name in enumerate(self.names):
            x = conv_layer_nhwc(getattr(self, name)[0], x, x_is_nchw=(i == 0))
            feats.append(x)
        flat = K.nhwc_to_nchw(x).reshape(x.shape[0], -1)             # the reference flattens NCHW, :683
        z = equal_linear_lrelu(self.final_linear[0], flat)
        noise = [f for f in feats for _ in range(2)][::-1][1:]       # :686-687
        w = self.generator.get_latent(z)
        if truncation < 1:
            w = truncation_latent + truncation * (w - truncation_latent)
        image = self.generator.decode(w, noise)
        return image, (w.unsqueeze(1).repeat(1, self.generator.n_latent, 1) if return_latents else None)
