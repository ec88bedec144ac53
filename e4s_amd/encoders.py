"""Regional style encoder FSEncoder_PSP -- MI355X-native.

Module tree / state_dict identical to the reference (src/models/encoders/psp_encoders.py:238-262,
src/models/encoders/helpers.py:56-72,122-144): `input_layer.{0,2}`, `body.N.res_layer.{1,2,3,5.fc1,5.fc2}`,
`body.N.shortcut_layer.0`.  Execution is a fixed schedule of HIP kernels on NHWC tensors:

    IN(x) -> conv3x3 -> PReLU          e4s_instnorm_stats/apply + e4s_conv_mfma_f32 (PReLU in the epilogue)
    conv3x3(stride) -> IN -> SE        e4s_conv_mfma_f32 + stats(+pooled) + e4s_se_gate_f32
    (+ shortcut) gate * IN(r) + sc     e4s_instnorm_apply_f32 (one fused pass)
    36 masked_select chains (264-283)  e4s_region_mean_f32 x3 (no host syncs)
"""
from collections import namedtuple

import torch
from torch import nn
from torch.nn import Conv2d, InstanceNorm2d, MaxPool2d, Module, PReLU, ReLU, Sequential, Sigmoid, AdaptiveAvgPool2d

from . import kernels as K
from .packs import param_key


class Bottleneck(namedtuple("Block", ["in_channel", "depth", "stride"])):
    """helpers.py:21-22"""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


def _pack3x3(conv):
    """[Cout,Cin,3,3] -> [1,9,Cout,Cin] (cached on the module, invalidated by in-place updates)."""
    w = conv.weight
    key = param_key(w)
    if getattr(conv, "_e4s_pack", None) is None or conv._e4s_pack[0] != key:
        with torch.no_grad():
            conv._e4s_pack = (key, K.pack_taps(w.detach().float().contiguous()))
    return conv._e4s_pack[1]


def _conv3x3(x, conv, cout, in_stats=None, want_stats=False, **kw):
    """Stride-1 3x3 conv (+ fused epilogue) in the configured arithmetic: the split-bf16 kernel where it applies and
    K.PRECISION asks for it, the exact fp32-MFMA kernel otherwise.  in_stats: InstanceNorm statistics of x -- the
    normalisation (helpers.py:128-131) is folded into the split-bf16 kernel's halo staging, or runs as its own pass
    (e4s_instnorm_apply_f32; the same two fp32 operations per element) in front of the fp32 kernel."""
    w = _pack3x3(conv)
    if K.wino_eligible(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout, in_stats=in_stats is not None) and set(kw) <= {"bias", "slope", "act", "alpha", "gain", "se"} \
            and not (want_stats and kw.get("act", 0) != 0):
        # Winograd F(2,3) along the rows: 1.5x fewer MFMAs than the direct split-bf16 kernel (csrc/conv_wino.hip)
        if getattr(conv, "_e4s_wino", None) is None or conv._e4s_wino[0] != conv._e4s_pack[0]:
            conv._e4s_wino = (conv._e4s_pack[0], K.wino_weights(w))
        return K.conv_wino(x, conv._e4s_wino[1], cout, in_stats=in_stats, want_stats=want_stats, **kw)
    if K.want_bf16x3(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout):
        if getattr(conv, "_e4s_split", None) is None or conv._e4s_split[0] != conv._e4s_pack[0]:
            conv._e4s_split = (conv._e4s_pack[0], K.split_bf16x2(w))
        return K.conv_mfma(x, w, cout, w_split=conv._e4s_split[1], in_stats=in_stats, want_stats=want_stats, **kw)
    if in_stats is not None:
        x = K.instnorm_apply(x, in_stats)
    return K.conv_mfma(x, w, cout, want_stats=want_stats, **kw)


def _conv_strided(x, conv, cout, stride, ntaps, want_stats=False, se=None):
    """The unit's stride-2 3x3 conv / 1x1 shortcut conv (helpers.py:125-137): per-tap gather kernels -- split-bf16 where
    K.PRECISION asks for it and the launch has enough 256-pixel tiles, exact fp32 otherwise."""
    w = _pack3x3(conv)
    b, h, wd, _ = x.shape
    covered = cout % 128 == 0 or cout == 64                      # (64: one half-used 128-column tile of the gather kernel)
    tiles = (b * (h // stride) * (wd // stride) + 255) // 256 * ((cout + 127) // 128 if covered else 0)
    if K.PRECISION != "f32" and covered and (K.PRECISION == "bf16x3" or tiles >= K.BF16X3_MIN_BLOCKS):
        if getattr(conv, "_e4s_split", None) is None or conv._e4s_split[0] != conv._e4s_pack[0]:
            conv._e4s_split = (conv._e4s_pack[0], K.split_bf16x2(w))
        return K.conv_mfma(x, w, cout, istride=stride, ntaps=ntaps, w_split=conv._e4s_split[1], want_stats=want_stats, se=se)
    return K.conv_mfma(x, w, cout, istride=stride, ntaps=ntaps, want_stats=want_stats, se=se)


class SEModule(Module):
    """helpers.py:56-72 (parameter holder; evaluated by e4s_se_gate_f32)."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def forward(self, x):
        """torch semantics, for third-party modules that embed an SEModule (the native units call e4s_se_gate_f32)."""
        return x * torch.sigmoid(self.fc2(torch.relu(self.fc1(x.mean((2, 3), keepdim=True)))))


class bottleneck_IR_SE_Ours(Module):
    """helpers.py:122-144"""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.in_channel, self.depth, self.stride = in_channel, depth, stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False),
                                             InstanceNorm2d(depth))
        self.res_layer = Sequential(InstanceNorm2d(in_channel),
                                    Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
                                    PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
                                    InstanceNorm2d(depth),
                                    SEModule(depth, 16))

    def run_nhwc(self, x, st_x=None, want_stats=False):
        """x NHWC [B,H,W,Cin] -> NHWC [B,H/stride,W/stride,depth].  Every InstanceNorm statistic is produced by the kernel
        that writes the tensor it describes: st_x (of the input) by the previous unit's final pass (want_stats -> returns
        (out, st_out) for the next unit), those of the two conv outputs by the convs' epilogues."""
        conv1, prelu, conv2, se = self.res_layer[1], self.res_layer[2], self.res_layer[3], self.res_layer[5]
        if st_x is None:
            st_x, _ = K.instnorm_stats(x)
        r = _conv3x3(x, conv1, self.depth, in_stats=st_x, act=2, slope=prelu.weight)
        se_w = (se.fc1.weight.view(se.fc1.weight.shape[0], -1), se.fc2.weight.view(se.fc2.weight.shape[0], -1))
        # the SE gate leaves the launch that finishes conv2's statistics (e4s_instnorm_finalize_se_f32)
        if self.stride == 1:
            r, (st_r, gate) = _conv3x3(r, conv2, self.depth, want_stats=True, se=se_w)
        else:
            r, (st_r, gate) = _conv_strided(r, conv2, self.depth, self.stride, 9, want_stats=True, se=se_w)
        if self.in_channel == self.depth:
            return K.instnorm_apply(r, st_r, gate=gate, res=x, rs=self.stride, want_stats=want_stats)   # MaxPool2d(1, s)
        sc, (st_sc, _) = _conv_strided(x, self.shortcut_layer[0], self.depth, self.stride, 1, want_stats=True)
        return K.instnorm_apply(r, st_r, gate=gate, res=sc, res_stats=st_sc, want_stats=want_stats)

    def forward(self, x):
        return K.nhwc_to_nchw(self.run_nhwc(K.nchw_to_nhwc(x)))


class FSEncoder_PSP(Module):
    """psp_encoders.py:238-309"""

    def __init__(self, mode="ir_se", opts=None):
        super().__init__()
        assert mode in ["ir_se"], "E4S instantiates FSEncoder_PSP(mode='ir_se') only (networks.py:48)"
        blocks = [get_block(64, 128, 3), get_block(128, 256, 4), get_block(256, 512, 14), get_block(512, 512, 3)]
        self.n_styles = 11
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), InstanceNorm2d(64), PReLU(64))
        modules = []
        for block in blocks:
            for bt in block:
                modules.append(bottleneck_IR_SE_Ours(bt.in_channel, bt.depth, bt.stride))
        self.body = Sequential(*modules)

    def get_per_comp_styleCode(self, style_feats, segmap):
        """psp_encoders.py:264-283 (NCHW feats, one-hot segmap) -> [B,R,C]."""
        labels, _ = K.mask_labels(segmap)
        feats = K.nchw_to_nhwc(style_feats)
        b, c = style_feats.shape[:2]
        out = torch.empty(b, segmap.shape[1], c, device=feats.device, dtype=torch.float32)
        K.region_mean_into(feats, labels, out, segmap.shape[1], 0)
        return out

    def encode_nhwc(self, x256, labels, num_regions):
        """x256: NHWC [B,256,256,3]; labels uint8 [B,Hm,Wm].  Returns codes [B,R,1280]."""
        x, st = K.conv3x3_small(x256, self.input_layer[0].weight, want_stats=True)      # statistics from the stem's own epilogue
        x, st_x = K.instnorm_apply(x, st, slope=self.input_layer[2].weight, want_stats=True)
        b = x.shape[0]
        codes = torch.empty(b, num_regions, 256 + 512 + 512, device=x.device, dtype=torch.float32)
        off = {6: 0, 20: 256, 23: 768}
        last = len(self.body) - 1
        for i, unit in enumerate(self.body):
            if i < last:
                x, st_x = unit.run_nhwc(x, st_x, want_stats=True)
            else:
                x = unit.run_nhwc(x, st_x)
            if i in off:
                K.region_mean_into(x, labels, codes, num_regions, off[i])
        return codes, x

    @torch.no_grad()
    def forward(self, x, segmap):
        """x NCHW [B,3,256,256], segmap one-hot [B,R,Hm,Wm] -> (codes [B,R,1280], zeros [B,512,16,16])."""
        labels, _ = K.mask_labels(segmap)
        codes, last = self.encode_nhwc(K.nchw_to_nhwc(x), labels, segmap.shape[1])
        b, h, w, c = last.shape
        return codes, torch.zeros(b, c, h, w, device=x.device, dtype=torch.float32)
