"""Loss networks of the optimisation loop -- MI355X-native.

`IDLoss` (src/criteria/id_loss.py:6-57 over the IR-SE50 `Backbone`, src/models/encoders/model_irse.py:10-69,
helpers.py:97-119), `LPIPS` (src/criteria/lpips/lpips.py:8-35 over torchvision's AlexNet `features`,
src/criteria/lpips/networks.py:28-83) and `FaceParsingLoss` (src/criteria/face_parsing/face_parsing_loss.py:20-78 over the
encoder half of the parsing `unet`, unet.py:6-92) as scripts/optimization.py:88-122 calls them: frozen networks whose only gradient is
the one back to the generated image.  Module trees / state_dict keys equal the reference's, so its checkpoints load
unchanged; execution is a fixed schedule of HIP kernels on NHWC tensors with a tape:

    pooling + crop + z-score        e4s_adaptive_pool_f32 (one pass per resolution, reads the NCHW image directly)
    3-channel stems                 e4s_conv3x3_small_f32 / e4s_conv_smallcin_f32 (+ their image gradients)
    3x3 / 5x5 / 1x1 convs           e4s_conv_mfma_f32 / e4s_conv_bf16x3_f32; dgrad = the same kernels on the flipped,
                                    transposed taps (stride 2: on the zero-inserted gradient)
    BatchNorm2d (eval)              folded into the conv's operand staging as (x - mu') * rho' -- the InstanceNorm slot of
                                    the encoder kernels with constant statistics; backward e4s_norm_bwd_frozen_f32
    SE, PReLU, MaxPool2d(3, 2)      e4s_se_gate_f32, e4s_prelu_f32, e4s_maxpool3s2_f32 (+ backward)
    Linear(25088, 512) + BN1d       one e4s_grouped_linear_f32 with both BatchNorms folded into the packed weight
    cosine / LPIPS distance         e4s_cosine_f32, e4s_lpips_layer_f32 (+ backward)

    parsing UNet encoder            conv + BatchNorm(eval) folded into ONE conv with bias + ReLU epilogue (16-channel maps
                                    zero-padded to the 32-channel K step), e4s_maxpool2_f32, e4s_relu_bwd_f32

The target image's features are cached per target tensor (the reference recomputes them every step, id_loss.py:33-35).
"""
import os

import torch
from torch import nn
from torch.autograd.function import once_differentiable
from torch.nn import (BatchNorm1d, BatchNorm2d, Conv2d, Dropout, Linear, MaxPool2d, Module, PReLU, ReLU, Sequential)

import functools

from . import kernels as K
from .encoders import SEModule, _conv3x3, _conv_strided, get_block
from .packs import param_key


def _plain_split(fn):
    """The frozen loss networks accept e4s_conv_mfma_f32's K split on plain maps with few blocks per sample (kernels.f32_plain_split; their
    28^2 / 56^2 layers at batch 1-2 are chains of exposed stage latencies otherwise).  Forward, feature extraction AND each backward: autograd runs
    the backward on its own thread, where the forward's thread-local switch is not set."""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with K.f32_plain_split():
            return fn(*a, **k)
    return wrapped

# The reference ALWAYS loads trained weights into its loss networks (id_loss.py:15, face_parsing_loss.py:29, lpips/utils.py:11-20
# + torchvision's pretrained AlexNet) and fails when they are missing.  Same here: a loss network without weights is an error
# unless this switch is on -- benchmarks and tests set it because they load seeded synthetic state dicts right after construction.
ALLOW_UNINITIALIZED = os.environ.get("E4S_ALLOW_UNINITIALIZED_LOSS_NETS", "0") == "1"


def _have_weights(what, path):
    if path and os.path.exists(path):
        return True
    if ALLOW_UNINITIALIZED:
        return False
    raise FileNotFoundError(
        f"{what}: no weights at {path!r}.  The reference loads trained weights here; optimising against a randomly initialised "
        "loss network is silently meaningless.  Point the option at the checkpoint, or set e4s_amd.criteria.ALLOW_UNINITIALIZED "
        "= True (env E4S_ALLOW_UNINITIALIZED_LOSS_NETS=1) when a state dict is loaded afterwards (synthetic-weight runs).")


def _target_key(y):
    """Cache key of a target image's features.  The entry also HOLDS y (so its storage cannot be freed and handed to a new
    tensor with the same pointer and version 0 -- the caching allocator does exactly that); _version catches in-place edits."""
    return (y.data_ptr(), y._version, tuple(y.shape), y.device)


# ---------------------------------------------------------------------------------------------------------------
# frozen-parameter packs
# ---------------------------------------------------------------------------------------------------------------
class _Taps:
    """A conv weight in the role `_conv3x3` / `_conv_strided` expect of a module (they cache their packs on it)."""

    def __init__(self, weight):
        self.weight = weight


def _transposed(conv):
    """The dgrad operand of a k x k conv: taps flipped, in/out channels swapped (cached; the networks are frozen)."""
    key = param_key(conv.weight)
    if getattr(conv, "_e4s_t", None) is None or conv._e4s_t[0] != key:
        with torch.no_grad():
            conv._e4s_t = (key, _Taps(conv.weight.detach().float().flip(2, 3).transpose(0, 1).contiguous()))
    return conv._e4s_t[1]


def _bn_stats(bn, batch):
    """BatchNorm in eval mode as the kernels' {mean, rstd} operand: gamma*(x-m)/sqrt(v+eps)+beta = (x - mu') * rho' with
    rho' = gamma/sqrt(v+eps), mu' = m - beta/rho'.  [B,C,2], cached per (parameters, batch)."""
    key = (param_key(bn.weight), param_key(bn.bias), bn.running_mean._version, bn.running_var._version, batch)
    if getattr(bn, "_e4s_stats", None) is None or bn._e4s_stats[0] != key:
        with torch.no_grad():
            rho = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            if bool((rho.abs() < 1e-30).any()):
                raise RuntimeError("BatchNorm with a zero scale cannot be folded into the conv's operand staging")
            mu = bn.running_mean.float() - bn.bias.float() / rho
            bn._e4s_stats = (key, torch.stack([mu, rho], 1)[None].expand(batch, -1, -1).contiguous())
    return bn._e4s_stats[1]


def _se_weights(se):
    return se.fc1.weight.detach().view(se.fc1.weight.shape[0], -1), se.fc2.weight.detach().view(se.fc2.weight.shape[0], -1)


# ---------------------------------------------------------------------------------------------------------------
# IR-SE50 (ArcFace) backbone
# ---------------------------------------------------------------------------------------------------------------
class Flatten(Module):
    """helpers.py:10-12"""

    def forward(self, x):
        return x.view(x.size(0), -1)


class bottleneck_IR_SE(Module):
    """helpers.py:97-119"""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.in_channel, self.depth, self.stride = in_channel, depth, stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        self.res_layer = Sequential(BatchNorm2d(in_channel),
                                    Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
                                    PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
                                    BatchNorm2d(depth),
                                    SEModule(depth, 16))

    def run_nhwc(self, x, tape=None):
        bn1, conv1, prelu, conv2, bn2, se = self.res_layer
        b = x.shape[0]
        st_x = _bn_stats(bn1, b)
        u1 = _conv3x3(x, conv1, self.depth, in_stats=st_x)
        r1 = K.prelu(u1, prelu.weight.detach())
        r2 = _conv3x3(r1, conv2, self.depth) if self.stride == 1 else _conv_strided(r1, conv2, self.depth, self.stride, 9)
        st_r = _bn_stats(bn2, b)
        inst, _ = K.instnorm_stats(r2)                                  # its mean column = the spatial average of r2
        pooled = (inst[:, :, 0] - st_r[:, :, 0]) * st_r[:, :, 1]        # avg_pool(BN(r2)), [B,C]
        fc1, fc2 = _se_weights(se)
        gate = K.se_gate(pooled.contiguous(), fc1, fc2)
        rec = dict(unit=self, x=x, u1=u1, r2=r2, pooled=pooled, gate=gate) if tape is not None else None
        if self.in_channel == self.depth:
            out = K.instnorm_apply(r2, st_r, gate=gate, res=x, rs=self.stride)
        else:
            sc = _conv_strided(x, self.shortcut_layer[0], self.depth, self.stride, 1)
            out = K.instnorm_apply(r2, st_r, gate=gate, res=sc, res_stats=_bn_stats(self.shortcut_layer[1], b))
        if tape is not None:
            tape.append(rec)
        return out

    def backward_nhwc(self, rec, dout):
        """dL/dx of run_nhwc (parameters are frozen)."""
        bn1, conv1, prelu, conv2, bn2, se = self.res_layer
        x, u1, r2, gate, pooled = rec["x"], rec["u1"], rec["r2"], rec["gate"], rec["pooled"]
        b, s = x.shape[0], self.stride
        st_r = _bn_stats(bn2, b)
        # out = gate * BN(r2) + shortcut; gate = sigmoid(fc2 relu(fc1 pooled)), pooled = mean_p BN(r2)
        dgate = K.instnorm_bwd_sums(dout, r2, st_r)[:, :, 1]
        fc1, fc2 = _se_weights(se)
        # [B,C]-sized chain rule on the native grouped kernels (no library GEMM inside a captured step: kernels.sum_all)
        hidden = K.grouped_linear(pooled.contiguous().unsqueeze(1), fc1.unsqueeze(0).contiguous(), None, None, 1.0, act=1, alpha=0.0)
        dz = (dgate * gate * (1.0 - gate)).unsqueeze(1).contiguous()
        dh = K.grouped_linear_t(dz, fc2.unsqueeze(0).contiguous(), 1.0, ref=hidden, alpha=0.0)
        dpooled = K.grouped_linear_t(dh, fc1.unsqueeze(0).contiguous(), 1.0).squeeze(1)
        extra = (dpooled / float(r2.shape[1] * r2.shape[2])).contiguous()
        dr2 = K.norm_bwd_frozen(dout, st_r, gate=gate, extra=extra)
        gz2 = dr2 if s == 1 else K.strided_scatter(dr2, s)
        dr1 = _conv3x3(gz2, _transposed(conv2), self.depth)
        du1, _ = K.prelu_bwd(dr1, u1, prelu.weight.detach())
        dxn = _conv3x3(du1, _transposed(conv1), self.in_channel)
        dx = K.norm_bwd_frozen(dxn, _bn_stats(bn1, b))
        if self.in_channel == self.depth:
            K.strided_scatter(dout, s, out=dx)                         # MaxPool2d(1, s) backward (s = 1: plain add)
        else:
            sconv, sbn = self.shortcut_layer
            dsc = K.norm_bwd_frozen(dout, _bn_stats(sbn, b))
            t = _conv_strided(dsc, _transposed(sconv), self.in_channel, 1, 1)
            K.strided_scatter(t, s, out=dx)
        return dx

    def forward(self, x):
        return K.nhwc_to_nchw(self.run_nhwc(K.nchw_to_nhwc(x)))


class Backbone(Module):
    """model_irse.py:10-69 (input_size 112, ir_se)."""

    TAPS = {2: 0, 6: 1, 20: 2, 23: 3}          # body indices whose outputs the multi-scale identity loss compares

    def __init__(self, input_size=112, num_layers=50, mode="ir_se", drop_ratio=0.4, affine=True):
        super().__init__()
        assert input_size == 112 and num_layers == 50 and mode == "ir_se", "IDLoss instantiates Backbone(112, 50, 'ir_se')"
        blocks = [get_block(64, 64, 3), get_block(64, 128, 4), get_block(128, 256, 14), get_block(256, 512, 3)]
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.output_layer = Sequential(BatchNorm2d(512), Dropout(drop_ratio), Flatten(), Linear(512 * 7 * 7, 512),
                                       BatchNorm1d(512, affine=affine))
        self.body = Sequential(*[bottleneck_IR_SE(bt.in_channel, bt.depth, bt.stride) for blk in blocks for bt in blk])

    def _head(self):
        """BatchNorm2d(512) -> Flatten (NCHW order) -> Linear -> BatchNorm1d as ONE affine map on the NHWC-flattened feature:
        (W'' [1,512,25088], b'' [1,512]), cached."""
        bn2, _, _, lin, bn1 = self.output_layer
        key = (param_key(lin.weight), param_key(lin.bias), param_key(bn2.weight), param_key(bn2.bias), bn2.running_mean._version,
               bn2.running_var._version, bn1.running_mean._version, bn1.running_var._version,
               param_key(bn1.weight) if bn1.affine else None)
        if getattr(self, "_e4s_head", None) is None or self._e4s_head[0] != key:
            with torch.no_grad():
                s2 = bn2.weight.float() / torch.sqrt(bn2.running_var.float() + bn2.eps)           # [512] per channel
                t2 = bn2.bias.float() - bn2.running_mean.float() * s2
                w = lin.weight.float().view(512, 512, 49)                                          # [o, c, p]
                bias = lin.bias.float() + (w * t2[None, :, None]).sum((1, 2))
                w = (w * s2[None, :, None]).permute(0, 2, 1).reshape(512, 49 * 512)               # [o, p*512 + c]
                s1 = torch.rsqrt(bn1.running_var.float() + bn1.eps)
                t1 = -bn1.running_mean.float() * s1
                if bn1.affine:
                    s1, t1 = s1 * bn1.weight.float(), t1 * bn1.weight.float() + bn1.bias.float()
                self._e4s_head = (key, (w * s1[:, None]).contiguous()[None], (bias * s1 + t1).contiguous()[None])
        return self._e4s_head[1], self._e4s_head[2]

    def features_nhwc(self, x112, multi_scale, tape=None):
        """x112 NHWC [B,112,112,3] -> the compared features, NOT yet l2-normalised (the cosine kernel normalises): the four
        tapped maps as NHWC tensors (model_irse.py:53-60 flattens them channel-major; a cosine does not see the order) and
        the [B,512] embedding."""
        conv0, bn0, prelu0 = self.input_layer
        b = x112.shape[0]
        c0 = K.conv3x3_small(x112, conv0.weight.detach())
        x = K.instnorm_apply(c0, _bn_stats(bn0, b), slope=prelu0.weight.detach())
        if tape is not None:
            tape.append(dict(x112=x112, c0=c0))
        feats = [None] * 4
        for i, unit in enumerate(self.body):
            x = unit.run_nhwc(x, tape)
            if multi_scale and i in self.TAPS:
                feats[self.TAPS[i]] = x
        w, bias = self._head()
        emb = K.grouped_linear(x.view(b, 1, -1), w, bias, None, 1.0).view(b, -1)
        return (feats if multi_scale else []) + [emb]

    def backward_nhwc(self, tape, dfeats, multi_scale):
        """dfeats: gradients of the rows features_nhwc returned (None = zero) -> dL/d(x112) NHWC."""
        w, _ = self._head()
        stem = tape[0]
        b = stem["x112"].shape[0]
        last = tape[-1]["r2"].shape                                      # [B,7,7,512]
        dx = K.grouped_linear_t(dfeats[-1].view(b, 1, -1), w, 1.0).view(last)
        for i in reversed(range(len(self.body))):
            if multi_scale and i in self.TAPS and dfeats[self.TAPS[i]] is not None:
                dx = K.add_scale(dx, dfeats[self.TAPS[i]], 1.0)
            dx = self.body[i].backward_nhwc(tape[1 + i], dx)
        conv0, bn0, prelu0 = self.input_layer
        st0 = _bn_stats(bn0, b)
        n0 = K.instnorm_apply(stem["c0"], st0)
        dn0, _ = K.prelu_bwd(dx, n0, prelu0.weight.detach())
        dc0 = K.norm_bwd_frozen(dn0, st0)
        return K.conv_smallcin_bwd(dc0, _smallcin_pack(conv0), tuple(stem["x112"].shape), 3, 1, 1, cache=_smallcin_cache(conv0))

    def forward(self, x, multi_scale=False):
        """model_irse.py:44-69: NCHW [B,3,112,112] -> list of l2-normalised feature rows."""
        with torch.no_grad():
            feats = self.features_nhwc(K.nchw_to_nhwc(x), multi_scale)
            rows = [K.nhwc_to_nchw(f).view(f.shape[0], -1) if f.dim() == 4 else f for f in feats]
            return [r / torch.norm(r, 2, 1, True) for r in rows]


def _smallcin_pack(conv):
    key = param_key(conv.weight)
    if getattr(conv, "_e4s_small", None) is None or conv._e4s_small[0] != key:
        conv._e4s_small = (key, K.pack_smallcin(conv.weight), {})      # the dict: derived images of this pack (conv_smallcin_bwd's GEMM form)
    return conv._e4s_small[1]


def _smallcin_cache(conv):
    """The per-pack cache handed to K.conv_smallcin_bwd (lives and dies with the pack of THIS conv)."""
    _smallcin_pack(conv)
    return conv._e4s_small[2]


class _IDLossFn(torch.autograd.Function):
    """sum over scales of mean_i (1 - cos(feat_k(y_hat_i), feat_k(y_i))) with the gradient back to y_hat (NCHW)."""

    @staticmethod
    @_plain_split
    def forward(ctx, y_hat, mod, y_feats):
        multi = mod.opts_multiscale
        tape = []
        x112, pre = mod._prep(y_hat)
        feats = mod.facenet.features_nhwc(x112, multi, tape)
        n = y_hat.shape[0]
        coefs = [K.cosine(f, t) for f, t in zip(feats, y_feats)]
        sims = torch.stack([c[:, 0] for c in coefs])                                # [scales, B]
        loss = (1.0 - sims).mean(1).sum()
        ctx.mod, ctx.tape, ctx.feats, ctx.y_feats, ctx.coefs, ctx.pre, ctx.n = mod, tape, feats, y_feats, coefs, pre, n
        ctx.in_shape = tuple(y_hat.shape)
        ctx.mark_non_differentiable(sims)
        return loss, sims

    @staticmethod
    @once_differentiable
    @_plain_split
    def backward(ctx, gloss, _gsims):
        mod = ctx.mod
        if ctx.tape is None:
            raise RuntimeError("IDLoss: the activation tape was released by the first backward; run the forward again "
                               "(retain_graph / double backward through this node are not supported)")
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dfeats = [K.cosine_bwd(f, t, c, g, -1.0 / ctx.n) for f, t, c in zip(ctx.feats, ctx.y_feats, ctx.coefs)]
        if not mod.opts_multiscale:
            dfeats = [None] * 4 + dfeats
        dx112 = mod.facenet.backward_nhwc(ctx.tape, dfeats, mod.opts_multiscale)
        ctx.tape = None
        return mod._prep_bwd(dx112, ctx.pre, ctx.in_shape), None, None


class IDLoss(Module):
    """src/criteria/id_loss.py:6-57.  `opts.ir_se50_path` is loaded when it names an existing file (the reference always
    loads it); without it the backbone keeps its initialisation (synthetic-weight benchmarks and tests)."""

    CROP = (35, 32, 188, 188)                   # x[:, :, 35:223, 32:220] (id_loss.py:28)

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.face_pool_1 = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
        path = getattr(opts, "ir_se50_path", None)
        if _have_weights("IDLoss (opts.ir_se50_path)", path):
            self.facenet.load_state_dict(torch.load(path, map_location="cpu"))
        self.face_pool_2 = torch.nn.AdaptiveAvgPool2d((112, 112))
        self.facenet.eval()
        self.set_requires_grad(False)
        self._target = None

    @property
    def opts_multiscale(self):
        return bool(getattr(self.opts, "id_loss_multiscale", True))

    def set_requires_grad(self, flag=True):
        for p in self.parameters():
            p.requires_grad = flag

    def _prep(self, x):
        """id_loss.py:26-29: pool to 256 (if needed), crop, pool to 112 -> NHWC [B,112,112,3]."""
        x = x.detach()
        if x.shape[2] != 256:
            p1 = K.adaptive_pool(x, (256, 256))
            return K.adaptive_pool(p1, (112, 112), crop=self.CROP, in_nchw=False), ("two", tuple(p1.shape))
        return K.adaptive_pool(x, (112, 112), crop=self.CROP), ("one", None)

    def _prep_bwd(self, dx112, pre, in_shape):
        if pre[0] == "two":
            dp1 = K.adaptive_pool_bwd(dx112, pre[1], crop=self.CROP, in_nchw=False)
            return K.adaptive_pool_bwd(dp1, in_shape)
        return K.adaptive_pool_bwd(dx112, in_shape, crop=self.CROP)

    @_plain_split
    def extract_feats(self, x):
        """l2-normalised feature rows of x (NCHW), no gradient."""
        with torch.no_grad():
            feats = self.facenet.features_nhwc(self._prep(x)[0], self.opts_multiscale)
            rows = [K.nhwc_to_nchw(f).view(f.shape[0], -1) if f.dim() == 4 else f for f in feats]
            return [r / torch.norm(r, 2, 1, True) for r in rows]

    def _target_feats(self, y):
        key = _target_key(y)
        if self._target is None or self._target[0] != key or self._target[1] is not y:
            with torch.no_grad():
                self._target = (key, y, self.facenet.features_nhwc(self._prep(y)[0], self.opts_multiscale))
        return self._target[2]

    def forward(self, y_hat, y):
        """-> (loss, sim_improvement, None).  sim_improvement is a 0-dim tensor (float() of it is the reference's number;
        the reference pays two host syncs per sample for it, id_loss.py:49)."""
        loss, sims = _IDLossFn.apply(y_hat, self, self._target_feats(y))
        return loss, (sims - 1.0).mean(1).sum(), None


# ---------------------------------------------------------------------------------------------------------------
# LPIPS (AlexNet)
# ---------------------------------------------------------------------------------------------------------------
def alexnet_features():
    """torchvision.models.alexnet().features (torchvision 0.13, the reference's pinned stack; torchvision itself is not a
    dependency here): indices and shapes as its state_dict has them."""
    return Sequential(
        Conv2d(3, 64, kernel_size=11, stride=4, padding=2), ReLU(inplace=True), MaxPool2d(kernel_size=3, stride=2),
        Conv2d(64, 192, kernel_size=5, padding=2), ReLU(inplace=True), MaxPool2d(kernel_size=3, stride=2),
        Conv2d(192, 384, kernel_size=3, padding=1), ReLU(inplace=True),
        Conv2d(384, 256, kernel_size=3, padding=1), ReLU(inplace=True),
        Conv2d(256, 256, kernel_size=3, padding=1), ReLU(inplace=True), MaxPool2d(kernel_size=3, stride=2))


class LinLayers(nn.ModuleList):
    """src/criteria/lpips/networks.py:23-34"""

    def __init__(self, n_channels_list):
        super().__init__([Sequential(nn.Identity(), Conv2d(nc, 1, 1, 1, 0, bias=False)) for nc in n_channels_list])
        for p in self.parameters():
            p.requires_grad = False


class AlexNet(Module):
    """src/criteria/lpips/networks.py:37-83 (BaseNet + AlexNet)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("mean", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("std", torch.Tensor([.458, .448, .450])[None, :, None, None])
        self.layers = alexnet_features()
        self.target_layers = [2, 5, 8, 10, 12]
        self.n_channels_list = [64, 192, 384, 256, 256]
        for p in self.parameters():
            p.requires_grad = False

    def _affine(self):
        inv = (1.0 / self.std.float()).view(3).contiguous()
        return inv, (-self.mean.float().view(3) * inv).contiguous()

    def _conv(self, x, conv, taps):
        """conv + bias + ReLU on the MFMA kernels (k 3: halo-tiled, k 5: per-tap gather)."""
        cout = conv.out_channels
        if taps == 9:
            return _conv3x3(x, conv, cout, bias=conv.bias.detach(), act=1, alpha=0.0, gain=1.0)
        return K.conv_mfma(x, _pack(conv), cout, ntaps=taps, spatial=False, bias=conv.bias.detach(), act=1, alpha=0.0, gain=1.0)

    def features_nhwc(self, img, size, tape=None):
        """img NCHW -> the five ReLU outputs (NHWC, not yet unit-normalised) of the image pooled to size x size."""
        L = self.layers
        scale, shift = self._affine()
        x0 = K.adaptive_pool(img.detach(), (size, size), scale=scale, shift=shift)
        f1 = K.conv_smallcin(x0, _smallcin_pack(L[0]), L[0].bias.detach(), 64, 11, 4, 2, relu=True)
        p1, i1 = K.maxpool3s2(f1)
        f2 = self._conv(p1, L[3], 25)
        p2, i2 = K.maxpool3s2(f2)
        f3 = self._conv(p2, L[6], 9)
        f4 = self._conv(f3, L[8], 9)
        f5 = self._conv(f4, L[10], 9)
        if tape is not None:
            tape.update(x0=tuple(x0.shape), i1=i1, i2=i2, size=size)
        return [f1, f2, f3, f4, f5]

    def backward_nhwc(self, tape, feats, dist_bwd, img_shape, dimg):
        """dist_bwd(k, acc): adds dL/d(feats[k]) of the k-th distance layer to acc (None: returns it) -> accumulates
        dL/d(img) (NCHW) into dimg."""
        L = self.layers
        f1, f2, f3, f4, f5 = feats
        zero = {c: torch.zeros(c, device=f1.device, dtype=torch.float32) for c in (64, 192, 384, 256)}
        d5, _ = K.prelu_bwd(dist_bwd(4, None), f5, zero[256])                                   # ReLU backward
        d4, _ = K.prelu_bwd(dist_bwd(3, _conv3x3(d5, _transposed(L[10]), 256)), f4, zero[256])
        d3, _ = K.prelu_bwd(dist_bwd(2, _conv3x3(d4, _transposed(L[8]), 384)), f3, zero[384])
        dp2 = _conv3x3(d3, _transposed(L[6]), 192)
        d2, _ = K.prelu_bwd(dist_bwd(1, K.maxpool3s2_bwd(dp2, tape["i2"], tuple(f2.shape))), f2, zero[192])
        dp1 = K.conv_mfma(d2, _pack(_transposed(L[3])), 64, ntaps=25, spatial=False)
        d1, _ = K.prelu_bwd(dist_bwd(0, K.maxpool3s2_bwd(dp1, tape["i1"], tuple(f1.shape))), f1, zero[64])
        dx0 = K.conv_smallcin_bwd(d1, _smallcin_pack(L[0]), tape["x0"], 11, 4, 2, cache=_smallcin_cache(L[0]))
        scale, _ = self._affine()
        return K.adaptive_pool_bwd(dx0, img_shape, scale=scale, dx_acc=dimg)

    def forward(self, x):
        """networks.py:53-64: unit-normalised NCHW activations of the five target layers."""
        with torch.no_grad():
            out = []
            for f in self.features_nhwc(x, x.shape[2]):
                n = K.nhwc_to_nchw(f)
                out.append(n / (torch.sqrt(torch.sum(n ** 2, dim=1, keepdim=True)) + 1e-10))
            return out


def _pack(conv):
    """[1, k*k, Cout, Cin] taps of a conv (cached on the module)."""
    key = param_key(conv.weight)
    if getattr(conv, "_e4s_pack", None) is None or conv._e4s_pack[0] != key:
        with torch.no_grad():
            conv._e4s_pack = (key, K.pack_taps(conv.weight.detach().float().contiguous()))
    return conv._e4s_pack[1]


class _LPIPSFn(torch.autograd.Function):
    """sum over `sizes` of LPIPS(pool(x, s), pool(y, s)) with the gradient back to x (NCHW)."""

    @staticmethod
    @_plain_split
    def forward(ctx, x, mod, sizes, y_feats):
        b = x.shape[0]
        tapes, feats_all = [], []
        total = None
        for size, fy in zip(sizes, y_feats):
            tape = {}
            fx = mod.net.features_nhwc(x, size, tape)
            tapes.append(tape)
            feats_all.append(fx)
            for k in range(5):
                d = K.lpips_layer(fx[k], fy[k], mod.lin_weight(k))
                total = d if total is None else total + d
        ctx.mod, ctx.tapes, ctx.feats, ctx.y_feats, ctx.sizes, ctx.b = mod, tapes, feats_all, y_feats, sizes, b
        ctx.in_shape = tuple(x.shape)
        return total.sum() / b

    @staticmethod
    @once_differentiable
    @_plain_split
    def backward(ctx, gout):
        mod = ctx.mod
        if ctx.tapes is None:
            raise RuntimeError("LPIPS: the activation tape was released by the first backward; run the forward again "
                               "(retain_graph / double backward through this node are not supported)")
        g = gout.reshape(1).to(torch.float32).contiguous()
        dimg = None
        for tape, fx, fy in zip(ctx.tapes, ctx.feats, ctx.y_feats):
            def dist_bwd(k, acc, fx=fx, fy=fy):
                return K.lpips_layer_bwd(fx[k], fy[k], mod.lin_weight(k), g, 1.0 / ctx.b, dfx_acc=acc)
            dimg = mod.net.backward_nhwc(tape, fx, dist_bwd, ctx.in_shape, dimg)
        ctx.tapes = ctx.feats = None
        return dimg, None, None, None


class LPIPS(Module):
    """src/criteria/lpips/lpips.py:8-35, net_type 'alex'.  The linear-layer weights come from the LPIPS release the reference
    downloads (lpips/utils.py:11-20); `weights` may name a local copy of that state dict or of the whole module, otherwise
    the initialisation is kept (there is no network here)."""

    def __init__(self, net_type="alex", version="0.1", weights=None):
        assert version in ["0.1"], "v0.1 is only supported now"
        assert net_type == "alex", "scripts/optimization.py:79 instantiates LPIPS(net_type='alex')"
        super().__init__()
        self.net = AlexNet()
        self.lin = LinLayers(self.net.n_channels_list)
        weights = weights or os.environ.get("E4S_LPIPS_WEIGHTS")
        if _have_weights("LPIPS (weights= / $E4S_LPIPS_WEIGHTS: torchvision AlexNet + the LPIPS v0.1 lin layers)", weights):
            sd = torch.load(weights, map_location="cpu")
            (self if any(k.startswith("net.") for k in sd) else self.lin).load_state_dict(sd)
        self._target = None

    def lin_weight(self, k):
        return self.lin[k][1].weight.detach().view(-1)

    def _target_feats(self, y, sizes):
        key = _target_key(y) + (tuple(sizes),)
        if self._target is None or self._target[0] != key or self._target[1] is not y:
            with torch.no_grad():
                self._target = (key, y, [self.net.features_nhwc(y, s) for s in sizes])
        return self._target[2]

    def forward_pooled(self, x, y, sizes):
        """sum_s LPIPS(adaptive_avg_pool2d(x, s), adaptive_avg_pool2d(y, s)) -- the three-scale term of
        scripts/optimization.py:100-108 with the pooling fused into the networks' first pass."""
        sizes = tuple(int(s) for s in sizes)
        return _LPIPSFn.apply(x, self, sizes, self._target_feats(y, sizes))

    def forward(self, x, y):
        return self.forward_pooled(x, y, (x.shape[2],))


# ---------------------------------------------------------------------------------------------------------------
# Face-parsing loss (UNet encoder features)
# ---------------------------------------------------------------------------------------------------------------
class unetConv2(Module):
    """src/criteria/face_parsing/model_utils.py:177-203 (is_batchnorm=True)"""

    def __init__(self, in_size, out_size, is_batchnorm=True):
        super().__init__()
        assert is_batchnorm
        self.conv1 = Sequential(Conv2d(in_size, out_size, 3, 1, 1), BatchNorm2d(out_size), ReLU())
        self.conv2 = Sequential(Conv2d(out_size, out_size, 3, 1, 1), BatchNorm2d(out_size), ReLU())

    def forward(self, x):
        return self.conv2(self.conv1(x))


class unetUp(Module):
    """model_utils.py:206-221 (is_deconv=True): decoder half, a parameter holder with the torch forward -- the loss never runs
    it (face_parsing_loss.py:47-50 uses extract_feats only)."""

    def __init__(self, in_size, out_size, is_deconv=True, is_batchnorm=True):
        super().__init__()
        self.conv = unetConv2(in_size, out_size, is_batchnorm)
        self.up = nn.ConvTranspose2d(in_size, out_size, kernel_size=2, stride=2)

    def forward(self, inputs1, inputs2):
        outputs2 = self.up(inputs2)
        offset = outputs2.size()[2] - inputs1.size()[2]
        return self.conv(torch.cat([nn.functional.pad(inputs1, 2 * [offset // 2, offset // 2]), outputs2], 1))


def _folded(seq, cin_pad, cout_pad):
    """Conv2d -> BatchNorm2d(eval) of a unetConv2 stage as ONE conv: W' = W * s[co], b' = (b - m) * s + beta, zero-padded to
    (cout_pad, cin_pad) channels (16-channel maps ride the 32-channel K step; the padded channels stay exactly 0 through
    bias-free ReLU, so cosines over the padded maps equal those over the real ones).  Returns a `_Taps` holder with .bias."""
    conv, bn = seq[0], seq[1]
    key = (param_key(conv.weight), param_key(conv.bias), param_key(bn.weight), param_key(bn.bias), bn.running_mean._version,
           bn.running_var._version, cin_pad, cout_pad)
    if getattr(seq, "_e4s_fold", None) is None or seq._e4s_fold[0] != key:
        with torch.no_grad():
            s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w = conv.weight.float() * s[:, None, None, None]
            b = (conv.bias.float() - bn.running_mean.float()) * s + bn.bias.float()
            cout, cin = w.shape[:2]
            w = nn.functional.pad(w, (0, 0, 0, 0, 0, cin_pad - cin, 0, cout_pad - cout)).contiguous()
            t = _Taps(w)
            t.bias = nn.functional.pad(b, (0, cout_pad - cout)).contiguous()
            seq._e4s_fold = (key, t)
    return seq._e4s_fold[1]


class unet(Module):
    """src/criteria/face_parsing/unet.py:6-92 (feature_scale 4, 19 classes, deconv, batchnorm)."""

    def __init__(self, feature_scale=4, n_classes=19, is_deconv=True, in_channels=3, is_batchnorm=True):
        super().__init__()
        assert is_deconv and is_batchnorm and in_channels == 3
        f = [int(x / feature_scale) for x in (64, 128, 256, 512, 1024)]
        self.filters = f
        self.conv1, self.maxpool1 = unetConv2(in_channels, f[0]), MaxPool2d(kernel_size=2)
        self.conv2, self.maxpool2 = unetConv2(f[0], f[1]), MaxPool2d(kernel_size=2)
        self.conv3, self.maxpool3 = unetConv2(f[1], f[2]), MaxPool2d(kernel_size=2)
        self.conv4, self.maxpool4 = unetConv2(f[2], f[3]), MaxPool2d(kernel_size=2)
        self.center = unetConv2(f[3], f[4])
        self.up_concat4, self.up_concat3 = unetUp(f[4], f[3]), unetUp(f[3], f[2])
        self.up_concat2, self.up_concat1 = unetUp(f[2], f[1]), unetUp(f[1], f[0])
        self.final = Conv2d(f[0], n_classes, 1)

    def forward(self, inputs):
        """label logits (unet.py:47-66): the module tree's torch forward -- inference is off the optimisation path."""
        c1 = self.conv1(inputs)
        c2 = self.conv2(self.maxpool1(c1))
        c3 = self.conv3(self.maxpool2(c2))
        c4 = self.conv4(self.maxpool3(c3))
        center = self.center(self.maxpool4(c4))
        up = self.up_concat4(c4, center)
        up = self.up_concat3(c3, up)
        up = self.up_concat2(c2, up)
        return self.final(self.up_concat1(c1, up))

    def _stages(self):
        pad = lambda c: (c + 31) // 32 * 32
        f = self.filters
        blocks = [self.conv1, self.conv2, self.conv3, self.conv4, self.center]
        cins = [3] + [pad(c) for c in f[:4]]
        return [(blk, cins[i], pad(f[i])) for i, blk in enumerate(blocks)]

    def features_nhwc(self, x512, tape=None):
        """x512 NHWC [B,512,512,3] -> the five encoder maps (unet.py:69-91), NHWC, channels zero-padded to 32."""
        feats = []
        x = x512
        for i, (blk, cin, cout) in enumerate(self._stages()):
            t1, t2 = _folded(blk.conv1, cin, cout), _folded(blk.conv2, cout, cout)
            if i == 0:
                a = K.conv_smallcin(x, _smallcin_pack(t1), t1.bias, cout, 3, 1, 1, relu=True)
            else:
                a = _conv3x3(x, t1, cout, bias=t1.bias, act=1, alpha=0.0, gain=1.0)
            f = _conv3x3(a, t2, cout, bias=t2.bias, act=1, alpha=0.0, gain=1.0)
            feats.append(f)
            rec = dict(a=a, f=f, x_shape=tuple(x.shape))
            if i < 4:
                x, rec["idx"] = K.maxpool2(f)
            if tape is not None:
                tape.append(rec)
        return feats

    def backward_nhwc(self, tape, dfeats):
        """dfeats[k] = dL/d(feats[k]) -> dL/d(x512) NHWC."""
        stages = self._stages()
        d = None
        for i in reversed(range(5)):
            blk, cin, cout = stages[i]
            rec = tape[i]
            g = dfeats[i] if d is None else K.add_scale(d, dfeats[i], 1.0)
            t1, t2 = _folded(blk.conv1, cin, cout), _folded(blk.conv2, cout, cout)
            da = _conv3x3(K.relu_bwd(g, rec["f"]), _transposed(t2), cout)
            du = K.relu_bwd(da, rec["a"])
            if i == 0:
                return K.conv_smallcin_bwd(du, _smallcin_pack(t1), rec["x_shape"], 3, 1, 1, cache=_smallcin_cache(t1))
            dp = _conv3x3(du, _transposed(t1), cin)
            d = K.maxpool2_bwd(dp, tape[i - 1]["idx"], tuple(tape[i - 1]["f"].shape))

    @_plain_split
    def extract_feats(self, inputs):
        """unet.py:69-91: l2-normalised rows of the five encoder maps (NCHW flatten order), no gradient."""
        with torch.no_grad():
            feats = self.features_nhwc(K.nchw_to_nhwc(inputs))
            rows = [K.nhwc_to_nchw(f)[:, :c].reshape(f.shape[0], -1) for f, c in zip(feats, self.filters)]
            return [r / torch.norm(r, 2, 1, True) for r in rows]


class _ParsingLossFn(torch.autograd.Function):
    @staticmethod
    @_plain_split
    def forward(ctx, y_hat, mod, y_feats):
        tape = []
        x512 = mod._prep(y_hat)
        feats = mod.G.features_nhwc(x512, tape)
        coefs = [K.cosine(f, t) for f, t in zip(feats, y_feats)]
        sims = torch.stack([c[:, 0] for c in coefs])
        ctx.mod, ctx.tape, ctx.feats, ctx.y_feats, ctx.coefs, ctx.n = mod, tape, feats, y_feats, coefs, y_hat.shape[0]
        ctx.in_shape = tuple(y_hat.shape)
        ctx.mark_non_differentiable(sims)
        return (1.0 - sims).mean(1).sum(), sims

    @staticmethod
    @once_differentiable
    @_plain_split
    def backward(ctx, gloss, _gsims):
        mod = ctx.mod
        if ctx.tape is None:
            raise RuntimeError("FaceParsingLoss: the activation tape was released by the first backward; run the forward again "
                               "(retain_graph / double backward through this node are not supported)")
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dfeats = [K.cosine_bwd(f, t, c, g, -1.0 / ctx.n) for f, t, c in zip(ctx.feats, ctx.y_feats, ctx.coefs)]
        dx512 = mod.G.backward_nhwc(ctx.tape, dfeats)
        ctx.tape = None
        return K.adaptive_pool_bwd(dx512, ctx.in_shape), None, None


class FaceParsingLoss(Module):
    """src/criteria/face_parsing/face_parsing_loss.py:20-78: cosine distance between the parsing UNet's five encoder maps of
    y_hat and y (pooled to 512^2).  `opts.face_parsing_model_path` is loaded when it names an existing file."""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.face_pool = torch.nn.AdaptiveAvgPool2d((512, 512))
        self.G = unet()
        path = getattr(opts, "face_parsing_model_path", None)
        if _have_weights("FaceParsingLoss (opts.face_parsing_model_path)", path):
            self.G.load_state_dict(torch.load(path, map_location="cpu"))
        self.G.eval()
        self.set_requires_grad(False)
        self._target = None

    def set_requires_grad(self, flag=True):
        for p in self.parameters():
            p.requires_grad = flag

    def _prep(self, x):
        return K.adaptive_pool(x.detach(), (512, 512))          # NCHW -> NHWC [B,512,512,3] (identity bins at 512^2)

    @_plain_split
    def extract_feats(self, x):
        with torch.no_grad():
            feats = self.G.features_nhwc(self._prep(x))
            rows = [K.nhwc_to_nchw(f)[:, :c].reshape(f.shape[0], -1) for f, c in zip(feats, self.G.filters)]
            return [r / torch.norm(r, 2, 1, True) for r in rows]

    def inference(self, x):
        raise NotImplementedError("label-map inference (face_parsing_loss.py:37-45: numpy / colour-map post-processing) is "
                                  "off the optimisation path; call self.G(x) for the logits")

    def _target_feats(self, y):
        key = _target_key(y)
        if self._target is None or self._target[0] != key or self._target[1] is not y:
            with torch.no_grad():
                self._target = (key, y, self.G.features_nhwc(self._prep(y)))
        return self._target[2]

    def forward(self, y_hat, y):
        """-> (loss, sim_improvement) as the reference; sim_improvement is a 0-dim tensor."""
        loss, sims = _ParsingLossFn.apply(y_hat, self, self._target_feats(y))
        return loss, (sims - 1.0).mean(1).sum()
