"""Autograd of the regional style encoder (config 5, src/training/coach.py:340-356: the encoder and the LocalMLPs are what
E4S trains; SURVEY.md 8(a) a1-a4 backward).  One torch.autograd.Function around FSEncoder_PSP.encode_nhwc: the forward is the
same kernel schedule with a tape, the backward walks the 24 IR-SE units in reverse:

    regional pooling            e4s_region_mean_bwd_f32
    gate * IN(r2) + shortcut     e4s_instnorm_bwd_f32 (its second sum IS dL/dgate), SE chain rule on [B,C] vectors
    conv 3x3 (stride 1 / 2)      dgrad: e4s_conv_bwd_mfma_f32 (stride 2: on the zero-inserted gradient);
    conv 1x1 stride 2            dgrad: e4s_conv_mfma_f32 with the transposed weights + e4s_strided_scatter_f32
    PReLU                        e4s_prelu_bwd_f32
    IN(x)                        e4s_instnorm_bwd_f32
    weight gradients             e4s_conv_wgrad_f32 (fp32 MFMA over the pixels, all 9 taps per block); the 3 -> 64 stem on the
                                 same kernel with the image zero-padded to 32 channels

The exact-zero SE input (the spatial mean of an instance-normalised map, helpers.py:64-66) makes dL/d(pooled) flow back
only through rounding residue; that path (O(1e-8) of the gradient) is dropped."""
import torch
from torch.autograd.function import once_differentiable

from .tape import Tape, pack as pack_record
from . import kernels as K
from .ddp import notify_grad
from .encoders import _pack3x3, _conv3x3, _conv_strided


def _wgrad(gz, xin, stride, ntaps):
    """dW [Cout,Cin,k,k] of y = conv(xin, W, stride, padding=k//2): gz NHWC [B,Ho,Wo,Cout], xin NHWC [B,Hi,Wi,Cin], on the
    fp32-MFMA weight-gradient kernel (e4s_conv_wgrad_f32: the contraction over the pixels, split-K, ordered reduction)."""
    cout, cin = gz.shape[3], xin.shape[3]
    dw = K.conv_wgrad(gz, xin, ntaps=ntaps, istride=stride)                # [ntaps, Cout, Cin]
    k = 3 if ntaps == 9 else 1
    return dw.permute(1, 2, 0).reshape(cout, cin, k, k)


def _wt(conv):
    """backward-packed (flipped, transposed) taps of a 3x3 conv, cached with the forward pack."""
    w = _pack3x3(conv)
    key = conv._e4s_pack[0]
    if getattr(conv, "_e4s_wt", None) is None or conv._e4s_wt[0] != key:
        conv._e4s_wt = (key, K.pack_taps_bwd(w))
    return conv._e4s_wt[1]


def _dgrad3x3(gz, conv, x):
    """dx of y = conv3x3(x, W, stride 1, padding 1): the 'same' 3x3 convolution of gz with the flipped, transposed taps.  Under
    E4S_PRECISION=auto/bf16x3 this is the FORWARD split-bf16 halo kernel on re-packed weights (364 TFLOP/s algorithmic on the
    512-channel layers) -- e4s_conv_bwd_mfma_f32, the exact-fp32 dx + ds kernel the generator needs for its per-region styles,
    ran these plain dgrads at ~66 TFLOP/s: 41 launches x ~190 us = 7.9 ms of a 72 ms G step (profiles/r03_train_kernel_stats.csv).
    f32: that kernel, as before."""
    b, h, w, cy = gz.shape
    cx = conv.weight.shape[1]
    if K.want_bf16x3(b, h, w, cy, cx):
        key = _pack3x3(conv) is not None and conv._e4s_pack[0]
        cached = getattr(conv, "_e4s_wt_fwd", None)
        if cached is None or cached[0] != key:
            wp = K.pack_taps_bwd(_pack3x3(conv))      # [1,9,Cin,Cout], taps flipped: one launch from the forward pack (was flip + transpose copy + pack)
            conv._e4s_wt_fwd = cached = (key, wp, K.split_bf16x2(wp), {})
        if K.wino_eligible(b, h, w, cy, cx):       # Winograd F(2,3) form of the same convolution (csrc/conv_wino.hip): 1.5x fewer MFMAs
            if "u" not in cached[3]:
                cached[3]["u"] = K.wino_weights(cached[1])
            return K.conv_wino(gz.contiguous(), cached[3]["u"], cx)
        return K.conv_mfma(gz.contiguous(), cached[1], cx, w_split=cached[2])
    dx, _ = K.conv_bwd(gz, _wt(conv), x, None, None, None, 1, 1, want_ds=False)       # x: shape only (no ds asked for)
    return dx


def unit_forward(unit, x, tape):
    """bottleneck_IR_SE_Ours.run_nhwc with a tape (the PReLU runs as its own pass so that its input is saved)."""
    conv1, prelu, conv2, se = unit.res_layer[1], unit.res_layer[2], unit.res_layer[3], unit.res_layer[5]
    st_x, _ = K.instnorm_stats(x)
    u1 = _conv3x3(x, conv1, unit.depth, in_stats=st_x)
    r1 = K.prelu(u1, prelu.weight)
    r2 = _conv3x3(r1, conv2, unit.depth) if unit.stride == 1 else _conv_strided(r1, conv2, unit.depth, unit.stride, 9)
    st_r, pooled = K.instnorm_stats(r2, want_pooled=True)
    fc1, fc2 = se.fc1.weight.view(se.fc1.weight.shape[0], -1), se.fc2.weight.view(se.fc2.weight.shape[0], -1)
    gate = K.se_gate(pooled, fc1, fc2)
    rec = dict(unit=unit, x=x, st_x=st_x, u1=u1, r1=r1, r2=r2, st_r=st_r, pooled=pooled, gate=gate)
    if unit.in_channel == unit.depth:
        out = K.instnorm_apply(r2, st_r, gate=gate, res=x, rs=unit.stride)
    else:
        sc = _conv_strided(x, unit.shortcut_layer[0], unit.depth, unit.stride, 1)
        st_sc, _ = K.instnorm_stats(sc)
        out = K.instnorm_apply(r2, st_r, gate=gate, res=sc, res_stats=st_sc)
        rec.update(sc=sc, st_sc=st_sc)
    tape.append(rec)
    return out


def unit_backward(rec, dout, give):
    unit = rec["unit"]
    conv1, prelu, conv2, se = unit.res_layer[1], unit.res_layer[2], unit.res_layer[3], unit.res_layer[5]
    x, r1, r2, gate = rec["x"], rec["r1"], rec["r2"], rec["gate"]
    s = unit.stride
    # ---- gate * IN(r2): dr2 and dL/dgate ----
    dr2, sums = K.instnorm_bwd(dout, r2, rec["st_r"], gate)
    dgate = sums[:, :, 1]
    fc1, fc2 = se.fc1.weight.detach().view(se.fc1.weight.shape[0], -1), se.fc2.weight.detach().view(se.fc2.weight.shape[0], -1)
    # the SE chain rule on [B,C]-sized operands, on the native grouped kernels (ordered sums; no library GEMM inside a captured step:
    # kernels.sum_all)
    pooled = rec["pooled"].contiguous().unsqueeze(1)                          # [B,1,C]
    hidden = K.grouped_linear(pooled, fc1.unsqueeze(0).contiguous(), None, None, 1.0, act=1, alpha=0.0)      # relu(pooled fc1^T) [B,1,Cr]
    dz = (dgate * gate * (1.0 - gate)).unsqueeze(1).contiguous()              # [B,1,C]
    give(se.fc2.weight, K.grouped_outer(dz, hidden, 1.0)[0].view_as(se.fc2.weight))
    dh = K.grouped_linear_t(dz, fc2.unsqueeze(0).contiguous(), 1.0, ref=hidden, alpha=0.0)                   # (dz fc2) * (hidden > 0)
    give(se.fc1.weight, K.grouped_outer(dh, pooled, 1.0)[0].view_as(se.fc1.weight))
    # ---- conv2 (3x3, stride s) ----
    give(conv2.weight, _wgrad(dr2, r1, s, 9))
    gz2 = dr2 if s == 1 else K.strided_scatter(dr2, s)
    dr1 = _dgrad3x3(gz2, conv2, r1)
    # ---- PReLU ----
    du1, dslope = K.prelu_bwd(dr1, rec["u1"], prelu.weight)
    give(prelu.weight, dslope)
    # ---- conv1 on IN(x) ----
    xn = K.instnorm_apply(x, rec["st_x"])
    give(conv1.weight, _wgrad(du1, xn, 1, 9))
    del xn
    dxn = _dgrad3x3(du1, conv1, x)
    dx, _ = K.instnorm_bwd(dxn, x, rec["st_x"])
    # ---- shortcut ----
    if unit.in_channel == unit.depth:
        K.strided_scatter(dout, s, out=dx)                                   # MaxPool2d(1, s) backward (s = 1: plain add)
    else:
        sconv = unit.shortcut_layer[0]
        dsc, _ = K.instnorm_bwd(dout, rec["sc"], rec["st_sc"])
        give(sconv.weight, _wgrad(dsc, x, s, 1))
        wT = sconv.weight.detach().view(unit.depth, unit.in_channel).t().contiguous().view(1, 1, unit.in_channel, unit.depth)
        t = K.conv_mfma(dsc, wT, unit.in_channel, ntaps=1, spatial=False)
        K.strided_scatter(t, s, out=dx)
    return dx


class EncoderFn(torch.autograd.Function):
    """codes [B,R,1280] = FSEncoder_PSP(x256, labels) with gradients w.r.t. every encoder parameter (passed as extra inputs
    so that autograd routes them)."""

    @staticmethod
    def forward(ctx, enc, x256, labels, num_regions, *params):
        tape = Tape()                                   # (activations stored as bf16 under tape.storage(): configs[4])
        conv0, prelu0 = enc.input_layer[0], enc.input_layer[2]
        c0 = K.conv3x3_small(x256, conv0.weight.detach())
        st0, _ = K.instnorm_stats(c0)
        x = K.instnorm_apply(c0, st0, slope=prelu0.weight.detach())
        b = x.shape[0]
        codes = torch.empty(b, num_regions, 256 + 512 + 512, device=x.device, dtype=torch.float32)
        off = {6: 0, 20: 256, 23: 768}
        shapes = {}
        for i, unit in enumerate(enc.body):
            x = unit_forward(unit, x, tape)
            if i in off:
                K.region_mean_into(x, labels, codes, num_regions, off[i])
                shapes[i] = tuple(x.shape)
        ctx.enc, ctx.tape, ctx.labels, ctx.R = enc, tape, labels, num_regions
        ctx.stem = pack_record(dict(x256=x256, c0=c0, st0=st0), tape._cache)
        ctx.off, ctx.shapes = off, shapes
        ctx.pidx = {id(p): i for i, p in enumerate(params)}
        ctx.nparams = len(params)
        return codes

    @staticmethod
    @once_differentiable
    def backward(ctx, dcodes):
        enc, tape = ctx.enc, ctx.tape
        if tape is None or any(t is None for t in tape):
            raise RuntimeError("EncoderFn: the activation tape was released by the first backward; run the forward again "
                               "(retain_graph / double backward through this node are not supported)")
        grads = [None] * ctx.nparams

        def give(p, g):
            i = ctx.pidx.get(id(p))
            if i is not None:
                if grads[i] is not None:
                    raise RuntimeError("EncoderFn: every encoder parameter receives exactly one gradient contribution")
                grads[i] = g
                notify_grad(p, g)      # data-parallel runs: the bucket holding p may leave now, under the remaining units
        dcodes = dcodes.contiguous().to(torch.float32)
        dout = None
        for i in reversed(range(len(tape))):
            if i in ctx.off:
                dout = K.region_mean_bwd(dcodes, ctx.labels, ctx.R, ctx.shapes[i], ctx.off[i], dfeat_acc=dout)
            if dout is None:
                continue
            dout = unit_backward(tape[i], dout, give)
            tape[i] = None                                                  # free the unit's activations
        # ---- stem: PReLU(IN(conv3x3(img))) ----
        conv0, prelu0 = enc.input_layer[0], enc.input_layer[2]
        st = ctx.stem
        n0 = K.instnorm_apply(st["c0"], st["st0"])
        dn0, dslope0 = K.prelu_bwd(dout, n0, prelu0.weight)
        give(prelu0.weight, dslope0)
        dc0, _ = K.instnorm_bwd(dn0, st["c0"], st["st0"])
        if id(conv0.weight) in ctx.pidx:
            # 3-channel input: the same fp32-MFMA weight-gradient kernel as every other conv, on the image zero-padded to the
            # kernel's 32-channel K step (8 MB per sample at 256^2; the padded channels' gradients are dropped)
            x256 = st["x256"]
            b, h, w, c = x256.shape
            xpad = torch.zeros(b, h, w, 32, device=x256.device, dtype=torch.float32)
            xpad[..., :c] = x256
            dw0 = K.conv_wgrad(dc0, xpad, ntaps=9, istride=1)                 # [9, 64, 32]
            give(conv0.weight, dw0[:, :, :c].permute(1, 2, 0).reshape(conv0.weight.shape).contiguous())
        return (None, None, None, None) + tuple(grads)
