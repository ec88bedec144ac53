"""Autograd around the fused HIP generator and LocalMLPs.

scripts/optimization.py optimises the regional style vectors [1,12,1280] with Adam through
`cal_style_codes` -> `gen_img` with the generator frozen (`train_G=False`, networks.py:63-66).  These
Functions provide exactly that gradient: d(loss)/d(latent) for the generator and d/d(style_vectors) for the
MLPs.  No weight gradients (config 5) yet.  The heavy lifting is native (e4s_conv_bwd_mfma_f32,
e4s_demod_grad_f32, e4s_torgb_bwd_*_f32, e4s_upfirdn2d_f32, e4s_fused_bias_act_f32); the [G,C]x[C,512]
chain-rule GEMVs of the style prologue and the MLP transposes use torch.matmul (plain library GEMMs).
"""
import math

import torch

from . import kernels as K


def styled_conv_backward(rec, dy, num_regions):
    """Backward of one fused StyledConv launch.  rec: the forward tape record (layer, x, y, s, d, noise, labels);
    dy: dL/dy NHWC.  Returns (dL/dx NHWC, dL/ds [G,Cin] including the path through the demodulation d(s))."""
    layer = rec["layer"]
    conv, act = layer.conv, layer.activate
    y, s, d, labels = rec["y"], rec["s"], rec["d"], rec["labels"]
    gz = K.fused_bias_act(dy, None, y, 3, 1, act.negative_slope, act.scale)     # lrelu'(y) * sqrt(2) * dy
    dd = K.demod_grad(gz, y, rec["noise"], layer.noise.weight, act.bias, act.negative_slope, act.scale, labels,
                      num_regions) / d                                          # dL/dd  (out_pre = d * c)
    pk = conv.packed()
    if "wt" not in pk:
        pk["wt"] = K.pack_taps_bwd(pk["w"])
    dx, ds = K.conv_bwd(gz, pk["wt"], rec["x"], s, d, labels, num_regions, 4 if conv.upsample else 1)
    # d = scale * rsqrt(scale^2 sum_ci s^2 Wsq + eps)  =>  dd/ds_ci = -d^3 s_ci Wsq[co,ci]
    ds = ds - s * ((dd * d * d * d) @ pk["wsq"])
    return dx, ds


class GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gen, latent, mask, noise):
        tape = []
        lat = latent.detach().to(torch.float32).contiguous()
        image, feats = gen._fused_forward(lat, mask, noise, tape=tape)
        ctx.gen, ctx.tape, ctx.lat_shape = gen, tape, tuple(lat.shape)
        ctx.set_materialize_grads(False)
        return image, feats

    @staticmethod
    def backward(ctx, dimage, dfeats):
        gen, tape = ctx.gen, ctx.tape
        b, r, nl, _ = ctx.lat_shape
        if dimage is None:
            dimage = torch.zeros_like(tape[-1]["out"])
        dev = dimage.device
        dlat = torch.zeros(ctx.lat_shape, device=dev, dtype=torch.float32)
        dskip = dimage.contiguous().to(torch.float32)
        dact = None                                     # grad w.r.t. the activation feeding the current rgb / next conv

        def add_style_grad(rec, ds_total):
            mod = rec["layer"].conv.modulation
            dstyle = (ds_total @ mod.weight.detach()) * mod.scale            # [G,512]
            if rec["masked"]:
                dlat[:, :, rec["idx"]] += dstyle.view(b, r, -1)
            else:
                dlat[:, 0, rec["idx"]] += dstyle

        for rec in reversed(tape):
            layer = rec["layer"]
            if rec["kind"] == "rgb":
                labels = rec["labels"]
                dact, dws = K.torgb_bwd(dskip, rec["x"], rec["ws"], labels, r, dx_acc=dact)
                w3 = layer.conv.weight.detach()[0, :, :, 0, 0]                  # [3,Cin]
                ds = layer.conv.scale * (dws * w3.unsqueeze(0)).sum(1)         # [G,Cin]
                add_style_grad(rec, ds)
                if rec["has_skip"]:                     # Upsample backward = FIR-downsample of the incoming grad
                    k4 = layer.upsample.kernel
                    n, c, h, w = dskip.shape
                    g = K.upfirdn2d_raw(dskip.reshape(n * c, h, w, 1), torch.flip(k4, [0, 1]), 1, 1, 2, 2, 1, 1, 1, 1)
                    dskip = g.view(n, c, h // 2, w // 2)
                else:
                    dskip = None
                continue
            # styled conv
            if rec.get("is_feats") and dfeats is not None:
                df = K.nchw_to_nhwc(dfeats.contiguous())
                dact = df if dact is None else dact + df
            dact, ds = styled_conv_backward(rec, dact, r)
            add_style_grad(rec, ds)
        return None, dlat, None, None


class StyleCodesFn(torch.autograd.Function):
    """cal_style_codes (networks.py:135-158) with d/d(style_vectors)."""

    @staticmethod
    def forward(ctx, style_vectors, w0, b0, w2, b2, add):
        sv = style_vectors.detach().to(torch.float32).contiguous()
        h = K.grouped_linear(sv, w0, b0, None, 1.0 / math.sqrt(w0.shape[2]), act=1, alpha=0.01)
        codes = K.grouped_linear(h, w2, b2, add, 1.0 / math.sqrt(w2.shape[2]))
        ctx.save_for_backward(h, w0, w2)
        return codes

    @staticmethod
    def backward(ctx, dcodes):
        h, w0, w2 = ctx.saved_tensors
        g = dcodes.to(torch.float32).transpose(0, 1).contiguous()                 # [R,B,O]
        dh = torch.bmm(g, w2) * (1.0 / math.sqrt(w2.shape[2]))                   # [R,B,512]
        dh = dh * torch.where(h.transpose(0, 1) > 0, 1.0, 0.01)
        dsv = torch.bmm(dh, w0) * (1.0 / math.sqrt(w0.shape[2]))                 # [R,B,1280]
        return dsv.transpose(0, 1).contiguous(), None, None, None, None, None
