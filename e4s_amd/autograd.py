"""Autograd around the fused HIP generator and LocalMLPs.

scripts/optimization.py optimises the regional style vectors [1,12,1280] with Adam through
`cal_style_codes` -> `gen_img` with the generator frozen (`train_G=False`, networks.py:63-66).  These
Functions provide that gradient -- d(loss)/d(latent) for the generator and d/d(style_vectors) for the MLPs -- and,
for config 5 (`train_G=True`), the gradients of every generator / LocalMLP parameter.  The heavy lifting is native (e4s_conv_bwd_mfma_f32,
e4s_demod_grad_f32, e4s_torgb_bwd_*_f32, e4s_upfirdn2d_f32, e4s_fused_bias_act_f32), including the transposed
contractions of the style prologue's chain rule and of the LocalMLP backward (e4s_grouped_linear_t_f32,
e4s_grouped_outer_f32); every split reduction adds its partial sums in a fixed order, so gradients are bit-reproducible.
Conv weight gradients (config 5) run on the fp32-MFMA weight-gradient kernel (e4s_conv_wgrad_f32); the modulation-weight and
demodulation outer products on e4s_grouped_outer_f32, the fold of the polyphase gradients on e4s_polyphase_fold_f32, the bias / noise-strength
sums on the ordered native column sums -- no library GEMM and no ATen global reduction (whose semaphore memset must not enter a captured step).
"""
import math

import torch
from torch.autograd.function import once_differentiable

from .tape import Tape
from . import kernels as K


def styled_conv_backward(rec, dy, num_regions, extras=None, defer=None):
    """Backward of one fused StyledConv launch.  rec: the forward tape record (layer, x, y, s, d, noise, labels);
    dy: dL/dy NHWC.  Returns (dL/dx NHWC, dL/ds [G,Cin] including the path through the demodulation d(s)).
    defer (a list): the demodulation path and everything behind dL/ds is left to ONE batched launch pair for all layers
    (kernels.style_grad_multi); the pieces are appended as a job and the second return value is None."""
    layer = rec["layer"]
    conv, act = layer.conv, layer.activate
    y, s, d, labels = rec["y"], rec["s"], rec["d"], rec["labels"]
    fused = K.act_bwd_demod(dy, y, rec["noise"], layer.noise.weight, act.bias, act.negative_slope, act.scale, labels, num_regions)
    if fused is not None:                  # gz = lrelu'(y) * sqrt(2) * dy and d * dL/dd (out_pre = d * c) in one pass over dy and y
        gz, dd_d = fused
    else:
        gz = K.fused_bias_act(dy, None, y, 3, 1, act.negative_slope, act.scale)
        dd_d = K.demod_grad(gz, y, rec["noise"], layer.noise.weight, act.bias, act.negative_slope, act.scale, labels, num_regions)
    pk = conv.packed()
    x = rec["x"]
    b, h, w, cin = x.shape
    cout = gz.shape[3]
    if labels is None and not conv.upsample and x.is_contiguous() and K.want_bf16x3(b, h, w, cout, cin):
        # one style per sample: dx = s * conv(gz * d, flipped W^T) is the FORWARD split-bf16 contraction on re-packed weights (the
        # exact-fp32 dx + ds kernel ran the 64->64@512^2 / 32->32@1024^2 dgrads of the optimisation step at 0.69 / 0.71 ms, the forward
        # kernels take ~0.1); ds = sum_p x * (the same contraction) leaves the pass that applies s
        if "wt_fwd" not in pk:
            wp = K.pack_taps_bwd(pk["w"], out=conv._buf("wt_fwd", (1, 9, cin, cout), x.device))           # [1,9,Cin,Cout], taps flipped (one launch)
            pk["wt_fwd"] = (wp, K.split_bf16x2(wp, out=conv._buf("wt_fwd_split", (1, 9, cin, cout), x.device)))
        wp, wps = pk["wt_fwd"]
        if cout == 32 and cin % 32 == 0:
            u = K.conv_c32(gz.contiguous(), wps, cin, in_scale=d)
        else:
            u = K.conv_mfma(gz.contiguous(), wp, cin, w_split=wps, in_scale=d)
        dx, ds = K.scale_dot(u, x, s)
    elif labels is None and conv.upsample and x.is_contiguous() and cout % 8 == 0 and K.want_bf16x3(b, h, w, 4 * cout, cin):
        # unmasked up-conv (model.py:287-300 backward; the 128->64->512^2 and 64->32->1024^2 layers): in polyphase form the four output
        # phases are four 3x3 correlations over the INPUT grid, so with the phases of gz laid side by side in the channel dimension
        # (e4s_pixel_unshuffle2_f32) the dgrad is ONE 'same' 3x3 convolution with 4*Cout input channels on the forward split-bf16 kernel:
        # dx = s * conv(unshuffle(gz) * d, flipped Weff^T), ds from the pass that applies s.  (The exact-fp32 dx + ds kernel ran these two
        # dgrads at ~0.5 / 1.0 ms per image: 36 multiply-adds per input pixel and channel pair on v_mfma_f32_32x32x2_f32.)
        if "wt_up_fwd" not in pk:
            wp = pk["w"].flip(1).permute(1, 3, 0, 2).reshape(1, 9, cin, 4 * cout).contiguous()     # [1,9,Cin,(phase,Cout)]
            keep = conv._buf("wt_up_fwd", tuple(wp.shape), wp.device)
            keep.copy_(wp)
            pk["wt_up_fwd"] = (keep, K.split_bf16x2(keep, out=conv._buf("wt_up_fwd_split", tuple(wp.shape), wp.device)))
        wp, wps = pk["wt_up_fwd"]
        u = K.conv_mfma(K.pixel_unshuffle2(gz), wp, cin, w_split=wps, in_scale=d.repeat(1, 4).contiguous())
        dx, ds = K.scale_dot(u, x, s)
    elif labels is not None and x.is_contiguous() and K.scatter_dgrad_wanted(b, h, w, cout, cin):
        # masked layer (per-pixel region styles, model.py:386-400) in SCATTER form (csrc/dgrad_scatter.hip): rows = source pixels, each
        # with ONE region, so the contraction G[m, t, ci] = sum_co (gz d[r(m)])[m, co] W[t][co][ci] is a plain 1x1 split-bf16 launch
        # with the nine taps stacked in its columns; the region-dependent parts -- u = gz * d[r], dx = sum_t s[r(m_t)] G[m_t, t] and the
        # ordered dL/ds sums -- are two streaming passes.  (The exact-fp32 dx + ds kernel below contracted these layers at ~1/3 of
        # this rate: 6.0 ms of a 16.9 ms optimisation step.)  Polyphase up-convs: one 1x1 launch per output phase.
        ncls = 4 if conv.upsample else 1
        wg, wgs = conv.scatter_taps()
        u = K.region_scale(gz, d, labels, num_regions, ncls)
        G = torch.empty(ncls, b, h, w, 9 * cin, device=x.device, dtype=torch.float32)
        uu = u if ncls == 4 else u.unsqueeze(0)
        for ph in range(ncls):
            K.conv_mfma(uu[ph], wg[ph:ph + 1], 9 * cin, ntaps=1, spatial=False, w_split=wgs[ph:ph + 1], out=G[ph])
        dx, ds = K.col2im_region(G, x, s, labels, num_regions, ncls)
    else:
        dx, ds = K.conv_bwd(gz, conv.bwd_taps(), x, s, d, labels, num_regions, 4 if conv.upsample else 1)
    # d = scale * rsqrt(scale^2 sum_ci s^2 Wsq + eps)  =>  dd/ds_ci = -d^3 s_ci Wsq[co,ci];  dL/dd = dd_d / d, so dL/dd * d^3 = dd_d * d^2
    if defer is not None:
        defer.append(dict(ds_raw=ds, dd_d=dd_d, d=d, s=s, wsq=pk["wsq"], G=ds.shape[0], Cin=cin, Cout=cout))
        if extras is not None:
            extras.update(gz=gz, dd3=dd_d * (d * d))
        return dx, None
    dd3 = dd_d * (d * d)
    # ds - s * (dd3 @ Wsq): the [G,Cout] x [Cout,Cin] contraction on e4s_grouped_linear_t_f32 with the combine fused
    ds = K.grouped_linear_t(dd3.unsqueeze(1), pk["wsq"].unsqueeze(0), -1.0, base=ds.unsqueeze(1),
                            mul=s.unsqueeze(1)).squeeze(1)
    if extras is not None:
        extras.update(gz=gz, dd3=dd3)
    return dx, ds


def styled_conv_weight_grad(rec, extras, num_regions):
    """dL/d(conv.weight) [1,Cout,Cin,3,3] of one fused StyledConv (config 5, train_G=True).
    out_pre = d * sum W s x:  dW = (gz*d)^T (s*x shifted)  [the contraction over all pixels: e4s_conv_wgrad_f32, region
    scales applied to the operands on the fly]  -  W * ((dd*d^3)^T s^2)  [through the demodulation]."""
    layer = rec["layer"]
    conv = layer.conv
    x, s, d, labels = rec["x"], rec["s"], rec["d"], rec["labels"]
    gz, dd3 = extras["gz"], extras["dd3"]
    b, ho, wo, cout = gz.shape
    _, h, w, cin = x.shape
    wraw = conv.weight.detach()[0]                                   # [Cout,Cin,3,3]
    kw = dict(s=s, d=d, labels=labels, num_regions=num_regions)
    if not conv.upsample:
        dw = K.conv_wgrad(gz, x, **kw).permute(1, 2, 0).reshape(cout, cin, 3, 3)
    else:
        # gradient w.r.t. the 4 x 9 polyphase kernels (one e4s_conv_wgrad_f32 call per output phase), folded back onto
        # the 3 x 3 weight with the transpose of the polyphase map
        deff = torch.stack([K.conv_wgrad(gz, x, ostride=2, phase=(ph >> 1, ph & 1), anchors=(h, w), **kw)
                            for ph in range(4)])                              # [4, 9, Cout, Cin]
        dw = K.polyphase_fold(deff, conv.blur.kernel, cout, cin)              # native transpose of the polyphase map (no library GEMM)
    # through the demodulation: - W * ((dd d^3)^T s^2); the [Cout, G] x [G, Cin] contraction on e4s_grouped_outer_f32 (one group, the
    # G rows as its batch): like every reduction of a captured step, native and ordered (kernels.sum_all)
    dw = dw - wraw * K.grouped_outer(dd3.unsqueeze(1).contiguous(), (s * s).unsqueeze(1).contiguous(), 1.0)[0].view(cout, cin, 1, 1)
    return dw.unsqueeze(0)


class GeneratorFn(torch.autograd.Function):
    """Fused generator with gradients w.r.t. the latent and (config 5, train_G=True) w.r.t. the generator's
    trainable parameters, which are passed as extra inputs so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, gen, latent, mask, noise, *params):
        tape = Tape()                                   # (activations stored as bf16 under tape.storage(): configs[4])
        lat = latent.detach().to(torch.float32).contiguous()
        image, feats = gen._fused_forward(lat, mask, noise, tape=tape)
        ctx.gen, ctx.tape, ctx.lat = gen, tape, lat
        ctx.need_lat = latent.requires_grad
        ctx.pidx = {id(p_): i for i, p_ in enumerate(params)}
        ctx.set_materialize_grads(False)
        return image, feats

    @staticmethod
    @once_differentiable
    def backward(ctx, dimage, dfeats):
        gen, tape, lat = ctx.gen, ctx.tape, ctx.lat
        b, r, nl, _ = lat.shape
        if dimage is None:
            dimage = torch.zeros_like(tape[-1]["out"])
        dev = dimage.device
        pgrads = [None] * len(ctx.pidx)
        jobs = []                                       # per layer: the pieces of its dL/ds -> dL/dlatent tail (one batched launch pair below)

        def want(p_):
            return id(p_) in ctx.pidx

        def give(p_, g):
            i = ctx.pidx[id(p_)]
            pgrads[i] = g if pgrads[i] is None else pgrads[i] + g

        dskip = dimage.contiguous().to(torch.float32)
        dact = None                                     # grad w.r.t. the activation feeding the current rgb / next conv

        def style_rows(rec):
            return lat[:, :, rec["idx"]].reshape(b * r, -1) if rec["masked"] else lat[:, 0, rec["idx"]]

        multi = K.STYLE_GRAD_MULTI
        dlat = None if multi else torch.zeros_like(lat)

        def mod_grads(rec, ds_total):
            mod = rec["layer"].conv.modulation
            if want(mod.weight):                                             # s = style @ Wm^T * scale + bias
                give(mod.weight, K.grouped_outer(ds_total.unsqueeze(1).contiguous(), style_rows(rec).unsqueeze(1).contiguous(),
                                                 mod.scale)[0])
            if want(mod.bias):
                give(mod.bias, K.batch_sum(ds_total.contiguous()))

        def add_style_grad(rec, job):
            """dL/ds -> dL/dlatent through the layer's modulation EqualLinear (s = style @ Wm^T * scale + bias): deferred to
            kernels.style_grad_multi, which also finishes dL/ds itself (demodulation path / ToRGB's ws = scale * w3 * s).
            (E4S_STYLE_GRAD_MULTI=0: the per-layer chain of rounds 2-4, kept for A/B runs: `job` is then the finished dL/ds.)"""
            mod = rec["layer"].conv.modulation
            if multi:
                job.update(wmod=mod.weight.detach(), mod_scale=mod.scale, slot=rec["idx"], masked=bool(rec["masked"]), rec=rec)
                jobs.append(job)
                return
            dstyle = K.grouped_linear_t(job.unsqueeze(1).contiguous(), mod.weight.detach().unsqueeze(0), mod.scale).squeeze(1)
            if rec["masked"]:
                dlat[:, :, rec["idx"]] += dstyle.view(b, r, -1)
            else:
                dlat[:, 0, rec["idx"]] += dstyle
            mod_grads(rec, job)

        for rec in reversed(tape):
            layer = rec["layer"]
            if rec["kind"] == "rgb":
                labels = rec["labels"]
                dact, dws = K.torgb_bwd(dskip, rec["x"], rec["ws"], labels, r, dx_acc=dact)
                w3 = layer.conv.weight.detach()[0, :, :, 0, 0]                  # [3,Cin]; dL/ds = scale * sum_c dws[:, c] * w3[c] (batched below)
                if want(layer.conv.weight):                                     # ws = scale * w * s
                    give(layer.conv.weight, (layer.conv.scale * (dws * rec["s"].unsqueeze(1)).sum(0)).view(1, 3, -1, 1, 1))
                if want(layer.bias):
                    give(layer.bias, K.channel_sum(dskip).view(1, 3, 1, 1))
                add_style_grad(rec, dict(dws=dws, w3=w3, conv_scale=layer.conv.scale, G=dws.shape[0], Cin=dws.shape[2]) if multi
                               else layer.conv.scale * (dws * w3.unsqueeze(0)).sum(1))
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
            conv = layer.conv
            train_w = want(conv.weight)
            extras = {} if (train_w or want(layer.noise.weight) or want(layer.activate.bias)) else None
            conv_job = [] if multi else None
            dact, ds = styled_conv_backward(rec, dact, r, extras, defer=conv_job)
            if extras is not None:
                gz = extras["gz"]
                if want(layer.activate.bias):
                    # native ordered sums, not ATen's global reduce (whose semaphore memset must not end up in a captured train step:
                    # kernels.sum_all)
                    give(layer.activate.bias, K.colsum(gz))
                if want(layer.noise.weight):
                    nz = rec["noise"]                                           # [Bn,1,H,W]
                    give(layer.noise.weight, K.sum_all(gz.sum(-1) * nz[:, 0]).view(1))
                if train_w:
                    give(conv.weight, styled_conv_weight_grad(rec, extras, r))
            add_style_grad(rec, conv_job[0] if multi else ds)
        if want(gen.input.input) and dact is not None:
            give(gen.input.input, dact.sum(0, keepdim=True).permute(0, 3, 1, 2).contiguous())
        # every layer's dL/ds (demodulation path included) and dL/dlatent in two launches; the modulation layers' own gradients need dL/ds
        if multi:
            dlat, ds_totals = K.style_grad_multi(jobs, b, r, nl, lat.shape[3], dev)
            for job, ds_total in zip(jobs, ds_totals):
                mod_grads(job["rec"], ds_total)
        return (None, dlat if ctx.need_lat else None, None, None) + tuple(pgrads)


class StyleCodesFn(torch.autograd.Function):
    """cal_style_codes (networks.py:135-158): the two stacked LocalMLP layers on e4s_grouped_linear_f32.
    Backward w.r.t. the style vectors (config 3) and the stacked weights/biases (config 5) on the native transposed /
    outer-product kernels (e4s_grouped_linear_t_f32, e4s_grouped_outer_f32, e4s_batch_sum_f32)."""

    @staticmethod
    def forward(ctx, style_vectors, w0, b0, w2, b2, add):
        sv = style_vectors.detach().to(torch.float32).contiguous()
        w0d, b0d, w2d, b2d = (t_.detach().contiguous() for t_ in (w0, b0, w2, b2))
        h = K.grouped_linear(sv, w0d, b0d, None, 1.0 / math.sqrt(w0.shape[2]), act=1, alpha=0.01)
        codes = K.grouped_linear(h, w2d, b2d, add, 1.0 / math.sqrt(w2.shape[2]))
        ctx.save_for_backward(sv, h, w0d, w2d)
        return codes

    @staticmethod
    @once_differentiable
    def backward(ctx, dcodes):
        sv, h, w0, w2 = ctx.saved_tensors
        s0, s2 = 1.0 / math.sqrt(w0.shape[2]), 1.0 / math.sqrt(w2.shape[2])
        g = dcodes.to(torch.float32).contiguous()                                 # [B,R,O]
        # dh = (g @ W2) * s2 * lrelu'(h): 163 MB of W2 streamed once, the activation gate fused into the second stage
        dh = K.grouped_linear_t(g, w2, s2, ref=h, alpha=0.01)                     # [B,R,512]
        need = ctx.needs_input_grad
        dsv = K.grouped_linear_t(dh, w0, s0) if need[0] else None                 # [B,R,1280]
        dw0 = K.grouped_outer(dh, sv, s0) if need[1] else None                    # [R,512,1280]
        db0 = K.batch_sum(dh) if need[2] else None
        dw2 = K.grouped_outer(g, h, s2) if need[3] else None                      # [R,O,512]
        db2 = K.batch_sum(g) if need[4] else None
        return dsv, dw0, db0, dw2, db2, None
