"""CPU oracle for the E4S hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 *restatement* of the reference algorithm
(e4s2022/e4s) for the path named in BASELINE.json: the Net3 regional style
encoder, the 12 LocalMLPs and the mask-guided StyleGAN2 generator.  It is
written functionally over a ``state_dict`` (no nn.Module tree) and keeps the
reference's *redundant* formulation on purpose (12 full region passes, blur as
a separate FIR, materialised modulated weights) so that it is an independent
check of the fused HIP kernels.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package ``e4s_amd`` never
does, and has no CPU fallback.

Parity pin: ``tests/golden/*.pt`` were produced by running the REAL reference
modules (imported from /root/reference with the op shim of SURVEY.md 8(c)) on
the seeded synthetic weights of ``e4s_amd.synth``; ``tests/test_oracle_golden.py``
checks this restatement against them.  The script that made them is
``tests/golden/make_golden.py``.

Each function cites the reference file:line (relative to the reference root)
it restates.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# L0/L1 operators
# --------------------------------------------------------------------------
def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Zero-insert upsample, pad/crop, true 2-D convolution with ``kernel``, decimate.

    Restates src/models/stylegan2/op/upfirdn2d_kernel.cu:52-137 (kernel flipped
    on load, :77; out size :167-168) as executed by the pure-PyTorch fallback
    src/pretrained/gpen/face_model/op/upfirdn2d.py:159-193.
    x: [N,C,H,W]; kernel: [kh,kw]; same pad on both axes (pad0 before, pad1 after).
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    kernel = kernel.to(x.dtype)
    p0, p1 = pad
    y = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = y.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = y
        y = z
    y = F.pad(y, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    y = y[:, :, max(-p0, 0): y.shape[2] - max(-p1, 0), max(-p0, 0): y.shape[3] - max(-p1, 0)]
    y = F.conv2d(y, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw))
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """scale * lrelu(x + b_c).  src/models/stylegan2/op/fused_bias_act_kernel.cu:19-49
    (act=3, grad=0), fallback form src/pretrained/gpen/face_model/op/fused_act.py:92-96."""
    shape = (1, -1) + (1,) * (x.ndim - 2)
    return scale * F.leaky_relu(x + bias.view(shape), negative_slope)


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """The native op itself, all six (act, grad) modes.
    src/models/stylegan2/op/fused_bias_act_kernel.cu:27-47."""
    v = x
    if bias is not None and bias.numel() > 0:
        shape = (1, -1) + (1,) * (x.ndim - 2)
        v = v + bias.view(shape)
    code = act * 10 + grad
    if code in (12, 32):
        y = torch.zeros_like(v)
    elif code == 30:
        y = torch.where(v > 0, v, v * alpha)
    elif code == 31:
        y = torch.where(ref > 0, v, v * alpha)
    else:  # 10, 11, default
        y = v
    return y * scale


def make_blur_kernel(k=(1, 3, 3, 1)):
    """src/models/stylegan2/model.py:22-31."""
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


def equal_linear(x, weight, bias, lr_mul=1.0):
    """src/models/stylegan2/model.py:135-169 (activation=None branch)."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    return F.linear(x, weight * scale, None if bias is None else bias * lr_mul)


# --------------------------------------------------------------------------
# Generator
# --------------------------------------------------------------------------
def modulated_conv2d(x, style, weight, mod_w, mod_b, demodulate=True, upsample=False):
    """One region pass.  src/models/stylegan2/model.py:276-320 (fused branch).
    x [B,Cin,H,W]; style [B,512]; weight [1,Cout,Cin,k,k]."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_w, mod_b).view(b, 1, cin, 1, 1)        # :276, bias_init=1 :231
    wgt = (1.0 / math.sqrt(cin * k * k)) * weight * s                  # :277
    if demodulate:
        d = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)              # :279-281
        wgt = wgt * d.view(b, cout, 1, 1, 1)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(b * cin, cout, k, k)          # :289-295
        y = F.conv_transpose2d(x.reshape(1, b * cin, h, w), wt, padding=0, stride=2, groups=b)
        y = y.view(b, cout, y.shape[2], y.shape[3])
        y = upfirdn2d(y, make_blur_kernel() * 4.0, pad=(1, 1))         # Blur :206-213,300
    else:
        y = F.conv2d(x.reshape(1, b * cin, h, w), wgt.view(b * cout, cin, k, k),
                     padding=k // 2, groups=b)                         # :312-318
        y = y.view(b, cout, h, w)
    return y


def _region_compose(conv_fn, style, mask, out_hw):
    """Mask-guided injection: sum_r conv(x, style[:, r]) * nearest(mask)[:, r].
    src/models/stylegan2/model.py:386-400 / :426-439."""
    seg = F.interpolate(mask, size=out_hw, mode="nearest")
    acc = None
    for r in range(style.shape[1]):
        y = conv_fn(style[:, r]) * seg[:, r].unsqueeze(1)
        acc = y if acc is None else acc + y
    return acc


def styled_conv(sd, pfx, x, style, mask, noise, upsample, masked):
    """src/models/stylegan2/model.py:382-406."""
    w, mw, mb = sd[pfx + "conv.weight"], sd[pfx + "conv.modulation.weight"], sd[pfx + "conv.modulation.bias"]
    fn = lambda st: modulated_conv2d(x, st, w, mw, mb, True, upsample)
    if masked:
        h, wd = x.shape[2:]
        out = _region_compose(fn, style, mask, (h * 2, wd * 2) if upsample else (h, wd))
    else:
        out = fn(style)
    out = out + sd[pfx + "noise.weight"] * noise                       # NoiseInjection :329-335
    return fused_leaky_relu(out, sd[pfx + "activate.bias"])            # :404


def to_rgb(sd, pfx, x, style, mask, skip, masked):
    """src/models/stylegan2/model.py:422-448."""
    w, mw, mb = sd[pfx + "conv.weight"], sd[pfx + "conv.modulation.weight"], sd[pfx + "conv.modulation.bias"]
    fn = lambda st: modulated_conv2d(x, st, w, mw, mb, False, False)
    out = _region_compose(fn, style, mask, x.shape[2:]) if masked else fn(style)
    out = out + sd[pfx + "bias"]
    if skip is not None:
        out = out + upfirdn2d(skip, make_blur_kernel() * 4.0, up=2, pad=(2, 1))   # Upsample :34-53
    return out


def generator_forward(sd, latent, mask, noise, size, remaining_layer_idx, pfx="G."):
    """Generator.forward with input_is_latent=True, one 4-D latent.
    src/models/stylegan2/model.py:576-667.  latent [B,R,n_latent,512]; mask one-hot
    [B,R,Hm,Wm]; noise: list of (2*log2(size)-3) tensors broadcastable to [B,1,H,W].
    Returns (image [B,3,size,size], feats at 16x16 after convs[2])."""
    K = remaining_layer_idx
    log_size = int(math.log2(size))
    b = latent.shape[0]
    x = sd[pfx + "input.input"].repeat(b, 1, 1, 1)
    x = styled_conv(sd, pfx + "conv1.", x, latent[:, :, 0], mask, noise[0], False, True)   # :529 mask_op=True
    skip = to_rgb(sd, pfx + "to_rgb1.", x, latent[:, :, 1], mask, None, True)
    feats = None
    i = 1
    for j, res_log in enumerate(range(3, log_size + 1)):
        conv_masked = not (res_log > 2 + K // 2)                       # :537,545
        rgb_masked = not (K != 17 and res_log >= 2 + K // 2)           # :553
        c1, c2, tr = f"{pfx}convs.{2 * j}.", f"{pfx}convs.{2 * j + 1}.", f"{pfx}to_rgbs.{j}."
        if i < K:                                                      # :639
            x = styled_conv(sd, c1, x, latent[:, :, i], mask, noise[i], True, conv_masked)
            if i + 2 == 5:                                             # split_layer_idx=5, networks.py:52
                feats = x
            x = styled_conv(sd, c2, x, latent[:, :, i + 1], mask, noise[i + 1], False, conv_masked)
            if K == 17 or i + 2 != K:
                skip = to_rgb(sd, tr, x, latent[:, :, i + 2], mask, skip, rgb_masked)
            else:
                skip = to_rgb(sd, tr, x, latent[:, 0, i + 2], mask, skip, rgb_masked)    # :650-653
        else:
            x = styled_conv(sd, c1, x, latent[:, 0, i], mask, noise[i], True, conv_masked)
            x = styled_conv(sd, c2, x, latent[:, 0, i + 1], mask, noise[i + 1], False, conv_masked)
            skip = to_rgb(sd, tr, x, latent[:, 0, i + 2], mask, skip, rgb_masked)
        i += 2
    return skip, feats


# --------------------------------------------------------------------------
# Discriminator (config 5 only; second consumer of upfirdn2d mode 1 / fused_bias_act)
# --------------------------------------------------------------------------
def _d_conv_layer(sd, pfx, x, kernel_size, downsample, bias=True, activate=True):
    """ConvLayer.  src/models/stylegan2/model.py:670-716: [Blur pad ((p+1)//2, p//2), p = 2 + (k-1)] ->
    EqualConv2d(stride 2, padding 0 | stride 1, padding k//2; scale 1/sqrt(Cin*k*k), :97-132) ->
    FusedLeakyReLU (bias) | ScaledLeakyReLU(0.2) (no bias)."""
    i = 0
    if downsample:
        p = 2 + (kernel_size - 1)
        x = upfirdn2d(x, sd[pfx + "0.kernel"], pad=((p + 1) // 2, p // 2))
        i = 1
    w = sd[pfx + f"{i}.weight"]
    scale = 1.0 / math.sqrt(w.shape[1] * kernel_size ** 2)
    cb = sd.get(pfx + f"{i}.bias") if (bias and not activate) else None
    x = F.conv2d(x, w * scale, cb, stride=2 if downsample else 1, padding=0 if downsample else kernel_size // 2)
    if activate:
        if bias:
            x = fused_leaky_relu(x, sd[pfx + f"{i + 1}.bias"])
        else:
            x = F.leaky_relu(x, 0.2) * math.sqrt(2)
    return x


def discriminator_forward(sd, x, size, pfx=""):
    """Discriminator.forward.  src/models/stylegan2/model.py:740-799.  x [B,3,size,size] -> logits [B,1]."""
    log_size = int(math.log2(size))
    out = _d_conv_layer(sd, pfx + "convs.0.", x, 1, False)
    for j in range(1, log_size - 1):                                 # ResBlock, :719-737
        c = f"{pfx}convs.{j}."
        r = _d_conv_layer(sd, c + "conv1.", out, 3, False)
        r = _d_conv_layer(sd, c + "conv2.", r, 3, True)
        sk = _d_conv_layer(sd, c + "skip.", out, 1, True, bias=False, activate=False)
        out = (r + sk) / math.sqrt(2)
    b, ch, h, w = out.shape                                          # minibatch stddev, :783-790
    group = min(b, 4)
    std = out.view(group, -1, 1, ch, h, w)
    std = torch.sqrt(std.var(0, unbiased=False) + 1e-8)
    std = std.mean([2, 3, 4], keepdim=True).squeeze(2)
    out = torch.cat([out, std.repeat(group, 1, h, w)], 1)
    out = _d_conv_layer(sd, pfx + "final_conv.", out, 3, False)
    out = out.reshape(b, -1)
    out = fused_leaky_relu(equal_linear(out, sd[pfx + "final_linear.0.weight"], None),
                           sd[pfx + "final_linear.0.bias"])         # EqualLinear(activation="fused_lrelu"), :159-164
    return equal_linear(out, sd[pfx + "final_linear.1.weight"], sd[pfx + "final_linear.1.bias"])


# --------------------------------------------------------------------------
# GPEN FullGenerator (the other consumer of modulated conv / upfirdn2d / fused_bias_act)
# --------------------------------------------------------------------------
def gpen_full_generator(sd, x, size, n_mlp=8):
    """FullGenerator.forward(inputs) with isconcat=True.  src/pretrained/gpen/face_model/gpen_model.py:671-690
    (encoder + latent), :488-555 (decoder), :318-357 (StyledConv: conv -> cat(out, w*noise) -> FusedLeakyReLU over
    2C channels), :359-377 (ToRGB).  x [B,3,size,size] -> (image [B,3,size,size], style latent w [B,style_dim]).
    The "noise" of every decoder layer is the encoder feature map of the same resolution (:684-686)."""
    log_size = int(math.log2(size))
    feats = []
    h = _d_conv_layer(sd, "ecd0.0.", x, 1, False)                       # ConvLayer == the Discriminator's, :558-606
    feats.append(h)
    for j in range(1, log_size - 1):
        h = _d_conv_layer(sd, f"ecd{j}.0.", h, 3, True)
        feats.append(h)
    z = fused_leaky_relu(equal_linear(h.reshape(h.shape[0], -1), sd["final_linear.0.weight"], None),
                         sd["final_linear.0.bias"])
    noise = [f for f in feats for _ in range(2)][::-1][1:]              # :686-687
    g = "generator."
    w = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)     # PixelNorm, :18-23
    for i in range(n_mlp):                                              # EqualLinear(lr_mul=0.01, fused_lrelu), :139-172
        w = fused_leaky_relu(equal_linear(w, sd[f"{g}style.{i + 1}.weight"], None, lr_mul=0.01),
                             sd[f"{g}style.{i + 1}.bias"] * 0.01)

    def styled(pfx, inp, nz, up):
        out = modulated_conv2d(inp, w, sd[pfx + "conv.weight"], sd[pfx + "conv.modulation.weight"],
                               sd[pfx + "conv.modulation.bias"], True, up)
        out = torch.cat((out, sd[pfx + "noise.weight"] * nz), dim=1)
        return fused_leaky_relu(out, sd[pfx + "activate.bias"])

    def torgb(pfx, inp, skip):
        out = modulated_conv2d(inp, w, sd[pfx + "conv.weight"], sd[pfx + "conv.modulation.weight"],
                               sd[pfx + "conv.modulation.bias"], False, False) + sd[pfx + "bias"]
        if skip is not None:
            out = out + upfirdn2d(skip, sd[pfx + "upsample.kernel"], up=2, pad=(2, 1))
        return out
    out = sd[g + "input.input"].repeat(x.shape[0], 1, 1, 1)
    out = styled(g + "conv1.", out, noise[0], False)
    skip = torgb(g + "to_rgb1.", out, None)
    for j in range(log_size - 2):
        out = styled(f"{g}convs.{2 * j}.", out, noise[1 + 2 * j], True)
        out = styled(f"{g}convs.{2 * j + 1}.", out, noise[2 + 2 * j], False)
        skip = torgb(f"{g}to_rgbs.{j}.", out, skip)
    return skip, w


# --------------------------------------------------------------------------
# Regional style encoder
# --------------------------------------------------------------------------
ENC_BLOCKS = ((64, 128, 3), (128, 256, 4), (256, 512, 14), (512, 512, 3))    # psp_encoders.py:242-247


def encoder_unit_plan():
    """(in_channel, depth, stride) of the 24 units. helpers.py:25-26."""
    units = []
    for cin, depth, n in ENC_BLOCKS:
        units.append((cin, depth, 2))
        units += [(depth, depth, 1)] * (n - 1)
    return units


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)                                # InstanceNorm2d defaults


def encoder_unit(sd, pfx, x, cin, depth, stride):
    """bottleneck_IR_SE_Ours.  src/models/encoders/helpers.py:122-144, SE :56-72."""
    if cin == depth:
        sc = x[:, :, ::stride, ::stride]                               # MaxPool2d(1, stride)
    else:
        sc = _inorm(F.conv2d(x, sd[pfx + "shortcut_layer.0.weight"], stride=stride))
    r = _inorm(x)
    r = F.conv2d(r, sd[pfx + "res_layer.1.weight"], padding=1)
    r = F.prelu(r, sd[pfx + "res_layer.2.weight"])
    r = F.conv2d(r, sd[pfx + "res_layer.3.weight"], stride=stride, padding=1)
    r = _inorm(r)
    se = r.mean((2, 3), keepdim=True)
    se = F.relu(F.conv2d(se, sd[pfx + "res_layer.5.fc1.weight"]))
    se = torch.sigmoid(F.conv2d(se, sd[pfx + "res_layer.5.fc2.weight"]))
    return r * se + sc


def region_mean(feats, mask):
    """get_per_comp_styleCode.  src/models/encoders/psp_encoders.py:264-283.
    Exact zeros for empty regions."""
    seg = F.interpolate(mask, size=feats.shape[2:], mode="nearest").bool()
    b, c = feats.shape[:2]
    out = feats.new_zeros(b, seg.shape[1], c)
    for i in range(b):
        for j in range(seg.shape[1]):
            area = int(seg[i, j].sum())
            if area > 0:
                out[i, j] = feats[i].masked_select(seg[i, j]).reshape(c, area).mean(1)
    return out


def encoder_forward(sd, x256, mask, pfx="encoder."):
    """FSEncoder_PSP.forward.  src/models/encoders/psp_encoders.py:285-309."""
    x = F.conv2d(x256, sd[pfx + "input_layer.0.weight"], padding=1)
    x = F.prelu(_inorm(x), sd[pfx + "input_layer.2.weight"])
    taps = {}
    for i, (cin, depth, stride) in enumerate(encoder_unit_plan()):
        x = encoder_unit(sd, f"{pfx}body.{i}.", x, cin, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    codes = torch.cat([region_mean(taps[6], mask), region_mean(taps[20], mask),
                       region_mean(taps[23], mask)], dim=2)
    return codes, torch.zeros_like(x)


# --------------------------------------------------------------------------
# Net3 API
# --------------------------------------------------------------------------
def get_style_vectors(sd, img, mask):
    """Net3.get_style_vectors.  src/models/networks.py:121-133."""
    return encoder_forward(sd, F.interpolate(img, (256, 256), mode="bilinear"), mask)


def cal_style_codes(sd, style_vectors, latent_avg, remaining_layer_idx):
    """Net3.cal_style_codes, start_from_latent_avg=True, learn_in_w=False.
    src/models/networks.py:135-158; LocalMLP :15-39 (nn.LeakyReLU default slope 0.01)."""
    K = remaining_layer_idx
    nw = K if K != 17 else 18
    b, r, _ = style_vectors.shape
    codes = []
    for i in range(r):
        h = equal_linear(style_vectors[:, i], sd[f"MLPs.{i}.mlp.0.weight"], sd[f"MLPs.{i}.mlp.0.bias"])
        h = F.leaky_relu(h, 0.01)
        h = equal_linear(h, sd[f"MLPs.{i}.mlp.2.weight"], sd[f"MLPs.{i}.mlp.2.bias"])
        codes.append(h.view(b, nw, 512))
    codes = torch.stack(codes, dim=1)
    if K != 17:
        codes = codes + latent_avg[:K].view(1, 1, K, 512)
        rest = latent_avg[K:].view(1, 1, -1, 512).expand(b, r, -1, -1)
        return torch.cat([codes, rest], dim=2)
    return codes + latent_avg.view(1, 1, -1, 512)


def gen_img(sd, style_codes, mask, noise, size, remaining_layer_idx):
    """Net3.gen_img.  src/models/networks.py:160-182."""
    return generator_forward(sd, style_codes, mask, noise, size, remaining_layer_idx)


def swap_style_vectors(target_sv, driven_sv, num_cls=12):
    """swap_comp_style_vector with the face-swap defaults.
    scripts/face_swap.py:117-146, call site :261-262 (keep 0,4,10,11 from target)."""
    out = target_sv.clone()
    for c in sorted(set(range(num_cls)) - {0, 4, 11, 10}):
        out[:, c] = driven_sv[:, c]
    if torch.sum(driven_sv[:, 7]) == 0:
        out[:, 7] = (target_sv[:, 7] + driven_sv[:, 7]) / 2
    if torch.sum(driven_sv[:, 9]) == 0:
        out[:, 9] = target_sv[:, 9]
    return out


def face_swap_core(sd, driven, driven_mask, target, target_mask, swapped_mask, latent_avg, noise,
                   size, remaining_layer_idx):
    """The E4S-core unit of work: scripts/face_swap.py:237-273."""
    d_sv, _ = get_style_vectors(sd, driven, driven_mask)
    t_sv, _ = get_style_vectors(sd, target, target_mask)
    sv = swap_style_vectors(t_sv, d_sv)
    codes = cal_style_codes(sd, sv, latent_avg, remaining_layer_idx)
    img, _ = gen_img(sd, codes, swapped_mask, noise, size, remaining_layer_idx)
    return img


# ---- pre/post-processing of scripts/face_swap.py, CPU restatements --------------------------------
def morph_flat(x, radius, op):
    """src/utils/morphology.py:23-198 for a flat (2r+1)^2 structuring element with the default geodesic border: samples
    outside the image never win (they are padded with -/+1e4).  x [B,C,H,W]; op 'dilation' (max) | 'erosion' (min)."""
    k = 2 * radius + 1
    if op == "dilation":
        return F.max_pool2d(F.pad(x, (radius,) * 4, value=-1e4), k, stride=1)
    return -F.max_pool2d(F.pad(-x, (radius,) * 4, value=-1e4), k, stride=1)


def create_masks(mask, outer_dilation=0, operation="dilation"):
    """scripts/face_swap.py:30-48."""
    if operation == "dilation":
        full = morph_flat(mask, outer_dilation, "dilation")
        border = full - mask
    elif operation == "erosion":
        full = morph_flat(mask, outer_dilation, "erosion")
        border = mask - full
    else:
        full = morph_flat(mask, outer_dilation, "dilation")
        border = full - morph_flat(mask, outer_dilation, "erosion")
    return mask, border.clip(0, 1), full


def swap_head_mask(source, target):
    """src/utils/swap_face_mask.py:33-82 (hair_first=True) on integer label tensors; returns (swapped, hole 0/255)."""
    res = torch.zeros_like(target)
    res[target == 0] = 99
    for c in (8, 7, 11, 4):
        res[target == c] = c
    for c in (1, 2, 3, 5, 6, 9):
        res[(source == c) & (res != 99)] = c
    res[target == 10] = 10
    hole = (res == 0).to(target.dtype) * 255
    res[res == 0] = 6
    res[res == 99] = 0
    return res, hole


def tensor2im_u8(img):
    """src/utils/torch_utils.py:63-69 without the PIL wrapper: [B,3,H,W] fp32 -> uint8 [B,H,W,3]."""
    v = ((img.permute(0, 2, 3, 1) + 1) / 2).clamp(0, 1) * 255
    return v.to(torch.uint8)


def paste_u8(face_u8, target_u8, content_mask):
    """scripts/face_swap.py:291-292,301-303: bilinear resize of the mask, float32 lerp, np.uint8 truncation."""
    h, w = face_u8.shape[1:3]
    m = F.interpolate(content_mask, (h, w), mode="bilinear", align_corners=False)[:, 0, :, :, None]
    return (face_u8.float() * m + target_u8.float() * (1 - m)).to(torch.uint8)


# ---- stitching (scripts/face_swap.py:81-97, 276-310; src/utils/multi_band_blending.py) ---------------------------------------
# OpenCV (opencv-python 4.7.0.72, e4s_env.yaml:96) is a third-party dependency that is NOT in this container: cv2.erode,
# cv2.GaussianBlur (CV_8U fixed-point path), cv2.pyrDown and cv2.pyrUp are restated below from OpenCV 4.x's published
# algorithms (imgproc/src/morph, smooth.dispatch.cpp, pyramids.cpp).  PARITY UNPINNED for these four: no cv2 here to
# generate golden vectors.  PIL IS here: the alpha composite below calls the reference's own PIL code path.
def cv2_erode_u8(mask, radius, border_value=255):
    """cv2.erode(mask, np.ones((2r+1,2r+1)), borderType=BORDER_CONSTANT, borderValue): uint8 [B,H,W] tensor."""
    x = F.pad(mask.float()[:, None], (radius,) * 4, value=float(border_value))
    return (-F.max_pool2d(-x, 2 * radius + 1, stride=1))[:, 0].to(torch.uint8)


def cv2_gaussian_taps_fixed8(ksize, sigma=0.0):
    """getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED(fractionBits = 8) -> integer taps summing to 256."""
    import math
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize <= 7:
        k = small[ksize]
    else:
        sg = sigma if sigma > 0 else 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        k = [math.exp(-(i - (ksize - 1) / 2) ** 2 / (2 * sg * sg)) for i in range(ksize)]
        k = [v / sum(k) for v in k]
    taps, err = [0] * ksize, 0.0
    for i in range(ksize // 2):
        adj = k[i] * 256 + err
        taps[i] = taps[-1 - i] = int(round(adj))
        err = adj - taps[i]
    taps[ksize // 2] = 256 - sum(taps)
    return taps


def cv2_gaussian_blur_u8(img, ksize, sigma=0.0):
    """cv2.GaussianBlur(img, (k,k), sigma) for CV_8U [B,H,W]: exact integer row and column passes with the 8.8 taps,
    BORDER_REFLECT_101, one rounding: (acc + 2^15) >> 16."""
    import numpy as np
    taps = np.array(cv2_gaussian_taps_fixed8(ksize, sigma), dtype=np.int64)
    r = ksize // 2
    x = np.pad(img.numpy().astype(np.int64), ((0, 0), (r, r), (r, r)), mode="reflect")
    h, w = img.shape[1:]
    rows = sum(taps[i] * x[:, :, i:i + w] for i in range(ksize))
    acc = sum(taps[j] * rows[:, j:j + h, :] for j in range(ksize))
    return torch.from_numpy(np.clip((acc + (1 << 15)) >> 16, 0, 255).astype(np.uint8))


def smooth_face_boundry(image_u8, dst_u8, mask_u8, radius=0, sigma=0.0):
    """scripts/face_swap.py:81-97 per sample, composite through PIL itself (as the reference does); uint8 HWC tensors in,
    RGB uint8 [B,H,W,3] of the pasted RGBA image out."""
    import numpy as np
    from PIL import Image
    out = []
    for b in range(image_u8.shape[0]):
        image_masked = Image.fromarray(image_u8[b].numpy()).convert("RGBA")
        pasted = Image.fromarray(dst_u8[b].numpy()).convert("RGBA")
        m = mask_u8[b:b + 1]
        if radius != 0:
            m = cv2_gaussian_blur_u8(cv2_erode_u8(m, radius, 255), 2 * radius + 1, sigma)
        image_masked.putalpha(Image.fromarray(m[0].numpy()))
        pasted.alpha_composite(image_masked)
        out.append(torch.from_numpy(np.array(pasted)[:, :, :3].copy()))
    return torch.stack(out)


def mask_to_u8(mask, size):
    """255 * F.interpolate(mask, size, 'bilinear')[b, 0].numpy().astype(np.uint8) (face_swap.py:291-294): [B,1,h,w] -> [B,H,W]."""
    import numpy as np
    m = F.interpolate(mask, size, mode="bilinear", align_corners=False)[:, 0].numpy()
    return torch.from_numpy((255 * m.astype(np.uint8)).astype(np.uint8))


def _reflect101(i, n):
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * n - 2 - i
    return i


def cv2_pyrdown(x):
    """cv2.pyrDown on a numpy HWC image: uint8 -> (sum + 128) >> 8 of the [1 4 6 4 1]^2 window; float32 -> row pass
    src[2x]*6 + (src[2x-1] + src[2x+1])*4 + src[2x-2] + src[2x+2], column pass row2*6 + (row1 + row3)*4 + row0 + row4, * 1/256;
    BORDER_REFLECT_101; output ((H+1)//2, (W+1)//2)."""
    import numpy as np
    h, w = x.shape[:2]
    ho, wo = (h + 1) // 2, (w + 1) // 2
    xs = [[_reflect101(2 * i + k - 2, w) for i in range(wo)] for k in range(5)]
    ys = [[_reflect101(2 * j + k - 2, h) for j in range(ho)] for k in range(5)]
    if x.dtype == np.uint8:
        v = x.astype(np.int64)
        row = v[:, xs[2]] * 6 + (v[:, xs[1]] + v[:, xs[3]]) * 4 + v[:, xs[0]] + v[:, xs[4]]
        acc = row[ys[2]] * 6 + (row[ys[1]] + row[ys[3]]) * 4 + row[ys[0]] + row[ys[4]]
        return ((acc + 128) >> 8).astype(np.uint8)
    f = np.float32
    v = x.astype(f)
    row = (v[:, xs[2]] * f(6) + (v[:, xs[1]] + v[:, xs[3]]) * f(4)) + v[:, xs[0]] + v[:, xs[4]]
    acc = (row[ys[2]] * f(6) + (row[ys[1]] + row[ys[3]]) * f(4)) + row[ys[0]] + row[ys[4]]
    return (acc * f(1.0 / 256.0)).astype(f)


def cv2_pyrup(x):
    """cv2.pyrUp on a float32 numpy HWC image (pyramids.cpp): columns -- even: s[x-1] + s[x]*6 + s[x+1], odd: (s[x] + s[x+1])*4,
    first column s0*6 + s1*2 / (s0 + s1)*4, last s[w-2] + s[w-1]*7 / s[w-1]*8; rows -- the generic form on rows (y-1, y, y+1) with
    row -1 = row 1 and row h = row h-1; * 1/64."""
    import numpy as np
    f = np.float32
    v = x.astype(f)
    h, w = v.shape[:2]
    row = np.empty((h, 2 * w) + v.shape[2:], dtype=f)
    for xx in range(w):
        if xx == 0:
            s1 = v[:, 1 if w > 1 else 0]
            row[:, 0] = v[:, 0] * f(6) + s1 * f(2)
            row[:, 1] = (v[:, 0] + s1) * f(4)
        elif xx == w - 1:
            row[:, 2 * xx] = v[:, xx - 1] + v[:, xx] * f(7)
            row[:, 2 * xx + 1] = v[:, xx] * f(8)
        else:
            row[:, 2 * xx] = (v[:, xx - 1] + v[:, xx] * f(6)) + v[:, xx + 1]
            row[:, 2 * xx + 1] = (v[:, xx] + v[:, xx + 1]) * f(4)
    out = np.empty((2 * h,) + row.shape[1:], dtype=f)
    for yy in range(h):
        ym, yp = (yy - 1 if yy > 0 else (1 if h > 1 else 0)), (yy + 1 if yy < h - 1 else h - 1)
        out[2 * yy] = ((row[ym] + row[yy] * f(6)) + row[yp]) * f(1.0 / 64.0)
        out[2 * yy + 1] = ((row[yy] + row[yp]) * f(4)) * f(1.0 / 64.0)
    return out


def laplacian_blend_u8(full_img, ori_img, mask, num_levels=10):
    """multi_band_blending.py:4-75 (`blending` for 1024^2 inputs, whose cv2.resize calls are identities): numpy uint8 HWC
    full_img / ori_img, float32 HWC mask -> uint8 HWC."""
    import numpy as np
    GA, GB, GM = full_img.copy(), ori_img.copy(), np.float32(mask)
    gpA, gpB, gpM = [GA], [GB], [GM]
    for _ in range(num_levels):
        GA, GB, GM = cv2_pyrdown(GA), cv2_pyrdown(GB), cv2_pyrdown(GM)
        gpA.append(np.float32(GA)); gpB.append(np.float32(GB)); gpM.append(np.float32(GM))
    lpA, lpB, gpMr = [gpA[num_levels - 1]], [gpB[num_levels - 1]], [gpM[num_levels - 1]]
    for i in range(num_levels - 1, 0, -1):
        lpA.append(np.subtract(gpA[i - 1], cv2_pyrup(gpA[i])))
        lpB.append(np.subtract(gpB[i - 1], cv2_pyrup(gpB[i])))
        gpMr.append(gpM[i - 1])
    LS = [la * gm + lb * (np.float32(1.0) - gm) for la, lb, gm in zip(lpA, lpB, gpMr)]
    ls_ = LS[0]
    for i in range(1, num_levels):
        ls_ = cv2_pyrup(ls_) + LS[i]
    return np.uint8(np.clip(ls_, 0, 255))


def stitch(swapped_face, target_u8, swapped_labels, hole, lap_bld=False, outer_dilation=5):
    """scripts/face_swap.py:276-310 for a batch (tensors in, uint8 [B,H,W,3] out); swapped_labels / hole uint8 [B,512,512]."""
    import numpy as np
    b, _, h, w = swapped_face.shape
    face = tensor2im_u8(swapped_face)
    lab = swapped_labels.long()
    bg = (lab == 0) | (lab == 11) | (lab == 4)
    fg = (~bg) | (hole == 255)
    fg = fg.float()[:, None]
    content, border, full = create_masks(fg, outer_dilation, "expansion" if lap_bld else "dilation")
    if lap_bld:
        pasted = paste_u8(face, target_u8, content)
        bm = F.interpolate(border, (h, w), mode="bilinear", align_corners=False)[:, 0, :, :, None].numpy()
        bm = np.repeat(bm, 3, axis=-1)
        return torch.stack([torch.from_numpy(laplacian_blend_u8(target_u8[i].numpy(), pasted[i].numpy(), bm[i])) for i in range(b)])
    mask_img = mask_to_u8(content if outer_dilation == 0 else full, (h, w))
    return smooth_face_boundry(face, target_u8, mask_img, radius=outer_dilation)


# --------------------------------------------------------------------------
# N3: loss networks of the optimisation loop (scripts/optimization.py:88-122)
# --------------------------------------------------------------------------
def _bn_eval(sd, pfx, x, eps=1e-5):
    """nn.BatchNorm2d / BatchNorm1d in eval mode (running statistics)."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    rm, rv = sd[pfx + "running_mean"].to(x.dtype), sd[pfx + "running_var"].to(x.dtype)
    y = (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + eps)
    if pfx + "weight" in sd:
        y = y * sd[pfx + "weight"].to(x.dtype).view(shape) + sd[pfx + "bias"].to(x.dtype).view(shape)
    return y


def irse50_unit(sd, pfx, x, cin, depth, stride):
    """bottleneck_IR_SE, src/models/encoders/helpers.py:97-119 (SEModule :56-72)."""
    w = lambda k: sd[pfx + k].to(x.dtype)
    if cin == depth:
        sc = x[:, :, ::stride, ::stride]                                   # MaxPool2d(1, stride)
    else:
        sc = _bn_eval(sd, pfx + "shortcut_layer.1.", F.conv2d(x, w("shortcut_layer.0.weight"), stride=stride))
    r = _bn_eval(sd, pfx + "res_layer.0.", x)
    r = F.conv2d(r, w("res_layer.1.weight"), padding=1)
    r = F.prelu(r, w("res_layer.2.weight"))
    r = F.conv2d(r, w("res_layer.3.weight"), stride=stride, padding=1)
    r = _bn_eval(sd, pfx + "res_layer.4.", r)
    g = r.mean((2, 3), keepdim=True)
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, w("res_layer.5.fc1.weight"))), w("res_layer.5.fc2.weight")))
    return r * g + sc


def irse50_plan():
    """get_blocks(50), helpers.py:30-53: (in_channel, depth, stride) of the 24 units."""
    plan = []
    for cin, depth, n in ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3)):
        plan += [(cin, depth, 2)] + [(depth, depth, 1)] * (n - 1)
    return plan


def irse50_features(sd, x112, multi_scale=True, pfx="facenet."):
    """Backbone.forward, src/models/encoders/model_irse.py:44-69 (eval: Dropout is the identity): l2-normalised rows."""
    x = F.conv2d(x112, sd[pfx + "input_layer.0.weight"].to(x112.dtype), padding=1)
    x = F.prelu(_bn_eval(sd, pfx + "input_layer.1.", x), sd[pfx + "input_layer.2.weight"].to(x112.dtype))
    taps = []
    for i, (cin, depth, stride) in enumerate(irse50_plan()):
        x = irse50_unit(sd, f"{pfx}body.{i}.", x, cin, depth, stride)
        if i in (2, 6, 20, 23):
            taps.append(x.reshape(x.shape[0], -1))
    x = _bn_eval(sd, pfx + "output_layer.0.", x).reshape(x.shape[0], -1)
    x = F.linear(x, sd[pfx + "output_layer.3.weight"].to(x.dtype), sd[pfx + "output_layer.3.bias"].to(x.dtype))
    x = _bn_eval(sd, pfx + "output_layer.4.", x)
    rows = (taps if multi_scale else []) + [x]
    return [r / torch.norm(r, 2, 1, True) for r in rows]                  # l2_norm, helpers.py:14-17


def id_loss(sd, y_hat, y, multi_scale=True):
    """IDLoss.forward, src/criteria/id_loss.py:24-57 -> (loss, sim_improvement)."""
    def extract(x):
        if x.shape[2] != 256:
            x = F.adaptive_avg_pool2d(x, (256, 256))
        x = F.adaptive_avg_pool2d(x[:, :, 35:223, 32:220], (112, 112))
        return irse50_features(sd, x, multi_scale)
    fy = [f.detach() for f in extract(y)]
    fh = extract(y_hat)
    loss, imp = 0.0, 0.0
    for a, b in zip(fh, fy):
        sim = (a * b).sum(1)
        loss = loss + (1 - sim).mean()
        imp = imp + (sim - (b * b).sum(1)).mean()
    return loss, imp


def alexnet_features(sd, x, pfx="net."):
    """BaseNet.forward over AlexNet, src/criteria/lpips/networks.py:50-83: z-score, torchvision alexnet.features (conv
    11/4/2 -> ReLU -> maxpool 3/2 -> conv 5/1/2 -> ReLU -> maxpool -> 3 x (conv 3/1/1 -> ReLU)), outputs after modules
    2, 5, 8, 10, 12 (1-based), each unit-normalised over channels (utils.py normalize_activation, eps 1e-10)."""
    w = lambda k: sd[pfx + k].to(x.dtype)
    x = (x - w("mean")) / w("std")
    outs = []
    x = F.relu(F.conv2d(x, w("layers.0.weight"), w("layers.0.bias"), stride=4, padding=2)); outs.append(x)
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, w("layers.3.weight"), w("layers.3.bias"), padding=2)); outs.append(x)
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, w("layers.6.weight"), w("layers.6.bias"), padding=1)); outs.append(x)
    x = F.relu(F.conv2d(x, w("layers.8.weight"), w("layers.8.bias"), padding=1)); outs.append(x)
    x = F.relu(F.conv2d(x, w("layers.10.weight"), w("layers.10.bias"), padding=1)); outs.append(x)
    return [o / (torch.sqrt(torch.sum(o ** 2, dim=1, keepdim=True)) + 1e-10) for o in outs]


def lpips(sd, x, y):
    """LPIPS.forward, src/criteria/lpips/lpips.py:29-35."""
    fx, fy = alexnet_features(sd, x), alexnet_features(sd, y)
    res = [F.conv2d((a - b) ** 2, sd[f"lin.{k}.1.weight"].to(x.dtype)).mean((2, 3), True)
           for k, (a, b) in enumerate(zip(fx, fy))]
    return torch.sum(torch.cat(res, 0)) / x.shape[0]


def lpips_multiscale(sd, x, y, sizes=(1024, 512, 256)):
    """the LPIPS term of Optimizer.calc_loss, scripts/optimization.py:100-108."""
    return sum(lpips(sd, F.adaptive_avg_pool2d(x, (s, s)), F.adaptive_avg_pool2d(y, (s, s))) for s in sizes)


def unet_encoder_features(sd, x, pfx="G."):
    """unet.extract_feats, src/criteria/face_parsing/unet.py:69-91: five unetConv2 stages (conv3x3 + BatchNorm(eval) + ReLU,
    twice; model_utils.py:177-203) with MaxPool2d(2) between them; l2-normalised flattened maps."""
    feats = []
    for i, name in enumerate(("conv1", "conv2", "conv3", "conv4", "center")):
        if i:
            x = F.max_pool2d(x, 2)
        for sub in ("conv1", "conv2"):
            p = f"{pfx}{name}.{sub}."
            x = F.conv2d(x, sd[p + "0.weight"].to(x.dtype), sd[p + "0.bias"].to(x.dtype), padding=1)
            x = F.relu(_bn_eval(sd, p + "1.", x))
        feats.append(x)
    return [f.reshape(f.shape[0], -1) / torch.norm(f.reshape(f.shape[0], -1), 2, 1, True) for f in feats]


def face_parsing_loss(sd, y_hat, y):
    """FaceParsingLoss.forward, src/criteria/face_parsing/face_parsing_loss.py:52-78 -> (loss, sim_improvement)."""
    def extract(x):
        if x.shape[2] != 512:
            x = F.adaptive_avg_pool2d(x, (512, 512))
        return unet_encoder_features(sd, x)
    fy = [f.detach() for f in extract(y)]
    fh = extract(y_hat)
    loss, imp = 0.0, 0.0
    for a, b in zip(fh, fy):
        sim = (a * b).sum(1)
        loss = loss + (1 - sim).mean()
        imp = imp + (sim - (b * b).sum(1)).mean()
    return loss, imp
