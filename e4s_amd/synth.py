"""Seeded synthetic weights / inputs for the E4S hot path.

The reference ships no checkpoints, so parity, smoke and
bench all run on synthetic weights.  Every tensor is drawn from its own
``torch.Generator`` seeded by crc32(key): the values do not depend on module
construction order, so the reference modules (when generating golden
fixtures), the CPU oracle and the HIP engine can all be loaded with bit-equal
weights on any box.

Key names and shapes follow the reference ``Net3.state_dict()``
(src/models/networks.py:41-82, src/models/stylegan2/model.py:451-571,
src/models/encoders/psp_encoders.py:238-262) -- SURVEY.md 8(b).
"""
import math
import zlib

import torch

STYLE_DIM = 512
GEN_CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
ENC_BLOCKS = ((64, 128, 3), (128, 256, 4), (256, 512, 14), (512, 512, 3))


def encoder_units():
    units = []
    for cin, depth, n in ENC_BLOCKS:
        units.append((cin, depth, 2))
        units += [(depth, depth, 1)] * (n - 1)
    return units


def net3_param_spec(out_size=1024, remaining_layer_idx=13, num_seg_cls=12, n_mlp=8):
    """[(key, shape, kind)] in reference state_dict order.  kind selects the synthetic
    distribution: 'randn', 'conv' (randn/sqrt(fan_in)), 'prelu', 'bias', 'modbias',
    'noisew', 'blur' (fixed FIR buffer), 'noisebuf'."""
    spec = []
    # encoder ------------------------------------------------------------
    spec.append(("encoder.input_layer.0.weight", (64, 3, 3, 3), "conv"))
    spec.append(("encoder.input_layer.2.weight", (64,), "prelu"))
    for i, (cin, d, _s) in enumerate(encoder_units()):
        p = f"encoder.body.{i}."
        if cin != d:
            spec.append((p + "shortcut_layer.0.weight", (d, cin, 1, 1), "conv"))
        spec.append((p + "res_layer.1.weight", (d, cin, 3, 3), "conv"))
        spec.append((p + "res_layer.2.weight", (d,), "prelu"))
        spec.append((p + "res_layer.3.weight", (d, d, 3, 3), "conv"))
        spec.append((p + "res_layer.5.fc1.weight", (d // 16, d, 1, 1), "conv"))
        spec.append((p + "res_layer.5.fc2.weight", (d, d // 16, 1, 1), "conv"))
    # LocalMLPs ----------------------------------------------------------
    nw = remaining_layer_idx if remaining_layer_idx != 17 else 18
    for i in range(num_seg_cls):
        p = f"MLPs.{i}.mlp."
        spec.append((p + "0.weight", (512, 1280), "randn"))
        spec.append((p + "0.bias", (512,), "bias"))
        spec.append((p + "2.weight", (512 * nw, 512), "randn"))
        spec.append((p + "2.bias", (512 * nw,), "bias"))
    # generator ----------------------------------------------------------
    for i in range(n_mlp):
        spec.append((f"G.style.{i + 1}.weight", (512, 512), "randn_lr"))
        spec.append((f"G.style.{i + 1}.bias", (512,), "bias"))
    spec.append(("G.input.input", (1, 512, 4, 4), "randn"))

    def styled(p, cin, cout, up):
        out = [(p + "conv.weight", (1, cout, cin, 3, 3), "randn")]
        if up:
            out.append((p + "conv.blur.kernel", (4, 4), "blur"))
        out += [(p + "conv.modulation.weight", (cin, 512), "randn"),
                (p + "conv.modulation.bias", (cin,), "modbias"),
                (p + "noise.weight", (1,), "noisew"),
                (p + "activate.bias", (cout,), "bias")]
        return out

    def torgb(p, cin, up):
        out = [(p + "bias", (1, 3, 1, 1), "bias")]
        if up:
            out.append((p + "upsample.kernel", (4, 4), "blur"))
        out += [(p + "conv.weight", (1, 3, cin, 1, 1), "randn"),
                (p + "conv.modulation.weight", (cin, 512), "randn"),
                (p + "conv.modulation.bias", (cin,), "modbias")]
        return out

    spec += styled("G.conv1.", 512, 512, False)
    spec += torgb("G.to_rgb1.", 512, False)
    log_size = int(math.log2(out_size))
    convs, rgbs = [], []
    cin = 512
    for j, rl in enumerate(range(3, log_size + 1)):
        cout = GEN_CHANNELS[2 ** rl]
        convs += styled(f"G.convs.{2 * j}.", cin, cout, True)
        convs += styled(f"G.convs.{2 * j + 1}.", cout, cout, False)
        rgbs += torgb(f"G.to_rgbs.{j}.", cout, True)
        cin = cout
    spec += convs + rgbs
    for l in range((log_size - 2) * 2 + 1):
        r = 2 ** ((l + 5) // 2)
        spec.append((f"G.noises.noise_{l}", (1, 1, r, r), "noisebuf"))
    return spec


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, kind, seed=0):
    g = _gen(key, seed)
    if kind in ("blur", "blur1"):               # Blur buffer: x4 when it follows an up-sampling conv (model.py:85-86)
        k = torch.tensor([1.0, 3.0, 3.0, 1.0])
        k = k[None, :] * k[:, None]
        return k / k.sum() * (4.0 if kind == "blur" else 1.0)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind in ("randn", "noisebuf"):
        return x
    if kind == "randn_lr":                      # EqualLinear(lr_mul=0.01): randn / lr_mul
        return x / 0.01
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        return x / math.sqrt(fan_in)
    if kind == "prelu":
        return 0.25 + 0.05 * x
    if kind == "bias":
        return 0.1 * x
    if kind == "modbias":                       # reference bias_init=1
        return 1.0 + 0.1 * x
    if kind == "noisew":
        return 0.1 + 0.02 * x
    raise ValueError(kind)


def synth_state_dict(out_size=1024, remaining_layer_idx=13, num_seg_cls=12, seed=0):
    return {k: synth_tensor(k, s, kind, seed)
            for k, s, kind in net3_param_spec(out_size, remaining_layer_idx, num_seg_cls)}


def disc_param_spec(size=64, channel_multiplier=2):
    """(key, shape, kind) of Discriminator(size) -- src/models/stylegan2/model.py:740-775 (buffers included)."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
          256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    spec = [("convs.0.0.weight", (ch[size], 3, 1, 1), "randn"), ("convs.0.1.bias", (ch[size],), "bias")]
    cin = ch[size]
    log_size = int(math.log2(size))
    for j, i in enumerate(range(log_size, 2, -1), start=1):
        cout = ch[2 ** (i - 1)]
        c = f"convs.{j}."
        spec += [(c + "conv1.0.weight", (cin, cin, 3, 3), "randn"), (c + "conv1.1.bias", (cin,), "bias"),
                 (c + "conv2.0.kernel", (4, 4), "blur1"), (c + "conv2.1.weight", (cout, cin, 3, 3), "randn"),
                 (c + "conv2.2.bias", (cout,), "bias"),
                 (c + "skip.0.kernel", (4, 4), "blur1"), (c + "skip.1.weight", (cout, cin, 1, 1), "randn")]
        cin = cout
    spec += [("final_conv.0.weight", (ch[4], cin + 1, 3, 3), "randn"), ("final_conv.1.bias", (ch[4],), "bias"),
             ("final_linear.0.weight", (ch[4], ch[4] * 16), "randn"), ("final_linear.0.bias", (ch[4],), "bias"),
             ("final_linear.1.weight", (1, ch[4]), "randn"), ("final_linear.1.bias", (1,), "bias")]
    return spec


def synth_disc_state_dict(size=64, seed=0):
    return {k: synth_tensor("D." + k, s, kind, seed) for k, s, kind in disc_param_spec(size)}


def gpen_channels(channel_multiplier=2, narrow=1.0):
    """src/pretrained/gpen/face_model/gpen_model.py:411-422."""
    return {4: int(512 * narrow), 8: int(512 * narrow), 16: int(512 * narrow), 32: int(512 * narrow),
            64: int(256 * channel_multiplier * narrow), 128: int(128 * channel_multiplier * narrow),
            256: int(64 * channel_multiplier * narrow), 512: int(32 * channel_multiplier * narrow),
            1024: int(16 * channel_multiplier * narrow), 2048: int(8 * channel_multiplier * narrow)}


def gpen_param_spec(size=512, style_dim=512, n_mlp=8, channel_multiplier=2, narrow=1.0):
    """(key, shape, kind) of GPEN FullGenerator(size, style_dim, n_mlp, isconcat=True) -- gpen_model.py:380-690
    (SURVEY.md 8(f) N2).  isconcat doubles the channels every StyledConv hands on (feat_multiplier = 2)."""
    ch = gpen_channels(channel_multiplier, narrow)
    log_size = int(math.log2(size))
    g = "generator."
    spec = []
    for i in range(n_mlp):
        spec += [(f"{g}style.{i + 1}.weight", (style_dim, style_dim), "randn_lr"), (f"{g}style.{i + 1}.bias", (style_dim,), "bias")]

    def styled(pfx, cin, cout, up):
        out = [(pfx + "conv.weight", (1, cout, cin, 3, 3), "randn")]
        if up:
            out.append((pfx + "conv.blur.kernel", (4, 4), "blur"))
        out += [(pfx + "conv.modulation.weight", (cin, style_dim), "randn"), (pfx + "conv.modulation.bias", (cin,), "modbias"),
                (pfx + "noise.weight", (1,), "noisew"), (pfx + "activate.bias", (2 * cout,), "bias")]
        return out

    def torgb(pfx, cin, up):
        out = [(pfx + "bias", (1, 3, 1, 1), "bias")]
        if up:
            out.append((pfx + "upsample.kernel", (4, 4), "blur"))
        out += [(pfx + "conv.weight", (1, 3, cin, 1, 1), "randn"), (pfx + "conv.modulation.weight", (cin, style_dim), "randn"),
                (pfx + "conv.modulation.bias", (cin,), "modbias")]
        return out
    spec.append((g + "input.input", (1, ch[4], 4, 4), "randn"))
    spec += styled(g + "conv1.", ch[4], ch[4], False) + torgb(g + "to_rgb1.", 2 * ch[4], False)
    cin = ch[4]
    for j, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        spec += styled(f"{g}convs.{2 * j}.", 2 * cin, cout, True) + styled(f"{g}convs.{2 * j + 1}.", 2 * cout, cout, False)
        cin = cout
    cin = ch[4]
    for j, i in enumerate(range(3, log_size + 1)):
        spec += torgb(f"{g}to_rgbs.{j}.", 2 * ch[2 ** i], True)
    spec += [("ecd0.0.0.weight", (ch[size], 3, 1, 1), "randn"), ("ecd0.0.1.bias", (ch[size],), "bias")]
    cin = ch[size]
    for j, i in enumerate(range(log_size, 2, -1), start=1):
        cout = ch[2 ** (i - 1)]
        spec += [(f"ecd{j}.0.0.kernel", (4, 4), "blur1"), (f"ecd{j}.0.1.weight", (cout, cin, 3, 3), "randn"),
                 (f"ecd{j}.0.2.bias", (cout,), "bias")]
        cin = cout
    spec += [("final_linear.0.weight", (style_dim, ch[4] * 16), "randn"), ("final_linear.0.bias", (style_dim,), "bias")]
    return spec


def synth_gpen_state_dict(size=512, seed=0, **kw):
    return {k: synth_tensor("GPEN." + k, s, kind, seed) for k, s, kind in gpen_param_spec(size, **kw)}


def synth_latent_avg(out_size=1024, seed=0):
    n_latent = int(math.log2(out_size)) * 2 - 2
    return 0.1 * torch.randn(n_latent, 512, generator=_gen("latent_avg", seed))


def synth_noise(out_size=1024, seed=0, batch=1):
    """17 noise maps [batch,1,2^r,2^r] (model.py:512-516)."""
    log_size = int(math.log2(out_size))
    out = []
    for l in range((log_size - 2) * 2 + 1):
        r = 2 ** ((l + 5) // 2)
        out.append(torch.randn(batch, 1, r, r, generator=_gen(f"noise{l}", seed)))
    return out


def synth_image(batch=1, size=1024, seed=0, tag="img"):
    return torch.randn(batch, 3, size, size, generator=_gen(tag, seed)).clamp_(-1, 1)


def onehot(labels, num_cls=12):
    """labelMap2OneHot (src/utils/torch_utils.py:166-172). labels [B,1,H,W] int64."""
    b, _, h, w = labels.shape
    out = torch.zeros(b, num_cls, h, w, device=labels.device)
    return out.scatter_(1, labels, 1.0)


def synth_labels_blocks(batch=1, size=512, cells=64, num_cls=12, seed=0, tag="blk"):
    """Pessimistic mask: random label per (size/cells)^2 block -- every region populated,
    boundaries everywhere (SURVEY.md 8(d) synthetic mask (ii))."""
    lab = torch.randint(0, num_cls, (batch, 1, cells, cells), generator=_gen(tag, seed))
    rep = size // cells
    return lab.repeat_interleave(rep, 2).repeat_interleave(rep, 3)


def synth_labels_face(batch=1, size=512, num_cls=12, seed=0, tag="face"):
    """Face-like label map built from ellipses: background 0, hair, skin, brows, eyes,
    nose, mouth, neck, ears ... -- large smooth regions with a few small ones, the
    realistic case for region-select.  Deterministic per (seed, sample)."""
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    out = torch.zeros(batch, 1, size, size, dtype=torch.int64)
    for b in range(batch):
        g = _gen(f"{tag}{b}", seed)
        j = (torch.rand(8, generator=g) - 0.5) * 0.08
        lab = torch.zeros(size, size, dtype=torch.int64)

        def ell(cx, cy, rx, ry):
            return ((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2 <= 1.0

        cx, cy = float(j[0]), float(j[1])
        lab[ell(cx, cy - 0.15, 0.62, 0.75)] = 4                 # hair
        lab[ell(cx, cy + 0.95, 0.30, 0.45)] = 8                 # neck
        lab[ell(cx - 0.52, cy + 0.05, 0.08, 0.16)] = 7          # ears
        lab[ell(cx + 0.52, cy + 0.05, 0.08, 0.16)] = 7
        lab[ell(cx, cy + 0.08, 0.48, 0.62)] = 1                 # skin
        lab[ell(cx - 0.2, cy - 0.18, 0.13, 0.035)] = 2          # brows
        lab[ell(cx + 0.2, cy - 0.18, 0.13, 0.035)] = 2
        lab[ell(cx - 0.2, cy - 0.06, 0.10, 0.05)] = 3           # eyes
        lab[ell(cx + 0.2, cy - 0.06, 0.10, 0.05)] = 3
        lab[ell(cx, cy + 0.12 + float(j[2]), 0.09, 0.17)] = 5   # nose
        lab[ell(cx, cy + 0.40, 0.20, 0.085)] = 6                # mouth / lips
        lab[ell(cx, cy + 0.40, 0.12, 0.03)] = 9                 # teeth
        if num_cls > 11:
            lab[ell(cx - 0.55, cy + 0.27, 0.03, 0.05)] = 11     # ear ring
        out[b, 0] = lab.clamp_(max=num_cls - 1)
    return out


def synth_module_state_dict(module, seed=0, tag="crit."):
    """Seeded values for every entry of a frozen loss network's state_dict (IDLoss / LPIPS; their real
    weights are downloads).  Chosen by key and shape only, so the reference module and e4s_amd.criteria's get identical
    tensors: conv / linear weights fan-in scaled (x sqrt(2) in front of a ReLU), BatchNorm scale 1 +- 0.1 with running
    statistics near (0, 1), PReLU slopes near 0.25, LPIPS lin weights non-negative; registered constants (mean / std) and
    counters keep their values."""
    out = {}
    for k, v in module.state_dict().items():
        g = _gen(tag + k, seed)
        shape = tuple(v.shape)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked" or k.endswith("net.mean") or k.endswith("net.std"):
            out[k] = v.clone()
            continue
        x = torch.randn(shape, generator=g, dtype=torch.float32)
        if leaf == "running_var":
            out[k] = 0.75 + 0.5 * torch.rand(shape, generator=g, dtype=torch.float32)
        elif leaf == "running_mean":
            out[k] = 0.1 * x
        elif leaf == "bias":
            out[k] = 0.1 * x
        elif len(shape) == 1:                       # BatchNorm scale / PReLU slope
            is_prelu = "res_layer.2." in k or k.endswith("input_layer.2.weight")
            out[k] = 0.25 + 0.05 * x if is_prelu else 1.0 + 0.1 * x
        elif ".lin." in k or k.startswith("lin."):
            out[k] = 0.1 * x.abs()
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = math.sqrt(2.0) if ".layers." in k else 1.0
            out[k] = gain * x / math.sqrt(fan_in)
    return out


def synth_image_pair(batch=1, size=1024, seed=0):
    """(y_hat, y): a smooth seeded image in [-1, 1] and a perturbed copy (what a loss network compares during inversion)."""
    import torch.nn.functional as F
    base = torch.randn(batch, 3, 16, 16, generator=_gen("pair.base", seed))
    mid = torch.randn(batch, 3, 64, 64, generator=_gen("pair.mid", seed))
    pert = torch.randn(batch, 3, 32, 32, generator=_gen("pair.pert", seed))
    up = lambda t: F.interpolate(t, size=(size, size), mode="bilinear", align_corners=False)
    y = (0.6 * up(base) + 0.2 * up(mid)).clamp_(-1, 1)
    return (y + 0.15 * up(pert)).clamp_(-1, 1), y
