"""Net3: regional style encoder + 12 LocalMLPs + mask-guided StyleGAN2 generator.

Same public surface as the reference (src/models/networks.py:41-183): constructor `Net3(opts)`,
`forward`, `get_style_vectors`, `cal_style_codes`, `gen_img`, plain attribute `latent_avg`, identical
state_dict keys -- so scripts/face_swap.py, scripts/face_edit.py and checkpoints drop in.  Every
method is a schedule of HIP kernels (libe4s_hip.so); there is no torch-eager fallback.
"""
import math

import torch
from torch import nn

from . import kernels as K
from .encoders import FSEncoder_PSP
from .packs import param_key
from .stylegan2 import EqualLinear, Generator


class LocalMLP(nn.Module):
    """networks.py:15-39: EqualLinear(1280->512) -> LeakyReLU(0.01) -> EqualLinear(512 -> 512*num_w_layers)."""

    def __init__(self, dim_component=512, dim_style=512, num_w_layers=18, latent_squeeze_ratio=1):
        super().__init__()
        self.dim_component, self.dim_style, self.num_w_layers = dim_component, dim_style, num_w_layers
        self.mlp = nn.Sequential(EqualLinear(dim_component, dim_style // latent_squeeze_ratio, lr_mul=1),
                                 nn.LeakyReLU(),
                                 EqualLinear(dim_style // latent_squeeze_ratio, dim_style * num_w_layers, lr_mul=1))

    def forward(self, x):
        """x [B, dim_component] -> [B, num_w_layers, 512] (single-region drop-in)."""
        l0, l2 = self.mlp[0], self.mlp[2]
        h = K.grouped_linear(x.unsqueeze(1).contiguous(), l0.weight.unsqueeze(0), l0.bias.unsqueeze(0), None,
                             l0.scale, act=1, alpha=self.mlp[1].negative_slope)
        y = K.grouped_linear(h, l2.weight.unsqueeze(0), l2.bias.unsqueeze(0), None, l2.scale)
        return y.view(-1, self.num_w_layers, self.dim_style)


class Net3(nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        assert self.opts.fsencoder_type in ["psp"]
        self.encoder = FSEncoder_PSP(mode="ir_se", opts=self.opts)
        dim_s_code = 256 + 512 + 512
        self.split_layer_idx = 5
        self.remaining_layer_idx = self.opts.remaining_layer_idx
        K_ = self.remaining_layer_idx
        self.MLPs = nn.ModuleList([LocalMLP(dim_component=dim_s_code, dim_style=512,
                                            num_w_layers=K_ if K_ != 17 else 18)
                                   for _ in range(self.opts.num_seg_cls)])
        self.G = Generator(size=self.opts.out_size, style_dim=512, n_mlp=8, split_layer_idx=self.split_layer_idx,
                           remaining_layer_idx=K_)
        # freeze policy, networks.py:63-82
        if not self.opts.train_G:
            for p in self.G.parameters():
                p.requires_grad = False
        else:
            for p in self.G.style.parameters():
                p.requires_grad = False
        if K_ != 17:
            for p in self.G.convs[-(17 - K_):].parameters():
                p.requires_grad = False
            for p in self.G.to_rgbs[-(17 - K_) // 2 - 1:].parameters():
                p.requires_grad = False
        self._mlp_pack = None

    # ---- stacked LocalMLP weights ([R,O,K]) so the 24 small GEMMs become 2 launches -------------
    def _mlp_weights(self):
        ps = [t for m in self.MLPs for t in (m.mlp[0].weight, m.mlp[0].bias, m.mlp[2].weight, m.mlp[2].bias)]
        key = param_key(*ps)
        if self._mlp_pack is None or self._mlp_pack[0] != key:
            with torch.no_grad():
                w0 = torch.stack([m.mlp[0].weight.detach() for m in self.MLPs]).contiguous()
                b0 = torch.stack([m.mlp[0].bias.detach() for m in self.MLPs]).contiguous()
                w2 = torch.stack([m.mlp[2].weight.detach() for m in self.MLPs]).contiguous()
                b2 = torch.stack([m.mlp[2].bias.detach() for m in self.MLPs]).contiguous()
            self._mlp_pack = (key, w0, b0, w2, b2)
        return self._mlp_pack[1:]

    def _encoder_needs_grad(self, img):
        """True when get_style_vectors must build an autograd graph: grad mode on and some encoder parameter trainable
        (Net3 leaves them trainable, networks.py:48; coach.py:340-356 trains them).  Gradients w.r.t. the IMAGE are not
        provided -- nothing in the reference asks for them -- and are refused loudly instead of detaching silently."""
        if not torch.is_grad_enabled():
            return False
        if img.requires_grad:
            raise NotImplementedError("gradients w.r.t. the input image of the regional encoder are not implemented "
                                      "(parameter gradients are: e4s_amd/encoder_autograd.py)")
        return any(p.requires_grad for p in self.encoder.parameters())

    # ---- API ------------------------------------------------------------------------------------
    def get_style_vectors(self, img, mask):
        """networks.py:121-133: img [B,3,H,W] in [-1,1], one-hot mask [B,R,Hm,Wm] ->
        ([B,R,1280], zeros [B,512,16,16])."""
        train = self._encoder_needs_grad(img)
        with torch.no_grad():
            labels, flags = K.mask_labels(mask)
            if self.G.strict_mask and not torch.cuda.is_current_stream_capturing() and bool(flags.item()):
                raise NotImplementedError("get_style_vectors needs a one-hot parsing mask (labelMap2OneHot); the "
                                          "regional pooling of soft masks is not implemented")
            x256 = K.resize_bilinear_to_nhwc(img, 256, 256)                  # F.interpolate(...,'bilinear') :131
            if not train:
                codes, last = self.encoder.encode_nhwc(x256, labels, mask.shape[1])
                b, h, w, c = last.shape
                return codes, torch.zeros(b, c, h, w, device=img.device, dtype=torch.float32)
        from .encoder_autograd import EncoderFn
        params = [p for p in self.encoder.parameters() if p.requires_grad]
        codes = EncoderFn.apply(self.encoder, x256, labels, mask.shape[1], *params)
        return codes, torch.zeros(img.shape[0], 512, 16, 16, device=img.device, dtype=torch.float32)

    def cal_style_codes(self, style_vectors):
        """networks.py:135-158 -> [B,R,n_latent,512]."""
        if not self.opts.start_from_latent_avg or self.opts.learn_in_w:
            raise NotImplementedError("only start_from_latent_avg=True, learn_in_w=False (the shipped configs)")
        K_ = self.remaining_layer_idx
        nw = K_ if K_ != 17 else 18
        if torch.is_grad_enabled() and any(p.requires_grad for m in self.MLPs for p in m.parameters()):
            # training: stack the live parameters so that autograd routes the stacked gradients back to the 12 MLPs
            w0 = torch.stack([m.mlp[0].weight for m in self.MLPs])
            b0 = torch.stack([m.mlp[0].bias for m in self.MLPs])
            w2 = torch.stack([m.mlp[2].weight for m in self.MLPs])
            b2 = torch.stack([m.mlp[2].bias for m in self.MLPs])
        else:
            w0, b0, w2, b2 = self._mlp_weights()
        b, r, _ = style_vectors.shape
        lat = self.latent_avg.to(device=style_vectors.device, dtype=torch.float32)
        add = lat[:nw].reshape(-1).contiguous()
        from .autograd import StyleCodesFn
        codes = StyleCodesFn.apply(style_vectors, w0, b0, w2, b2, add).view(b, r, nw, 512)
        if K_ != 17:
            rest = lat[K_:].view(1, 1, -1, 512).expand(b, r, -1, -1)
            codes = torch.cat([codes, rest], dim=2)
        return codes

    def gen_img(self, struc_codes, style_codes, mask, randomize_noise=True, noise=None, return_latents=False):
        """networks.py:160-182 -> (images, latent | -1, feats16)."""
        images, result_latent, feats = self.G([style_codes], struc_codes, mask, input_is_latent=True,
                                              randomize_noise=randomize_noise, noise=noise,
                                              return_latents=return_latents, use_structure_code=False)
        if return_latents:
            return images, result_latent, feats
        return images, -1, feats

    def forward(self, img, mask, resize=False, randomize_noise=True, return_latents=False):
        """networks.py:85-119 -> (images, feats16[, latent])."""
        sv, struct = self.get_style_vectors(img, mask)
        codes = self.cal_style_codes(sv)
        images, latent, feats = self.G([codes], struct, mask, input_is_latent=True, randomize_noise=randomize_noise,
                                       return_latents=return_latents, use_structure_code=False)
        if return_latents:
            return images, feats, latent
        return images, feats


_SWAP_SEL = {}


def swap_comp_style_vector(style_vectors1, style_vectors2, comp_indices, belowFace_interpolation=False):
    """scripts/face_swap.py:117-146, applied per sample so that batches work.
    style_vectors1 = target, style_vectors2 = source/driven."""
    r = style_vectors1.shape[1]
    if (style_vectors1.is_cuda and style_vectors1.dtype == torch.float32 and style_vectors2.dtype == torch.float32
            and style_vectors2.device == style_vectors1.device and style_vectors2.shape == style_vectors1.shape and 9 < r <= 32):
        # one native launch (the torch statement below is ~13 tiny ATen launches in the replayed graph); same results bit for bit
        return K.swap_styles(style_vectors1, style_vectors2, comp_indices, belowFace_interpolation)
    key = (style_vectors1.device, r, tuple(sorted(comp_indices)))
    sel = _SWAP_SEL.get(key)
    if sel is None:                    # built once, eagerly (an H2D copy is not capturable in a HIP graph)
        sel = torch.zeros(1, r, 1, dtype=torch.bool)
        sel[0, list(key[2]), 0] = True
        sel = _SWAP_SEL[key] = sel.to(style_vectors1.device)
    out = torch.where(sel, style_vectors2, style_vectors1)
    # torch.where instead of boolean indexing: no host sync, HIP-graph capturable
    no_ear = (style_vectors2[:, 7].sum(1, keepdim=True) == 0)
    out[:, 7] = torch.where(no_ear, (style_vectors1[:, 7] + style_vectors2[:, 7]) / 2, out[:, 7])
    no_teeth = (style_vectors2[:, 9].sum(1, keepdim=True) == 0)
    out[:, 9] = torch.where(no_teeth, style_vectors1[:, 9], out[:, 9])
    if belowFace_interpolation:
        out[:, 8] = (style_vectors1[:, 8] + style_vectors2[:, 8]) / 2
    return out


@torch.no_grad()
def face_swap_core(net, driven, driven_mask, target, target_mask, swapped_mask, noise=None, randomize_noise=False):
    """The E4S-core unit of work (scripts/face_swap.py:237-273) on a batch of B swaps:
    2 encoder passes (fused into one batched pass), regional style swap, LocalMLPs, generator."""
    b = driven.shape[0]

    def stacked(a, c):
        """[a; c] along dim 0 -- as a VIEW when the two already sit back to back in one buffer (GraphedFaceSwap lays its static
        inputs out that way: the 2 x 200 MB concatenation copies of a batch of 8 were 0.18 ms of every step)."""
        if (a.is_contiguous() and c.is_contiguous() and a.shape == c.shape and a.dtype == c.dtype
                and a.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()
                and c.storage_offset() == a.storage_offset() + a.numel()):
            return a.as_strided((2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
        return torch.cat([a, c], 0)
    sv, _ = net.get_style_vectors(stacked(driven, target), stacked(driven_mask, target_mask))
    d_sv, t_sv = sv[:b], sv[b:]
    comp = set(range(net.opts.num_seg_cls)) - {0, 4, 11, 10}
    swapped = swap_comp_style_vector(t_sv, d_sv, comp)
    codes = net.cal_style_codes(swapped)
    img, _, _ = net.gen_img(None, codes, swapped_mask, randomize_noise=randomize_noise, noise=noise)
    return img


class GraphedFaceSwap:
    """`face_swap_core` for fixed shapes captured once into a HIP graph and replayed.

    One swap is ~700 short kernel launches; at batch 1 the Python/ctypes enqueue cost (not the GPU) sets
    the latency.  Capture removes it: inputs are copied into static device buffers, the whole schedule
    (encoder x2, style swap, MLPs, generator) replays as one graph launch.  Weight re-packing caches must
    be warm, so the first call runs eagerly twice before capturing.

    Preconditions a replay cannot check on the host without a sync: (1) the three parsing masks are one-hot
    (labelMap2OneHot) -- the captured kernels OR a not-one-hot flag into `self.flags`; call `validate()` (one sync)
    whenever convenient, it raises if any replay since the last call saw a soft mask (eager calls fall back to the
    reference's R-pass formulation instead; a graph cannot); (2) the weights have not changed since capture -- the
    graph bakes in the packed-weight pointers: build a new GraphedFaceSwap after an optimizer step / EMA update /
    load_state_dict."""

    def __init__(self, net, batch, img_size=1024, mask_size=512, noise_batch=None):
        dev = next(net.parameters()).device
        self.net, self.batch = net, batch
        r = net.opts.num_seg_cls
        # driven | target images (and their masks) back to back in ONE buffer each: face_swap_core's batched encoder pass then
        # takes a view instead of concatenating
        imgs = torch.zeros(2 * batch, 3, img_size, img_size, device=dev)
        masks = torch.zeros(2 * batch, r, mask_size, mask_size, device=dev)
        self.static = [imgs[:batch], masks[:batch], imgs[batch:], masks[batch:],
                       torch.zeros(batch, r, mask_size, mask_size, device=dev)]
        for m in self.static[1::2] + [self.static[4]]:
            m[:, 0] = 1.0                                      # a valid one-hot mask for the warm-up runs
        nb = batch if noise_batch is None else noise_batch
        self.noise = [torch.zeros(nb, 1, n.shape[2], n.shape[3], device=dev) for n in net.G.make_noise()]
        self.flags = torch.zeros(1, device=dev, dtype=torch.int32)
        self.graph = None
        self.out = None

    def _load(self, driven, dm, target, tm, sm, noise):
        for dst, src in zip(self.static, (driven, dm, target, tm, sm)):
            dst.copy_(src)
        for dst, src in zip(self.noise, noise):
            dst.copy_(src)

    def __call__(self, driven, dm, target, tm, sm, noise):
        self._load(driven, dm, target, tm, sm, noise)
        return self.replay()

    def load(self, driven, dm, target, tm, sm, noise):
        """Fill the graph's input buffers once; `replay()` then runs the swap on whatever they hold.  A producer that writes its batches
        straight into `self.static` (driven, driven mask, target, target mask, swapped mask) / `self.noise` -- a data loader, the previous
        pipeline stage -- saves the 22 device-to-device copies per step that `__call__` makes (0.34 ms of an 18 ms step at batch 8)."""
        self._load(driven, dm, target, tm, sm, noise)

    def replay(self):
        """The swap of the CURRENT contents of the input buffers (captured on first use); returns the static output tensor."""
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    face_swap_core(self.net, *self.static, noise=self.noise)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # with a process group up (N > 1) other threads of this process (the RCCL watchdog) may touch the HIP
            # runtime while the capture runs: only this thread's calls are policed then
            mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
            from .shard import quiesce_before_capture
            quiesce_before_capture()
            with K.flag_sink(self.flags):
                with torch.cuda.graph(self.graph, capture_error_mode=mode):
                    self.out = face_swap_core(self.net, *self.static, noise=self.noise)
        self.graph.replay()
        return self.out

    def validate(self):
        """One host sync: raises if any replay since the last validate() was fed a mask that is not one-hot."""
        bad = bool(self.flags.item())
        self.flags.zero_()
        if bad:
            raise RuntimeError("GraphedFaceSwap was replayed with a parsing mask that is not one-hot: its outputs used "
                               "hard argmax regions; run face_swap_core eagerly for soft masks")
