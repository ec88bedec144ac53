"""One training iteration of the joint Net3 + Discriminator loop (BASELINE.json configs[4]) -- the body of Coach.train()
(src/training/coach.py:280-398) as a host-side schedule over the native kernels:

    D step   (coach.py:290-307, every `d_every`-th iteration)   net forward without a graph -> D(recon), D(img) on the native
                                                               Discriminator graph -> AdvDLoss (adv_loss.py:18-29) -> backward
                                                               through disc_autograd's closed Function families -> fused Adam
    R1 step  (coach.py:309-319, adv_loss.py:33-45, when `d_reg_every` != -1 and the D step's batch index is a multiple of it)
                                                               dD/d(img) with create_graph -> |grad|^2 penalty -> second-order
                                                               backward through the same families (weight gradients only)
    G step   (coach.py:324-357)                                 Net3.forward (encoder + LocalMLPs [+ G] trainable) -> AdvGLoss * g_adv_lambda
                                                               + calc_loss (coach.py:403-453: parsing, ID, l2, LPIPS x3) -> backward through
                                                               the loss networks, the generator, the LocalMLPs and the encoder -> [bucketed
                                                               all-reduce overlapped with the encoder backward] -> fused Adam
    EMA      (coach.py:396-398, torch_utils.py:189-194)         e4s_ema_f32 per tensor

The scalar glue on [B,1] logits (softplus, means, the loss sum) stays in torch: a handful of one-element launches that
autograd differentiates as is.  Nothing here touches the control plane of the Coach (data loading, logging, checkpoints)."""
import contextlib
import os

import torch
import torch.nn.functional as F

from . import kernels as K
from . import tape as _tape


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


EMA_DECAY = 0.5 ** (32 / (100 * 1000))      # coach.py:29 `ACCUM` = 0.99977822... (its inline comment "0.9977..." drops a digit)


def accumulate(model1, model2, decay=0.999):
    """torch_utils.accumulate (src/utils/torch_utils.py:189-194; same default decay): EMA of model2's parameters into model1 as
    multi-tensor launches (48 tensors each); version counters advance (kernels.ema_multi_), so cached weight packs of model1 are
    rebuilt on next use."""
    p2 = dict(model2.named_parameters())
    dst, src = [], []
    for k, p in model1.named_parameters():
        dst.append(p.detach())
        src.append(p2[k].detach().contiguous())
    with torch.no_grad():
        K.ema_multi_(dst, src, decay)


class _MseFn(torch.autograd.Function):
    """F.mse_loss (mean reduction) with the sum on the native ordered column sum: bit-reproducible, and free of the hipMemsetAsync that
    ATen's global reduce issues for its semaphores (a memset NODE in a captured step; kernels.sum_all says why that matters here)."""

    @staticmethod
    def forward(ctx, a, b):
        d = a.detach() - b.detach()
        ctx.save_for_backward(d)
        ctx.need = tuple(ctx.needs_input_grad[:2])
        return K.sum_all(d * d) / d.numel()

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        ga = d * (g * (2.0 / d.numel()))
        return (ga if ctx.need[0] else None), (-ga if ctx.need[1] else None)


def mse_loss(input, target):
    """nn.MSELoss() of coach.py:147 / F.mse_loss of scripts/optimization.py:95 on device tensors of one shape."""
    native = (input.shape == target.shape and input.is_cuda and input.device == target.device
              and input.dtype == torch.float32 and target.dtype == torch.float32)
    if not native:                    # other dtypes / mixed devices / broadcasting: ATen's rules (and its errors), as before the native path existed
        return F.mse_loss(input, target)
    return _MseFn.apply(input, target)


def adv_g_loss(fake_pred):
    """AdvGLoss, adv_loss.py:8-16."""
    return F.softplus(-fake_pred).mean()


def adv_d_loss(real_pred, fake_pred):
    """AdvDLoss, adv_loss.py:18-29."""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """DR1Loss, adv_loss.py:33-45 (the reference's conv2d_gradfix.no_weight_gradients() is inert on every supported torch,
    SURVEY.md 8(a) item 10; here the first-order weight-gradient nodes are simply never asked for)."""
    grad_real, = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def w_norm_loss(latent, latent_avg=None, start_from_latent_avg=True):
    """WNormLoss, src/criteria/w_norm.py:5-14 on the regional latent [B,R,n_latent,512] (coach.py:438-445; w_norm_lambda is 0 as
    shipped): a scalar reduction of a ~1 MB tensor, left to autograd's ATen ops."""
    if start_from_latent_avg:
        latent = latent - latent_avg
    return torch.sum(latent.norm(2, dim=(2, 3))) / (latent.shape[0] * latent.shape[1])


class LossOpts:
    """The loss weights of train_options.py:44-57 (defaults as shipped; style_lambda -- the VGG16 gram-matrix loss of
    coach.py:446-450, 0 as shipped, needs torchvision's VGG16 -- is not built)."""

    def __init__(self, face_parsing_lambda=0.1, id_lambda=0.1, l2_lambda=1.0, lpips_lambda=0.8, g_adv_lambda=0.01,
                 r1_lambda=10.0, d_every=15, d_reg_every=-1, lpips_sizes=(1024, 512, 256), w_norm_lambda=0.0):
        self.face_parsing_lambda, self.id_lambda, self.l2_lambda = face_parsing_lambda, id_lambda, l2_lambda
        self.lpips_lambda, self.g_adv_lambda, self.r1_lambda = lpips_lambda, g_adv_lambda, r1_lambda
        self.d_every, self.d_reg_every, self.lpips_sizes = d_every, d_reg_every, tuple(lpips_sizes)
        self.w_norm_lambda = w_norm_lambda


class TrainIteration:
    """net: Net3 in train mode; disc: Discriminator or None (train_D False); crit: dict with optional 'lpips', 'id', 'parsing'
    (e4s_amd.criteria modules); opt / opt_d: optimisers over net's / disc's trainable parameters; averager / averager_d:
    ddp.GradAverager (N > 1) or None; net_ema: EMA copy of net or None; ema_decay: coach.py:29's ACCUM by default."""

    def __init__(self, net, disc, crit, opt, opt_d, lo=None, averager=None, averager_d=None, net_ema=None, ema_decay=EMA_DECAY,
                 fork_losses="id", bf16_storage=False):
        self.net, self.disc, self.crit, self.opt, self.opt_d = net, disc, crit, opt, opt_d
        # BASELINE.json configs[4] names bf16: True stores what the forward passes park in HBM for their backward (encoder / generator tapes,
        # the loss networks' and D's saved tensors) as bf16 (e4s_amd/tape.py); arithmetic, weights, moments and gradients stay fp32.  Pair it
        # with ddp.GradAverager(payload_dtype=torch.bfloat16) for the all-reduce payload.  The reference itself trains fp32: default False.
        self.bf16_storage = bf16_storage
        # calc_loss's loss networks (parsing UNet, IR-SE50, LPIPS x3) are independent chains of small launches: each on its own stream,
        # forked from and joined to the step's stream (optim.forked_sum; eager and inside a capture) -- same terms, same order of addition
        # fork_losses: "id" (default; True means the same) = ONE side stream, for the identity network, the longest chain -- the others run on the
        # step's stream under it; "all" = every term on its own side stream; False = one chain.  Eager and captured steps use the SAME placement:
        # autograd accumulates the terms' image gradients in an order that follows the placement, so a capture with another placement than its eager
        # twin differs from it in the last bit (test_graphed_g_step_equals_the_eager_loop).
        if os.environ.get("E4S_FORK_LOSSES", "1") == "0":          # (env: A/B switch, tools/)
            fork_losses = False
        self.fork_losses = "id" if fork_losses is True else fork_losses
        self.lo = lo or LossOpts()
        self.averager, self.averager_d, self.net_ema, self.ema_decay = averager, averager_d, net_ema, ema_decay
        # `net` may be the reference's wrapper, nn.parallel.DistributedDataParallel(net, find_unused_parameters=True,
        # broadcast_buffers=False) (coach.py:74-85): the forward goes through the wrapper (its reducer averages the gradients), the
        # EMA, the latent average and the freeze policy address the module inside, as coach.py does with `self.net.module`
        self.core = getattr(net, "module", net)
        self._net_trainable = [p for p in net.parameters() if p.requires_grad]
        self.global_step = 0

    # ---- coach.py:403-453 -------------------------------------------------------------------------------------------
    def calc_loss(self, img, recon, latent=None):
        lo, terms, fns, keys = self.lo, {}, [], []

        def term(key, weight, fn):
            def run():
                terms[key] = fn()
                return terms[key] * weight
            fns.append(run)
            keys.append(key)
        if lo.face_parsing_lambda > 0 and "parsing" in self.crit:
            term("parsing", lo.face_parsing_lambda, lambda: self.crit["parsing"](recon, img)[0])
        if lo.id_lambda > 0 and "id" in self.crit:
            term("id", lo.id_lambda, lambda: self.crit["id"](recon, img)[0])
        if lo.l2_lambda > 0:
            term("l2", lo.l2_lambda, lambda: mse_loss(recon, img))
        if lo.lpips_lambda > 0 and "lpips" in self.crit:
            # the three adaptive_avg_pool2d scales of coach.py:425-434, pooled inside the networks' first pass
            term("lpips", lo.lpips_lambda, lambda: self.crit["lpips"].forward_pooled(recon, img, lo.lpips_sizes))
        if lo.w_norm_lambda > 0:
            if latent is None:
                raise RuntimeError("w_norm_lambda > 0 needs the latent (Net3.forward(..., return_latents=True))")
            term("w_norm", lo.w_norm_lambda,
                 lambda: w_norm_loss(latent, self.core.latent_avg, getattr(self.core.opts, "start_from_latent_avg", True)))
        if self.fork_losses and recon.is_cuda and len(fns) > 1:
            from .optim import forked_sum
            # fork_losses == "id": ONE side branch, the identity network (the longest chain), the rest on the current stream under it
            side = [keys.index("id")] if (self.fork_losses == "id" and "id" in keys and not os.environ.get("E4S_FORK_SIDE")) else None
            if self.fork_losses == "id" and "id" not in keys:
                side = []                                         # no identity term: one chain
            loss = forked_sum(0.0, fns, inputs=(recon, img), side_terms=side)      # ((0 + parsing) + id) + l2 + lpips: coach.py:403-453's order
        else:
            loss = 0.0
            for fn in fns:
                loss = loss + fn()
        return loss, terms

    def _storage(self):
        return _tape.storage(torch.bfloat16) if self.bf16_storage else contextlib.nullcontext()

    def generator_loss(self, img, onehot, **fwd):
        latent = None
        with self._storage():
            if self.lo.w_norm_lambda > 0:
                recon, _, latent = self.net(img, onehot, return_latents=True, **fwd)
            else:
                recon, _ = self.net(img, onehot, **fwd)
            loss, terms = self.calc_loss(img, recon, latent)
            if self.disc is not None:
                terms["g_adv"] = adv_g_loss(self.disc(recon))
                loss = loss + self.lo.g_adv_lambda * terms["g_adv"]
        return loss, terms, recon

    # ---- coach.py:290-307 -------------------------------------------------------------------------------------------
    def d_step(self, img, onehot, **fwd):
        requires_grad(self.disc, True)
        with torch.no_grad():                       # torch_utils.requires_grad(self.net, False): no graph through the net
            recon, _ = self.net(img, onehot, **fwd)
        with self._storage():
            d_loss = adv_d_loss(self.disc(img), self.disc(recon))
        self.disc.zero_grad()
        if self.averager_d is not None:
            self.averager_d.arm()
        d_loss.backward()
        if self.averager_d is not None:
            self.averager_d.finish()
        self.opt_d.step()
        return d_loss.detach()

    # ---- coach.py:309-319 -------------------------------------------------------------------------------------------
    def r1_step(self, img):
        requires_grad(self.disc, True)
        img = img.detach().requires_grad_(True)
        real_pred = self.disc(img)
        r1 = d_r1_loss(real_pred, img)
        self.disc.zero_grad()
        if self.averager_d is not None:
            self.averager_d.arm()
        (self.lo.r1_lambda / 2 * r1 * max(self.lo.d_reg_every, 1) + 0 * real_pred[0]).sum().backward()
        if self.averager_d is not None:
            self.averager_d.finish()
        self.opt_d.step()
        return r1.detach()

    # ---- coach.py:324-357, 396-398 ----------------------------------------------------------------------------------
    def g_step(self, img, onehot, **fwd):
        if self.disc is not None:
            requires_grad(self.disc, False)
        for p in self._net_trainable:
            p.requires_grad = True
        self.net.zero_grad()
        loss, terms, _ = self.generator_loss(img, onehot, **fwd)
        if self.averager is not None:
            self.averager.arm()                     # bucket all-reduces fire from gradient hooks, under the backward
        loss.backward()
        if self.averager is not None:
            self.averager.finish()
        self.opt.step()
        if self.net_ema is not None:
            accumulate(self.net_ema, self.core, self.ema_decay)
        # detached: a caller that keeps `terms` around must not keep this step's autograd graph (and with it the AccumulateGrad node of
        # every parameter, pinned to the stream of THIS step) alive into the next one -- see optim.GraphedStep
        return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in terms.items()}

    def forget_targets(self):
        """A new batch: drop the loss networks' cached target features.  criteria.py caches them per target TENSOR (right for the
        latent optimisation, whose target never changes); a training loop that refills one static image buffer, or a benchmark that
        feeds the same tensor every step, must not be served the previous batch's features -- the reference recomputes them in every
        calc_loss (id_loss.py:33-35, lpips.py:27-35, face_parsing_loss.py:60-78)."""
        for c in self.crit.values():
            if hasattr(c, "_target"):
                c._target = None

    def graphed_g_step(self, img, onehot, warmup=2, fork_losses_in_graph=None, **fwd):
        """The G step (forward, every loss term incl. the target features, backward, [bucketed gradient all-reduces,] fused Adam, EMA)
        captured as ONE HIP graph: returns an optim.GraphedStep; `.step()` replays it on whatever `img` / `onehot` hold then (static
        buffers, refilled in place).  Needs FusedAdam(capturable=True).  With a ddp.GradAverager the RCCL all-reduces are captured with
        the rest (coach.py:74-85,340-357: DDP's bucket all-reduces overlapped with the backward): they sit on RCCL's own stream in the
        graph, forked where a bucket fills and joined before the write-back, so a replay overlaps them with the remaining backward
        exactly as the eager step does.  The eager G step is ~2 500 launches, host-bound: 65-97 ms per step depending on the host
        against ~55 ms of kernel time.  Every weight pack the step reads from a TRAINED network is rebuilt inside the graph (the net's are
        stale at capture time -- the warm-up steps just updated them --, D's are invalidated here); after a replay the version counters
        of everything the graph wrote (parameters, EMA copy) are advanced, so eager consumers between replays (D steps, net_ema
        evaluation) re-pack from the current weights.  A trainable generator (train_G=True, the reference's default:
        train_options.py:32-33, coach.py:324-331) is captured like the rest: its packs are rebuilt inside the graph, the style
        prologue's job tables depend on addresses only (Generator._style_plan) and are built by the warm-up steps."""
        from .optim import GraphedStep
        from . import disc_autograd
        if self.core is not self.net:
            raise RuntimeError("graphed_g_step: torch's DistributedDataParallel reducer cannot be held by a stream capture; pass the "
                               "bare Net3 and a ddp.GradAverager (its bucket all-reduces are captured with the step), or use g_step()")

        def body():
            self.forget_targets()
            if self.disc is not None:
                disc_autograd.invalidate_packs(self.disc)
            # Inside the capture only ONE loss network -- the identity net, the longest chain -- gets a side stream (the eager step forks them all):
            # a replayed hipGraph is spread over several hardware queues and pays for every cross-queue edge.  Batch-2 step at 1024^2, alternations on
            # one box, ms: all four terms forked 60.6-62.6 (bimodal), one chain 53.5-54.4, LPIPS alone on the side 52.3-52.5, **ID alone 51.2-51.3**,
            # parsing alone 55.2-55.6 (DEBUG_HIP_FORCE_GRAPH_QUEUES=1 gives the fully forked graph the one-chain time: the diagnosis).  Same terms, same
            # order of addition either way (optim.forked_sum's contract).  fork_losses_in_graph: None (default: the iteration's own placement, "id")
            # | "id" | "all" | False (one chain) -- anything but None makes the capture differ from the eager twin in the last bit.
            in_graph = "all" if os.environ.get("E4S_G_FORK_IN_GRAPH") == "1" else fork_losses_in_graph          # (env: experiments with E4S_FORK_SIDE)
            fork = self.fork_losses
            if in_graph is not None and self.fork_losses:
                self.fork_losses = in_graph
            try:
                loss, _ = self.g_step(img, onehot, **fwd)
            finally:
                self.fork_losses = fork
            return loss
        ema = [p.detach() for p in self.net_ema.parameters()] if self.net_ema is not None else []
        return GraphedStep(self.opt, body, warmup=warmup, also_written=ema)

    def graphed_d_step(self, img, onehot, warmup=1, **fwd):
        """The D step (coach.py:290-307: Net3 forward without a graph, D(real), D(fake), AdvDLoss backward through the closed Function
        families of disc_autograd, [bucketed all-reduces,] fused Adam on D) captured as ONE HIP graph -- ~2 000 launches eagerly.  Needs
        opt_d = FusedAdam(capturable=True).  The net's weight packs are rebuilt inside the graph when the G step trains it between replays
        (packs are keyed on versions; GraphedStep advances them)."""
        from .optim import GraphedStep
        from . import disc_autograd

        def body():
            disc_autograd.invalidate_packs(self.disc)
            return self.d_step(img, onehot, **fwd)
        return GraphedStep(self.opt_d, body, warmup=warmup)

    def graphed_r1_step(self, img, warmup=1):
        """The R1 step (coach.py:309-319: the second-order pass) as ONE HIP graph; same requirements as graphed_d_step."""
        from .optim import GraphedStep
        from . import disc_autograd

        def body():
            disc_autograd.invalidate_packs(self.disc)
            return self.r1_step(img)
        return GraphedStep(self.opt_d, body, warmup=warmup)

    def iteration(self, img, onehot, batch_idx=0, **fwd):
        """One pass of the loop body at self.global_step (coach.py:281-398)."""
        out = {}
        lo = self.lo
        if self.disc is not None and self.opt_d is not None and self.global_step % lo.d_every == 0:
            out["d_loss"] = self.d_step(img, onehot, **fwd)
            if lo.d_reg_every != -1 and batch_idx % lo.d_reg_every == 0:
                out["r1_loss"] = self.r1_step(img)
        out["loss"], out["terms"] = self.g_step(img, onehot, **fwd)
        self.global_step += 1
        return out
