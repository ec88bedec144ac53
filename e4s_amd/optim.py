"""Fused optimiser step for the latent-optimisation loop (scripts/optimization.py:125-161 builds torch.optim.Adam over the
[1,12,1280] style vectors; SURVEY.md 8(f) N1 'Adam step fusion').  Same hyper-parameters and state as torch.optim.Adam
(no amsgrad); each parameter is updated by ONE kernel (e4s_adam_step_f32) instead of ~10 elementwise launches.
`capturable=True` keeps the step count on the device (e4s_adam_step_dev_f32), so the whole optimisation step -- forward,
backward and update -- can be captured in a HIP graph (`GraphedStep`) and replayed.  Measured at 1024^2: 11.6 ms replayed vs
11.9 ms eager on the l2-only step, 20.4 vs ~25 ms with the LPIPS and identity terms."""
import torch

from . import kernels as K


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable      # step counts live in state[p]["step"] as device int64[1] tensors (as torch's
                                          # capturable Adam keeps them): per parameter, and part of state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam handles contiguous fp32 parameters")
                if p.grad.is_sparse or p.grad.device != p.device or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdam needs a dense fp32 gradient on the parameter's device")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros(1, device=p.device, dtype=torch.int64) if self.capturable else 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                if self.capturable:
                    if not torch.is_tensor(st["step"]):         # state loaded from a non-capturable optimiser
                        st["step"] = torch.full((1,), int(st["step"]), device=p.device, dtype=torch.int64)
                    K.adam_step_dev(p, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                                    group["weight_decay"], st["step"], True)
                    torch.autograd.graph.increment_version(p)
                    continue
                if torch.is_tensor(st["step"]):                 # state loaded from a capturable optimiser (one sync, once)
                    st["step"] = int(st["step"].item())
                st["step"] += 1
                K.adam_step(p, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                            group["weight_decay"], st["step"])
                # the kernel wrote through the raw pointer: advance the version counter like an in-place torch op would, or
                # every weight pack keyed on (data_ptr, _version) (e4s_amd/packs.py) would keep serving the OLD weights
                torch.autograd.graph.increment_version(p)
        return loss


class GraphedStep:
    """One optimisation step captured in a HIP graph.  `body()` runs forward + backward + `opt.step()` for a capturable
    optimiser and returns the loss tensor; it must be free of host synchronisation and read its inputs from tensors whose
    storage does not change between steps (update them in place).  The constructor runs `warmup` REAL steps eagerly on a side
    stream (they count as optimisation steps), then captures one more without executing it; `step()` replays it and returns
    the (static) loss tensor."""

    def __init__(self, opt, body, warmup=2):
        if not getattr(opt, "capturable", False):
            raise RuntimeError("GraphedStep needs FusedAdam(capturable=True): the host-side step count cannot be replayed")
        self.opt = opt
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                     # also fills the weight-pack / target-feature caches
                opt.zero_grad(set_to_none=True)
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.steps_done = max(1, warmup)
        self.graph = torch.cuda.CUDAGraph()
        flags = torch.zeros(1, device=opt.param_groups[0]["params"][0].device, dtype=torch.int32)
        opt.zero_grad(set_to_none=True)
        with K.flag_sink(flags):                                # this thread's mask checks accumulate here instead of syncing
            with torch.cuda.graph(self.graph):
                self.loss = body()
        self.flags = flags

    def step(self):
        self.graph.replay()
        self.steps_done += 1
        return self.loss

    def validate(self):
        """One host sync: raises if any replay since the last validate() saw a parsing mask that is not one-hot (the replayed
        kernels use hard argmax regions; the eager path falls back to the reference's R-pass formulation) or a non-finite loss."""
        bad = bool(self.flags.item())
        self.flags.zero_()
        if bad:
            raise RuntimeError("GraphedStep was replayed with a parsing mask that is not one-hot")
        if not bool(torch.isfinite(self.loss).all()):
            raise RuntimeError("GraphedStep: non-finite loss")
