"""Fused optimiser step for the latent-optimisation loop (scripts/optimization.py:125-161 builds torch.optim.Adam over the
[1,12,1280] style vectors; SURVEY.md 8(f) N1 'Adam step fusion') and the joint train step (coach.py:119-150).  Same
hyper-parameters and state as torch.optim.Adam (no amsgrad).

* default: each parameter is updated by ONE kernel (e4s_adam_step_f32) instead of ~10 elementwise launches; step counts are host ints.
* `capturable=True`: nothing step-dependent is computed on the host, so the whole optimisation step -- forward, backward and update -- can
  be captured in a HIP graph (`GraphedStep`) and replayed.  The step counts of a parameter group live in ONE flat device int64 tensor
  (`state[p]["step"]` is a one-element view of it, so `state_dict()` round-trips per parameter as torch's capturable Adam does) advanced by
  one launch per group; the learning rate lives in a device float64 (`sync_hyper()` uploads `group["lr"]` when it changed: coach.py:377-381's
  schedule reaches a captured graph); the updates of ALL parameters go out as multi-tensor launches (e4s_adam_multi_dev_f32, 48 tensors
  per launch with their pointers in the kernel arguments): Net3's 344 tensors are 1 + 8 launches per step, not 688.
Measured at 1024^2: 11.6 ms replayed vs 11.9 ms eager on the l2-only step, 20.4 vs ~25 ms with the LPIPS and identity terms."""
import os

import torch

from . import kernels as K


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable
        self._dev = {}                    # group index -> {"lr": device f64[1], "lr_host": float, "hyper": tuple, "flat": int64 [n] | None}

    # ---- capturable plumbing ----------------------------------------------------------------------------------------
    def _group_dev(self, gi, group, device):
        d = self._dev.get(gi)
        if d is None or d["lr"].device != device:
            d = {"lr": torch.full((1,), float(group["lr"]), device=device, dtype=torch.float64), "lr_host": float(group["lr"]),
                 "hyper": None, "flat": None}
            self._dev[gi] = d
        return d

    def sync_hyper(self):
        """Upload every group's learning rate if it changed on the host (one fill launch per changed group; call it OUTSIDE a capture --
        GraphedStep.step() does).  betas / eps / weight_decay travel by value in the launches: a captured graph keeps the values it was
        captured with, so GraphedStep.step() refuses a replay after they changed (hyper_by_value) instead of silently ignoring them."""
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is None:
                continue
            if float(group["lr"]) != d["lr_host"]:
                d["lr"].fill_(float(group["lr"]))
                d["lr_host"] = float(group["lr"])

    def hyper_by_value(self):
        """What a captured step bakes into its kernel arguments besides addresses: (betas, eps, weight_decay) of every group.  GraphedStep
        compares it before every replay and refuses a change; the optimiser itself never refuses to step (a sticky refusal here
        blocked eager stepping and the very re-capture the message asked for)."""
        return tuple((tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])) for g in self.param_groups)

    def load_state_dict(self, state_dict):
        """torch semantics; the flat step tensors are re-packed by the next step().  A GraphedStep captured before the reload notices the
        new state addresses in its step() and raises; the optimiser steps eagerly and can be captured again."""
        super().load_state_dict(state_dict)
        for d in self._dev.values():
            d["flat"] = None
            d["hyper"] = None

    def _flat_steps(self, d, plist, device):
        """One int64 tensor holding the step counts of `plist` (state[p]['step'] are views of it).  Re-packed when the state was
        loaded from a checkpoint (separate tensors / host ints) or the set of parameters with state changed."""
        flat = d["flat"]
        ok = flat is not None and flat.numel() == len(plist)
        if ok:
            base = flat.data_ptr()
            for i, p in enumerate(plist):
                st = self.state[p].get("step")
                if not torch.is_tensor(st) or st.data_ptr() != base + 8 * i:
                    ok = False
                    break
        if not ok:
            # (a GraphedStep captured earlier holds the OLD flat tensor and the exp_avg / exp_avg_sq addresses in its kernel arguments -- and
            # references to them, so the memory stays valid; its step() compares captured_state_ptrs() and refuses to replay)
            vals = []
            for p in plist:
                st = self.state[p].get("step", 0)
                vals.append(st.reshape(1).to(device=device, dtype=torch.int64) if torch.is_tensor(st)
                            else torch.full((1,), int(st), device=device, dtype=torch.int64))
            flat = torch.cat(vals) if vals else torch.zeros(0, device=device, dtype=torch.int64)
            for i, p in enumerate(plist):
                self.state[p]["step"] = flat[i:i + 1]
            d["flat"] = flat
        return flat

    def captured_state_ptrs(self):
        """Addresses a captured step has baked into its kernel arguments: per group the flat step tensor, per parameter exp_avg /
        exp_avg_sq (GraphedStep compares them before every replay)."""
        out = []
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            out.append(d["flat"].data_ptr() if d is not None and d.get("flat") is not None else 0)
            for p in group["params"]:
                st = self.state.get(p)
                if st and "exp_avg" in st:
                    out.append(st["exp_avg"].data_ptr())
                    out.append(st["exp_avg_sq"].data_ptr())
        return tuple(out)

    def written_tensors(self):
        """Every tensor step() writes through raw pointers (GraphedStep advances their version counters after a replay)."""
        return [p for g in self.param_groups for p in g["params"] if p in self.state]

    def _check(self, p):
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("FusedAdam handles contiguous fp32 parameters")
        if p.grad.is_sparse or p.grad.device != p.device or p.grad.dtype != torch.float32:
            raise RuntimeError("FusedAdam needs a dense fp32 gradient on the parameter's device")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            for p in live:
                self._check(p)
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            if self.capturable:
                dev = live[0].device
                d = self._group_dev(gi, group, dev)
                capturing = torch.cuda.is_current_stream_capturing()
                if not capturing:
                    self.sync_hyper()
                d["hyper"] = (tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
                with_state = [p for p in group["params"] if p in self.state and "exp_avg" in self.state[p]]
                flat = self._flat_steps(d, with_state, dev)
                if len(live) == len(with_state):
                    K.advance_steps(flat)                        # one launch for the whole group
                else:                                            # some parameters have no gradient this step: theirs do not advance
                    for p in live:
                        K.advance_steps(self.state[p]["step"])
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
                K.adam_multi_dev(live, grads, [self.state[p]["exp_avg"] for p in live], [self.state[p]["exp_avg_sq"] for p in live],
                                 [self.state[p]["step"] for p in live], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                 lr_dev=d["lr"])
                torch.autograd.graph.increment_version(live)
                continue
            for p in live:
                st = self.state[p]
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
    the (static) loss tensor.

    `also_written`: tensors other than the optimiser's parameters that the captured body writes through raw pointers (the EMA copy
    of the weights).  A replay changes them without passing through torch, so step() advances the version counter of every written
    tensor afterwards -- the weight packs of eager consumers (a D step between two replays, an evaluation of net_ema) are keyed on
    (data_ptr, _version) and would otherwise keep serving the weights of the capture."""

    def __init__(self, opt, body, warmup=2, also_written=(), capture_error_mode=None):
        if not getattr(opt, "capturable", False):
            raise RuntimeError("GraphedStep needs FusedAdam(capturable=True): the host-side step count cannot be replayed")
        self.opt = opt
        side = torch.cuda.Stream()
        self._stream = side                                     # warm-ups AND capture run on it (see below)
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
        if capture_error_mode is None:       # a live process group's watchdog thread polls events: "global" would fail the capture
            import torch.distributed as dist
            capture_error_mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        from .shard import quiesce_before_capture
        quiesce_before_capture()                                # (a live "nccl" group: its watchdog's list is empty before the capture)
        import os
        dump = os.environ.get("E4S_GRAPH_DEBUG_DUMP")           # diagnostics: write the captured graph as a DOT file (node kinds: the
        if dump:                                                # captured step must hold kernel nodes only, no MEMSET nodes -- kernels.sum_all)
            self.graph.enable_debug_mode()
        with K.flag_sink(flags):                                # this thread's mask checks accumulate here instead of syncing
            # ... on the SAME side stream as the warm-up steps.  Autograd's AccumulateGrad nodes keep the stream they were created on, and
            # nodes that outlive the warm-ups (anything still referencing last step's graph keeps them alive; torch warns "The
            # AccumulateGrad node's stream does not match ...") would otherwise run on a second stream inside the capture: their branch of
            # the replayed graph is ordered against the main branch only through the engine's event edges, not through the caching
            # allocator's per-stream reuse -- a train_G step captured that way came back with NaNs once all nine generator conv weights
            # were trainable (tools/debug_graphed_trainG.py, profiles/r05_graphed_trainG_bisect.json).  One stream: one branch.
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=capture_error_mode):
                self.loss = body()
        if dump:
            self.graph.debug_dump(dump)
        self.flags = flags
        self._written = list(opt.written_tensors()) + list(also_written)
        self._state_ptrs = opt.captured_state_ptrs()
        self._hyper = opt.hyper_by_value()
        # the graph's kernel arguments point INTO these tensors: holding them keeps the addresses from being handed to anybody else if the
        # optimiser drops them (load_state_dict), so that a stale replay can only be refused below, never scribble over foreign memory
        self._state_refs = [d.get("flat") for d in opt._dev.values()] + [t for st in opt.state.values() for t in st.values() if torch.is_tensor(t)]

    def step(self):
        self.opt.sync_hyper()                                   # a changed group["lr"] reaches the replay through device memory
        if self.opt.captured_state_ptrs() != self._state_ptrs:
            raise RuntimeError("GraphedStep: the optimiser's state tensors are not the ones this step was captured with "
                               "(load_state_dict / new state after the capture); re-capture the GraphedStep")
        if self.opt.hyper_by_value() != self._hyper:
            raise RuntimeError("GraphedStep: betas / eps / weight_decay changed after the step was captured (they travel by value in the "
                               "captured launches); re-capture the GraphedStep")
        self.graph.replay()
        self.steps_done += 1
        torch.autograd.graph.increment_version(self._written)
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


_FORK_STREAMS = {}


def forked_sum(base, terms, inputs=(), side_terms=None):
    """base + terms[0]() + terms[1]() + ... (added in that order), each term evaluated on ITS OWN side stream forked from the current
    one and joined before the sum.  For loss terms that are independent chains of small launches -- the LPIPS / identity / parsing
    networks of scripts/optimization.py:88-122 on one 1024^2 image: ~1 100 batch-1 launches that a single stream serialises (8.7 ms of a
    15.8 ms optimisation step; forked: 14.5 ms, same loss and same latent bit for bit after 200 replayed steps).  Works eagerly and inside
    a GraphedStep capture (the side streams fork from and join the capturing stream, so they are part of the capture; autograd runs each
    term's backward on the stream of its forward and orders the gradient hand-offs itself).
    inputs: tensors the terms read that were produced on the current stream (recorded on the side streams for the allocator).
    side_terms: indices of the terms that get a side stream (default: all); the others run on the current stream under them.  Inside a capture
    fewer branches can be faster than more: a replayed hipGraph pays for every cross-queue edge (config 3: the identity network alone on a side
    stream 12.0 ms per step, all three loss networks forked 12.9, none 14.5; the batch-2 train step is fastest as ONE chain)."""
    main = torch.cuda.current_stream()
    key = (main.device, len(terms))
    side = _FORK_STREAMS.get(key)
    if side is None:
        side = _FORK_STREAMS[key] = [torch.cuda.Stream(device=main.device) for _ in terms]
    if side_terms is None:
        env = os.environ.get("E4S_FORK_SIDE")                # (experiments: comma-separated indices of the terms that get a side stream)
        side_terms = range(len(terms)) if env is None else [int(t) for t in env.split(",") if t != ""]
    on_side = set(side_terms)
    parts = []
    for i, (fn, st) in enumerate(zip(terms, side)):
        if i not in on_side:
            parts.append(None)
            continue
        st.wait_stream(main)
        with torch.cuda.stream(st):
            for t in inputs:
                t.record_stream(st)
            parts[len(parts):] = [fn()]
    for i, fn in enumerate(terms):                            # the rest on the current stream, under the side branches
        if i not in on_side:
            parts[i] = fn()
    out = base
    for i, (part, st) in enumerate(zip(parts, side)):
        if i in on_side:
            main.wait_stream(st)
            part.record_stream(main)
        out = out + part
    return out
