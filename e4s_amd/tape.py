"""Per-layer records ("tape") that the fused forward passes keep for their backward (autograd.GeneratorFn, encoder_autograd.EncoderFn),
with an optional REDUCED-PRECISION STORAGE of the recorded activations.

BASELINE.json configs[4] names a bf16 train step.  The reference trains fp32 (coach.py has no autocast) and so does the default here; with
`storage(torch.bfloat16)` active while a forward records its tape, every large fp32 activation a record holds for the backward (layer inputs
/ outputs: x, y, out, u1, r1, r2, sc) is STORED as bf16 (round to nearest even, one conversion per distinct tensor) and widened back to
fp32 when a backward reads it -- the arithmetic of both passes stays fp32 / split-bf16, master weights, Adam moments and gradients stay
fp32; only what is parked in HBM between the passes (and, with ddp.GradAverager(payload_dtype=torch.bfloat16), what crosses xGMI) is bf16.
Tensors saved through ctx.save_for_backward (the loss networks, the Discriminator families) are covered by the same context through
torch.autograd.graph.saved_tensors_hooks."""
import contextlib
import weakref

import torch

STORAGE_DTYPE = None                   # None: records keep the fp32 tensors themselves
ACT_KEYS = ("x", "y", "out", "u1", "r1", "r2", "sc", "c0")
MIN_NUMEL = 1 << 16                    # smaller tensors (styles, statistics, gates) stay fp32


class _Packed(dict):
    """A record whose large activations are stored in STORAGE_DTYPE; reading one widens it (the fp32 copy lives as long as its reader)."""

    def __init__(self, rec, cache, dtype):
        super().__init__(rec)
        self._lo = set()
        for k in ACT_KEYS:
            t = rec.get(k)
            if torch.is_tensor(t) and t.dtype == torch.float32 and t.is_cuda and t.numel() >= MIN_NUMEL and not t.requires_grad:
                # one conversion per distinct tensor (a layer's output is the next layer's input).  Keyed on the OBJECT, checked through a
                # weak reference: an address-keyed cache served the bf16 image of a FREED activation to the next one the allocator placed
                # at the same address with the same shape (every unit of the encoder) -- gradients off by 100 %
                hit = cache.get(id(t))
                q = hit[1] if hit is not None and hit[0]() is t and hit[2] == t._version else None
                if q is None:
                    q = t.to(dtype)
                    cache[id(t)] = (weakref.ref(t), q, t._version)
                dict.__setitem__(self, k, q)
                self._lo.add(k)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v.float() if k in self._lo else v

    def get(self, k, default=None):
        return self[k] if k in self else default


class Tape(list):
    def __init__(self):
        super().__init__()
        self._cache = {}

    def append(self, rec):
        super().append(pack(rec, self._cache))


def pack(rec, cache=None):
    """rec as it should be kept for the backward under the storage policy of the moment."""
    if STORAGE_DTYPE is None:
        return rec
    return _Packed(rec, cache if cache is not None else {}, STORAGE_DTYPE)


@contextlib.contextmanager
def storage(dtype=torch.bfloat16):
    """Activations recorded for the backward inside this context are stored as `dtype` (see the module docstring)."""
    global STORAGE_DTYPE
    if dtype not in (None, torch.bfloat16, torch.float16):
        raise ValueError("tape.storage: bf16 / fp16 / None")
    saved = STORAGE_DTYPE
    STORAGE_DTYPE = dtype

    def to_lo(t):
        if dtype is not None and t.is_cuda and t.dtype == torch.float32 and t.numel() >= MIN_NUMEL and not isinstance(t, torch.nn.Parameter) \
                and not getattr(t, "_e4s_keep_fp32", False):
            return (t.to(dtype), True)
        return (t, False)

    def to_hi(packed):
        t, lo = packed
        return t.float() if lo else t
    try:
        with torch.autograd.graph.saved_tensors_hooks(to_lo, to_hi):
            yield
    finally:
        STORAGE_DTYPE = saved
