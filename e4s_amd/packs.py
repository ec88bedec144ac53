"""Cache keys of the re-packed weight images (tap-packed, polyphase, split-bf16, stacked LocalMLPs).

A pack is valid for one (storage, version) of its source parameters.  `Tensor._version` advances on every in-place
update made THROUGH the tensor (optimizer steps, `load_state_dict`, `p.copy_()` under no_grad) but NOT on updates made
through `p.data` (the reference's EMA, src/utils/torch_utils.py:189-194, writes `p.data.mul_().add_()`).  Code that
mutates weights behind autograd's back must call `invalidate_packs()` afterwards (the overlay's
`src.utils.torch_utils.accumulate` does); it bumps a process-wide generation that is part of every key.  A captured
HIP graph (networks.GraphedFaceSwap) bakes the pack pointers in: re-capture it after any weight change."""

_GENERATION = 0


def invalidate_packs():
    """Drop every cached weight pack of every module (they are rebuilt on next use)."""
    global _GENERATION
    _GENERATION += 1


def param_key(*tensors):
    return (_GENERATION,) + tuple((t.data_ptr(), t._version) for t in tensors)
