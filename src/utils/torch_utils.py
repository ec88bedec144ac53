"""src/utils/torch_utils.py overlay.  The reference's helpers (tensor2im, tensor2map, vis_faces, ... :10-165) are used as
they are when the reference checkout and its third-party imports (torchvision, matplotlib) are available; the four
functions on the hot path's input side / checkpoint plumbing are (re)defined here, `accumulate` in a form that advances
the parameters' version counters: the reference's `par.data.mul_()` (:189-194) leaves `_version` unchanged, and the
native modules key their packed-weight caches on (data_ptr, _version)."""
import torch

from .._overlay import exec_reference_module as _exec

_exec("utils/torch_utils.py", globals())


def labelMap2OneHot(label, num_cls):
    """[B,1,H,W] int64 label map -> one-hot float [B,num_cls,H,W]  (torch_utils.py:166-172)."""
    b, _, h, w = label.size()
    return torch.zeros(b, num_cls, h, w, device=label.device).scatter_(1, label, 1.0)


def remove_module_prefix(state_dict, prefix):
    return {k.replace(prefix, "", 1): v for k, v in state_dict.items()}


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def accumulate(model1, model2, decay=0.999):
    """EMA of the parameters (coach.py:67,396), in place THROUGH the parameter so `_version` advances."""
    p2 = dict(model2.named_parameters())
    with torch.no_grad():
        for k, p in model1.named_parameters():
            p.mul_(decay).add_(p2[k].detach(), alpha=1 - decay)
    from e4s_amd.packs import invalidate_packs
    invalidate_packs()
