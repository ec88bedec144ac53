"""The tensor helpers the face-swap / training scripts take from src/utils/torch_utils.py that sit on the
hot path's input side (labelMap2OneHot :166-172) or its checkpoint plumbing (:175-194)."""
import torch


def labelMap2OneHot(label, num_cls):
    """[B,1,H,W] int64 label map -> one-hot float [B,num_cls,H,W]."""
    b, _, h, w = label.size()
    return torch.zeros(b, num_cls, h, w, device=label.device).scatter_(1, label, 1.0)


def remove_module_prefix(state_dict, prefix):
    return {k.replace(prefix, "", 1): v for k, v in state_dict.items()}


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def accumulate(model1, model2, decay=0.999):
    p2 = dict(model2.named_parameters())
    for k, p in model1.named_parameters():
        p.data.mul_(decay).add_(p2[k].data, alpha=1 - decay)
