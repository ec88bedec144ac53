"""conv2d_gradfix boundary (src/models/stylegan2/op/conv2d_gradfix.py:22-92).

In the reference the custom gradient path is only enabled for torch 1.7/1.8
(`could_use_op`, :85-92); on every other version -- including the pinned 1.12.1 -- it falls
through to F.conv2d / F.conv_transpose2d (:34-42, :78-83).  The API (including the
`no_weight_gradients()` context DR1Loss enters, src/criteria/adv_loss.py:34) is kept; these
entry points serve the Discriminator (config 5), which is not on the generator hot path.
"""
import contextlib

import torch.nn.functional as F

enabled = False
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation,
                    groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return F.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                              output_padding=output_padding, groups=groups, dilation=dilation)
