"""fused_leaky_relu / FusedLeakyReLU on the HIP kernel e4s_fused_bias_act_f32.

Mirrors the reference operator (src/models/stylegan2/op/fused_act.py:18-85): same signature,
same autograd structure (backward gates on the sign of the saved OUTPUT and re-applies `scale`,
fused_bias_act_kernel.cu:43,47; grad_bias = grad_input summed over all dims but 1, :33-38;
double backward, :41-47).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import kernels as K


class _FusedLeakyReLUBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = K.fused_bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
        if grad_input.ndim >= 2:
            grad_bias = K.channel_sum(grad_input)
        else:
            grad_bias = grad_input.sum(0)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        gg = K.fused_bias_act(gradgrad_input.contiguous(), gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gg, None, None, None


class _FusedLeakyReLU(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = K.fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        grad_input, grad_bias = _FusedLeakyReLUBackward.apply(grad_output.contiguous(), out, ctx.negative_slope,
                                                              ctx.scale)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _FusedLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
