"""``fused_leaky_relu`` / ``FusedLeakyReLU`` backed by the HIP kernel ``e4s_fused_bias_act_f32``.

Operator contract of the reference (src/models/stylegan2/op/fused_act.py:50-85, native kernel
fused_bias_act_kernel.cu:19-49):  y = gain * leaky_relu(x + bias[channel], slope).
Differentiable twice, like the reference (:18-47): the gradient gates on the sign of the saved OUTPUT and
re-applies the gain (kernel mode act=3, grad=1, .cu:43,47), and d/d(bias) is that gradient summed over every
axis but the channel axis (:33-38), here by ``e4s_channel_sum_f32`` instead of ``Tensor.sum``.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import kernels as K

_LRELU, _FWD, _GRAD = 3, 0, 1            # (act, grad) codes of the native op


def _bias_grad(g):
    """Sum over all axes except axis 1 (1-D/2-D inputs have no spatial axes)."""
    return K.channel_sum(g) if g.ndim >= 2 else g.sum(0)


class _GatedGrad(Function):
    """g_in = gain * (out > 0 ? g : slope * g); its own derivative w.r.t. g is the same gating, which is what a
    second backward (R1 / path-length regularisers, src/criteria/adv_loss.py:34-60) needs."""

    @staticmethod
    def forward(ctx, g, out, slope, gain):
        ctx.slope, ctx.gain = slope, gain
        ctx.save_for_backward(out)
        g_in = K.fused_bias_act(g, None, out, _LRELU, _GRAD, slope, gain)
        return g_in, _bias_grad(g_in)

    @staticmethod
    def backward(ctx, gg_in, gg_bias):
        (out,) = ctx.saved_tensors
        # d(g_in)/d(g) and d(g_bias)/d(g) share the gate: feed gg_in + broadcast(gg_bias) through the same kernel
        gg = K.fused_bias_act(gg_in.contiguous(), gg_bias, out, _LRELU, _GRAD, ctx.slope, ctx.gain)
        return gg, None, None, None


class _BiasLeakyReLU(Function):
    @staticmethod
    def forward(ctx, x, bias, slope, gain):
        y = K.fused_bias_act(x, bias, None, _LRELU, _FWD, slope, gain)
        ctx.slope, ctx.gain = slope, gain
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gx, gb = _GatedGrad.apply(g.contiguous(), y, ctx.slope, ctx.gain)
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """Holds the per-channel ``bias`` parameter (state_dict key ``...activate.bias``)."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
