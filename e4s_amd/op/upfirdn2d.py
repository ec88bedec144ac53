"""upfirdn2d on the HIP kernel e4s_upfirdn2d_f32.

Mirrors src/models/stylegan2/op/upfirdn2d.py:17-147: input [N,C,H,W] is viewed as
[N*C, H, W, 1]; backward is the same op with the flipped kernel, up<->down swapped and the
g_pad of :108-113; double backward re-applies the forward op (:59-82).
"""
import torch
from torch.autograd import Function

from .. import kernels as K


class _UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        gx0, gx1, gy0, gy1 = g_pad
        g = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        gi = K.upfirdn2d_raw(g, grad_kernel, down_x, down_y, up_x, up_y, gx0, gx1, gy0, gy1)
        gi = gi.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.in_size, ctx.out_size = in_size, out_size
        return gi

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        gg = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
        px0, px1, py0, py1 = ctx.pad
        out = K.upfirdn2d_raw(gg, kernel, ctx.up[0], ctx.up[1], ctx.down[0], ctx.down[1], px0, px1, py0, py1)
        out = out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
        return out, None, None, None, None, None, None, None, None


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        x = input.reshape(-1, in_h, in_w, 1)
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
        out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (px0, px1, py0, py1)
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        out = K.upfirdn2d_raw(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        gi = _UpFirDn2dBackward.apply(grad_output.contiguous(), kernel, grad_kernel, ctx.up, ctx.down, ctx.pad,
                                      ctx.g_pad, ctx.in_size, ctx.out_size)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
