"""Operator boundary -- the same names the reference exports from src/models/stylegan2/op/__init__.py:1-2,
backed by libe4s_hip.so instead of JIT-compiled CUDA extensions."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from . import conv2d_gradfix

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "conv2d_gradfix"]
