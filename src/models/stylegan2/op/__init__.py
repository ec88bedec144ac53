from e4s_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix  # noqa: F401
