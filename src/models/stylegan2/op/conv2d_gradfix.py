from e4s_amd.op.conv2d_gradfix import conv2d, conv_transpose2d, no_weight_gradients  # noqa: F401
