from e4s_amd.op.fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
