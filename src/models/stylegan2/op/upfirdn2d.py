from e4s_amd.op.upfirdn2d import upfirdn2d  # noqa: F401
