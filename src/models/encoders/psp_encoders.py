from e4s_amd.encoders import FSEncoder_PSP  # noqa: F401
