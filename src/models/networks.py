from e4s_amd.networks import LocalMLP, Net3  # noqa: F401
