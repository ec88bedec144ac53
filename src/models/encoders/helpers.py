from e4s_amd.encoders import Bottleneck, SEModule, bottleneck_IR_SE_Ours, get_block  # noqa: F401
