"""src/models/encoders/helpers.py overlay: the reference's helpers (Flatten, l2_norm, bottleneck_IR, bottleneck_IR_SE,
get_blocks -- used by the ID-loss backbone, src/models/encoders/model_irse.py:2, and the parsing UNet,
src/criteria/face_parsing/unet.py:4) stay the reference's own; the hot-path block types are the native ones."""
from ..._overlay import exec_reference_module as _exec

_exec("models/encoders/helpers.py", globals())
from e4s_amd.encoders import Bottleneck, SEModule, bottleneck_IR_SE_Ours, get_block  # noqa: E402,F401
