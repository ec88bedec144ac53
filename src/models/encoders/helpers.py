"""src/models/encoders/helpers.py overlay: the reference's helpers (Flatten, l2_norm, SEModule, bottleneck_IR, get_blocks
-- used by the parsing UNet, src/criteria/face_parsing/unet.py:4) stay the reference's own; the block types on the hot path
(the regional encoder's bottleneck_IR_SE_Ours, the identity loss's bottleneck_IR_SE) are the native ones."""
from ..._overlay import exec_reference_module as _exec

if not _exec("models/encoders/helpers.py", globals()):
    from e4s_amd.encoders import Bottleneck, SEModule, get_block  # noqa: F401  (no reference checkout: native holders)
    from e4s_amd.criteria import Flatten  # noqa: F401
from e4s_amd.encoders import bottleneck_IR_SE_Ours  # noqa: E402,F401
from e4s_amd.criteria import bottleneck_IR_SE  # noqa: E402,F401
