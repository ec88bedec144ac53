"""src/criteria/lpips/lpips.py overlay (scripts/optimization.py:23): the native LPIPS (AlexNet)."""
from e4s_amd.criteria import LPIPS  # noqa: F401
