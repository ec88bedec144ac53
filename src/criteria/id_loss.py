"""src/criteria/id_loss.py overlay (scripts/optimization.py:24, src/training/coach.py): the native IDLoss."""
from e4s_amd.criteria import IDLoss  # noqa: F401
