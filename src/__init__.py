"""Import-path overlay: the reference's scripts do `from src.models.networks import Net3`
(scripts/face_swap.py:26, scripts/optimization.py:19, scripts/face_edit.py:11, src/training/coach.py:24-25).
The hot-path modules under this package re-export the MI355X-native implementations in `e4s_amd`; every other
`src.*` import falls through to the reference checkout at $E4S_REFERENCE_ROOT (see src/_overlay.py), so the scripts
run unchanged with this repo first on sys.path.  Nothing is implemented here."""
from ._overlay import extend as _extend

_extend(__path__, "")
