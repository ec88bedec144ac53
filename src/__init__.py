"""Import-path shim: the reference's scripts do `from src.models.networks import Net3`
(scripts/face_swap.py:26, scripts/optimization.py:19, scripts/face_edit.py:11, src/training/coach.py:24-25).
These modules re-export the MI355X-native implementations in `e4s_amd` under the same paths so those
scripts drop in unchanged.  Nothing is implemented here."""
