"""src/criteria/face_parsing/face_parsing_loss.py overlay (scripts/optimization.py:25): the native FaceParsingLoss."""
from e4s_amd.criteria import FaceParsingLoss  # noqa: F401
