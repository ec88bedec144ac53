"""src/models/encoders/model_irse.py overlay: the IR-SE50 backbone of the identity loss, native (e4s_amd.criteria)."""
from e4s_amd.criteria import Backbone  # noqa: F401
