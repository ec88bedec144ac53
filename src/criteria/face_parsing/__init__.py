from ..._overlay import extend as _extend

_extend(__path__, "criteria/face_parsing")
