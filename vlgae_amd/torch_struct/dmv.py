"""`from ...torch_struct.dmv import LEFT, RIGHT, NOCHILD, ...` keeps working for callers of the reference
layout (src/model/ldndmv.py:22, src/model/dmv_helper/*.py); the values live in constants.py."""
from .constants import DEC_NUM, DIR_NUM, GO, HASCHILD, LEFT, NOCHILD, RIGHT, STOP, VAL_NUM  # noqa: F401
