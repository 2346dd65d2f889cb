"""Index constants of the DMV potentials -- same names and values as the reference
(src/model/torch_struct/dmv.py:7-15); imported by callers such as ldndmv.py:22 and dmv_helper/*."""
NOCHILD = 1
HASCHILD = 0
LEFT = 0
RIGHT = 1
GO = 0
STOP = 1
DIR_NUM = 2
VAL_NUM = 2
DEC_NUM = 2
