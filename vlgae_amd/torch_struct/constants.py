"""Index conventions of the DMV potentials (same meaning and values as the reference's module-level
constants, src/model/torch_struct/dmv.py:7-15):

    dec    [B, N, direction, valence, decision]      attach [B, N(head), N(child), valence]

direction: LEFT / RIGHT of the head.  valence: whether the head has ALREADY generated a child further out
on that side (children are generated outside-in).  decision: GO on generating another child, or STOP.
"""
HASCHILD, NOCHILD = 0, 1      # valence
LEFT, RIGHT = 0, 1            # direction
GO, STOP = 0, 1               # decision
DIR_NUM = VAL_NUM = DEC_NUM = 2
