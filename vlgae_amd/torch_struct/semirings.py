"""Semiring tags.  The reference's semirings are classes of torch ops (semirings/semirings.py:127-207);
here the two that any caller selects -- Log and Max -- are template parameters of the HIP kernels and
these classes only carry the tag.  The other reference semirings (Std, KMax, KL, CrossEntropy, Entropy,
TempMax, Risk, Sampled, SparseMax, Checkpoint*, Fast*) are never selected by any caller in the reference
 and are out of scope."""

NEGINF = -1e12  # semirings.py:16 -- finite sentinel, also the kernels' VLG_NEGINF


class Semiring:
    zero = None
    one = 0.0
    kernel_id = None

    @classmethod
    def size(cls):
        return 1


class LogSemiring(Semiring):
    """(logsumexp, +, NEGINF, 0): gradients of the total are marginals (semirings.py:173-184)."""
    zero = NEGINF
    kernel_id = 0


class MaxSemiring(Semiring):
    """(max, +, NEGINF, 0): gradients of the total are the arg-max structure (semirings.py:187-207)."""
    zero = NEGINF
    kernel_id = 1
