"""Drop-in for the reference's `src/model/torch_struct/distributions.py` on the path VLGAE uses.

Same class names, constructor signatures, lazy properties, shapes and constants as the reference
(distributions.py:25-53,116-133,162-174,190-193,245-298), so `from vlgae_amd.torch_struct import
DMV1o, DependencyCRF` replaces `from src.model.torch_struct import DMV1o, DependencyCRF`
(ldndmv.py:21, joint.py:20, dmv.py:16) with no other change.  Every property evaluates through the
HIP kernels; there is no PyTorch implementation of the DP in this package.
"""
import torch
from torch import Tensor
from torch.distributions.distribution import Distribution
from torch.distributions.utils import lazy_property

from . import functional as F
from .dmv import NOCHILD, RIGHT  # noqa: F401  (re-exported like the reference)
from .semirings import NEGINF, LogSemiring, MaxSemiring


def _out_of_scope(name):
    raise NotImplementedError(
        f"StructDistribution.{name}: no caller in VLGAE uses it (only partition / max / argmax / marginals are "
        "reached; SURVEY.md section 2) -- outside the MI355X hot path this package implements.")


class StructDistribution(Distribution):
    """Base structured distribution (reference: distributions.py:25-243)."""

    has_enumerate_support = False
    struct = None

    def __init__(self, log_potentials, lengths=None, args={}):
        batch_shape = log_potentials.shape[:1]
        event_shape = log_potentials.shape[1:]
        self.log_potentials = log_potentials
        self.lengths = lengths
        self.args = args
        super().__init__(batch_shape=batch_shape, event_shape=event_shape, validate_args=False)

    # -- the four quantities VLGAE reads ------------------------------------------------------------
    def _sum(self, semiring):
        raise NotImplementedError

    def _marginals(self, semiring):
        raise NotImplementedError

    @lazy_property
    def partition(self):
        "Log-partition function (distributions.py:190-193)."
        return self._sum(LogSemiring)

    @lazy_property
    def max(self):
        "Score of the best structure (distributions.py:116-124)."
        return self._sum(MaxSemiring)

    @lazy_property
    def argmax(self):
        "Best structure as 0/1 parts = Max-semiring marginals (distributions.py:126-133)."
        return self._marginals(MaxSemiring)

    @lazy_property
    def marginals(self):
        "Posterior marginals of the parts (distributions.py:162-174)."
        return self._marginals(LogSemiring)

    @lazy_property
    def mode(self):
        return self.argmax

    # -- reference API with no caller in VLGAE ------------------------------------------------------
    @property
    def entropy(self):
        _out_of_scope("entropy")

    def cross_entropy(self, other):
        _out_of_scope("cross_entropy")

    def kl(self, other):
        _out_of_scope("kl")

    def risk(self, cost):
        _out_of_scope("risk")

    def kmax(self, k):
        _out_of_scope("kmax")

    def topk(self, k):
        _out_of_scope("topk")

    @property
    def count(self):
        _out_of_scope("count")

    def gumbel_crf(self, temperature=1.0):
        _out_of_scope("gumbel_crf")

    def sample(self, sample_shape=torch.Size()):
        _out_of_scope("sample")

    def enumerate_support(self, expand=True):
        _out_of_scope("enumerate_support")


class DMV1o(StructDistribution):
    """First-order DMV with valence (reference: distributions.py:245-265, dmv.py:18-69).

    log_potentials = [dec [B,N,2,2,2], attach [B,N,N,2]] (root-merged, see `merge`); lengths [B].
    partition / max: [B,1].  argmax / marginals: the attach part only, [B,N,N,2] (dmv.py:68-69).
    """

    def __init__(self, log_potentials, lengths, args={}):
        super().__init__(log_potentials[0], lengths=lengths, args=args)
        self.log_potentials = log_potentials

    def _sum(self, semiring):
        dec, attach = self.log_potentials
        return F.dmv1o_sum(dec, attach, self.lengths, semiring.kernel_id)

    def _marginals(self, semiring):
        # helpers.py:118-154 differentiates the semiring sum w.r.t. the potentials and keeps the attach
        # part; the fused kernel returns exactly that gradient.  (Not differentiable a second time:
        # no caller does -- inputs are always detached leaves, joint.py:252-253.)
        dec, attach = self.log_potentials
        _, _, gatt = F.dmv1o_run(dec, attach, self.lengths, semiring.kernel_id, True, want_dec=False)
        return gatt.to(attach.dtype) if attach.dtype == torch.float64 else gatt

    @lazy_property
    def argmax_heads(self):
        """Extension (not in the reference): the best tree as `predicted` heads [B,N], computed on the device.
        Same result as `arc = self.argmax.sum(-1).nonzero(); predicted[arc[:,0], arc[:,2]] = arc[:,1]`
        (joint.py:256-258) without the host synchronisation of `nonzero()`."""
        dec, attach = self.log_potentials
        return F.dmv1o_decode(dec, attach, self.lengths)[1]

    def marginals_and_heads(self, keep_viterbi=False):
        """Extension: (`marginals`, `argmax_heads`) with the two DPs overlapped on two HIP streams -- the pair
        lang_feat_max_tree asks for every step (joint.py:251-258).  keep_viterbi=True: the Viterbi pass also produces the
        tree counts and is remembered, so a later `DMV1o(same potentials).max` (the parser's `-max` loss, ldndmv.py:277-281)
        and its backward launch nothing (see functional.dmv1o_marginals_and_heads)."""
        dec, attach = self.log_potentials
        _, gatt, heads = F.dmv1o_marginals_and_heads(dec, attach, self.lengths, keep_viterbi)
        return gatt, heads

    def marginals_and_heads_async(self, keep_viterbi=False):
        """`marginals_and_heads` with BOTH DPs on side streams: returns a handle at once; `handle.wait()` -> (logZ, marginals,
        heads) joins them into the current stream.  Work enqueued in between overlaps the DPs (functional.dmv1o_structure_async)."""
        dec, attach = self.log_potentials
        return F.dmv1o_structure_async(dec, attach, self.lengths, keep_viterbi)

    @staticmethod
    def merge(dec: Tensor, attach: Tensor, root: Tensor, one=0, zero=NEGINF):
        """Root-augmented potentials (distributions.py:253-265): the root is token 0, generates only to
        the right, takes exactly the `root` scores with valence NOCHILD; everything else is `zero`."""
        return F.dmv1o_merge_autograd(dec, attach, root, one, zero)


class DMV1oRules(StructDistribution):
    """Extension: the DMV of `DMV1o`, parameterised by the scorer's rule tables instead of
    merged per-position potentials.  It replaces this block of DiscriminativeNDMV._forward + loss
    (src/model/ldndmv.py:185-209, 277-281)

        attach = attach_rule.gather(2, token...)  -> tril/triu direction select -> function mask
        root   = root_rule.gather(1, token);  merged = DMV1o.merge(dec, attach, root);  DMV1o(merged, lengths)

    with one kernel launch that gathers inside its load stage and returns gradients in rule space.

      attach_rule [B,L,T,2(dir),2(valence)]   log-probs over the T child tokens (ldndmv.py:185)
      dec         [B,L,2,2,2]                 root_rule [T] / [1,T] / [B,T]      token [B,L] int64
      head_mask   [B,L] bool, True = the word takes no children (`function_mask`, ldndmv.py:194-198)
    partition / max: [B,1], differentiable w.r.t. attach_rule, dec, root_rule.  argmax_heads: [B,L+1]."""

    def __init__(self, attach_rule, dec, root_rule, token, lengths, head_mask=None, mask_fill=-1e20, args={}):
        super().__init__(dec, lengths=lengths, args=args)
        self.log_potentials = [attach_rule, dec, root_rule]
        self.token, self.head_mask, self.mask_fill = token, head_mask, mask_fill

    def _sum(self, semiring):
        a, d, r = self.log_potentials
        return F.dmv1o_rules_sum(a, d, r, self.token, self.lengths, self.head_mask, semiring.kernel_id, self.mask_fill)

    def _marginals(self, semiring):
        "Rule-space expected counts (Log) or best-tree rule indicators (Max): the attach_rule part."
        a, d, r = self.log_potentials
        return F.dmv1o_rules_run(a, d, r, self.token, self.lengths, semiring.kernel_id, True, self.head_mask,
                                 False, self.mask_fill)["grad_rule"]

    @lazy_property
    def argmax_heads(self):
        a, d, r = self.log_potentials
        return F.dmv1o_rules_run(a, d, r, self.token, self.lengths, MaxSemiring.kernel_id, False, self.head_mask, True,
                                 self.mask_fill)["heads"]


class DependencyCRF(StructDistribution):
    """Projective single-root dependency CRF (reference: distributions.py:269-298, deptree.py:14-76).

    log_potentials [B,N,N] head -> child with the root at index 0; lengths [B] or None (= N-1).
    partition / max: [B].  argmax / marginals: [B,N,N].
    """

    def __init__(self, log_potentials, lengths=None, args={}, multiroot=False):
        super().__init__(log_potentials, lengths, args)
        assert not multiroot, "multiroot is asserted False by the reference's DP (deptree.py:26-27)"
        self.multiroot = multiroot

    def _sum(self, semiring):
        return F.deptree_sum(self.log_potentials, self.lengths, semiring.kernel_id)

    def _marginals(self, semiring):
        arc = self.log_potentials
        _, garc = F.deptree_run(arc, self.lengths, semiring.kernel_id, True)
        return garc.to(arc.dtype) if arc.dtype == torch.float64 else garc

    @lazy_property
    def argmax_heads(self):
        "Extension: best tree as heads [B,N] on the device (see DMV1o.argmax_heads; MBR decode of ldndmv.py:294-303)."
        return F.deptree_decode(self.log_potentials, self.lengths)[1]

    def log_prob(self, value):
        "log p(tree) for 0/1 arc indicators `value` [..., B, N, N] (distributions.py:55-75)."
        score = (self.log_potentials * value.type_as(self.log_potentials)).flatten(-2).sum(-1)
        return score - self.partition
