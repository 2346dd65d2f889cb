"""ctypes binding of the C ABI in include/vlgae_amd.h.

The HIP library is the ONLY compute path of this package: if it is missing or fails to load,
importing an op raises.  There is no CPU / eager-PyTorch fallback.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLGAE_AMD_LIB") or os.path.join(_PKG, "_lib", "libvlgae_amd.so")

F32, BF16 = 0, 1
SEMIRING_LOG, SEMIRING_MAX = 0, 1
OP_DMV1O_INSIDE, OP_DMV1O_INSIDE_OUTSIDE, OP_DEPTREE_INSIDE, OP_DEPTREE_INSIDE_OUTSIDE = 0, 1, 2, 3

_vp, _i, _f, _sz, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_longlong

class SmallGemm(ctypes.Structure):
    """VlgSmallGemm of include/vlgae_amd.h: one problem of vlg_small_gemm_group."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("a", "b", "bias", "u", "v", "c")]
                + [(n, ctypes.c_longlong) for n in ("sab", "sam", "sak", "sbb", "sbk", "sbn", "scb", "ldc", "sbias", "su", "sv")]
                + [(n, ctypes.c_int) for n in ("batch", "M", "N", "K", "accumulate", "in_dtype", "out_dtype")] + [("alpha", ctypes.c_float)])


class WgradReduce(ctypes.Structure):
    """VlgWgradReduce of include/vlgae_amd.h: one deferred split-K reduction of vlg_linear_wgrad_reduce_group."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("ws", "d_weight", "d_bias", "x_colsum")] + [(n, ctypes.c_int) for n in ("K", "M", "N", "ld_dw", "out_dtype", "in_dtype")]


class FfStage(ctypes.Structure):
    """VlgFfStage of include/vlgae_amd.h: one stage of vlg_ff_linear_act_chain2."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("w", "bias", "mask", "rng", "out", "act", "sum")] + [("mask_scale", ctypes.c_float), ("p", ctypes.c_float),
                ("site", ctypes.c_uint), ("J", ctypes.c_int), ("swap", ctypes.c_int), ("accumulate", ctypes.c_int)])


class WgradPartial(ctypes.Structure):
    """VlgWgradPartial of include/vlgae_amd.h: one split-K product of vlg_linear_wgrad_partial_group."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("dy", "x", "ws")] + [("ws_bytes", ctypes.c_size_t)]
                + [(n, ctypes.c_int) for n in ("ld_dy", "ld_x", "K", "M", "N", "in_dtype", "want_bias", "want_x_colsum")])


# symbol -> (restype, argtypes); one entry per declaration in include/vlgae_amd.h
SIGNATURES = {
    "vlg_dmv1o_inside": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vlg_dmv1o_inside_outside": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vlg_deptree_inside": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vlg_deptree_inside_outside": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vlg_dmv1o_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "vlg_deptree_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "vlg_dmv1o_marginals_viterbi_supported": (_i, [_i]),
    "vlg_dmv1o_marginals_viterbi": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vlg_dmv1o_rules": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             _sz, _vp]),
    "vlg_dmv1o_merge": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "vlg_dmv1o_count_sum": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "vlg_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vlg_bilinear_align": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "vlg_bilinear_align_backward_workspace": (_sz, [_i, _i, _i, _i, _i, _i]),
    "vlg_bilinear_align_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "vlg_grounding_loss_workspace": (_sz, [_i, _i, _i]),
    "vlg_grounding_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _sz, _vp, _vp,
                                _vp, _vp]),
    "vlg_align_reduced_workspace": (_sz, [_i, _i]),
    "vlg_align_reduced": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp, _vp]),
    "vlg_align_reduced_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "vlg_grounding_decode_workspace": (_sz, [_i, _i]),
    "vlg_grounding_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "vlg_trilinear": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "vlg_trilinear_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "vlg_trilinear_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "vlg_trilinear_backward_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "vlg_trilinear_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "vlg_trilinear_backward_g": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "vlg_attn_fuse_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "vlg_attn_fuse_saved_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "vlg_attn_fuse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "vlg_attn_fuse_backward_workspace": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "vlg_attn_fuse_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp]),
    "vlg_box_rel_pairwise": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "vlg_box_rel_pairwise_backward_workspace": (_sz, [_i, _i, _i]),
    "vlg_box_rel_pairwise_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp, _vp, _vp]),
    "vlg_linear_wgrad_workspace": (_sz, [_i, _i, _i]),
    "vlg_linear_wgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp, _i, _vp, _vp, _vp]),
    "vlg_dropout": (_i, [_vp, _vp, _i, _vp, ctypes.c_uint, _f, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "vlg_rng_advance": (_i, [_vp, _vp]),
    "vlg_vis_encoder": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "vlg_vis_encoder_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "vlg_linear_wgrad_partial": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _i, _vp]),
    "vlg_linear_wgrad_reduce_group": (_i, [_vp, _i, _vp]),
    "vlg_linear_wgrad_partial_group": (_i, [_vp, _i, _vp]),
    "vlg_langfeat_root_cat": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "vlg_langfeat_root_cat_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "vlg_langfeat_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "vlg_langfeat_split_backward": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "vlg_langfeat_marginal": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "vlg_langfeat_arc_out": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vlg_langfeat_rowscale": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "vlg_small_gemm": (_i, [_vp, _ll, _ll, _ll, _vp, _ll, _ll, _ll, _vp, _ll, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "vlg_small_gemm_group": (_i, [_vp, _i, _vp]),
    "vlg_ff_context_mean": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "vlg_ff_mlp_act": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "vlg_ff_act": (_i, [_vp, _vp, _vp, _f, _vp, ctypes.c_uint, _f, _vp, _ll, _i, _i, _i, _i, _f, _vp]),
    "vlg_ff_act_backward": (_i, [_vp, _vp, _vp, _f, _vp, ctypes.c_uint, _f, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _vp]),
    "vlg_ff_root_rule": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "vlg_ff_root_rule_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "vlg_ff_linear_act": (_i, [_vp, _i, _vp, _vp, _ll, _i, _vp, _i, _i, _i, _vp, _f, _vp, ctypes.c_uint, _f, _vp, _f, _vp]),
    "vlg_ff_linear_act_backward": (_i, [_vp, _i, _vp, _i, _i, _ll, _i, _vp, _vp, _f, _vp, ctypes.c_uint, _f, _vp, _vp, _i, _i, _f, _vp]),
    "vlg_ff_linear_act_chain2": (_i, [_vp, _i, _ll, _i, _vp, _vp, _f, _vp]),
    "vlg_ff_linear_kn": (_i, [_vp, _i, _vp, _i, _ll, _i, _vp, ctypes.c_uint, _f, _vp, _i, _vp]),
    "vlg_ff_linear_mlp_act_backward": (_i, [_vp, _i, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _i, _vp, _f, _vp]),
    "vlg_ff_transpose256": (_i, [_vp, _i, _vp, _vp]),
    "vlg_dropout_mask": (_i, [_vp, ctypes.c_uint, _f, _vp, _ll, _vp]),
    "vlg_ff_mlp_act_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "vlg_ndmv_potentials": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "vlg_ndmv_potentials_backward_workspace": (_sz, [_i, _i, _i, _i]),
    "vlg_ndmv_potentials_backward": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp, _i, _vp,
                                          _vp, _i, _vp, _vp, _vp]),
    "vlg_dmv1o_viterbi": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vlg_scale_counts": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "vlg_feed_kmeans": (_i, [_vp, ctypes.c_int64, _vp, _i, _i, _vp, _vp, _vp]),
    "vlg_feed_batches": (_i, [_vp, ctypes.c_int64, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "vlg_feed_npy_shape": (_i, [ctypes.c_char_p, _vp, _vp]),
    "vlg_feed_collate_npy": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i]),
    "vlg_selftest_xlane": (_i, [_vp, _vp]),
    "vlg_last_error": (ctypes.c_char_p, []),
    "vlg_version": (_i, []),
}

_lib = None
ABI_VERSION = 144   # what include/vlgae_amd.h declares at this revision; lib() refuses any other library (argument lists differ between versions)


def lib():
    """Load libvlgae_amd.so (once).  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"vlgae_amd: HIP extension not built ({LIB_PATH} missing). Run `python -m vlgae_amd.build` "
                "(needs hipcc, cross-compiles for gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        partial = bool(os.environ.get("VLGAE_AMD_LIB"))   # tools/ A/B variants may hold a subset of the translation units
        for name, (res, args) in SIGNATURES.items():
            if partial and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)   # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        # same symbol names, different argument lists across ABI versions: an older library (a tools/_v A/B variant through VLGAE_AMD_LIB, a
        # stale build) would be called with the wrong pointers and strides -- refuse it, in partial mode too
        got = handle.vlg_version() if hasattr(handle, "vlg_version") else None
        if got != ABI_VERSION:
            raise RuntimeError(f"vlgae_amd: {LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it "
                               "(`python -m vlgae_amd.build --force`; A/B variants: tools/build_variant.sh from this tree)")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vlg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"vlgae_amd.{what} failed (code {rc:#x}): {msg}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


try:   # raw handles without building torch.cuda.Stream objects (every launch asks; the small ops are host-bound)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:   # pragma: no cover
    _raw_stream = lambda index: torch.cuda.current_stream(index).cuda_stream
    _cur_device = torch.cuda.current_device


def mask_u8(t, device):
    """A keep-mask as contiguous uint8 on `device`.  bool and uint8 share their storage: a view, not a conversion kernel (the
    `.to(torch.uint8)` of a bool mask is a 5 us launch, two per grounding-loss call)."""
    if t is None:
        return None
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    elif t.dtype != torch.uint8:
        t = t.to(torch.uint8)
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


def stream_of(t):
    """The HIP stream the launch goes to: the current stream of t's device.  hipLaunchKernel acts on the process's CURRENT
    device, so a tensor that lives elsewhere must not get this far (it would launch onto the wrong GPU or fail with an
    invalid handle): fail loudly instead."""
    index = t.device.index
    if index is None:
        index = _cur_device()
    elif index != _cur_device():
        raise RuntimeError(f"vlgae_amd: tensor on {t.device} but the current device is cuda:{_cur_device()}; "
                           "wrap the call in `with torch.cuda.device(t.device):` (one process per GPU sets it once)")
    return ctypes.c_void_p(_raw_stream(index))


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"vlgae_amd.{what}: tensors must live on an MI355X (got device '{t.device}'). This package has no CPU path.")


def in_dtype(t):
    """Kernel input element type for tensor t and the tensor to hand over (contiguous)."""
    if t.dtype == torch.bfloat16:
        return BF16, t if t.is_contiguous() else t.contiguous()
    if t.dtype == torch.float32:
        return F32, t if t.is_contiguous() else t.contiguous()
    return F32, t.detach().float().contiguous()


def alloc_f32(device, shapes, extra_bytes=0, cast=None):
    """One float32 allocation carved into 256-byte aligned views of the given shapes (None entries are skipped and
    returned as None), plus a trailing scratch view of `extra_bytes`.  The ops here are small enough that a handful of
    separate torch.empty calls costs as much host time as the kernels take.

    Returns (views, scratch).  `views.cast(n, dtype)` (see Carved) converts the first n tensors with ONE elementwise launch."""
    import math
    sizes, padded = [], []
    for shp in shapes:
        n = 0 if shp is None else math.prod(shp)
        sizes.append(n)
        padded.append((n + 63) & ~63)
    total = sum(padded)
    flat = torch.empty(total + (extra_bytes + 3) // 4, dtype=torch.float32, device=device)
    return Carved(flat, shapes, sizes, padded), flat[total:]


class Carved(list):
    """The views of alloc_f32, remembering the flat buffer so that a dtype change of several of them is one launch."""

    def __init__(self, flat, shapes, sizes, padded):
        self.flat, self.shapes, self.sizes, self.padded = flat, shapes, sizes, padded
        super().__init__(self._carve(flat, len(shapes)))

    def _carve(self, flat, count):
        parts = flat[:sum(self.padded[:count])].split_with_sizes(self.padded[:count]) if count else ()
        return [None if shp is None else (p if n == m else p[:n]).view(shp)
                for p, shp, n, m in zip(parts, self.shapes, self.sizes, self.padded)]

    def cast(self, count, dtype):
        """The first `count` tensors in `dtype` (one conversion kernel over their common storage)."""
        if dtype == torch.float32:
            return list(self[:count])
        return self._carve(self.flat[:sum(self.padded[:count])].to(dtype), count)
