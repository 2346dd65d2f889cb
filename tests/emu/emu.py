"""ctypes front-end of the host phase emulator (tests/emu/emu_dp.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("VLG_EMU_SO") or os.path.join(_HERE, "_build", "libvlg_emu.so")   # override: a sanitizer build (tests/test_sanitizers.py)
_CORE = os.path.join(_HERE, "..", "..", "vlgae_amd", "csrc", "vlg_dp_core.h")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "emu_dp.cpp")
        stale = (not os.path.exists(_SO)) or any(os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, _CORE))
        if stale and not os.environ.get("VLG_EMU_SO"):
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", _SO], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _bf16_bits(a):
    """round-to-nearest-even fp32 -> bf16 bit patterns (uint16)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def dmv1o(dec, attach, lengths, semiring=0, grad=True, glogZ=None, nt=16, order=0, bf16=False, mode=0):
    """mode: DmvLayout placement mode to emulate (0 = everything in one arena; 1 = the overlay mode: value charts in the
    first arena during the inside pass, copied to a second arena -- the workspace -- before the outside pass)."""
    B, N = dec.shape[:2]
    lib().emu_set_dmv_mode(int(mode))
    if bf16:
        dec, attach = _bf16_bits(dec), _bf16_bits(attach)
    else:
        dec, attach = np.ascontiguousarray(dec, np.float32), np.ascontiguousarray(attach, np.float32)
    ln = np.ascontiguousarray(lengths, np.int64)
    lz = np.full(B, np.nan, np.float32)
    gd = np.full((B, N, 2, 2, 2), np.nan, np.float32) if grad else None
    ga = np.full((B, N, N, 2), np.nan, np.float32) if grad else None
    g = None if glogZ is None else np.ascontiguousarray(glogZ, np.float32)
    rc = lib().emu_dmv1o(_p(dec), _p(attach), _p(ln), B, N, int(bf16), semiring, _p(g), _p(lz), _p(gd), _p(ga), nt, order)
    lib().emu_set_dmv_mode(0)
    assert rc == 0
    return lz, gd, ga


def deptree(arc, lengths, semiring=0, grad=True, glogZ=None, nt=16, order=0):
    B, N = arc.shape[:2]
    arc = np.ascontiguousarray(arc, np.float32)
    ln = None if lengths is None else np.ascontiguousarray(lengths, np.int64)
    lz = np.full(B, np.nan, np.float32)
    ga = np.full((B, N, N), np.nan, np.float32) if grad else None
    g = None if glogZ is None else np.ascontiguousarray(glogZ, np.float32)
    rc = lib().emu_deptree(_p(arc), _p(ln), B, N, 0, semiring, _p(g), _p(lz), _p(ga), nt, order)
    assert rc == 0
    return lz, ga


def canary_trips():
    return int(lib().emu_canary_trips())


def dmv1o_decode(dec, attach, lengths, nt=16, order=0):
    B, N = dec.shape[:2]
    dec, attach = np.ascontiguousarray(dec, np.float32), np.ascontiguousarray(attach, np.float32)
    ln = np.ascontiguousarray(lengths, np.int64)
    best = np.full(B, np.nan, np.float32)
    heads = np.full((B, N), -1, np.int64)
    assert lib().emu_dmv1o_decode(_p(dec), _p(attach), _p(ln), B, N, _p(best), _p(heads), nt, order) == 0
    return best, heads


def deptree_decode(arc, lengths, nt=16, order=0):
    B, N = arc.shape[:2]
    arc = np.ascontiguousarray(arc, np.float32)
    ln = np.ascontiguousarray(lengths, np.int64)
    best = np.full(B, np.nan, np.float32)
    heads = np.full((B, N), -1, np.int64)
    assert lib().emu_deptree_decode(_p(arc), _p(ln), B, N, _p(best), _p(heads), nt, order) == 0
    return best, heads


def dmv1o_rules(attach_rule, dec, root_rule, token, lengths, head_mask=None, semiring=0, grad=True, heads=False,
                fill=-1e20, nt=16, order=0):
    B, L, T = attach_rule.shape[:3]
    rule = np.ascontiguousarray(attach_rule, np.float32)
    dec = np.ascontiguousarray(dec, np.float32)
    root = np.ascontiguousarray(np.asarray(root_rule, np.float32).reshape(-1, T))
    per_sentence = int(root.shape[0] == B and B > 1)
    tok = np.ascontiguousarray(token, np.int64)
    ln = np.ascontiguousarray(lengths, np.int64)
    hm = None if head_mask is None else np.ascontiguousarray(head_mask, np.uint8)
    lz = np.full(B, np.nan, np.float32)
    g_rule = np.zeros((B, L, T, 2, 2), np.float32) if grad else None
    g_dec = np.zeros((B, L, 2, 2, 2), np.float32) if grad else None
    g_root = np.zeros((B, T), np.float32) if grad else None
    hd = np.full((B, L + 1), -1, np.int64) if heads else None
    import ctypes
    rc = lib().emu_dmv1o_rules(_p(rule), _p(dec), _p(root), per_sentence, _p(tok), _p(hm), _p(ln), B, L, T, semiring,
                               ctypes.c_float(fill), _p(lz), _p(g_rule), _p(g_dec), _p(g_root), _p(hd), nt, order)
    assert rc == 0
    return lz, g_rule, g_dec, g_root, hd
