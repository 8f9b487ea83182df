"""Several GPUs behind ONE call from one host thread: thin wrappers over
zkp_pairing_batch_multi / zkp_pairing_check_batch_multi.  Every PairingEngine is one context (normally one per GPU of
the node); the checks are split into contiguous blocks, one per context; the AND of the per-context flags is formed on
the host - the data path has no collective.  The one-process-per-GPU form lives in dist.py (torch.distributed / RCCL)."""
import ctypes

import numpy as np

from . import _lib
from .engine import _np, _ptr


def _handles(engines):
    if not engines:
        raise ValueError("at least one PairingEngine is needed")
    arr = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    return arr, len(engines)


def _inputs(g1, g2, inf1, inf2):
    g1, g2 = _np(g1, 12), _np(g2, 24)
    if g1.shape[0] != g2.shape[0]:
        raise ValueError("g1 and g2 hold different numbers of points")
    i1 = None if inf1 is None else _np(inf1, None, np.uint8)
    i2 = None if inf2 is None else _np(inf2, None, np.uint8)
    for i in (i1, i2):
        if i is not None and i.size != g1.shape[0]:
            raise ValueError("infinity flags and points differ in length")
    return g1, g2, i1, i2


def pairing_multi(engines, g1, g2, inf1=None, inf2=None):
    """-> (Gt (n,72), ok (n,) uint8 = Gt == identity, all_ok bool) over the engines' GPUs"""
    lib = _lib.load()
    g1, g2, i1, i2 = _inputs(g1, g2, inf1, inf2)
    n = g1.shape[0]
    out, ok, allok = np.empty((n, 72), dtype=np.uint64), np.empty(n, dtype=np.uint8), ctypes.c_int(1)
    arr, cnt = _handles(engines)
    rc = lib.zkp_pairing_batch_multi(arr, cnt, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n, _ptr(out), _ptr(ok), ctypes.byref(allok))
    if rc != 0:
        raise _lib.ZkpError(rc, "; ".join(lib.zkp_last_error(e._h).decode() for e in engines))
    return out, ok, bool(allok.value)


def pairing_check_multi(engines, g1, g2, k, inf1=None, inf2=None):
    """-> (ok (n_checks,) uint8, all_ok bool): groups of k consecutive pairs, one shared final exponentiation each"""
    lib = _lib.load()
    g1, g2, i1, i2 = _inputs(g1, g2, inf1, inf2)
    n = g1.shape[0]
    if k <= 0 or n % k:
        raise ValueError("the number of pairs must be a positive multiple of k")
    ok, allok = np.empty(n // k, dtype=np.uint8), ctypes.c_int(1)
    arr, cnt = _handles(engines)
    rc = lib.zkp_pairing_check_batch_multi(arr, cnt, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n // k, k, _ptr(ok), ctypes.byref(allok))
    if rc != 0:
        raise _lib.ZkpError(rc, "; ".join(lib.zkp_last_error(e._h).decode() for e in engines))
    return ok, bool(allok.value)
