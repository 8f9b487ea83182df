"""ctypes binding of oracle/_build/liborc.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liborc.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("bls12_381_oracle.c", "bls12_381_oracle.h", "orc_constants.h")]
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(os.path.join(ORACLE_DIR, "_build", "liborc_slow.so")) or \
            any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _b(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _arr(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _un(name, nin, nout):
    def f(a):
        a = _arr(a, nin)
        out = np.zeros(nout, dtype=np.uint64)
        getattr(lib(), name)(_p(a), _p(out))
        return out
    return f


def _bin(name, n):
    def f(a, b):
        a, b = _arr(a, n), _arr(b, n)
        out = np.zeros(n, dtype=np.uint64)
        getattr(lib(), name)(_p(a), _p(b), _p(out))
        return out
    return f


def _un_ok(name, n):
    def f(a):
        a = _arr(a, n)
        out = np.zeros(n, dtype=np.uint64)
        fn = getattr(lib(), name)
        fn.restype = ctypes.c_int
        ok = fn(_p(a), _p(out))
        return (out if ok else None)
    return f


fp_add, fp_sub, fp_mul = _bin("orc_fp_add", 6), _bin("orc_fp_sub", 6), _bin("orc_fp_mul", 6)
fp_neg, fp_square = _un("orc_fp_neg", 6, 6), _un("orc_fp_square", 6, 6)
fp_invert, fp_sqrt = _un_ok("orc_fp_invert", 6), _un_ok("orc_fp_sqrt", 6)
fp2_add, fp2_sub, fp2_mul = _bin("orc_fp2_add", 12), _bin("orc_fp2_sub", 12), _bin("orc_fp2_mul", 12)
fp2_neg, fp2_square = _un("orc_fp2_neg", 12, 12), _un("orc_fp2_square", 12, 12)
fp2_conjugate, fp2_mul_by_nonresidue = _un("orc_fp2_conjugate", 12, 12), _un("orc_fp2_mul_by_nonresidue", 12, 12)
fp2_invert, fp2_sqrt = _un_ok("orc_fp2_invert", 12), _un_ok("orc_fp2_sqrt", 12)
fp6_add, fp6_sub, fp6_mul = _bin("orc_fp6_add", 36), _bin("orc_fp6_sub", 36), _bin("orc_fp6_mul", 36)
fp6_neg, fp6_square = _un("orc_fp6_neg", 36, 36), _un("orc_fp6_square", 36, 36)
fp6_mul_by_nonresidue = _un("orc_fp6_mul_by_nonresidue", 36, 36)
fp6_frobenius_map = _un("orc_fp6_frobenius_map", 36, 36)
fp6_frobenius_map_refcompat = _un("orc_fp6_frobenius_map_refcompat", 36, 36)
fp6_invert = _un_ok("orc_fp6_invert", 36)
fp12_add, fp12_sub, fp12_mul = _bin("orc_fp12_add", 72), _bin("orc_fp12_sub", 72), _bin("orc_fp12_mul", 72)
fp12_square, fp12_conjugate = _un("orc_fp12_square", 72, 72), _un("orc_fp12_conjugate", 72, 72)
fp12_frobenius_map = _un("orc_fp12_frobenius_map", 72, 72)
fp12_frobenius_map_refcompat = _un("orc_fp12_frobenius_map_refcompat", 72, 72)
fp12_cyclotomic_square = _un("orc_fp12_cyclotomic_square", 72, 72)
fp12_invert = _un_ok("orc_fp12_invert", 72)


def fp_pow_vartime(a, e):
    a, e = _arr(a, 6), _arr(e, 6)
    out = np.zeros(6, dtype=np.uint64)
    lib().orc_fp_pow_vartime(_p(a), _p(e), _p(out))
    return out


def fp_is_canonical(a):
    return bool(lib().orc_fp_is_canonical(_p(_arr(a, 6))))


def fp_to_bytes_be(a):
    out = np.zeros(48, dtype=np.uint8)
    lib().orc_fp_to_bytes_be(_p(_arr(a, 6)), _b(out))
    return bytes(out)


def fp_from_bytes_be(b):
    inp = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    out = np.zeros(6, dtype=np.uint64)
    ok = lib().orc_fp_from_bytes_be(_b(inp), _p(out))
    return out if ok else None


def fp6_mul_by_1(a, c1):
    a, c1 = _arr(a, 36), _arr(c1, 12)
    out = np.zeros(36, dtype=np.uint64)
    lib().orc_fp6_mul_by_1(_p(a), _p(c1), _p(out))
    return out


def fp6_mul_by_01(a, c0, c1):
    a, c0, c1 = _arr(a, 36), _arr(c0, 12), _arr(c1, 12)
    out = np.zeros(36, dtype=np.uint64)
    lib().orc_fp6_mul_by_01(_p(a), _p(c0), _p(c1), _p(out))
    return out


def fp12_one():
    out = np.zeros(72, dtype=np.uint64)
    lib().orc_fp12_one(_p(out))
    return out


def fp12_mul_by_014(a, c0, c1, c4):
    a, c0, c1, c4 = _arr(a, 72), _arr(c0, 12), _arr(c1, 12), _arr(c4, 12)
    out = np.zeros(72, dtype=np.uint64)
    lib().orc_fp12_mul_by_014(_p(a), _p(c0), _p(c1), _p(c4), _p(out))
    return out


def fp12_pow_u64(a, e):
    a = _arr(a, 72)
    out = np.zeros(72, dtype=np.uint64)
    lib().orc_fp12_pow_u64(_p(a), ctypes.c_uint64(e), _p(out))
    return out


def g1_generator():
    out = np.zeros(12, dtype=np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g2_generator():
    out = np.zeros(24, dtype=np.uint64)
    lib().orc_g2_generator(_p(out))
    return out


def _pt_un(name, n):
    def f(p, inf=0):
        p = _arr(p, n)
        out = np.zeros(n, dtype=np.uint64)
        oi = ctypes.c_uint8(0)
        getattr(lib(), name)(_p(p), ctypes.c_uint8(inf), _p(out), ctypes.byref(oi))
        return out, int(oi.value)
    return f


def _pt_add(name, n):
    def f(p, pinf, q, qinf):
        p, q = _arr(p, n), _arr(q, n)
        out = np.zeros(n, dtype=np.uint64)
        oi = ctypes.c_uint8(0)
        getattr(lib(), name)(_p(p), ctypes.c_uint8(pinf), _p(q), ctypes.c_uint8(qinf), _p(out), ctypes.byref(oi))
        return out, int(oi.value)
    return f


def _pt_mul(name, n):
    def f(p, k, inf=0):
        p = _arr(p, n)
        kk = np.array([(int(k) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        out = np.zeros(n, dtype=np.uint64)
        oi = ctypes.c_uint8(0)
        getattr(lib(), name)(_p(p), ctypes.c_uint8(inf), _p(kk), _p(out), ctypes.byref(oi))
        return out, int(oi.value)
    return f


g1_double, g2_double = _pt_un("orc_g1_double", 12), _pt_un("orc_g2_double", 24)
g1_add, g2_add = _pt_add("orc_g1_add", 12), _pt_add("orc_g2_add", 24)
g1_mul, g2_mul = _pt_mul("orc_g1_mul", 12), _pt_mul("orc_g2_mul", 24)


def g1_is_on_curve(p):
    return bool(lib().orc_g1_is_on_curve(_p(_arr(p, 12))))


def g1_is_torsion_free(p):
    return bool(lib().orc_g1_is_torsion_free(_p(_arr(p, 12))))


def g1_is_valid(p, inf=0):
    return int(lib().orc_g1_is_valid(_p(_arr(p, 12)), ctypes.c_uint8(inf)))


def g2_is_on_curve(p):
    return bool(lib().orc_g2_is_on_curve(_p(_arr(p, 24))))


def g2_is_torsion_free(p):
    return bool(lib().orc_g2_is_torsion_free(_p(_arr(p, 24))))


def g2_is_valid(p, inf=0):
    return int(lib().orc_g2_is_valid(_p(_arr(p, 24)), ctypes.c_uint8(inf)))


def g2_psi(p):
    out = np.zeros(24, dtype=np.uint64)
    lib().orc_g2_psi(_p(_arr(p, 24)), _p(out))
    return out


def _infs(a, n):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.size == n
    return a


def multi_miller_loop_batch(g1, g2, n_checks, k, inf1=None, inf2=None):
    g1, g2 = _arr(g1, 12 * n_checks * k), _arr(g2, 24 * n_checks * k)
    inf1, inf2 = _infs(inf1, n_checks * k), _infs(inf2, n_checks * k)
    out = np.zeros(72 * n_checks, dtype=np.uint64)
    lib().orc_multi_miller_loop_batch(_p(g1), _p(g2), _b(inf1), _b(inf2), ctypes.c_size_t(n_checks), ctypes.c_size_t(k), _p(out))
    return out.reshape(n_checks, 72)


def final_exponentiation_batch(f):
    f = _arr(f)
    n = f.size // 72
    out = np.zeros(72 * n, dtype=np.uint64)
    lib().orc_final_exponentiation_batch(_p(f), ctypes.c_size_t(n), _p(out))
    return out.reshape(n, 72)


def pairing_batch(g1, g2, inf1=None, inf2=None, nthreads=1):
    g1, g2 = _arr(g1), _arr(g2)
    n = g1.size // 12
    assert g2.size == 24 * n
    inf1, inf2 = _infs(inf1, n), _infs(inf2, n)
    out = np.zeros(72 * n, dtype=np.uint64)
    lib().orc_pairing_batch_mt(_p(g1), _p(g2), _b(inf1), _b(inf2), ctypes.c_size_t(n), _p(out), ctypes.c_int(nthreads))
    return out.reshape(n, 72)


SLOW_LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liborc_slow.so")
_slow = None


def pairing_batch_slow(g1, g2, nthreads=1):
    """the same pairings through the reference-faithful slow build (-DORC_SLOW: canonical integers, schoolbook product +
    long division per Fp::mul as in the reference's src/fp.rs:416-434); only for bench.py's cpu_baseline leg and its test"""
    global _slow
    if _slow is None:
        build()
        if not os.path.exists(SLOW_LIB_PATH):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
        _slow = ctypes.CDLL(SLOW_LIB_PATH)
    g1, g2 = _arr(g1), _arr(g2)
    n = g1.size // 12
    assert g2.size == 24 * n
    out = np.zeros(72 * n, dtype=np.uint64)
    _slow.orc_pairing_batch_mt(_p(g1), _p(g2), None, None, ctypes.c_size_t(n), _p(out), ctypes.c_int(nthreads))
    return out.reshape(n, 72)


def pairing_check_batch(g1, g2, n_checks, k, inf1=None, inf2=None):
    g1, g2 = _arr(g1, 12 * n_checks * k), _arr(g2, 24 * n_checks * k)
    inf1, inf2 = _infs(inf1, n_checks * k), _infs(inf2, n_checks * k)
    ok = np.zeros(n_checks, dtype=np.uint8)
    lib().orc_pairing_check_batch(_p(g1), _p(g2), _b(inf1), _b(inf2), ctypes.c_size_t(n_checks), ctypes.c_size_t(k), _b(ok))
    return ok


def miller_loop_affine(g1, g2):
    out = np.zeros(72, dtype=np.uint64)
    lib().orc_miller_loop_affine(_p(_arr(g1, 12)), _p(_arr(g2, 24)), _p(out))
    return out


def g1_mul_batch(p, k, nthreads=1):
    p, k = _arr(p), _arr(k)
    n = p.size // 12
    out = np.zeros(12 * n, dtype=np.uint64)
    lib().orc_g1_mul_batch_mt(_p(p), _p(k), ctypes.c_size_t(n), _p(out), ctypes.c_int(nthreads))
    return out.reshape(n, 12)


def g2_mul_batch(p, k, nthreads=1):
    p, k = _arr(p), _arr(k)
    n = p.size // 24
    out = np.zeros(24 * n, dtype=np.uint64)
    lib().orc_g2_mul_batch_mt(_p(p), _p(k), ctypes.c_size_t(n), _p(out), ctypes.c_int(nthreads))
    return out.reshape(n, 24)


# ---- int <-> limb helpers shared by tests
def to_limbs(v, n=6):
    return np.array([(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def from_limbs(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a).reshape(-1)))


def ints_to_arr(vals):
    """list of Fp ints -> flat uint64 array of 6 limbs each"""
    return np.concatenate([to_limbs(v) for v in vals])


def arr_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 6)
    return [from_limbs(r) for r in a]
