"""Batched engine: thin, typed wrapper over the C ABI (one PairingEngine == one zkp_ctx == one GPU).

Array conventions (numpy uint64 or torch int64/uint64-viewed tensors, C-contiguous):
  g1 (n,12)  g2 (n,24)  inf (n,) uint8  fp12/Gt (n,72)  scalars (n,4)
Host numpy arrays go through the host-pointer entry points (copy in / copy out); torch tensors
that live on the engine's GPU go through the *_dev entry points and never leave HBM."""
import ctypes

import numpy as np

from . import _lib

KERNEL_AUTO, KERNEL_THREAD, KERNEL_COOP = 0, 1, 2


def _np(a, cols, dtype=np.uint64):
    a = np.ascontiguousarray(a, dtype=dtype)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _flags(inf, n, name):
    """optional per-point infinity flags: n bytes"""
    if inf is None:
        return None
    a = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
    if a.size != n:
        raise ValueError("%s: %d flags for %d points" % (name, a.size, n))
    return a


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class PairingEngine:
    def __init__(self, device=0, kernel=None, validate=False):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.zkp_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise _lib.ZkpError(rc, "zkp_init(device=%d)" % device)
        self._h = h
        self.device = int(device)
        if validate:
            self._chk(self._lib.zkp_set_validate(self._h, 1))
        if kernel is not None:
            self.set_kernel(kernel)

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc != 0:
            raise _lib.ZkpError(rc, self._lib.zkp_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zkp_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, kind):
        kind = {"auto": 0, "thread": 1, "coop": 2}.get(kind, kind)
        self._chk(self._lib.zkp_set_kernel(self._h, int(kind)))

    def set_validate(self, on):
        self._chk(self._lib.zkp_set_validate(self._h, 1 if on else 0))

    def take_validation_status(self):
        """validation mode on the device-tensor entry points: True if any such call since the last query saw a field
        element >= p (synchronises the current torch stream; clears the word)"""
        bad = ctypes.c_int(0)
        self._chk(self._lib.zkp_take_validation_status_dev(self._h, self._stream(), ctypes.byref(bad)))
        return bool(bad.value)

    def device_info(self):
        cus, clk = ctypes.c_int(), ctypes.c_int()
        name = ctypes.create_string_buffer(64)
        self._chk(self._lib.zkp_device_info(self._h, ctypes.byref(cus), ctypes.byref(clk), name, 64))
        return {"cus": cus.value, "clock_khz": clk.value, "arch": name.value.decode()}

    @staticmethod
    def gt_identity():
        p = _lib.load().zkp_gt_identity()
        return np.array([p[i] for i in range(72)], dtype=np.uint64)

    # ------------------------------------------------------------------ host-array API
    @staticmethod
    def host_array(shape, dtype=np.uint64):
        """numpy array in page-locked host memory (zkp_host_alloc): the host-pointer entry points copy from / into such an
        array by DMA, overlapped with the kernels; a pageable array works too, its copies block the calling thread.  The
        memory is released when the array (and every view of it) is gone."""
        import weakref
        lib = _lib.load()
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        st = lib.zkp_host_alloc(max(nbytes, 1), ctypes.byref(p))
        if st != 0:
            raise _lib.ZkpError(st, "zkp_host_alloc(%d bytes)" % nbytes)
        buf = (ctypes.c_ubyte * max(nbytes, 1)).from_address(p.value)
        weakref.finalize(buf, lib.zkp_host_free, ctypes.c_void_p(p.value))
        return np.frombuffer(buf, dtype=dtype, count=nbytes // np.dtype(dtype).itemsize).reshape(shape)

    def pairing(self, g1, g2, inf1=None, inf2=None, out=None):
        """out[i] = pairing(g1[i], g2[i])  -> (n,72) canonical Gt (into `out` when given, e.g. a host_array)"""
        if _is_torch(g1):
            if out is not None and not _is_torch(out):
                raise ValueError("out must be a tensor on the engine's GPU when the inputs are")
            return self._pairing_t(g1, g2, inf1, inf2, out)
        g1, g2 = _np(g1, 12), _np(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 hold different numbers of points")
        i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        if out is None:
            out = np.empty((n, 72), dtype=np.uint64)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.uint64 and out.shape == (n, 72) and out.flags["C_CONTIGUOUS"]):
            raise ValueError("out must be a C-contiguous (n, 72) uint64 array")
        self._chk(self._lib.zkp_pairing_batch(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n, _ptr(out)))
        return out

    def multi_miller_loop(self, g1, g2, k, inf1=None, inf2=None):
        """groups of k consecutive pairs -> (n_checks,72) MillerLoopResult"""
        if _is_torch(g1):
            return self._miller_t(g1, g2, k, inf1, inf2)
        g1, g2 = _np(g1, 12), _np(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n or k <= 0 or n % k:
            raise ValueError("g1 / g2 sizes do not match or are not a multiple of k")
        i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        out = np.empty((n // k, 72), dtype=np.uint64)
        self._chk(self._lib.zkp_multi_miller_loop_batch(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n // k, k, _ptr(out)))
        return out

    def final_exponentiation(self, f):
        if _is_torch(f):
            return self._fexp_t(f)
        f = _np(f, 72)
        out = np.empty_like(f)
        self._chk(self._lib.zkp_final_exponentiation_batch(self._h, _ptr(f), f.shape[0], _ptr(out)))
        return out

    def pairing_check(self, g1, g2, k, inf1=None, inf2=None):
        """-> (ok bytes (n_checks,), all_ok bool): prod_j e(g1[c*k+j], g2[c*k+j]) == Gt::identity()"""
        if _is_torch(g1):
            return self._check_t(g1, g2, k, inf1, inf2)
        g1, g2 = _np(g1, 12), _np(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n or k <= 0 or n % k:
            raise ValueError("g1 / g2 sizes do not match or are not a multiple of k")
        i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        ok = np.empty(n // k, dtype=np.uint8)
        allok = ctypes.c_int(1)
        self._chk(self._lib.zkp_pairing_check_batch(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n // k, k, _ptr(ok), ctypes.byref(allok)))
        return ok, bool(allok.value)

    # ---- the whole batch as ONE product check (one shared final exponentiation)
    def miller_product(self, g1, g2, inf1=None, inf2=None):
        """multi_miller_loop over ALL pairs of the batch -> (72,) MillerLoopResult"""
        if _is_torch(g1):
            import torch
            self._t_pairs(g1, g2, inf1, inf2)
            out = torch.empty(72, dtype=g1.dtype, device=g1.device)
            self._chk(self._lib.zkp_miller_product_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), g1.numel() // 12,
                                                       self._tp(out), self._stream()))
            return out
        g1, g2 = _np(g1, 12), _np(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 hold different numbers of points")
        i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        out = np.empty(72, dtype=np.uint64)
        self._chk(self._lib.zkp_miller_product(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n, _ptr(out)))
        return out

    def fp12_product(self, f):
        """f_0 * f_1 * ... * f_n-1 of (n,72) Fp12 values -> (72,)"""
        if _is_torch(f):
            import torch
            self._t_check(f, 72, "f")
            out = torch.empty(72, dtype=f.dtype, device=f.device)
            self._chk(self._lib.zkp_fp12_product_dev(self._h, self._tp(f), f.numel() // 72, self._tp(out), self._stream()))
            return out
        f = _np(f, 72)
        out = np.empty(72, dtype=np.uint64)
        self._chk(self._lib.zkp_fp12_product(self._h, _ptr(f), f.shape[0], _ptr(out)))
        return out

    def pairing_product_check(self, g1, g2, inf1=None, inf2=None):
        """-> (Gt (72,), is_one): prod_i e(g1[i], g2[i]) == Gt::identity() with one final exponentiation.
        Device tensors return (Gt tensor, int32 tensor(1,)) without synchronising."""
        if _is_torch(g1):
            import torch
            self._t_pairs(g1, g2, inf1, inf2)
            gt = torch.empty(72, dtype=g1.dtype, device=g1.device)
            one = torch.empty(1, dtype=torch.int32, device=g1.device)
            self._chk(self._lib.zkp_pairing_product_check_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), g1.numel() // 12,
                                                              self._tp(gt), self._tp(one), self._stream()))
            return gt, one
        g1, g2 = _np(g1, 12), _np(g2, 24)
        n = g1.shape[0]
        if g2.shape[0] != n:
            raise ValueError("g1 and g2 hold different numbers of points")
        i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        gt = np.empty(72, dtype=np.uint64)
        one = ctypes.c_int(0)
        self._chk(self._lib.zkp_pairing_product_check(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n, _ptr(gt), ctypes.byref(one)))
        return gt, bool(one.value)

    def g1_is_valid(self, g1, inf=None):
        if _is_torch(g1):
            return self._valid_t(g1, inf, 1)
        g1 = _np(g1, 12)
        i = _flags(inf, g1.shape[0], "inf")
        st = np.empty(g1.shape[0], dtype=np.uint8)
        self._chk(self._lib.zkp_g1_is_valid_batch(self._h, _ptr(g1), _ptr(i), g1.shape[0], _ptr(st)))
        return st

    def g2_is_valid(self, g2, inf=None):
        if _is_torch(g2):
            return self._valid_t(g2, inf, 2)
        g2 = _np(g2, 24)
        i = _flags(inf, g2.shape[0], "inf")
        st = np.empty(g2.shape[0], dtype=np.uint8)
        self._chk(self._lib.zkp_g2_is_valid_batch(self._h, _ptr(g2), _ptr(i), g2.shape[0], _ptr(st)))
        return st

    def g1_mul(self, base, scalars):
        """[k_i] base_i; base (12,) broadcasts. -> (points (n,12), inf (n,))"""
        if _is_torch(scalars):
            return self._mul_t(base, scalars, 1)
        sc = _np(scalars, 4)
        base = _np(base, 12)
        n = sc.shape[0]
        stride = 0 if base.shape[0] == 1 and n != 1 else 12
        if stride and base.shape[0] != n:
            raise ValueError("base points and scalars differ in number")
        out, oi = np.empty((n, 12), dtype=np.uint64), np.empty(n, dtype=np.uint8)
        self._chk(self._lib.zkp_g1_mul_batch(self._h, _ptr(base), stride, _ptr(sc), n, _ptr(out), _ptr(oi)))
        return out, oi

    def g2_mul(self, base, scalars):
        if _is_torch(scalars):
            return self._mul_t(base, scalars, 2)
        sc = _np(scalars, 4)
        base = _np(base, 24)
        n = sc.shape[0]
        stride = 0 if base.shape[0] == 1 and n != 1 else 24
        if stride and base.shape[0] != n:
            raise ValueError("base points and scalars differ in number")
        out, oi = np.empty((n, 24), dtype=np.uint64), np.empty(n, dtype=np.uint8)
        self._chk(self._lib.zkp_g2_mul_batch(self._h, _ptr(base), stride, _ptr(sc), n, _ptr(out), _ptr(oi)))
        return out, oi

    def decode_points(self, data, which):
        """uncompressed big-endian bytes -> (points, inf, status); which = 1 (G1, 96 B) or 2 (G2, 192 B)"""
        size = 96 if which == 1 else 192
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        if buf.size % size:
            raise ValueError("byte string is not a multiple of %d bytes" % size)
        n = buf.size // size
        pts = np.empty((n, size // 8), dtype=np.uint64)
        inf, st = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8)
        fn = self._lib.zkp_g1_decode_batch if which == 1 else self._lib.zkp_g2_decode_batch
        self._chk(fn(self._h, _ptr(buf), n, _ptr(pts), _ptr(inf), _ptr(st)))
        return pts, inf, st

    def encode_points(self, pts, which, inf=None):
        cols = 12 if which == 1 else 24
        pts = _np(pts, cols)
        n = pts.shape[0]
        i = _flags(inf, n, "inf")
        out = np.empty(n * cols * 8, dtype=np.uint8)
        fn = self._lib.zkp_g1_encode_batch if which == 1 else self._lib.zkp_g2_encode_batch
        self._chk(fn(self._h, _ptr(pts), _ptr(i), n, _ptr(out)))
        return out.tobytes()

    # ---- BASELINE config 5 in one call, and the codec on resident tensors
    POINT_STATUS = ("ok", "non_canonical", "malformed", "not_on_curve", "not_in_subgroup")

    def points_check(self, g1_bytes, g2_bytes, k, st1=None, st2=None, ok=None, all_ok=None):
        """raw uncompressed points -> decode -> is_valid -> pairing check (zkp_points_check_batch[_dev]).
        numpy uint8 arrays (n x 96, n x 192): returns (st1, st2, ok, all_ok bool).  torch uint8 tensors on the engine's GPU:
        fills the given st1 / st2 / ok (uint8) and all_ok (int32[1]) tensors - each optional - asynchronously, returns None."""
        if _is_torch(g1_bytes):
            import torch
            u8 = (torch.uint8,)
            self._t_check(g1_bytes, 96, "g1_bytes", dtypes=u8), self._t_check(g2_bytes, 192, "g2_bytes", dtypes=u8)
            n = g1_bytes.numel() // 96
            if g2_bytes.numel() // 192 != n or k <= 0 or n % k:
                raise ValueError("byte strings / k do not match")
            self._t_bytes(st1, n, "st1"), self._t_bytes(st2, n, "st2"), self._t_bytes(ok, n // k, "ok")
            if all_ok is not None:
                self._t_check(all_ok, None, "all_ok", rows=1, dtypes=(torch.int32,))
            self._chk(self._lib.zkp_points_check_batch_dev(self._h, self._tp(g1_bytes), self._tp(g2_bytes), n // k, k, self._tp(st1), self._tp(st2),
                                                           self._tp(ok), self._tp(all_ok), self._stream()))
            return None
        b1 = np.ascontiguousarray(g1_bytes, dtype=np.uint8).reshape(-1, 96)
        b2 = np.ascontiguousarray(g2_bytes, dtype=np.uint8).reshape(-1, 192)
        n = b1.shape[0]
        if b2.shape[0] != n or k <= 0 or n % k:
            raise ValueError("byte strings / k do not match")
        s1, s2, okb = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8), np.empty(n // k, dtype=np.uint8)
        allok = ctypes.c_int(1)
        self._chk(self._lib.zkp_points_check_batch(self._h, _ptr(b1), _ptr(b2), n // k, k, _ptr(s1), _ptr(s2), _ptr(okb), ctypes.byref(allok)))
        return s1, s2, okb, bool(allok.value)

    def decode_points_dev(self, data, which):
        """uint8 tensor of uncompressed points on the engine's GPU -> (points int64 (n, 12 | 24), inf uint8, status uint8) tensors"""
        import torch
        size = 96 if which == 1 else 192
        self._t_check(data, size, "bytes", dtypes=(torch.uint8,))
        n = data.numel() // size
        pts = torch.empty((n, size // 8), dtype=torch.int64, device=data.device)
        inf = torch.empty(n, dtype=torch.uint8, device=data.device)
        st = torch.empty(n, dtype=torch.uint8, device=data.device)
        fn = self._lib.zkp_g1_decode_batch_dev if which == 1 else self._lib.zkp_g2_decode_batch_dev
        self._chk(fn(self._h, self._tp(data), n, self._tp(pts), self._tp(inf), self._tp(st), self._stream()))
        return pts, inf, st

    def encode_points_dev(self, pts, which, inf=None):
        """(n, 12 | 24) point tensor on the engine's GPU -> (n, 96 | 192) uint8 tensor of uncompressed big-endian points"""
        import torch
        cols = 12 if which == 1 else 24
        self._t_check(pts, cols, "points")
        n = pts.numel() // cols
        self._t_bytes(inf, n, "inf")
        out = torch.empty((n, cols * 8), dtype=torch.uint8, device=pts.device)
        fn = self._lib.zkp_g1_encode_batch_dev if which == 1 else self._lib.zkp_g2_encode_batch_dev
        self._chk(fn(self._h, self._tp(pts), self._tp(inf), n, self._tp(out), self._stream()))
        return out

    # ---- one rank per GPU: the RCCL communicator behind the C ABI (the torch.distributed flavour is zkvm_pairings_amd/dist.py)
    @staticmethod
    def comm_unique_id():
        """128 bytes made on rank 0; every rank hands the same bytes to comm_init_rank"""
        buf = ctypes.create_string_buffer(128)
        st = _lib.load().zkp_comm_unique_id(buf)
        if st != 0:
            raise _lib.ZkpError(st, "zkp_comm_unique_id")
        return buf.raw

    def comm_init_rank(self, nranks, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        self._chk(self._lib.zkp_comm_init_rank(self._h, int(nranks), int(rank), ctypes.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        self._chk(self._lib.zkp_comm_destroy(self._h))

    def comm_info(self):
        n, r = ctypes.c_int(), ctypes.c_int()
        self._chk(self._lib.zkp_comm_info(self._h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    def _join_failed(self, entry, dev_flag=None):
        """a LOCAL argument error found on this side of the ABI (sizes that do not match, k that does not divide n) must not keep the
        rank out of the collective its peers are already waiting in: enter it through the same entry point with arguments the library
        itself refuses (no points, one check) - it then takes part with flag 0 / the zero record - and let the caller raise afterwards"""
        null = ctypes.c_void_p(None)
        out = ctypes.c_int(0)
        if entry == "check":
            rc = self._lib.zkp_pairing_check_batch_allreduce(self._h, null, null, null, null, 1, 1, null, ctypes.byref(out))
        elif entry == "check_dev":
            rc = self._lib.zkp_pairing_check_batch_allreduce_dev(self._h, null, null, null, null, 1, 1, null, self._tp(dev_flag), self._stream())
        elif entry == "gt_check_dev":
            rc = self._lib.zkp_pairing_gt_check_batch_allreduce_dev(self._h, null, null, null, null, 1, 1, null, null, self._tp(dev_flag), self._stream())
        elif entry == "points":
            rc = self._lib.zkp_points_check_batch_allreduce(self._h, null, null, 1, 1, null, null, null, ctypes.byref(out))
        else:
            rc = self._lib.zkp_pairing_product_check_allgather(self._h, null, null, null, null, 1, null, ctypes.byref(out))
        # the library refuses the arguments (ZKP_ERR_ARG = -1) AFTER it has taken part in the collective; ZKP_ERR_COMM (-6) means there
        # was no communicator / RCCL failed, i.e. nothing was joined - the caller's error should say so
        self.last_join_status = rc
        return rc

    def and_allreduce(self, flag):
        """in-place AND (all-reduce MIN) of an int32[1] tensor of {0,1} over the communicator's ranks, on the current stream"""
        import torch
        self._t_check(flag, None, "flag", rows=1, dtypes=(torch.int32,))
        self._chk(self._lib.zkp_and_allreduce_dev(self._h, self._tp(flag), self._stream()))

    def pairing_check_allreduce(self, g1, g2, k, inf1=None, inf2=None):
        """this rank's block of a sharded check + the AND over all ranks (zkp_pairing_check_batch_allreduce[_dev]):
        -> (ok bytes of this rank's checks, all_ok over ALL ranks: bool for host arrays, int32[1] tensor for device tensors)"""
        if _is_torch(g1):
            import torch
            allok = torch.empty(1, dtype=torch.int32, device=torch.device("cuda", self.device))
            try:
                n = self._t_pairs(g1, g2, inf1, inf2, k)
            except (TypeError, ValueError):
                self._join_failed("check_dev", allok)
                raise
            ok = torch.empty(n // k, dtype=torch.uint8, device=g1.device)
            self._chk(self._lib.zkp_pairing_check_batch_allreduce_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n // k, k,
                                                                      self._tp(ok), self._tp(allok), self._stream()))
            return ok, allok
        try:
            g1, g2 = _np(g1, 12), _np(g2, 24)
            n = g1.shape[0]
            if g2.shape[0] != n or k <= 0 or n % k:
                raise ValueError("g1 / g2 sizes do not match or are not a multiple of k")
            i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        except (TypeError, ValueError):
            self._join_failed("check")
            raise
        ok = np.empty(n // k, dtype=np.uint8)
        allok = ctypes.c_int(1)
        self._chk(self._lib.zkp_pairing_check_batch_allreduce(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n // k, k, _ptr(ok), ctypes.byref(allok)))
        return ok, bool(allok.value)

    def pairing_gt_check_allreduce(self, g1, g2, k, out_gt, ok, all_ok, inf1=None, inf2=None):
        """device tensors only: pairing_gt_check of this rank's block + the AND over all ranks in all_ok (int32[1], required) -
        zkp_pairing_gt_check_batch_allreduce_dev, BASELINE config 3 as one call per rank"""
        import torch
        try:
            self._t_check(all_ok, None, "all_ok", rows=1, dtypes=(torch.int32,))
        except (TypeError, ValueError, AttributeError):
            # a flag tensor this rank cannot even hand to the library (wrong dtype / shape / device, not a tensor): the peers are waiting
            # in the collective all the same - join it with a flag of the wrapper's own, then raise
            self._join_failed("gt_check_dev", torch.empty(1, dtype=torch.int32, device=torch.device("cuda", self.device)))
            raise
        try:
            n = self._t_pairs(g1, g2, inf1, inf2, k)
            if out_gt is not None:
                self._t_check(out_gt, 72, "out_gt", rows=n // k)
            self._t_bytes(ok, n // k, "ok")
        except (TypeError, ValueError):
            self._join_failed("gt_check_dev", all_ok)
            raise
        self._chk(self._lib.zkp_pairing_gt_check_batch_allreduce_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n // k, k,
                                                                     self._tp(out_gt), self._tp(ok), self._tp(all_ok), self._stream()))

    def points_check_allreduce(self, g1_bytes, g2_bytes, k):
        """config 5 on a node: this rank's block of points_check + the AND over all ranks (host arrays) -> (st1, st2, ok, all_ok)"""
        try:
            b1 = np.ascontiguousarray(g1_bytes, dtype=np.uint8).reshape(-1, 96)
            b2 = np.ascontiguousarray(g2_bytes, dtype=np.uint8).reshape(-1, 192)
            n = b1.shape[0]
            if b2.shape[0] != n or k <= 0 or n % k:
                raise ValueError("byte strings / k do not match")
        except (TypeError, ValueError):
            self._join_failed("points")
            raise
        s1, s2, okb = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8), np.empty(n // k, dtype=np.uint8)
        allok = ctypes.c_int(1)
        self._chk(self._lib.zkp_points_check_batch_allreduce(self._h, _ptr(b1), _ptr(b2), n // k, k, _ptr(s1), _ptr(s2), _ptr(okb), ctypes.byref(allok)))
        return s1, s2, okb, bool(allok.value)

    def pairing_product_check_allgather(self, g1, g2, inf1=None, inf2=None):
        try:
            g1, g2 = _np(g1, 12), _np(g2, 24)
            n = g1.shape[0]
            if g2.shape[0] != n:
                raise ValueError("g1 and g2 hold different numbers of points")
            i1, i2 = _flags(inf1, n, "inf1"), _flags(inf2, n, "inf2")
        except (TypeError, ValueError):
            self._join_failed("product")
            raise
        gt = np.empty(72, dtype=np.uint64)
        one = ctypes.c_int(0)
        self._chk(self._lib.zkp_pairing_product_check_allgather(self._h, _ptr(g1), _ptr(g2), _ptr(i1), _ptr(i2), n, _ptr(gt), ctypes.byref(one)))
        return gt, bool(one.value)

    FP_OPS = {"mul": 0, "add": 1, "sub": 2, "neg": 3, "square": 4, "invert": 5}
    TOWER_OPS = {"fp2_mul": 0, "fp2_square": 1, "fp6_mul": 2, "fp6_square": 3, "fp6_frobenius": 4, "fp12_mul": 5, "fp12_square": 6,
                 "fp12_mul_by_014": 7, "fp12_frobenius": 8, "fp12_conjugate": 9, "fp12_cyclotomic_square": 10, "fp12_cyclotomic_pow2k": 11, "fp12_cyclotomic_decompress": 12,
                 "fp2_invert": 13, "fp2_mul_by_nonresidue": 14, "fp2_mul_fp": 15, "fp6_mul_by_1": 16, "fp6_mul_by_01": 17, "fp6_mul_by_nonresidue": 18,
                 "fp6_invert": 19, "fp12_invert": 20}

    def fp_op(self, op, a, b=None, core28=False):
        """zkVM-precompile-shaped batched field op: op 0 = mul, 1 = add (reference src/fp.rs:376,443), 2 sub, 3 neg,
        4 square, 5 invert (or the names in FP_OPS); core28 runs it on the 28-bit core of the cooperative family."""
        op = self.FP_OPS.get(op, op)
        a = _np(a, 6)
        b = None if b is None else _np(b, 6)
        if b is not None and b.shape != a.shape:
            raise ValueError("fp_op: operand shapes differ")
        out = np.empty_like(a)
        self._chk(self._lib.zkp_fp_op_batch(self._h, int(op) | (16 if core28 else 0), _ptr(a), _ptr(b), a.shape[0], _ptr(out)))
        return out

    def tower_op(self, op, a, b=None, repeat=1):
        """one tower operation per (n,72) record on the engine's kernel family (zkp_tower_op_batch); op: name in TOWER_OPS"""
        op = self.TOWER_OPS.get(op, op)
        a = _np(a, 72)
        b = None if b is None else _np(b, 72)
        if b is not None and b.shape != a.shape:
            raise ValueError("tower_op: operand shapes differ")
        out = np.empty_like(a)
        self._chk(self._lib.zkp_tower_op_batch(self._h, int(op), _ptr(a), _ptr(b), a.shape[0], int(repeat), _ptr(out)))
        return out

    # ------------------------------------------------------------------ torch (device-resident) API
    def _t_check(self, t, cols, name="tensor", rows=None, dtypes=None):
        """every tensor handed to a *_dev entry point: on this engine's GPU, contiguous, 64-bit words (or bytes), and
        large enough - a wrong-sized or CPU tensor would otherwise become an out-of-bounds access in HBM"""
        import torch
        if not _is_torch(t):
            raise TypeError("%s: expected a torch tensor" % name)
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError("%s must live on cuda:%d" % (name, self.device))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
        if t.dtype not in (dtypes or (torch.int64, torch.uint64)):
            raise ValueError("%s has dtype %s" % (name, t.dtype))
        if cols is not None and t.numel() % cols:
            raise ValueError("%s: %d elements is not a multiple of %d" % (name, t.numel(), cols))
        if rows is not None and t.numel() < rows * (cols or 1):
            raise ValueError("%s: %d elements, %d needed" % (name, t.numel(), rows * (cols or 1)))
        return t

    def _t_bytes(self, t, n, name):
        import torch
        return None if t is None else self._t_check(t, None, name, rows=n, dtypes=(torch.uint8,))

    def _t_pairs(self, g1, g2, inf1, inf2, k=1):
        self._t_check(g1, 12, "g1"), self._t_check(g2, 24, "g2")
        n = g1.numel() // 12
        if g2.numel() // 24 != n:
            raise ValueError("g1 holds %d points, g2 %d" % (n, g2.numel() // 24))
        if k <= 0 or n % k:
            raise ValueError("%d pairs is not a positive multiple of k = %d" % (n, k))
        self._t_bytes(inf1, n, "inf1"), self._t_bytes(inf2, n, "inf2")
        return n

    @staticmethod
    def _tp(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _pairing_t(self, g1, g2, inf1, inf2, out=None):
        import torch
        n = self._t_pairs(g1, g2, inf1, inf2)
        if out is None:
            out = torch.empty((n, 72), dtype=g1.dtype, device=g1.device)
        self._t_check(out, 72, "out", rows=n)
        self._chk(self._lib.zkp_pairing_batch_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n, self._tp(out), self._stream()))
        return out

    def _miller_t(self, g1, g2, k, inf1, inf2):
        import torch
        n = self._t_pairs(g1, g2, inf1, inf2, k)
        out = torch.empty((n // k, 72), dtype=g1.dtype, device=g1.device)
        self._chk(self._lib.zkp_multi_miller_loop_batch_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n // k, k, self._tp(out), self._stream()))
        return out

    def _fexp_t(self, f):
        import torch
        self._t_check(f, 72, "f")
        out = torch.empty_like(f)
        self._chk(self._lib.zkp_final_exponentiation_batch_dev(self._h, self._tp(f), f.numel() // 72, self._tp(out), self._stream()))
        return out

    def _check_t(self, g1, g2, k, inf1, inf2):
        import torch
        n = self._t_pairs(g1, g2, inf1, inf2, k)
        ok = torch.empty(n // k, dtype=torch.uint8, device=g1.device)
        allok = torch.empty(1, dtype=torch.int32, device=g1.device)
        self._chk(self._lib.zkp_pairing_check_batch_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n // k, k, self._tp(ok), self._tp(allok), self._stream()))
        return ok, allok

    def pairing_gt_check(self, g1, g2, k, out_gt, ok, all_ok, inf1=None, inf2=None):
        """device tensors only: Gt out + ok bytes + AND flag in one fused pass (bench step); out_gt / ok / all_ok are each optional"""
        import torch
        n = self._t_pairs(g1, g2, inf1, inf2, k)
        if out_gt is not None:
            self._t_check(out_gt, 72, "out_gt", rows=n // k)
        self._t_bytes(ok, n // k, "ok")
        if all_ok is not None:
            self._t_check(all_ok, None, "all_ok", rows=1, dtypes=(torch.int32,))
        self._chk(self._lib.zkp_pairing_gt_check_batch_dev(self._h, self._tp(g1), self._tp(g2), self._tp(inf1), self._tp(inf2), n // k, k,
                                                           self._tp(out_gt), self._tp(ok), self._tp(all_ok), self._stream()))

    def _valid_t(self, pts, inf, which):
        import torch
        cols = 12 if which == 1 else 24
        self._t_check(pts, cols, "points")
        n = pts.numel() // cols
        self._t_bytes(inf, n, "inf")
        st = torch.empty(n, dtype=torch.uint8, device=pts.device)
        fn = self._lib.zkp_g1_is_valid_batch_dev if which == 1 else self._lib.zkp_g2_is_valid_batch_dev
        self._chk(fn(self._h, self._tp(pts), self._tp(inf), n, self._tp(st), self._stream()))
        return st

    def _mul_t(self, base, scalars, which):
        import torch
        cols = 12 if which == 1 else 24
        self._t_check(scalars, 4, "scalars")
        n = scalars.numel() // 4
        if not _is_torch(base):
            base = torch.from_numpy(np.ascontiguousarray(base, dtype=np.uint64).view(np.int64)).to(scalars.device)
        base = base.contiguous().view(-1)
        self._t_check(base, cols, "base")
        stride = 0 if base.numel() == cols and n != 1 else cols
        if stride and base.numel() != n * cols:
            raise ValueError("base holds %d points, %d scalars given" % (base.numel() // cols, n))
        out = torch.empty((n, cols), dtype=scalars.dtype, device=scalars.device)
        oi = torch.empty(n, dtype=torch.uint8, device=scalars.device)
        fn = self._lib.zkp_g1_mul_batch_dev if which == 1 else self._lib.zkp_g2_mul_batch_dev
        self._chk(fn(self._h, self._tp(base), stride, self._tp(scalars), n, self._tp(out), self._tp(oi), self._stream()))
        return out, oi

    def clock_probe(self, stream, spin_us=20000):
        """queue the one-wavefront clock probe on `stream` (a torch.cuda.Stream); -> (tensor of two int64: shader-clock
        ticks, wall-clock ticks; valid once the stream has drained, wall clock rate in kHz)"""
        import torch
        out = torch.empty(2, dtype=torch.int64, device=torch.device("cuda", self.device))   # no fill: nothing queued on another stream may touch it
        khz = ctypes.c_int(0)
        self._chk(self._lib.zkp_clock_probe_dev(self._h, ctypes.c_void_p(stream.cuda_stream), int(spin_us), self._tp(out), ctypes.byref(khz)))
        return out, khz.value

    def time_coop_step(self, which, n):
        """ms of one launch of a diagnostic / single-kernel timing run (zkp_time_coop_step)"""
        ms = ctypes.c_float()
        self._chk(self._lib.zkp_time_coop_step(self._h, int(which), int(n), ctypes.byref(ms)))
        return ms.value

    PROFILE_CLASSES = ("k_prep_lines", "k_coop<30,4> miller", "k_coop<24,34> fexp_a", "k_batch_inv", "k_ksq", "k_kdec_a", "k_kdec_b",
                       "k_coop<36,24> hard-part step programs", "k_coop<24,34> phase-C step programs")

    def profile_pairing(self, g1, g2, out):
        """one pass of the fused pairing with every launch timed on its own (zkp_profile_pairing_dev) -> {class: (ms, launches)}"""
        n = self._t_pairs(g1, g2, None, None)
        self._t_check(out, 72, "out", rows=n)
        k = len(self.PROFILE_CLASSES)
        ms, cnt = (ctypes.c_float * k)(), (ctypes.c_int * k)()
        self._chk(self._lib.zkp_profile_pairing_dev(self._h, self._tp(g1), self._tp(g2), n, self._tp(out), ms, cnt))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(self.PROFILE_CLASSES)}

    def time_pairing(self, g1, g2, out, reps):
        """avg ms per launch of the fused pairing kernel, HIP events on the engine's own stream."""
        n = self._t_pairs(g1, g2, None, None)
        self._t_check(out, 72, "out", rows=n)
        ms = ctypes.c_float()
        self._chk(self._lib.zkp_time_pairing_dev(self._h, self._tp(g1), self._tp(g2), g1.numel() // 12, self._tp(out), int(reps), ctypes.byref(ms)))
        return ms.value
