"""ctypes loader for libzkp_pairings.so (the C ABI declared in include/zkp_pairings.h).

No fallback of any kind: if the library is missing or a GPU is not usable the import / call
raises.  The product never imports anything under oracle/."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKP_LIB_PATH: A/B timing of two builds of the same library on one GPU box (development only)
LIB_PATH = os.environ.get("ZKP_LIB_PATH") or os.path.join(_HERE, "libzkp_pairings.so")

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_int = ctypes.c_int

# name -> (restype, argtypes); MUST list every symbol include/zkp_pairings.h declares
SIGNATURES = {
    "zkp_abi_version": (c_int, []),
    "zkp_strerror": (ctypes.c_char_p, [c_int]),
    "zkp_init": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "zkp_free": (None, [c_vp]),
    "zkp_last_error": (ctypes.c_char_p, [c_vp]),
    "zkp_set_validate": (c_int, [c_vp, c_int]),
    "zkp_set_kernel": (c_int, [c_vp, c_int]),
    "zkp_device_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.c_char_p, c_sz]),
    "zkp_gt_identity": (c_u64p, []),
    "zkp_pairing_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_multi_miller_loop_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp]),
    "zkp_final_exponentiation_batch": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "zkp_pairing_check_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, ctypes.POINTER(c_int)]),
    "zkp_miller_product": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_fp12_product": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "zkp_pairing_product_check": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, ctypes.POINTER(c_int)]),
    "zkp_g1_is_valid_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_g2_is_valid_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_g1_mul_batch": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_vp]),
    "zkp_g2_mul_batch": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_vp]),
    "zkp_g1_decode_batch": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "zkp_g2_decode_batch": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "zkp_g1_encode_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_g2_encode_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "zkp_fp_op_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "zkp_tower_op_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, ctypes.c_uint32, c_vp]),
    "zkp_pairing_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_multi_miller_loop_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp]),
    "zkp_final_exponentiation_batch_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_pairing_check_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp]),
    "zkp_pairing_gt_check_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "zkp_miller_product_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_fp12_product_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_pairing_product_check_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "zkp_g1_is_valid_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_g2_is_valid_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_g1_mul_batch_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "zkp_g2_mul_batch_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "zkp_g1_decode_batch_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "zkp_g2_decode_batch_dev": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "zkp_g1_encode_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_g2_encode_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "zkp_points_check_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, ctypes.POINTER(c_int)]),
    "zkp_points_check_batch_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "zkp_comm_unique_id": (c_int, [c_vp]),
    "zkp_comm_init_rank": (c_int, [c_vp, c_int, c_int, c_vp]),
    "zkp_comm_destroy": (c_int, [c_vp]),
    "zkp_comm_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "zkp_and_allreduce_dev": (c_int, [c_vp, c_vp, c_vp]),
    "zkp_pairing_check_batch_allreduce": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, ctypes.POINTER(c_int)]),
    "zkp_pairing_check_batch_allreduce_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp]),
    "zkp_pairing_gt_check_batch_allreduce_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "zkp_points_check_batch_allreduce": (c_int, [c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, ctypes.POINTER(c_int)]),
    "zkp_points_check_batch_allreduce_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "zkp_pairing_product_check_allgather": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, ctypes.POINTER(c_int)]),
    "zkp_take_validation_status_dev": (c_int, [c_vp, c_vp, ctypes.POINTER(c_int)]),
    "zkp_pairing_check_batch_multi": (c_int, [ctypes.POINTER(c_vp), c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_vp, ctypes.POINTER(c_int)]),
    "zkp_pairing_batch_multi": (c_int, [ctypes.POINTER(c_vp), c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, ctypes.POINTER(c_int)]),
    "zkp_host_alloc": (c_int, [c_sz, ctypes.POINTER(c_vp)]),
    "zkp_host_free": (c_int, [c_vp]),
    "zkp_host_register": (c_int, [c_vp, c_sz]),
    "zkp_host_unregister": (c_int, [c_vp]),
    "zkp_clock_probe_dev": (c_int, [c_vp, c_vp, ctypes.c_uint, c_vp, ctypes.POINTER(c_int)]),
    "zkp_time_coop_step": (c_int, [c_vp, c_int, c_sz, ctypes.POINTER(ctypes.c_float)]),
    "zkp_profile_pairing_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(c_int)]),
    "zkp_time_pairing_dev": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_int, ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


class ZkpError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = load().zkp_strerror(status).decode()
        super().__init__("zkp status %d (%s)%s" % (status, msg, (": " + detail) if detail else ""))


def load():
    """Load the shared library (no GPU is touched by loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C zkvm_pairings_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as the
        # system one).  Device pointers and streams are shared with torch tensors, so torch's copy must
        # be the one that is loaded first; without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
