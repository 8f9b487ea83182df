"""MI355X-native batched BLS12-381 pairing engine behind the zkvm-pairings API shape.

Loading this package loads libzkp_pairings.so (the C ABI in include/zkp_pairings.h); it raises if
the library has not been built.  There is no CPU fallback."""
from . import _lib
from ._lib import ZkpError

_lib.load()

from .engine import KERNEL_AUTO, KERNEL_COOP, KERNEL_THREAD, PairingEngine  # noqa: E402
from .pairings import (G1Affine, G2Affine, Gt, MillerLoopResult, final_exponentiation,  # noqa: E402
                       multi_miller_loop, pairing)
from . import synthetic  # noqa: E402

__all__ = ["PairingEngine", "G1Affine", "G2Affine", "Gt", "MillerLoopResult", "pairing", "multi_miller_loop",
           "final_exponentiation", "synthetic", "ZkpError", "KERNEL_AUTO", "KERNEL_THREAD", "KERNEL_COOP"]
