#!/usr/bin/env python3
"""PCIe-inclusive times of the host-pointer entry point zkp_pairing_batch (upload + kernels + download inside the call) from page-locked
arrays, from pageable arrays with a reused output, and from pageable arrays with a NEW output array per call (fresh pages) - the host_api
block of the bench line on its own; never bench.py's value.   tools/bench_host_api.py [pairs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic

eng = z.PairingEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
g1, g2, _, _ = synthetic.random_pairs(eng, n)
hp1, hp2, hgt = eng.host_array((n, 12)), eng.host_array((n, 24)), eng.host_array((n, 72))
hp1[:], hp2[:] = g1, g2
pgt = np.zeros((n, 72), dtype=np.uint64)


def ms(fn, reps=2):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) * 1e3 / reps


def fresh_call_only(reps=3):
    """a NEW output array per call, but only the library call is timed (np.empty before the clock starts, the array freed after it stops)"""
    tot = 0.0
    for _ in range(reps):
        o_ = np.empty((n, 72), dtype=np.uint64)
        t = time.perf_counter()
        eng.pairing(g1, g2, out=o_)
        tot += time.perf_counter() - t
        del o_
    return tot * 1e3 / reps


out = {"pairs": n, "knobs": {k: v for k, v in os.environ.items() if k.startswith("ZKP_")},
       "pinned_gt_out_ms": ms(lambda: eng.pairing(hp1, hp2, out=hgt)),
       "pageable_gt_out_reused_ms": ms(lambda: eng.pairing(g1, g2, out=pgt)),
       "pageable_gt_out_fresh_pages_ms": ms(lambda: eng.pairing(g1, g2)),
       "pageable_gt_out_fresh_pages_call_only_ms": fresh_call_only(),
       "pageable_flags_only_ms": ms(lambda: eng.pairing_check(g1, g2, 1)),
       "equal": bool(np.array_equal(pgt, hgt))}
print(json.dumps(out))
