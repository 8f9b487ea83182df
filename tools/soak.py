#!/usr/bin/env python3
"""One-off soak: large random batches through both kernel families, EVERY output compared with the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
eng = z.PairingEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
rng = np.random.default_rng(12345)
g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=0xABCDEF)
inf1 = (rng.random(n) < 0.01).astype(np.uint8)
inf2 = (rng.random(n) < 0.01).astype(np.uint8)
t = time.time(); want = o.pairing_batch(g1, g2, inf1, inf2, nthreads=16); print("oracle pairing %.1f s" % (time.time() - t), flush=True)
for kern in ("coop", "thread"):
    eng.set_kernel(kern)
    got = eng.pairing(g1, g2, inf1, inf2)
    bad = np.flatnonzero((got != want).any(axis=1))
    print(kern, "pairing mismatches:", bad.size, flush=True)
    assert bad.size == 0
    ml = eng.multi_miller_loop(g1[:n // 4 * 3], g2[:n // 4 * 3], 3, inf1[:n // 4 * 3], inf2[:n // 4 * 3])
    if kern == "coop":
        wml = o.multi_miller_loop_batch(g1[:3 * 8192], g2[:3 * 8192], 8192, 3, inf1[:3 * 8192], inf2[:3 * 8192])
        assert np.array_equal(ml[:8192], wml)
        ml_coop = ml
    else:
        assert np.array_equal(ml, ml_coop)
    fe = eng.final_exponentiation(ml)
    if kern == "coop":
        fe_coop = fe
        assert np.array_equal(fe[:8192], o.final_exponentiation_batch(wml))
    else:
        assert np.array_equal(fe, fe_coop)
    print(kern, "3-pair miller/final-exp consistent", flush=True)
print("SOAK OK", n)
