#!/usr/bin/env python3
"""debug: which checks / Fp coefficients of repeated compressed squarings (k_ksq + decompression) differ from the oracle"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
import oracle_lib as o
eng = z.PairingEngine(0, kernel="coop")
n = 40
g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=4711)
gt = o.pairing_batch(g1, g2, nthreads=8)
gt[0] = eng.gt_identity()
for rep in (1, 2, 3):
    got = eng.tower_op("fp12_cyclotomic_pow2k", gt, None, repeat=rep)
    rows = []
    for i in range(n):
        w = gt[i]
        for _ in range(rep):
            w = o.fp12_cyclotomic_square(w)
        bad = [c for c in range(12) if not np.array_equal(got[i][6 * c:6 * c + 6], w[6 * c:6 * c + 6])]
        rows.append(bad)
    print("rep", rep, "checks with wrong coefficients:", sum(1 for b in rows if b), "of", n)
    for i in range(min(n, 20)):
        print("   check %2d (lane quad %2d of its wavefront): wrong Fp coefficients %s" % (i, i % 16, rows[i]))
