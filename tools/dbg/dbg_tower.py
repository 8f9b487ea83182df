#!/usr/bin/env python3
"""debug: which checks / words of a tower op differ from the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import zkvm_pairings_amd as z
import oracle_lib as o
from test_gpu_parity import _rnd_records
eng = z.PairingEngine(0, kernel="coop")
for n in (40, 200):
    a2, b2 = _rnd_records(35, n, 2), _rnd_records(36, n, 2)
    a12, b12 = _rnd_records(31, n, 12), _rnd_records(32, n, 12)
    for rep in range(3):
        got = eng.tower_op("fp2_mul", a2, b2)
        bad = [(i, [w for w in range(72) if got[i][w] != (o.fp2_mul(a2[i, :12], b2[i, :12])[w] if w < 12 else 0)]) for i in range(n)]
        bad = [(i, ws) for i, ws in bad if ws]
        print("fp2_mul n=%d rep %d bad:" % (n, rep), [(i, ws[:4], len(ws)) for i, ws in bad][:12], len(bad))
        got = eng.tower_op("fp12_mul", a12, b12)
        bad = []
        for i in range(n):
            want = o.fp12_mul(a12[i], b12[i])
            ws = [w for w in range(72) if got[i][w] != want[w]]
            if ws: bad.append((i, ws[:4], len(ws)))
        print("fp12_mul n=%d rep %d bad:" % (n, rep), bad[:12], len(bad))
