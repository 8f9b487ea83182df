#!/usr/bin/env python3
"""multi_miller_loop() of checks with k pairs, k around and above the eight an unrolled Miller program takes (round 5: more pairs share ONE
accumulator through the run-time-k program; ZKP_COOP_NO_STREAM=1 = groups of eight joined by f12mul, rounds 1-4).  Prints ms per batch and
the sha256 of the Miller values, so that two runs (with and without the knob) can be compared:  python3 tools/time_kpairs.py [pairs_total]"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
eng = z.PairingEngine(0)
g1, g2, _, _ = synthetic.random_pairs(eng, total, seed=17, device_tensors=True)
out = {"pairs": total, "streaming": os.environ.get("ZKP_COOP_NO_STREAM") != "1"}
for k in (8, 9, 12, 16, 24, 32, 48, 64, 96):
    n = total // k * k
    a, b = g1[:n], g2[:n]
    ml = eng.multi_miller_loop(a, b, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        ml = eng.multi_miller_loop(a, b, k)
    e1.record()
    e1.synchronize()
    out["k%d" % k] = {"checks": n // k, "ms": e0.elapsed_time(e1) / 2, "sha256": hashlib.sha256(ml.cpu().numpy().tobytes()).hexdigest()[:16]}
print(json.dumps(out))
