#!/usr/bin/env python3
"""k_ksq on REAL data at several wavefront counts (run under rocprofv3 --kernel-trace and read the k_ksq durations): 57 compressed squarings of
n Gt elements through zkp_tower_op_batch(FP12_CYCLOTOMIC_POW2K) - against the synthetic timing hook (tools/occupancy_probe.py), whose
operands are a repeated byte."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
eng = z.PairingEngine(0)
g1, g2, _, _ = synthetic.random_pairs(eng, 49152, seed=5)
gt = eng.pairing(g1, g2)
for n in (256, 4096, 8192, 12288, 16384, 20480, 24576, 32768, 49152):
    for _ in range(2):
        out = eng.tower_op("fp12_cyclotomic_pow2k", gt[:n], repeat=57)
print("done", out.shape)
