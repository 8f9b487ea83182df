#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs 4 and 5) on one GPU: 3-pair Groth16-shaped checks with a
shared final exponentiation, and G1/G2 validity (on-curve + subgroup) of raw points.  Device-resident
inputs, HIP-event-free wall timing around torch.cuda.synchronize()."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 16)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
eng = z.PairingEngine(0)
dev = torch.device("cuda", 0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.reps


out = {}
n = a.n
g1, g2, _, _ = synthetic.random_pairs(eng, 3 * n, seed=7, device_tensors=True)
ok = torch.empty(n, dtype=torch.uint8, device=dev)
flag = torch.empty(1, dtype=torch.int32, device=dev)
for kern in ("coop", "thread"):
    eng.set_kernel(kern)
    dt = timed(lambda: eng.pairing_gt_check(g1, g2, 3, None, ok, flag))
    out["check3_%s_checks_per_s" % kern] = n / dt
# k = 8 pairs per check: two groups of four on the cooperative path, joined by one Fp12 multiplication
n8 = n // 4
g1k, g2k, _, _ = synthetic.random_pairs(eng, 8 * n8, seed=8, device_tensors=True)
ok8 = torch.empty(n8, dtype=torch.uint8, device=dev)
for kern in ("coop", "thread"):
    eng.set_kernel(kern)
    dt = timed(lambda: eng.pairing_gt_check(g1k, g2k, 8, None, ok8, flag))
    out["check8_%s_checks_per_s" % kern] = n8 / dt
del g1k, g2k
# the whole batch as ONE product check: Miller loops four pairs per accumulator, Fp12 product tree, one final exponentiation
np_ = 4 * n
g1p, g2p, _, _ = synthetic.random_pairs(eng, np_, seed=9, device_tensors=True)
for kern in ("coop", "thread"):
    eng.set_kernel(kern)
    dt = timed(lambda: eng.pairing_product_check(g1p, g2p))
    out["product_check_%s_pairs_per_s" % kern] = np_ / dt
del g1p, g2p
eng.set_kernel("auto")
p1, p2 = g1[:n].contiguous(), g2[:n].contiguous()
dt = timed(lambda: eng.g1_is_valid(p1))
out["g1_is_valid_points_per_s"] = n / dt
dt = timed(lambda: eng.g2_is_valid(p2))
out["g2_is_valid_points_per_s"] = n / dt
sc = torch.from_numpy(synthetic.scalars(3, n).view(np.int64)).to(dev)
dt = timed(lambda: eng.g1_mul(synthetic.G1_GENERATOR, sc))
out["g1_mul_points_per_s"] = n / dt
dt = timed(lambda: eng.g2_mul(synthetic.G2_GENERATOR, sc))
out["g2_mul_points_per_s"] = n / dt
print(json.dumps(out))
