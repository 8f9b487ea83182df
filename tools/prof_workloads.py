#!/usr/bin/env python3
"""The workloads of BASELINE configs 4 and 5 (and the multi_miller_loop() ABI) as a bare profiling target: only the kernels of the chosen
set reach the GPU behind the input generation, `--reps` launches each, so that a rocprofv3 trace / counter run of this command holds
exactly them (tools/collect_profiles2.sh; tools/pmc_workloads.py reduces the counter CSVs per kernel).

  --set a   config 4: 2^18 three-pair checks (k_prep_lines<true> k = 3, k_coop<30,4> miller3, one final exponentiation per check)
            config 5: G1 / G2 is_valid of 2^20 points (k_g1_valid28, k_g2_valid28), and the G1 / G2 scalar multiplication the
            input generation is made of (k_g1_mul28, k_g2_mul28: the 3 x 2^18 + ... points of this run)
  --set b   multi_miller_loop() of 2^20 pairs, k = 1: k_prep_lines<false> (upstream-shaped lines) + k_coop<30,4> miller1 to wire
  --set c   config 5 end to end: zkp_points_check_batch_dev over 2^20 encoded (G1, G2) pairs (decode, is_valid, pairing check)
Prints one JSON line with the wall time per launch (torch.cuda.synchronize around the loop)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
import zkvm_pairings_amd as z  # noqa: E402
from zkvm_pairings_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--set", default="a", choices=("a", "b", "c"))
ap.add_argument("--n", type=int, default=1 << 20, help="points / pairs (config 4 takes n / 4 three-pair checks)")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0, help="untimed launches ahead of the timed ones (workspace allocation happens in the first launch)")
a = ap.parse_args()
eng = z.PairingEngine(0)
dev = torch.device("cuda", 0)
n = a.n


def timed(fn):
    for _ in range(a.warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / a.reps


out = {"set": a.set, "n": n, "reps": a.reps, "launches": a.reps + a.warmup}
g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=synthetic.SEED, device_tensors=True)
if a.set == "a":
    nc = n // 4
    ok = torch.empty(nc, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    out["config4_ms"] = timed(lambda: eng.pairing_gt_check(g1[:3 * nc], g2[:3 * nc], 3, None, ok, flag))
    out["config4_checks"] = nc
    out["g1_valid_ms"] = timed(lambda: eng.g1_is_valid(g1))
    out["g2_valid_ms"] = timed(lambda: eng.g2_is_valid(g2))
elif a.set == "b":
    out["multi_miller_loop_ms"] = timed(lambda: eng.multi_miller_loop(g1, g2, 1))
else:
    b1 = eng.encode_points_dev(g1, 1)
    b2 = eng.encode_points_dev(g2, 2)
    st1 = torch.empty(n, dtype=torch.uint8, device=dev)
    st2 = torch.empty(n, dtype=torch.uint8, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    out["points_check_ms"] = timed(lambda: eng.points_check(b1, b2, 1, st1, st2, ok, flag))
print(json.dumps(out), flush=True)
eng.close()
