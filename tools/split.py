#!/usr/bin/env python3
"""Where a pass spends its time, for the library ZKP_LIB_PATH selects (A/B of builds on one box: tools/ab.sh times the
whole pass, this prints the kernels and phases).  Device-resident inputs, torch events on the current stream."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
eng = z.PairingEngine(0)
dev = torch.device("cuda", 0)
g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=11, device_tensors=True)


def timed(fn, reps=2):
    r = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps, r


out = {"n": n}
torch.cuda.synchronize()   # the input generation ran on torch's stream, the step timings use the engine's own
nk = min(n, 1 << 16)
out["prep_lines_ms"] = eng.time_coop_step(10, nk) * n / nk
out["miller_prog_ms"] = eng.time_coop_step(11, nk) * n / nk
ml_ms, ml = timed(lambda: eng.multi_miller_loop(g1, g2, 1))
out["multi_miller_loop_ms"] = ml_ms
fe_ms, _ = timed(lambda: eng.final_exponentiation(ml))
out["final_exponentiation_ms"] = fe_ms
out["ksq_400sq_ms_per_2p16"] = eng.time_coop_step(9, 1 << 16)
for i, nm in [(0, "T1"), (3, "T6"), (4, "T12"), (5, "LIN"), (7, "cyc_comp"), (8, "fill"), (12, "T6s"), (13, "T12s"), (14, "T12b")]:
    out["step_" + nm + "_ms"] = eng.time_coop_step(i, 1 << 16)
gt = torch.empty((n, 72), dtype=torch.int64, device=dev)
ok = torch.empty(n, dtype=torch.uint8, device=dev)
flag = torch.empty(1, dtype=torch.int32, device=dev)
p_ms, _ = timed(lambda: eng.pairing_gt_check(g1, g2, 1, gt, ok, flag))
out["pairing_ms"] = p_ms
out["pairings_per_s"] = n / p_ms * 1e3
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}))
