#!/usr/bin/env python3
"""G1 / G2 is_valid on resident points over batch sizes (ms per call, HIP events): tools/valid_sweep.py  ->  one JSON line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
eng = z.PairingEngine(0)
N = 1 << 18
g1, g2, _, _ = synthetic.random_pairs(eng, N, seed=3, device_tensors=True)
rows = []
for n in (4096, 16384, 32768, 49152, 57344, 65536, 98304, 131072, 196608, 262144):
    r = {"n": n}
    for name, fn, pts in (("g1", eng.g1_is_valid, g1), ("g2", eng.g2_is_valid, g2)):
        fn(pts[:n]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            st = fn(pts[:n])
        e1.record(); e1.synchronize()
        r[name + "_ms"] = e0.elapsed_time(e1) / 4                  # back to back: each launch behind an identical one
        lone = []
        for _ in range(5):                                        # ONE call behind an idle GPU / a kernel of another shape
            junk = torch.zeros(1 << 20, device=pts.device) + 1.0
            torch.cuda.synchronize()
            e0.record(); st = fn(pts[:n]); e1.record(); e1.synchronize()
            lone.append(e0.elapsed_time(e1))
        r[name + "_lone_ms"] = sorted(lone)[2]
        r[name + "_bad"] = int((st != 0).sum().item())
    rows.append(r)
print(json.dumps({"knobs": {k: v for k, v in os.environ.items() if k.startswith("ZKP_")}, "rows": rows}))
