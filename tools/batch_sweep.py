#!/usr/bin/env python3
"""Batch-size sweep of the fused pairing on resident inputs (round 6: the small / medium-batch regime).

  tools/batch_sweep.py [--sizes 1,4,16,...] [--k 1,3] [--reps R] [--tag TEXT]

For every n (checks) and k (pairs per check): ms per call as a host sees it (call + stream synchronisation, mean of the repetitions),
ms per call when calls are queued back to back (HIP-event time over the repetitions / repetitions) and checks/s from the latter.  One
JSON line on stdout; the environment (ZKP_COOP_* knobs, ZKP_LIB_PATH) selects what is measured, --tag labels the line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic

DEFAULT_SIZES = [1, 4, 16, 64, 256, 1024, 4096, 1 << 14, 1 << 16, 1 << 17, 1 << 18]


def sweep(eng, g1, g2, sizes, ks, reps=None, gt_out=True, graph=False):
    """rows of {n, k, ms_call_sync, ms_queued, checks_per_s} for the resident pair tensors g1 / g2 (at least max(n) * max(k) pairs)"""
    dev = g1.device
    rows = []
    for k in ks:
        for n in sizes:
            if n * k > g1.shape[0]:
                continue
            a, b = g1[: n * k], g2[: n * k]
            gt = torch.empty((n, 72), dtype=torch.int64, device=dev) if gt_out else None
            ok = torch.empty(n, dtype=torch.uint8, device=dev)
            flag = torch.empty(1, dtype=torch.int32, device=dev)
            call = lambda: eng.pairing_gt_check(a, b, k, gt, ok, flag)
            call()
            torch.cuda.synchronize()
            r = reps or (20 if n * k <= 4096 else 6 if n * k <= (1 << 16) else 3)
            t0 = time.perf_counter()
            for _ in range(r):
                call()
                torch.cuda.synchronize()
            ms_sync = (time.perf_counter() - t0) * 1e3 / r
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(r):
                call()
            e1.record()
            e1.synchronize()
            ms_q = e0.elapsed_time(e1) / r
            # the same call captured into a hipGraph and replayed (small n: where ~30 dependent launches could matter) - the measured answer to
            # "replay a graph of the launch sequence": the *_dev entry points are capturable once their workspaces exist
            ms_graph = None
            if graph and n * k <= 4096:
                try:
                    want = (gt.clone() if gt_out else None, ok.clone())
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        call()
                    ok.fill_(7)
                    g.replay()
                    torch.cuda.synchronize()
                    same = bool(torch.equal(ok, want[1]) and (not gt_out or torch.equal(gt, want[0])))
                    t0 = time.perf_counter()
                    for _ in range(r):
                        g.replay()
                        torch.cuda.synchronize()
                    ms_graph = {"ms_replay_sync": (time.perf_counter() - t0) * 1e3 / r, "bytes_equal_plain_call": same}
                    del g
                except Exception as ex:      # capture unsupported by the runtime at hand: reported, never fatal
                    ms_graph = {"error": repr(ex)}
            # fingerprint of the Gt block and the ok bytes (wrap-around sums on the GPU): equal across knob settings of one build
            w64 = torch.arange(1, 73, dtype=torch.int64, device=dev) * 0x1E3779B97F4A7C15 | 1
            fp = [int((gt * w64).sum().item()) if gt_out else 0, int(ok.sum().item()), int(flag.item())]
            rows.append({"n": n, "k": k, "ms_call_sync": ms_sync, "ms_queued": ms_q, "checks_per_s": n / ms_q * 1e3, "fp": fp})
            if ms_graph is not None:
                rows[-1]["hipgraph"] = ms_graph
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default=",".join(str(s) for s in DEFAULT_SIZES))
    ap.add_argument("--k", default="1")
    ap.add_argument("--reps", type=int, default=0)
    ap.add_argument("--tag", default="")
    ap.add_argument("--graph", action="store_true", help="also capture each small call into a hipGraph and time its replay")
    args = ap.parse_args()
    sizes = [int(s) for s in args.sizes.split(",") if s]
    ks = [int(s) for s in args.k.split(",") if s]
    eng = z.PairingEngine(0)
    g1, g2, _, _ = synthetic.random_pairs(eng, max(sizes) * max(ks), seed=synthetic.SEED, device_tensors=True)
    torch.cuda.synchronize()
    rows = sweep(eng, g1, g2, sizes, ks, args.reps or None, graph=args.graph)
    knobs = {k: v for k, v in os.environ.items() if k.startswith("ZKP_")}
    print(json.dumps({"tag": args.tag, "knobs": knobs, "rows": rows}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
