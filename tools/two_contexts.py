#!/usr/bin/env python3
"""A STREAM of medium batches: calls of n pairs queued back to back on ONE context / stream against the same calls alternating over TWO
contexts on two streams (each context has its own workspace and pipelines, so one call's latency-bound islands run beside the other call's
Miller loop).  tools/two_contexts.py [n] [calls]  ->  one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic


def run(n=1 << 16, calls=16, contexts=2):
    dev = torch.device("cuda", 0)
    engs = [z.PairingEngine(0) for _ in range(contexts)]
    g1, g2, _, _ = synthetic.random_pairs(engs[0], n, seed=synthetic.SEED, device_tensors=True)
    bufs = [(torch.empty((n, 72), dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev),
             torch.empty(1, dtype=torch.int32, device=dev)) for _ in range(contexts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(contexts)]
    torch.cuda.synchronize()

    def go(m):
        for i in range(calls):
            j = i % m
            with torch.cuda.stream(streams[j]):
                engs[j].pairing_gt_check(g1, g2, 1, *bufs[j])
    out = {"pairs_per_call": n, "calls": calls}
    for m in range(1, contexts + 1):
        go(m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams[:m]:
            s.wait_event(e0)
        go(m)
        for s in streams[:m]:
            e = torch.cuda.Event()
            e.record(s)
            torch.cuda.current_stream().wait_event(e)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        out["contexts_%d" % m] = {"ms_total": ms, "ms_per_call": ms / calls, "pairings_per_s": n * calls / ms * 1e3}
    out["gt_equal"] = bool(torch.equal(bufs[0][0], bufs[-1][0]))
    for e in engs:
        e.close()
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    print(json.dumps(run(n, calls, ctx)))
