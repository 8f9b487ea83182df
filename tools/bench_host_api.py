#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points zkp_pairing_batch / zkp_pairing_check_batch (H2D + kernels +
D2H from pageable numpy arrays; batches above ZKP_HOST_SLICE pairs are pipelined in slices), reported in DESIGN.md
only - never as bench.py's value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
eng = z.PairingEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
g1, g2, _, _ = synthetic.random_pairs(eng, n)
eng.pairing(g1[:1024], g2[:1024])
t = time.perf_counter(); out = eng.pairing(g1, g2); dt = time.perf_counter() - t
print("host-pointer zkp_pairing_batch: %d pairs in %.1f ms -> %.0f pairings/s (PCIe-inclusive, pageable host memory)" % (n, dt * 1e3, n / dt))
t = time.perf_counter(); ok, allok = eng.pairing_check(g1, g2, 1); dt = time.perf_counter() - t
print("host-pointer zkp_pairing_check_batch (flags only out): %.1f ms -> %.0f pairings/s" % (dt * 1e3, n / dt))
