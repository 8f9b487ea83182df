#!/usr/bin/env python3
"""Per-step cost of the cooperative interpreter from synthetic programs (400 iterations each)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkvm_pairings_amd as z
eng = z.PairingEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
names = ["T=1", "T=3", "T=3+epi", "T=6", "T=12", "LIN", "cyc_sqr", "cyc_sqr+companions", "spill12+fill12", "ksq (compressed, 4 lanes/check)", None, None, "T=6 one-slot", "T=12 one-slot", "T=12 B two-slot"]
res = {}
for i, nm in enumerate(names):
    if nm is None:
        continue
    ms = ctypes.c_float()
    rc = eng._lib.zkp_time_coop_step(eng._h, i, n, ctypes.byref(ms))
    assert rc == 0, rc
    per_wave = 16 if i == 9 else 5
    waves = (n + per_wave - 1) // per_wave
    per_simd = waves / 1024.0
    # cycles per step per SIMD-wave-slot at 2.1 GHz: time * f / (iterations * waves per SIMD)
    cyc = ms.value * 1e-3 * 2.1e9 / (400 * per_simd)
    res[nm] = cyc
    print("%-22s %.3f ms  => %.0f SIMD-cycles per step (at 2.1 GHz), %.0f per check" % (nm, ms.value, cyc, cyc / per_wave))
P = (res["T=12"] - res["T=6"]) / 6
print("P (per product block) = %.0f ; R (T=1 minus P) = %.0f ; epilogue = %.0f ; LIN = %.0f" % (P, res["T=1"] - P, res["T=3+epi"] - res["T=3"], res["LIN"]))
P1 = (res["T=12 one-slot"] - res["T=6 one-slot"]) / 6
print("one-slot forms: term = %.0f ; per-step part = %.0f ; B two-slot term = %.0f" % (P1, res["T=6 one-slot"] - 6 * P1, (res["T=12 B two-slot"] - (res["T=6 one-slot"] - 6 * P1)) / 12))
