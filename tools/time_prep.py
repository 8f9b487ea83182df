import sys, os
sys.path.insert(0, os.getcwd())
import zkvm_pairings_amd as z
eng = z.PairingEngine(0)
for w in (10, 15, 11):
    ms = [eng.time_coop_step(w, 1 << 16) for _ in range(3)]
    print(w, ["%.3f" % x for x in ms], "-> per 2^20: %.1f ms" % (min(ms) * 16))
