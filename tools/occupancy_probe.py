import sys, os
sys.path.insert(0, os.getcwd())
import zkvm_pairings_amd as z
eng = z.PairingEngine(0)
for n in (4096, 12288, 14336, 16384, 20480, 24576, 28672, 32768, 49152, 65536):
    ms9 = eng.time_coop_step(9, n)     # 400 compressed squarings: n/16 wavefronts
    ms11 = None
    try:
        eng.time_coop_step(10, min(n, 65536)); ms11 = eng.time_coop_step(11, min(n, 65536))   # Miller program: n/5 wavefronts
    except Exception as ex:
        ms11 = -1
    ms16, ms17 = eng.time_coop_step(16, n), eng.time_coop_step(17, n)
    ms18, ms19 = eng.time_coop_step(18, n), eng.time_coop_step(19, n)
    print("n=%6d  ksq waves %5d: %.3f ms per 400 sq (%.2f us/sq = %.0f us per 57)  57 sq behind a step program: %.0f us, behind an idle GPU: %.0f us, behind a 1-squaring launch: %.0f us, behind program + 1-squaring launch: %.0f us   miller waves %5d: %.3f ms" % (
        n, (n + 15) // 16, ms9, ms9 * 1e3 / 400, ms9 * 1e3 / 400 * 57, ms16 * 1e3, ms17 * 1e3, ms18 * 1e3, ms19 * 1e3, (n + 4) // 5, ms11))
