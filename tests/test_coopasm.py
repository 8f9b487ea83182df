"""CPU gate for the hand-scheduled inline-asm blocks (tools/coopasm.py): the generated instruction text is executed by
tools/asmemu.py - a model of the gfx950 instructions the generator uses, with EXEC masks, DPP quad permutations and an LDS
image - and must give the integers of tools/coopgen.py's limb-exact kernel models."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import asmemu  # noqa: E402
import coopasm  # noqa: E402
import coopgen as cg  # noqa: E402

NL = 14


def _p_limbs():
    return [(cg.P >> (28 * i)) & 0xfffffff for i in range(NL)]


def _subst():
    s = {"p%d" % i: v for i, v in enumerate(_p_limbs())}
    s["pinv"] = (-pow(cg.P, -1, 1 << 28)) % (1 << 28)
    return s


def test_generated_constants_match_the_step_programs():
    assert coopasm.P_BLS == cg.P
    assert coopasm.p_balanced() == list(cg.P_BAL)
    assert (coopasm.VRED_C, coopasm.VRED_SHIFT_IN, coopasm.VRED_SHIFT_OUT) == (cg.VRED_C, cg.VRED_SHIFT_IN, cg.VRED_SHIFT_OUT)
    inc = open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_coop_mulacc.inc")).read()
    g = coopasm.generate_ksq()
    assert all(('"%s\\n\\t"' % l) in inc for l in g.lines[:50] + g.lines[-50:]), "zkp_coop_mulacc.inc is not what tools/coopasm.py generates"


def test_ksq_body_equals_the_model_of_a_compressed_squaring():
    """three consecutive squarings of a random (z2..z5) on one lane quad: after each, the block's X registers hold the lane's new
    coefficient, its Y registers the pair partner's, and the LDS image the parked copy - all equal to emu_ksq's values."""
    rng = random.Random(20261004)
    g = coopasm.generate_ksq()
    for trial in range(3):
        state = [cg.to_limbs_balanced(rng.randrange(cg.P) - cg.P // 2 if trial else rng.randrange(cg.P // 2)) for _ in range(12)]
        state = [cg.vred(list(x)) if abs(cg.limbs_value(x)) >= 0.51 * cg.P else x for x in state]
        # lane r's coefficient and its pair partner's, as k_ksq sets them up (emu_ksq)
        pairv = []
        for pr in range(2):
            u = (state[2 * cg.KS_TU[pr]], state[2 * cg.KS_TU[pr] + 1])
            v = (state[2 * cg.KS_TV[pr]], state[2 * cg.KS_TV[pr] + 1])
            pairv.append((u, v))
        mine = [pairv[0][1], pairv[0][0], pairv[1][0], pairv[1][1]]
        other = [pairv[0][0], pairv[0][1], pairv[1][1], pairv[1][0]]

        def forms(r, mn, ot):
            (mr, mi), (o_r, oi) = mn, ot
            if r & 1:
                return [list(mr), list(mi), list(o_r), list(oi)]
            xr, xi = [a + b for a, b in zip(mr, o_r)], [a + b for a, b in zip(mi, oi)]
            v_mine = r == 0
            # A lanes hand the block -Y: their product then arrives as -A (tools/coopasm.py generate_ksq)
            return [xr, xi, [b - a for a, b in zip(xr, mi if v_mine else oi)], [-a - b for a, b in zip(xi, mr if v_mine else o_r)]]

        emu = asmemu.Emu(lanes=4, subst=_subst())

        def park(lane, val):
            flat = list(val[0]) + list(val[1])
            for k in range(28):
                emu.lds[(k // 4) * 1024 + 16 * lane + 4 * (k % 4)] = flat[k] & asmemu.M32

        for r in range(4):
            park(r, mine[r])
        for it in range(3):
            for r in range(4):
                f = forms(r, mine[r], other[r])
                for j, base in enumerate((g.XR, g.XI, g.YR, g.YI)):
                    for i in range(NL):
                        emu.v.setdefault(base + i, [None] * 4)[r] = f[j][i] & asmemu.M32
            for reg in range(g.S, g.vend):      # everything the block clobbers starts undefined
                emu.v.pop(reg, None)
            emu.exec = 0xf
            emu.run(g.lines)
            assert emu.exec == 0xf
            st = {i: x for i, x in enumerate(state)}
            for i in range(12, 24):
                st[i] = None
            cg.emu_ksq(st, 0, 12, 1, 1)
            want = {}
            for r in (0, 2):
                pr = r >> 1
                tm, to = (cg.KS_TV[pr], cg.KS_TU[pr]) if r == 0 else (cg.KS_TU[pr], cg.KS_TV[pr])
                want[r] = (st[12 + 2 * tm], st[12 + 2 * tm + 1])
                want[r ^ 1] = (st[12 + 2 * to], st[12 + 2 * to + 1])
            for r in range(4):
                got_m = ([asmemu.s32(emu.v[g.XR + i][r]) for i in range(NL)], [asmemu.s32(emu.v[g.XI + i][r]) for i in range(NL)])
                got_o = ([asmemu.s32(emu.v[g.YR + i][r]) for i in range(NL)], [asmemu.s32(emu.v[g.YI + i][r]) for i in range(NL)])
                assert got_m == (list(want[r][0]), list(want[r][1])), (trial, it, r, "new coefficient")
                assert got_o == (list(want[r ^ 1][0]), list(want[r ^ 1][1])), (trial, it, r, "partner")
                flat = got_m[0] + got_m[1]
                assert [asmemu.s32(emu.lds[(k // 4) * 1024 + 16 * r + 4 * (k % 4)]) for k in range(28)] == flat, (trial, it, r, "parked copy")
            # next squaring starts from the compressed value just produced (the state the model wrote at the snapshot position)
            for i in range(12):
                if st[12 + i] is not None:
                    state[i] = st[12 + i]
            mine = [want[r] for r in range(4)]
            other = [want[r ^ 1] for r in range(4)]
