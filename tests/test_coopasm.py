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
    # byte for byte: the emulator tests of this file run what the generator emits NOW, the compiler what is checked in
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tf:
        coopasm.write_inc(tf.name)
        assert open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_coop_mulacc.inc")).read() == open(tf.name).read(), \
            "zkp_coop_mulacc.inc is not what tools/coopasm.py generates: run tools/coopasm.py"


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


# ------------------------------------------------------------------------------------------------------------------------
# the interpreter's MULACC block: every MULACC step of a few step programs, on the lanes of one check, against the columns
# tools/coopgen.py's emulator accumulates and reduces for that step
def _resolved_rows(st, S, SC):
    """zkp_coop.hip coop_resolve_table for one step and the lanes of group 0: per term a row of (A1, B1, A2 | B2 << 16) byte
    addresses, then the row of flag words"""
    SG = S + ((4 - S % 8) + 8) % 8
    addr = lambda slot: 16 * ((0 if slot & 64 else SC) + (slot & 63))
    T = st["T"]
    rows = [[(0, 0, 0)] * 12 for _ in range(T + 1)]
    for lig in range(12):
        ln = st["lanes"][lig] if lig < len(st["lanes"]) else None
        f1 = f2 = 0
        for t in range(T):
            if ln is None or t >= len(ln["terms"]):
                a1 = a2 = b1 = b2 = cg.ZERO
                asub = bsub = neg = da = 0
            else:
                a1, a2, asub, b1, b2, bsub, neg, da = ln["terms"][t]
            rows[t][lig] = (addr(a1), addr(b1), addr(a2) | (addr(b2) << 16))
            f1 |= (int(neg) << t) | (int(da) << (12 + t))
            f2 |= (int(asub) << t) | (int(bsub) << (12 + t))
        # the flag row also carries the step's store words: LDS byte address of the result slot and of its companion slot (-1: none)
        sd = st.get("sd")
        rows[T][lig] = (f1, f2, addr(ln["dst"]) if ln is not None else 0xffffffff, addr(sd[lig]) if (sd and ln is not None) else 0xffffffff)
    return rows, SG


def _run_mulacc_block(g, out, st, slots, S, SC, active_mask=0xfff):
    """execute the generated block for one MULACC step on 12 lanes; slots: slot number -> 14 limbs (constants included)"""
    rows, SG = _resolved_rows(st, S, SC)
    PS = SC + 5 * SG
    T = st["T"]
    hdr1 = hdr3 = 0
    for t in range(T):
        ts = [ln["terms"][t] for ln in st["lanes"] if t < len(ln["terms"])]
        if all(x[1] == cg.ZERO and not x[2] for x in ts):
            hdr3 |= 1 << t
        if all(x[4] == cg.ZERO and not x[5] for x in ts):
            hdr3 |= 1 << (12 + t)
        if not any(x[6] for x in ts):
            hdr1 |= 1 << (4 + t)
        if any(x[7] for x in ts):
            hdr1 |= 1 << (16 + t)
    subst = _subst()
    hdr1 |= int(bool(st["epi"])) | (int(bool(st.get("sd"))) << 1)
    subst.update({"rt": "s[100:101]", "T": "s102", "h1": "s103", "h3": "s104", "lane16": "v1", "ps1": PS * 16, "ps2": PS * 32, "ps3": PS * 48, "row": 1024,
                  "act": "s[106:107]", "nost": "s108"})
    emu = asmemu.Emu(lanes=12, subst=subst)
    base = 0x40000
    emu.s[100], emu.s[101], emu.s[102], emu.s[103], emu.s[104] = base, 0, T, hdr1, hdr3
    emu.s[106], emu.s[107], emu.s[108] = active_mask, 0, int(bool(st["epi"]))
    emu.v[1] = [16 * lane for lane in range(12)]
    for r, row in enumerate(rows):
        for lane, ent in enumerate(row):
            for k in range(4):
                emu.mem[base + 1024 * r + 16 * lane + 4 * k] = (ent[k] if k < len(ent) else 0) & asmemu.M32
    for extra in range(2):       # the block reads one row past the flag row's predecessor ahead of itself (the table is padded)
        for lane in range(12):
            for k in range(4):
                emu.mem.setdefault(base + 1024 * (len(rows) + extra) + 16 * lane + 4 * k, 0)
    for slot, limbs in slots.items():
        idx = (0 if slot & 64 else SC) + (slot & 63)
        for i in range(NL):
            emu.lds[16 * idx + (i // 4) * PS * 16 + 4 * (i % 4)] = limbs[i] & asmemu.M32
        for i in (14, 15):
            emu.lds[16 * idx + 3 * PS * 16 + 4 * (i % 4)] = 0
    emu.run(g.lines)
    return [[asmemu.s32(emu.v[out + i][lane]) for i in range(NL)] for lane in range(12)], emu


def _check_program(builder, S, SC, em, max_steps=None):
    g, out = coopasm.generate()
    done = 0
    steps = builder.steps
    pc, loop_start, loop_left = 0, None, 0
    while pc < len(steps):
        st = steps[pc]
        if st["op"] == cg.OP_MULACC and (max_steps is None or done < max_steps):
            got, emu = _run_mulacc_block(g, out, st, em.slot, S, SC)
            for lig, ln in enumerate(st["lanes"]):
                col = [0] * (2 * NL - 1)
                for (a1, a2, asub, b1, b2, bsub, neg, da) in ln["terms"]:
                    a, b = em.form(a1, a2, asub), em.form(b1, b2, bsub)
                    a = [-x for x in a] if neg else a
                    a = [2 * x for x in a] if da else a
                    for i in range(NL):
                        for j in range(NL):
                            col[i + j] += a[i] * b[j]
                assert got[lig] == cg.acc_reduce(col), (pc, lig)
            done += 1
            em.run([st])
            # a step without epilogue stores from inside the block: the result slot and (companion steps) the companion slot of every
            # lane hold what the emulator's step leaves there; an epilogue step stores nothing (the compiled code behind the block does)
            PS = SC + 5 * (S + ((4 - S % 8) + 8) % 8)

            def lds_slot(slot):
                idx = SC + slot
                return [asmemu.s32(emu.lds.get(16 * idx + (i // 4) * PS * 16 + 4 * (i % 4), 0)) for i in range(NL)]

            for lig, ln in enumerate(st["lanes"]):
                if st["epi"]:
                    assert lds_slot(ln["dst"]) != got[lig] or not any(got[lig]), (pc, lig, "an epilogue step must not store from the block")
                else:
                    assert lds_slot(ln["dst"]) == list(em.slot[ln["dst"]]) == got[lig], (pc, lig, "result store")
                    if st.get("sd"):
                        assert lds_slot(st["sd"][lig]) == list(em.slot[st["sd"][lig]]), (pc, lig, "companion store")
            pc += 1
            continue
        em.run([st]) if st["op"] not in (cg.OP_LOOP, cg.OP_ENDLOOP) else None
        if st["op"] == cg.OP_LOOP:
            loop_start, loop_left = pc + 1, min(st["n"], 2)      # two iterations of a loop are enough here
        elif st["op"] == cg.OP_ENDLOOP:
            loop_left -= 1
            if loop_left > 0:
                pc = loop_start
                continue
        pc += 1
    return done


def test_mulacc_block_equals_the_emulator_on_the_tower_programs():
    """fp12_mul (T = 12, two-slot B operands), fp12_sqr (doubled A operands, negated terms), cyc_sqr (three terms; the epilogue is
    compiled code behind the block), fp12_014: the block's 14 result limbs per lane are the emulator's reduced columns"""
    rng = random.Random(5)
    for op in ("fp12_mul", "fp12_sqr", "fp12_014", "cyc_sqr", "fp6_mul"):
        b = cg.prog_tower(op)
        wire = [rng.randrange(cg.P) for _ in range(12)]
        wire2 = [rng.randrange(cg.P) for _ in range(12)]
        em = cg.Emu(wire_in=wire, wire_in2=wire2)
        assert _check_program(b, cg.LDS_SLOTS, cg.N_CONST, em) >= 1, op


def test_mulacc_block_equals_the_emulator_on_miller_steps():
    """the first steps of the Miller program of one pair (30-slot configuration, companion slots, loops): line products (T = 6)
    and accumulator squarings (T = 7)"""
    import bls12_381_model as m
    lines = cg.model_lines([(m.G1_GEN, m.G2_GEN)])
    b = cg.prog_miller(1, False)
    em = cg.Emu(lines=lines)
    assert _check_program(b, cg.LDS_WIDE_SLOTS, cg.LDS_WIDE_CONSTS, em, max_steps=6) == 6
