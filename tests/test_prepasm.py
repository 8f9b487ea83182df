"""CPU gate for the asm doubling step of k_prep_lines<true> (tools/prepasm.py): the generated instruction text, executed by
tools/asmemu.py on the two lanes of one pair, must give the values of the Costello-Lange-Naehrig doubling formulas in
big-integer arithmetic - the new point, the three line coefficients it stores - inside the bounds the next step needs."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import asmemu  # noqa: E402
import coopgen as cg  # noqa: E402
import prepasm  # noqa: E402

NL = 14
P = cg.P


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2k(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def f2xi(a):
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def doubling_model(X, Y, W, xP, yP):
    B, C = f2mul(Y, Y), f2mul(W, W)
    YW = f2add(Y, W)
    H2 = f2sub(f2sub(f2mul(YW, YW), B), C)
    E = f2k(f2xi(C), 3)
    F = f2k(E, 3)
    X2 = f2mul(X, X)
    XY = f2add(X, Y)
    XY2 = f2sub(f2sub(f2mul(XY, XY), X2), B)
    nx = f2mul(XY2, f2sub(B, F))
    BF = f2add(B, F)
    ny = f2sub(f2mul(BF, BF), f2k(f2mul(E, E), 12))
    nw = f2mul(B, f2k(H2, 4))
    l2 = f2k(f2sub(B, E), 2)
    l1 = f2k(f2k(X2, -6), xP)
    l0 = f2k(H2, yP)
    return nx, ny, nw, (l2, l1, l0)


def _subst():
    s = {"p%d" % i: (P >> (28 * i)) & 0xfffffff for i in range(NL)}
    s["pinv"] = (-pow(P, -1, 1 << 28)) % (1 << 28)
    s.update({"estride": "s110", "voff": "v1", "smask": "s[112:113]", "base": "s[114:115]"})
    return s


def test_doubling_step_block_equals_the_formulas():
    rng = random.Random(99)
    g = prepasm.generate()
    for trial in range(4):
        # reduced inputs as the previous step leaves them: values in (-0.05 p, 1.05 p); trial 0: the first step (W = 2, X, Y < p)
        def rv():
            return rng.randrange(P) if trial % 2 == 0 else rng.randrange(P) - P // 20

        X, Y = (rv(), rv()), (rv(), rv())
        W = (2, 0) if trial == 0 else (rv(), rv())
        xP, yP = rng.randrange(P), rng.randrange(P)

        def limbs(v):           # Montgomery form of the VALUE v (which may be a little outside [0, p)), balanced limbs
            sh = (v // P) * P if v < 0 or v >= P else 0
            l = cg.mont(v - sh)
            if sh:
                add = cg.to_limbs_balanced(sh)
                l = cg.weak_norm([a + b for a, b in zip(l, add)])
            return l

        emu = asmemu.Emu(lanes=2, subst=_subst())
        for c in range(2):
            for base, val in ((g.X, X), (g.Y, Y), (g.W, W)):
                for i, x in enumerate(cg.mont(val[c] % P)):
                    emu.v.setdefault(base + i, [None, None])[c] = x & asmemu.M32
        nc = 7
        estride = 2 * nc * 64
        emu.s[110] = estride
        emu.s[112], emu.s[113] = 0xffffffff, 0xffffffff
        emu.s[114], emu.s[115] = 0x100000, 0
        emu.v[1] = [0, nc * 64]                      # lane c writes record e + c
        for v, val in ((0, xP), (1, yP)):
            l = cg.mont(val)
            for lane in range(2):
                for i in range(NL):
                    emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = l[i] & asmemu.M32
                for i in (14, 15):
                    emu.lds[(v * 4 + 3) * 1024 + 16 * lane + 4 * (i % 4)] = 0
        emu.run(g.lines)
        assert emu.exec == 3
        nx, ny, nw, lines = doubling_model(X, Y, W, xP, yP)

        def out(base):
            ls = [[asmemu.s32(emu.v[base + i][c]) for i in range(NL)] for c in range(2)]
            for l in ls:
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]), "limb bound of a reduced value"
                assert -0.06 * P < cg.limbs_value(l) < 1.06 * P, "value bound of a reduced value"
            return tuple(cg.from_mont(l) for l in ls)

        assert out(g.X) == nx and out(g.Y) == ny and out(g.W) == nw, trial
        for k, want in enumerate(lines):             # records 0, 2, 4 (+ c)
            for c in range(2):
                addr = 0x100000 + (2 * k + c) * nc * 64
                l = [asmemu.s32(emu.mem[addr + 4 * i]) for i in range(NL)]
                assert emu.mem[addr + 56] == 0 and emu.mem[addr + 60] == 0
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1])
                assert abs(cg.limbs_value(l)) < 1.06 * P
                assert cg.from_mont(l) == want[c], (trial, k, c)


def test_generated_file_is_current():
    """byte for byte: the emulator tests of this file run what the generator emits NOW, the compiler what is checked in"""
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tf:
        prepasm.write_inc(tf.name)
        assert open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_prep_dbl.inc")).read() == open(tf.name).read(), \
            "zkp_prep_dbl.inc is not what tools/prepasm.py generates: run tools/prepasm.py"


def addition_model(X, Y, W, qx, qy, xP, yP):
    th = f2sub(f2k(Y, 2), f2mul(qy, W))
    la = f2sub(f2k(X, 2), f2mul(qx, W))
    l2 = f2sub(f2mul(th, qx), f2mul(la, qy))
    l1 = f2k(f2k(th, -1), xP)
    l0 = f2k(la, yP)
    C, D = f2mul(th, th), f2mul(la, la)
    E, F, G = f2mul(la, D), f2mul(W, C), f2mul(f2k(X, 2), D)
    H = f2sub(f2add(E, F), f2k(G, 2))
    nx = f2mul(la, H)
    ny = f2sub(f2mul(th, f2sub(G, H)), f2mul(f2k(Y, 2), E))
    nw = f2k(f2mul(W, E), 2)
    return nx, ny, nw, (l2, l1, l0)


def test_addition_step_block_equals_the_formulas():
    rng = random.Random(123)
    g = prepasm.generate_add()
    for trial in range(3):
        X, Y, W = [(rng.randrange(P), rng.randrange(P)) for _ in range(3)]
        qx, qy = (rng.randrange(P), rng.randrange(P)), (rng.randrange(P), rng.randrange(P))
        xP, yP = rng.randrange(P), rng.randrange(P)
        emu = asmemu.Emu(lanes=2, subst=_subst())
        for c in range(2):
            for base, val in ((g.X, X), (g.Y, Y), (g.W, W)):
                for i, x in enumerate(cg.mont(val[c])):
                    emu.v.setdefault(base + i, [None, None])[c] = x & asmemu.M32
        nc = 5
        emu.s[110] = 2 * nc * 64
        emu.s[112], emu.s[113] = 0xffffffff, 0xffffffff
        emu.s[114], emu.s[115] = 0x200000, 0
        emu.v[1] = [0, nc * 64]
        for v, val in ((0, (xP, xP)), (1, (yP, yP)), (2, qx), (3, qy)):
            for lane in range(2):
                l = cg.mont(val[lane])
                for i in range(NL):
                    emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = l[i] & asmemu.M32
                for i in (14, 15):
                    emu.lds[(v * 4 + 3) * 1024 + 16 * lane + 4 * (i % 4)] = 0
        emu.run(g.lines)
        assert emu.exec == 3
        nx, ny, nw, lines = addition_model(X, Y, W, qx, qy, xP, yP)

        def out(base):
            ls = [[asmemu.s32(emu.v[base + i][c]) for i in range(NL)] for c in range(2)]
            for l in ls:
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1])
                assert -0.06 * P < cg.limbs_value(l) < 1.06 * P
            return tuple(cg.from_mont(l) for l in ls)

        assert out(g.X) == nx and out(g.Y) == ny and out(g.W) == nw, trial
        for k, want in enumerate(lines):
            for c in range(2):
                addr = 0x200000 + (2 * k + c) * nc * 64
                l = [asmemu.s32(emu.mem[addr + 4 * i]) for i in range(NL)]
                assert emu.mem[addr + 56] == 0 and emu.mem[addr + 60] == 0
                assert abs(cg.limbs_value(l)) < 1.06 * P
                assert cg.from_mont(l) == want[c], (trial, k, c)


def test_jacobian_doubling_step_equals_alg26_of_the_model():
    """the doubling step of k_prep_lines<false> (prepasm.generate_jac): new point and the three stored line records against the
    LITERAL Alg. 26 of the big-integer model (bls12_381_model.doubling_step, which the oracle-pinned Miller value is built from),
    chained over three steps so that a step runs on what the previous one left (normalised X', Z')"""
    import bls12_381_model as m
    rng = random.Random(2026)
    g = prepasm.generate_jac()
    assert g.vend <= 256
    q = m.g2_mul(m.G2_GEN, rng.randrange(2, 1 << 60))
    r = (q[0], q[1], (1, 0))
    xP, yP = rng.randrange(P), rng.randrange(P)
    emu = asmemu.Emu(lanes=2, subst=_subst())
    for c in range(2):
        for base, val in ((g.X, r[0]), (g.Y, r[1]), (g.W, r[2])):
            for i, x in enumerate(cg.mont(val[c])):
                emu.v.setdefault(base + i, [None, None])[c] = x & asmemu.M32
    nc = 7
    emu.s[110] = 2 * nc * 64
    emu.s[112], emu.s[113] = 0xffffffff, 0xffffffff
    emu.s[115] = 0
    emu.v[1] = [0, nc * 64]
    for v, val in ((0, xP), (1, yP)):
        l = cg.mont(val)
        for lane in range(2):
            for i in range(16):
                emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = (l[i] if i < NL else 0) & asmemu.M32
    for step in range(3):
        emu.s[114] = 0x100000 * (step + 1)
        emu.run(g.lines)
        assert emu.exec == 3
        r, (c0, c1, c2) = m.doubling_step(r)
        for base, want, k in ((g.X, r[0], 2.2), (g.Y, r[1], 1.06), (g.W, r[2], 2.2)):
            for c in range(2):
                l = [asmemu.s32(emu.v[base + i][c]) for i in range(NL)]
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]) and abs(cg.limbs_value(l)) < k * P
                assert cg.from_mont(l) == want[c], (step, base, c)
        recs = (c2, ((c1[0] * xP) % P, (c1[1] * xP) % P), ((c0[0] * yP) % P, (c0[1] * yP) % P))     # records 0, 2, 4 (+ c)
        for k, want in enumerate(recs):
            for c in range(2):
                addr = 0x100000 * (step + 1) + (2 * k + c) * nc * 64
                l = [asmemu.s32(emu.mem[addr + 4 * i]) for i in range(NL)]
                assert emu.mem[addr + 56] == 0 and emu.mem[addr + 60] == 0
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]) and abs(cg.limbs_value(l)) < 1.06 * P
                assert cg.from_mont(l) == want[c], (step, k, c)


def test_jacobian_addition_step_equals_alg27_of_the_model():
    """the mixed-addition step of k_prep_lines<false> (prepasm.generate_jac_add) against the literal Alg. 27 of the model, behind
    two asm doublings (so that it runs on what a doubling leaves)"""
    import bls12_381_model as m
    rng = random.Random(427)
    gd, ga = prepasm.generate_jac(), prepasm.generate_jac_add()
    assert ga.vend <= 256
    q = m.g2_mul(m.G2_GEN, rng.randrange(2, 1 << 60))
    r = (q[0], q[1], (1, 0))
    xP, yP = rng.randrange(P), rng.randrange(P)
    emu = asmemu.Emu(lanes=2, subst=_subst())
    for c in range(2):
        for base, val in ((gd.X, r[0]), (gd.Y, r[1]), (gd.W, r[2])):
            for i, x in enumerate(cg.mont(val[c])):
                emu.v.setdefault(base + i, [None, None])[c] = x & asmemu.M32
    nc = 3
    emu.s[110] = 2 * nc * 64
    emu.s[112], emu.s[113] = 0xffffffff, 0xffffffff
    emu.s[115] = 0
    emu.v[1] = [0, nc * 64]
    for v, val in ((0, (xP, xP)), (1, (yP, yP)), (2, q[0]), (3, q[1])):
        for lane in range(2):
            l = cg.mont(val[lane])
            for i in range(16):
                emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = (l[i] if i < NL else 0) & asmemu.M32
    for step, kind in enumerate("ddadda"):
        emu.s[114] = 0x100000 * (step + 1)
        emu.run(gd.lines if kind == "d" else ga.lines)
        assert emu.exec == 3
        r, (c0, c1, c2) = m.doubling_step(r) if kind == "d" else m.addition_step(r, q)
        for base, want, k in ((gd.X, r[0], 2.2), (gd.Y, r[1], 1.06), (gd.W, r[2], 2.2)):
            for c in range(2):
                l = [asmemu.s32(emu.v[base + i][c]) for i in range(NL)]
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]) and abs(cg.limbs_value(l)) < k * P
                assert cg.from_mont(l) == want[c], (step, kind, base, c)
        recs = (c2, ((c1[0] * xP) % P, (c1[1] * xP) % P), ((c0[0] * yP) % P, (c0[1] * yP) % P))
        for k, want in enumerate(recs):
            for c in range(2):
                addr = 0x100000 * (step + 1) + (2 * k + c) * nc * 64
                l = [asmemu.s32(emu.mem[addr + 4 * i]) for i in range(NL)]
                assert emu.mem[addr + 56] == 0 and emu.mem[addr + 60] == 0
                assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]) and abs(cg.limbs_value(l)) < 1.06 * P
                assert cg.from_mont(l) == want[c], (step, kind, k, c)


def test_line_stores_obey_the_lane_mask():
    """a lane that the store mask excludes (a pair with an infinity, a lane behind the last pair) computes but stores nothing"""
    rng = random.Random(7)
    g = prepasm.generate()
    X, Y, W = [(rng.randrange(P), rng.randrange(P)) for _ in range(3)]
    emu = asmemu.Emu(lanes=2, subst=_subst())
    for c in range(2):
        for base, val in ((g.X, X), (g.Y, Y), (g.W, W)):
            for i, x in enumerate(cg.mont(val[c])):
                emu.v.setdefault(base + i, [None, None])[c] = x & asmemu.M32
    nc = 3
    emu.s[110] = 2 * nc * 64
    emu.s[112], emu.s[113] = 1, 0                       # lane 0 only
    emu.s[114], emu.s[115] = 0x300000, 0
    emu.v[1] = [0, nc * 64]
    for v in (0, 1):
        l = cg.mont(rng.randrange(P))
        for lane in range(2):
            for i in range(16):
                emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = (l[i] if i < NL else 0) & asmemu.M32
    emu.run(g.lines)
    stored = sorted({(a - 0x300000) // (nc * 64) for a in emu.mem})
    assert stored == [0, 2, 4], stored                 # records e + 0 of lane 0; nothing of lane 1 (records 1, 3, 5)
    assert emu.exec == 3


def _exec_discipline(path):
    """every write to EXEC inside a generated asm block takes its value from (a) the SGPR pair that saved EXEC at the block's entry,
    (b) a lane-role mask that was ANDed with that pair, (c) an operand the CALLER computed under its own EXEC (%[smask], a ballot;
    %[exec...]), or (d) vcc / another pair ANDed with (a).  Returns the number of EXEC writes checked."""
    import re
    inside, checked = set(), 0
    for raw in open(path):
        m = re.match(r'\s*"(.*?)\\n', raw)
        if raw.startswith("#define"):
            inside = set()                                   # a new block: nothing is known about EXEC yet
        if not m:
            continue
        ins = m.group(1).strip()
        mm = re.fullmatch(r"s_mov_b64 (s\[\d+:\d+\]), exec", ins)
        if mm:
            inside.add(mm.group(1))
            continue
        mm = re.fullmatch(r"s_and_b64 (s\[\d+:\d+\]|exec|vcc), (\S+), (\S+)", ins)
        if mm:
            dst, a, b = mm.groups()
            a = a.rstrip(",")
            ok = a in inside or b in inside or a == "exec" or b == "exec"
            if dst == "exec":
                assert ok, (path, ins)
                checked += 1
            elif ok and dst != "vcc":
                inside.add(dst)
            elif dst in inside:
                inside.discard(dst)
            continue
        mm = re.fullmatch(r"s_mov_b64 exec, (\S+)", ins)
        if mm:
            src = mm.group(1)
            assert src in inside or src.startswith("%["), "%s: `%s` loads EXEC with a mask that was never ANDed with the entry EXEC" % (path, ins)
            checked += 1
            continue
        assert not re.match(r"s_\w+ exec\b", ins), (path, ins)     # no other way into EXEC
        mm = re.match(r"s_\w+ (s\[\d+:\d+\]),", ins)
        if mm and mm.group(1) in inside:
            inside.discard(mm.group(1))                      # a pair that is overwritten by something else is no longer a mask
    return checked


def test_no_generated_block_loads_exec_with_an_absolute_mask():
    """the lane-role masks of every asm block stay INSIDE the EXEC the block was entered with - checked on the checked-in includes,
    i.e. on the text the compiler sees (the callers keep all 64 lanes alive today; nothing may depend on that)"""
    csrc = os.path.join(ROOT, "zkvm_pairings_amd", "csrc")
    n = sum(_exec_discipline(os.path.join(csrc, f)) for f in ("zkp_prep_dbl.inc", "zkp_valid_steps.inc", "zkp_coop_mulacc.inc"))
    assert n > 200


def test_doubling_step_on_half_a_wavefront_leaves_the_other_half_alone():
    """the same block on four lanes with the second pair switched off at entry: its registers are never written, its records never
    stored, the first pair's results are what the two-lane run gives, and EXEC comes back as it was"""
    rng = random.Random(5)
    g = prepasm.generate()
    X, Y, W = [(rng.randrange(P), rng.randrange(P)) for _ in range(3)]
    pts = [rng.randrange(P), rng.randrange(P)]

    def run(lanes, entry):
        emu = asmemu.Emu(lanes=lanes, subst=_subst())
        emu.exec = entry
        for c in range(2):
            for base, val in ((g.X, X), (g.Y, Y), (g.W, W)):
                for i, x in enumerate(cg.mont(val[c])):
                    emu.v.setdefault(base + i, [None] * lanes)[c] = x & asmemu.M32
        nc = 5
        emu.s[110] = 2 * nc * 64
        emu.s[112], emu.s[113] = entry, 0                   # the caller's ballot: a subset of its EXEC
        emu.s[114], emu.s[115] = 0x200000, 0
        emu.v[1] = [0, nc * 64] + [None] * (lanes - 2)
        for v in (0, 1):
            l = cg.mont(pts[v])
            for lane in range(2):
                for i in range(16):
                    emu.lds[(v * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = (l[i] if i < NL else 0) & asmemu.M32
        emu.run(g.lines)
        return emu

    two, four = run(2, 3), run(4, 3)
    assert four.exec == 3
    for r, vals in four.v.items():
        assert vals[2] is None and vals[3] is None, "v%d of a lane the caller had switched off was written" % r
        if r in two.v:
            assert vals[:2] == two.v[r][:2], r
    assert four.mem == two.mem and len(two.mem) > 0
