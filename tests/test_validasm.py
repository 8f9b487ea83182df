"""CPU gate for the asm steps of the subgroup checks (tools/validasm.py): the generated instruction text, executed by tools/asmemu.py,
must give the Jacobian doubling / mixed addition of the group law - checked twice: against the step's own formulas in big-integer
arithmetic (limb and value bounds of the next step included) and against the affine group law of the big-integer model
(tests/golden/bls12_381_model.py), chained over several steps the way the kernels chain them."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import asmemu  # noqa: E402
import bls12_381_model as m  # noqa: E402
import coopgen as cg  # noqa: E402
import validasm  # noqa: E402

NL = 14
P = cg.P


def _subst():
    s = {"p%d" % i: (P >> (28 * i)) & 0xfffffff for i in range(NL)}
    s["pinv"] = (-pow(P, -1, 1 << 28)) % (1 << 28)
    return s


def _put(emu, base, lane, limbs):
    for i, x in enumerate(limbs):
        emu.v.setdefault(base + i, [None] * emu.n)[lane] = x & asmemu.M32


def _get(emu, base, lane):
    return [asmemu.s32(emu.v[base + i][lane]) for i in range(NL)]


def _check_reduced(l, lo=-0.06, hi=1.06):
    assert all(abs(x) <= (1 << 27) + 16 for x in l[:NL - 1]), "limb bound of a reduced value"
    assert lo * P < cg.limbs_value(l) < hi * P, "value bound"


def _lds_put(emu, slot, lane, limbs):
    for i in range(16):
        emu.lds[(slot * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)] = (limbs[i] if i < NL else 0) & asmemu.M32


# ------------------------------------------------------------------------------------------------------------ G1
def _jac_to_affine1(X, Y, Z):
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


def test_g1_blocks_fit_three_waves_per_simd():
    d, a = validasm.g1_dbl(), validasm.g1_madd()
    assert d.vend <= 168 and a.vend <= 168 and d.vb == a.vb == 6
    # the point of the exercise: fewer instructions than the compiled step's ~5,700 (profiles/r04: 799 k per wavefront over 127 + 16 steps)
    assert sum(1 for l in d.lines if not l.endswith(":")) < 3100


def test_g1_doubling_and_addition_chain_equals_the_group_law():
    rng = random.Random(4)
    dbl, madd = validasm.g1_dbl(), validasm.g1_madd()
    lanes = 2
    # two points of E(Fp): multiples of the generator (one lane each); start at (x, y, 1)
    pts = [m.g1_mul(m.G1_GEN, rng.randrange(2, 1 << 64)) for _ in range(lanes)]
    jac = [(p[0], p[1], 1) for p in pts]
    emu = asmemu.Emu(lanes=lanes, subst=_subst())
    for lane, (x, y, z) in enumerate(jac):
        _put(emu, dbl.X, lane, cg.mont(x)); _put(emu, dbl.Y, lane, cg.mont(y)); _put(emu, dbl.Z, lane, cg.mont(z))
        _lds_put(emu, validasm.QX, lane, cg.mont(x))
        _lds_put(emu, validasm.QY, lane, cg.mont(y))
    acc = list(pts)            # the affine value the registers must represent
    for step, kind in enumerate("dddadaddadd"):
        before = [tuple(cg.from_mont(_get(emu, b, lane)) for b in (dbl.X, dbl.Y, dbl.Z)) for lane in range(lanes)]
        emu.run(dbl.lines if kind == "d" else madd.lines)
        for lane in range(lanes):
            X, Y, Z = (_get(emu, b, lane) for b in (dbl.X, dbl.Y, dbl.Z))
            # X' leaves the doubling with limbs of up to 3 units and |value| <= 2.2 p, the addition normalised with |value| <= 3.3 p
            assert all(abs(v) <= (3 if kind == "d" else 1) * ((1 << 27) + 16) for v in X[:NL - 1])
            assert abs(cg.limbs_value(X)) < (2.2 if kind == "d" else 3.3) * P
            _check_reduced(Y)
            _check_reduced(Z)
            x1, y1, z1 = before[lane]
            if kind == "d":     # the step's own formulas
                A, B = x1 * x1 % P, y1 * y1 % P
                S, M = 4 * x1 * B % P, 3 * A % P
                nx = (M * M - 2 * S) % P
                want = (nx, (M * (S - nx) - 8 * B * B) % P, 2 * y1 * z1 % P)
                acc[lane] = m.g1_add(acc[lane], acc[lane])
            else:
                qx, qy = pts[lane]
                zz = z1 * z1 % P
                H, r = (qx * zz - x1) % P, (qy * z1 * zz - y1) % P
                HH = H * H % P
                HHH, V = H * HH % P, x1 * HH % P
                nx = (r * r - HHH - 2 * V) % P
                want = (nx, (r * (V - nx) - y1 * HHH) % P, z1 * H % P)
                acc[lane] = m.g1_add(acc[lane], pts[lane])
            got = tuple(cg.from_mont(v) for v in (X, Y, Z))
            assert got == want, (step, kind, lane)
            assert _jac_to_affine1(*got) == acc[lane], (step, kind, lane)
    # the parked value of the addition is the lane's own: both lanes went through different data
    assert acc[0] != acc[1]


def test_g1_exceptional_cases_send_z_to_zero_and_keep_it_there():
    """P + P and P - P through the mixed addition, a point with Z = 0 through both steps: Z' = 0 mod p afterwards - the signal
    the kernel hands such points to the generic kernel on"""
    dbl, madd = validasm.g1_dbl(), validasm.g1_madd()
    p1 = m.g1_mul(m.G1_GEN, 77)
    for (y_sign, z0) in ((1, 1), (-1, 1), (1, 0)):
        emu = asmemu.Emu(lanes=1, subst=_subst())
        _put(emu, dbl.X, 0, cg.mont(p1[0])); _put(emu, dbl.Y, 0, cg.mont(y_sign * p1[1] % P)); _put(emu, dbl.Z, 0, cg.mont(z0))
        _lds_put(emu, validasm.QX, 0, cg.mont(p1[0]))
        _lds_put(emu, validasm.QY, 0, cg.mont(p1[1]))
        emu.run(madd.lines)
        assert cg.from_mont(_get(emu, dbl.Z, 0)) == 0
        for blk in (dbl, madd, dbl):
            emu.run(blk.lines)
            assert cg.from_mont(_get(emu, dbl.Z, 0)) == 0
            for b in (dbl.Y, dbl.Z):
                _check_reduced(_get(emu, b, 0))


# ------------------------------------------------------------------------------------------------------------ G2
def _jac_to_affine2(X, Y, Z):
    if Z == (0, 0):
        return None
    zi = m.f2_inv(Z)
    zi2 = m.f2_mul(zi, zi)
    return (m.f2_mul(X, zi2), m.f2_mul(Y, m.f2_mul(zi2, zi)))


def _f2k(a, k):
    return (a[0] * k % P, a[1] * k % P)


def test_g2_doubling_and_addition_chain_equals_the_group_law():
    rng = random.Random(5)
    dbl, madd = validasm.g2_dbl(), validasm.g2_madd()
    assert dbl.vend <= 256 and madd.vend <= 256
    pairs = 2
    pts = [m.g2_mul(m.G2_GEN, rng.randrange(2, 1 << 64)) for _ in range(pairs)]
    emu = asmemu.Emu(lanes=2 * pairs, subst=_subst())
    for k, (x, y) in enumerate(pts):
        for c in range(2):
            lane = 2 * k + c
            _put(emu, dbl.X, lane, cg.mont(x[c])); _put(emu, dbl.Y, lane, cg.mont(y[c])); _put(emu, dbl.W, lane, cg.mont(1 if c == 0 else 0))
            _lds_put(emu, 0, lane, cg.mont(x[c]))
            _lds_put(emu, 1, lane, cg.mont(y[c]))
    acc = list(pts)
    for step, kind in enumerate("ddadaddadd"):
        def val(base, k):
            return tuple(cg.from_mont(_get(emu, base, 2 * k + c)) for c in range(2))
        before = [(val(dbl.X, k), val(dbl.Y, k), val(dbl.W, k)) for k in range(pairs)]
        emu.run(dbl.lines if kind == "d" else madd.lines)
        assert emu.exec == (1 << (2 * pairs)) - 1
        for k in range(pairs):
            for c in range(2):
                # X' leaves the doubling normalised with |value| <= 2.2 p, the addition renormalised
                _check_reduced(_get(emu, dbl.X, 2 * k + c), *((-2.2, 2.2) if kind == "d" else (-0.52, 0.52)))
                _check_reduced(_get(emu, dbl.Y, 2 * k + c))
                _check_reduced(_get(emu, dbl.W, 2 * k + c))
            x1, y1, z1 = before[k]
            if kind == "d":
                A, B = m.f2_sqr(x1), m.f2_sqr(y1)
                S, M = _f2k(m.f2_mul(x1, B), 4), _f2k(A, 3)
                nx = m.f2_sub(m.f2_sqr(M), _f2k(S, 2))
                want = (nx, m.f2_sub(m.f2_mul(M, m.f2_sub(S, nx)), _f2k(m.f2_sqr(B), 8)), _f2k(m.f2_mul(y1, z1), 2))
                acc[k] = m.g2_add(acc[k], acc[k])
            else:
                qx, qy = pts[k]
                zz = m.f2_sqr(z1)
                H, r = m.f2_sub(m.f2_mul(qx, zz), x1), m.f2_sub(m.f2_mul(qy, m.f2_mul(z1, zz)), y1)
                HH = m.f2_sqr(H)
                HHH, V = m.f2_mul(H, HH), m.f2_mul(x1, HH)
                nx = m.f2_sub(m.f2_sub(m.f2_sqr(r), HHH), _f2k(V, 2))
                want = (nx, m.f2_sub(m.f2_mul(r, m.f2_sub(V, nx)), m.f2_mul(y1, HHH)), m.f2_mul(z1, H))
                acc[k] = m.g2_add(acc[k], pts[k])
            got = (val(dbl.X, k), val(dbl.Y, k), val(dbl.W, k))
            assert got == want, (step, kind, k)
            assert _jac_to_affine2(*got) == acc[k], (step, kind, k)


def _lds_get(emu, slot, lane):
    return [asmemu.s32(emu.lds[(slot * 4 + i // 4) * 1024 + 16 * lane + 4 * (i % 4)]) for i in range(NL)]


def test_g2_three_wave_steps_equal_the_group_law():
    """the five-block variants (three waves per SIMD): Y in registers, X and Z in LDS slots, the affine point from the scratch
    buffer - same formulas, same bounds, same chain as the two-wave blocks"""
    rng = random.Random(53)
    dbl, madd = validasm.g2_dbl3(), validasm.g2_madd3()
    assert dbl.vend <= 168 and madd.vend <= 168
    pairs = 2
    lanes = 2 * pairs
    pts = [m.g2_mul(m.G2_GEN, rng.randrange(2, 1 << 64)) for _ in range(pairs)]
    sub = _subst()
    sub.update({"qstride": "s120", "qlo": "s121", "qhi": "s122", "qoff": "v1"})
    emu = asmemu.Emu(lanes=lanes, subst=sub)
    qbase, qstride = 0x500000, 16 * lanes
    emu.s[120], emu.s[121], emu.s[122] = qstride, qbase, 0
    emu.v[1] = [16 * lane for lane in range(lanes)]
    Y = dbl.R[1]
    for k, (x, y) in enumerate(pts):
        for c in range(2):
            lane = 2 * k + c
            _put(emu, Y, lane, cg.mont(y[c]))
            _lds_put(emu, validasm.LX, lane, cg.mont(x[c]))
            _lds_put(emu, validasm.LZ, lane, cg.mont(1 if c == 0 else 0))
            for v, val in ((0, x[c]), (1, y[c])):
                l = cg.mont(val)
                for i in range(16):
                    emu.mem[qbase + ((v * 4 + i // 4) * lanes + lane) * 16 + 4 * (i % 4)] = (l[i] if i < NL else 0) & asmemu.M32
    acc = list(pts)

    def state(k):
        return (tuple(cg.from_mont(_lds_get(emu, validasm.LX, 2 * k + c)) for c in range(2)),
                tuple(cg.from_mont(_get(emu, Y, 2 * k + c)) for c in range(2)),
                tuple(cg.from_mont(_lds_get(emu, validasm.LZ, 2 * k + c)) for c in range(2)))

    for step, kind in enumerate("ddaddadda"):
        before = [state(k) for k in range(pairs)]
        emu.run(dbl.lines if kind == "d" else madd.lines)
        assert emu.exec == (1 << lanes) - 1
        for k in range(pairs):
            for c in range(2):
                _check_reduced(_lds_get(emu, validasm.LX, 2 * k + c), *((-2.2, 2.2) if kind == "d" else (-0.52, 0.52)))
                _check_reduced(_get(emu, Y, 2 * k + c))
                _check_reduced(_lds_get(emu, validasm.LZ, 2 * k + c))
            x1, y1, z1 = before[k]
            if kind == "d":
                A, B = m.f2_sqr(x1), m.f2_sqr(y1)
                S, M = _f2k(m.f2_mul(x1, B), 4), _f2k(A, 3)
                nx = m.f2_sub(m.f2_sqr(M), _f2k(S, 2))
                want = (nx, m.f2_sub(m.f2_mul(M, m.f2_sub(S, nx)), _f2k(m.f2_sqr(B), 8)), _f2k(m.f2_mul(y1, z1), 2))
                acc[k] = m.g2_add(acc[k], acc[k])
            else:
                qx, qy = pts[k]
                zz = m.f2_sqr(z1)
                H, r = m.f2_sub(m.f2_mul(qx, zz), x1), m.f2_sub(m.f2_mul(qy, m.f2_mul(z1, zz)), y1)
                HH = m.f2_sqr(H)
                HHH, V = m.f2_mul(H, HH), m.f2_mul(x1, HH)
                nx = m.f2_sub(m.f2_sub(m.f2_sqr(r), HHH), _f2k(V, 2))
                want = (nx, m.f2_sub(m.f2_mul(r, m.f2_sub(V, nx)), m.f2_mul(y1, HHH)), m.f2_mul(z1, H))
                acc[k] = m.g2_add(acc[k], pts[k])
            got = state(k)
            assert got == want, (step, kind, k)
            assert _jac_to_affine2(*got) == acc[k], (step, kind, k)
    # P + P through the addition: Z' = 0 and it stays 0
    for k, (x, y) in enumerate(pts):
        for c in range(2):
            lane = 2 * k + c
            _put(emu, Y, lane, cg.mont(y[c]))
            _lds_put(emu, validasm.LX, lane, cg.mont(x[c]))
            _lds_put(emu, validasm.LZ, lane, cg.mont(1 if c == 0 else 0))
    for blk in (madd, dbl, madd):
        emu.run(blk.lines)
        for k in range(pairs):
            assert state(k)[2] == (0, 0)


def test_g2_exceptional_cases_send_z_to_zero():
    dbl, madd = validasm.g2_dbl(), validasm.g2_madd()
    p2 = m.g2_mul(m.G2_GEN, 5)
    for (y_sign, z0) in ((1, 1), (-1, 1), (1, 0)):
        emu = asmemu.Emu(lanes=2, subst=_subst())
        for c in range(2):
            _put(emu, dbl.X, c, cg.mont(p2[0][c])); _put(emu, dbl.Y, c, cg.mont(y_sign * p2[1][c] % P))
            _put(emu, dbl.W, c, cg.mont(z0 if c == 0 else 0))
            _lds_put(emu, 0, c, cg.mont(p2[0][c]))
            _lds_put(emu, 1, c, cg.mont(p2[1][c]))
        emu.run(madd.lines)
        for blk in (None, dbl, madd):
            if blk:
                emu.run(blk.lines)
            assert [cg.from_mont(_get(emu, dbl.W, c)) for c in range(2)] == [0, 0]


def test_generated_file_is_current():
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tf:
        validasm.write_inc(tf.name)
        assert open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_valid_steps.inc")).read() == open(tf.name).read(), \
            "zkp_valid_steps.inc is not what tools/validasm.py generates: run tools/validasm.py"
