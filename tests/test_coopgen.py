"""CPU gate for the lane-cooperative kernels' step programs: tools/coopgen.py's bit-accurate emulator
(same limb arithmetic as the HIP interpreter, with overflow assertions) must reproduce the big-int
model on the generators and on a 3-pair check."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bls12_381_model as m  # noqa: E402
import coopgen as cg  # noqa: E402


def run_pairing(pairs):
    k = len(pairs)
    lines = cg.model_lines(pairs)
    em = cg.Emu(lines=lines).run(cg.prog_miller(k, False).steps)
    assert em.cursor == cg.n_line_steps()
    state = em.state
    f_true = [cg.from_mont(state[cg.ST_F + i]) for i in range(12)]
    ea = cg.Emu(state=state).run(cg.prog_fexp_a(False).steps)
    n = cg.from_mont(ea.state[cg.ST_N])
    ea.state[cg.ST_NINV] = cg.mont(m.fp_inv(n))       # the batched inversion kernel's job
    ec = cg.run_plan(cg.fexp_c_plan(), ea.state)
    return f_true, ec, (em, ea)


def test_emulated_pairing_of_generators_matches_model():
    f_true, ec, (em, ea) = run_pairing([(m.G1_GEN, m.G2_GEN)])
    ml = m.multi_miller_loop([(m.G1_GEN, m.G2_GEN)])
    assert f_true == m.f12_flat_ints(ml)
    assert ec.wire_out == m.f12_flat_ints(m.final_exponentiation(ml))
    assert ec.is_identity is False
    # headroom of the lazy accumulation: columns must stay well inside 63 bits
    assert max(em.max_col, ec.plan_stats["max_col"]).bit_length() <= 61
    # step census used by DESIGN.md: the x-power chains' 315 squarings run in k_ksq, not on the interpreter
    assert em.counts["mulacc_steps"] > 100 and 40 < ec.plan_stats["mulacc_steps"] < 80


@pytest.mark.parametrize("split", [2, 6])
def test_x_power_chain_split_points_give_the_same_gt(split, monkeypatch):
    """the point where an x-power chain leaves the compressed squaring run for Granger-Scott steps (KSQ_SPLIT; 3 is what ships,
    6 = rounds 2-3: all 63 squarings compressed, six snapshots) changes the plan, never the value"""
    monkeypatch.setattr(cg, "KSQ_SPLIT", split)
    plan = cg.prog_fexp_c(True)
    runs = [st for st in plan if st[0] == cg.PLAN_KSQ]
    nsq = cg.X_BITS[split - 1]
    assert [st[3] for st in runs] == [nsq - 1] + [nsq] * 4 and all(bin(st[4]).count("1") == split for st in runs)
    pairs = [(m.G1_GEN, m.G2_GEN)]
    em = cg.Emu(lines=cg.model_lines(pairs)).run(cg.prog_miller(1, False).steps)
    ea = cg.Emu(state=em.state).run(cg.prog_fexp_a(False).steps)
    ea.state[cg.ST_NINV] = cg.mont(m.fp_inv(cg.from_mont(ea.state[cg.ST_N])))
    ec = cg.run_plan(plan, ea.state)
    assert ec.wire_out == m.f12_flat_ints(m.final_exponentiation(m.multi_miller_loop(pairs))) and ec.is_identity is False


def test_emulated_wire_roundtrip_and_three_pair_check(model_vectors):
    H = lambda s: int(s, 16)
    c = model_vectors["pairing"]["multi3"]
    pairs = [((H(a[0]), H(a[1])), ((H(b[0]), H(b[1])), (H(b[2]), H(b[3])))) for a, b in zip(c["g1"], c["g2"])]
    lines = cg.model_lines(pairs)
    em = cg.Emu(lines=lines).run(cg.prog_miller(3, True).steps)
    assert em.wire_out == [H(x) for x in c["miller"]]
    # final exponentiation from the wire format
    ea = cg.Emu(wire_in=em.wire_out).run(cg.prog_fexp_a(True).steps)
    n = cg.from_mont(ea.state[cg.ST_N])
    ea.state[cg.ST_NINV] = cg.mont(m.fp_inv(n))
    ec = cg.run_plan(cg.fexp_c_plan(), ea.state)
    assert ec.wire_out == [1] + [0] * 11 and ec.is_identity is True


def test_infinity_pair_gives_neutral_lines():
    lines = cg.model_lines([(None, m.G2_GEN), (m.G1_GEN, m.G2_GEN)])
    em = cg.Emu(lines=lines).run(cg.prog_miller(2, True).steps)
    assert em.wire_out == m.f12_flat_ints(m.multi_miller_loop([(m.G1_GEN, m.G2_GEN)]))


def test_grouped_miller_join_matches_single_loop(model_vectors):
    """k > 4 pairs per check: groups of <= 4 pairs run separate Miller loops, f12mul joins them.
    Emulated on the 3-pair vector split 2 + 1."""
    H = lambda s: int(s, 16)
    c = model_vectors["pairing"]["multi3"]
    pairs = [((H(a[0]), H(a[1])), ((H(b[0]), H(b[1])), (H(b[2]), H(b[3])))) for a, b in zip(c["g1"], c["g2"])]
    e0 = cg.Emu(lines=cg.model_lines(pairs[:2])).run(cg.prog_miller(2, False).steps)
    e1 = cg.Emu(lines=cg.model_lines(pairs[2:])).run(cg.prog_miller(1, False).steps)
    state = dict(e0.state)
    for i in range(12):
        state[cg.ST_G + i] = e1.state[cg.ST_F + i]      # the kernel's st_off argument
    ew = cg.Emu(state=dict(state)).run(cg.prog_f12mul(True).steps)
    assert ew.wire_out == [H(x) for x in c["miller"]]
    es = cg.Emu(state=dict(state)).run(cg.prog_f12mul(False).steps)
    assert [cg.from_mont(es.state[cg.ST_F + i]) for i in range(12)] == [H(x) for x in c["miller"]]


def test_fp12_product_tree_level(model_vectors):
    """one level of the Fp12 product tree: wire x wire -> wire"""
    H = lambda s: int(s, 16)
    c = model_vectors["pairing"]["multi3"]
    a = [H(x) for x in c["miller"]]
    g = m.f12_flat_ints(m.multi_miller_loop([(m.G1_GEN, m.G2_GEN)]))
    em = cg.Emu(wire_in=a, wire_in2=g).run(cg.prog_f12mul_pairs().steps)
    assert em.wire_out == m.f12_flat_ints(m.f12_mul(m.f12_from_flat_ints(a), m.f12_from_flat_ints(g)))
    # x * 1 == x either way
    one = [1] + [0] * 11
    assert cg.Emu(wire_in=a, wire_in2=one).run(cg.prog_f12mul_pairs().steps).wire_out == a
    assert cg.Emu(wire_in=one, wire_in2=g).run(cg.prog_f12mul_pairs().steps).wire_out == g


def _cyclotomic(seed):
    g = m.SplitMix64(seed)
    f = [(g.below(m.P), g.below(m.P)) for _ in range(6)]
    t0 = f
    for _ in range(6):
        t0 = m.f12_frob(t0)
    t2 = m.f12_mul(t0, m.f12_inv(f))
    return m.f12_mul(m.f12_frob(m.f12_frob(t2)), t2)


def _state_of(v12, base):
    return {base + i: cg.mont(x) for i, x in enumerate(m.f12_flat_ints(v12))}


def test_compressed_squaring_run_and_decompression():
    """k_ksq / k_kdec_a / k_kdec_b (limb-exact models): snapshots of a compressed squaring run, completed by the
    decompression, equal g^(2^e); the identity (all four coefficients zero) and z2 == 0 take the 0 / 0 := 0 route."""
    g = _cyclotomic(77)
    state = _state_of(g, 0)
    mask = (1 << 0) | (1 << 2) | (1 << 6)
    cg.emu_ksq(state, 0, cg.ST_SNAP, 7, mask)
    cg.emu_kdec_a(state, cg.ST_SNAP, 3, cg.ST_KN)
    cg.emu_inv(state, cg.ST_KN, cg.ST_KNINV, 3)
    cg.emu_kdec_b(state, cg.ST_SNAP, 3, cg.ST_KNINV)
    h, want = g, {}
    for e in range(1, 8):
        h = m.f12_sqr(h)
        want[e] = m.f12_flat_ints(h)
    for k, e in enumerate((1, 3, 7)):
        got = [cg.from_mont(state[cg.ST_SNAP + 12 * k + i]) for i in range(12)]
        assert got == want[e], (k, e)
    # the identity: compressed form (0, 0, 0, 0) -> decompression gives 1
    one = m.f12_one()
    state = _state_of(one, 0)
    cg.emu_ksq(state, 0, cg.ST_SNAP, 3, 1 << 2)
    cg.emu_kdec_a(state, cg.ST_SNAP, 1, cg.ST_KN)
    cg.emu_inv(state, cg.ST_KN, cg.ST_KNINV, 1)
    cg.emu_kdec_b(state, cg.ST_SNAP, 1, cg.ST_KNINV)
    assert [cg.from_mont(state[cg.ST_SNAP + i]) for i in range(12)] == [1] + [0] * 11
    # the exceptional branch z2 == 0, z3 != 0 (z1 = 2 z4 z5 / z3) on an artificial record: the formulas, not a group element
    g = m.SplitMix64(5)
    z3, z4, z5 = [(g.below(m.P), g.below(m.P)) for _ in range(3)]
    flat = [0] * 12
    for pos, z in ((2, z3), (1, z4), (5, z5)):
        flat[2 * pos], flat[2 * pos + 1] = z
    state = {cg.ST_SNAP + i: cg.mont(v) for i, v in enumerate(flat)}
    cg.emu_kdec_a(state, cg.ST_SNAP, 1, cg.ST_KN)
    cg.emu_inv(state, cg.ST_KN, cg.ST_KNINV, 1)
    cg.emu_kdec_b(state, cg.ST_SNAP, 1, cg.ST_KNINV)
    sc = lambda a, k: ((a[0] * k) % m.P, (a[1] * k) % m.P)
    z1 = m.f2_mul(sc(m.f2_mul(z4, z5), 2), m.f2_inv(z3))
    z0 = m.f2_add(m.f2_mul_xi(m.f2_sub(sc(m.f2_sqr(z1), 2), sc(m.f2_mul(z3, z4), 3))), (1, 0))
    got = [cg.from_mont(state[cg.ST_SNAP + i]) for i in range(12)]
    assert (got[0], got[1]) == z0 and (got[8], got[9]) == z1
    # the plan's mask is the set bits of |x|
    assert cg.X_BITS == [16, 48, 57, 60, 62, 63] and cg.KSQ_NSQ == 63


def test_tower_op_programs_match_model():
    """the zkp_tower_op_batch programs (direct hooks for Fp2 / Fp6 / Fp12 primitives) against the big-int model"""
    g = m.SplitMix64(4242)
    rnd = lambda: [g.below(m.P) for _ in range(12)]
    a, b = rnd(), rnd()
    A, B = m.f12_from_flat_ints(a), m.f12_from_flat_ints(b)
    run = lambda op, x, y=None: cg.Emu(wire_in=x, wire_in2=y).run(cg.prog_tower(op).steps).wire_out
    assert run("fp12_mul", a, b) == m.f12_flat_ints(m.f12_mul(A, B))
    assert run("fp12_sqr", a) == m.f12_flat_ints(m.f12_sqr(A))
    assert run("fp12_frob", a) == m.f12_flat_ints(m.f12_frob(A))
    assert run("fp12_conj", a) == m.f12_flat_ints(m.f12_conj(A))
    c0, c1, c4 = (b[0], b[1]), (b[2], b[3]), (b[4], b[5])
    assert run("fp12_014", a, b) == m.f12_flat_ints(m.f12_mul(A, m._sparse_014(c0, c1, c4)))
    pad = lambda v: list(v) + [0] * (12 - len(v))
    f2 = lambda v: (v[0], v[1])
    assert run("fp2_mul", pad(a[:2]), pad(b[:2])) == pad(m.f2_mul(f2(a), f2(b)))
    assert run("fp2_sqr", pad(a[:2])) == pad(m.f2_sqr(f2(a)))
    f6 = lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
    flat6 = lambda t: [c for pr in t for c in pr]
    assert run("fp6_mul", pad(a[:6]), pad(b[:6])) == pad(flat6(m.f6_mul(f6(a), f6(b))))
    assert run("fp6_sqr", pad(a[:6])) == pad(flat6(m.f6_mul(f6(a), f6(a))))
    assert run("fp12_frob", pad(a[:6]))[:6] == flat6(m.f6_frob_true(f6(a)))
    cyc = m.f12_flat_ints(_cyclotomic(9))
    assert run("cyc_sqr", cyc) == m.f12_flat_ints(m.f12_sqr(m.f12_from_flat_ints(cyc)))
    # wire -> state -> (k_ksq, decompression) -> wire: g^(2^5)
    st = cg.Emu(wire_in=cyc).run(cg.prog_tower_to_state().steps).state
    cg.emu_ksq(st, 0, cg.ST_SNAP, 5, 1 << 4)
    cg.emu_kdec_a(st, cg.ST_SNAP, 1, cg.ST_KN)
    cg.emu_inv(st, cg.ST_KN, cg.ST_KNINV, 1)
    cg.emu_kdec_b(st, cg.ST_SNAP, 1, cg.ST_KNINV)
    h = m.f12_from_flat_ints(cyc)
    for _ in range(5):
        h = m.f12_sqr(h)
    assert cg.Emu(state=st).run(cg.prog_tower_from_snap().steps).wire_out == m.f12_flat_ints(h)


def test_remaining_tower_programs_match_model():
    """round 4: mul_by_nonresidue, Fp2 x Fp, Fp6::mul_by_1 / mul_by_01 and the three inversions (program A, the batched
    inversion's value, program B) against the big-int model; a zero input ends as the zero record"""
    g = m.SplitMix64(99)
    rnd = lambda: [g.below(m.P) for _ in range(12)]
    a, b = rnd(), rnd()
    pad = lambda v: list(v) + [0] * (12 - len(v))
    f2 = lambda v: (v[0], v[1])
    f6 = lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
    flat6 = lambda t: [c for pr in t for c in pr]
    z = (0, 0)
    run = lambda op, x, y=None: cg.Emu(wire_in=x, wire_in2=y).run(cg.prog_tower2(op).steps).wire_out
    assert run("fp2_nr", pad(a[:2])) == pad(m.f2_mul_xi(f2(a)))
    assert run("fp2_mulfp", pad(a[:2]), pad(b[:1])) == pad([a[0] * b[0] % m.P, a[1] * b[0] % m.P])
    assert run("fp6_by1", pad(a[:6]), pad(b[:2])) == pad(flat6(m.f6_mul(f6(a), (z, f2(b), z))))
    assert run("fp6_by01", pad(a[:6]), pad(b[:4])) == pad(flat6(m.f6_mul(f6(a), (f2(b), (b[2], b[3]), z))))
    assert run("fp6_nr", pad(a[:6])) == pad(flat6(m.f6_mul(f6(a), (z, (1, 0), z))))

    def inv(which, x):
        prog_a = cg.prog_fexp_a(True) if which == "fp12" else cg.prog_tower_inv_a(which)
        ea = cg.Emu(wire_in=x).run(prog_a.steps)
        nn = cg.from_mont(ea.state[cg.ST_N])
        ea.state[cg.ST_NINV] = cg.mont(m.fp_inv(nn) if nn else 0)      # k_batch_inv: 0 gives 0
        return cg.Emu(state=ea.state).run(cg.prog_tower_inv_b(which).steps).wire_out

    assert inv("fp2", pad(a[:2])) == pad(m.f2_inv(f2(a)))
    assert inv("fp6", pad(a[:6])) == pad(flat6(m.f6_inv(f6(a))))
    assert inv("fp12", a) == m.f12_flat_ints(m.f12_inv(m.f12_from_flat_ints(a)))
    for w in ("fp2", "fp6", "fp12"):
        assert inv(w, [0] * 12) == [0] * 12
        assert cg.PROGRAMS["tw_%s_inv_b" % w]().peak <= cg.LDS_SLOTS


def test_program_encoding_is_consistent():
    for name, mk in cg.PROGRAMS.items():
        b = mk()
        hdr, tbl = cg.encode(b)
        assert len(hdr) % 4 == 0 and hdr[-4] == cg.OP_END
        assert b.peak <= cg.LDS_DEEP_SLOTS, (name, b.peak)     # write_inc asserts the matching constants limit (lds_config)
        depth = 0
        for i in range(0, len(hdr), 4):
            op = hdr[i] & 0xFF
            if op == cg.OP_LOOP:
                depth += 1
                assert depth == 1 and hdr[i + 1] >= 2
            if op == cg.OP_ENDLOOP:
                depth -= 1
            if op in (cg.OP_MULACC, cg.OP_LIN, cg.OP_GLOAD, cg.OP_GSTORE):
                assert hdr[i + 2] < len(tbl)
        assert depth == 0
    # the generated include is in sync with the generator
    import tempfile
    with tempfile.NamedTemporaryFile("r", suffix=".inc") as tf:
        cg.write_inc(tf.name)
        with open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_coop_prog.inc")) as f:
            assert f.read() == open(tf.name).read(), "run tools/coopgen.py to regenerate zkp_coop_prog.inc"


def test_executed_multiply_add_count():
    """tools/executed_macs.py (what bench.py prints as executed_macs_per_pairing): the generators' own counts, below the algorithmic
    6.56 M of SURVEY 8(d) and dominated by the Miller program and the compressed squarings"""
    import executed_macs
    r = executed_macs.per_pairing()
    assert abs(r["total"] - sum(v for k, v in r.items() if k != "total")) < 1
    assert 4.4e6 < r["total"] < 5.6e6 < 6560700
    assert r["k_coop miller1"] > r["k_ksq"] > r["k_coop<36,24> hard-part step programs"] > r["k_prep_lines<true>"] > r["k_coop<24,34> phase-C step programs"] > r["k_coop fexp_a"]
    # the other BASELINE configs (bench.py secondary_workloads.*.executed_frac_of_peak): the subgroup checks execute what the algorithmic
    # count prices (bench.py FPMUL_G1_VALID / FPMUL_G2_VALID x 300) to within 6 % - their algorithmic fraction has no Karatsuba / lazy slack
    s2 = executed_macs.secondary()
    assert 0.94 < s2["g1_is_valid_point"] / (1002 * 300) < 1.06 and 0.90 < s2["g2_is_valid_point"] / (1346 * 300) < 1.0
    assert 0.72 < s2["config4_three_pair_check"] / (33221 * 300) < 0.82 and s2["multi_miller_loop_pair"] > r["k_coop miller1"]
    # a Karatsuba block is 147 multiply-adds, a reduction 196: the Fp12 product step of the interpreter
    prod = [st for st in cg.prog_tower("fp12_mul").steps if st["op"] == cg.OP_MULACC and st["T"] == 12]
    assert prod and executed_macs.program_macs_per_lane(prod) == len(prod) * (12 * 147 + 196)


def test_division_step_inversion_model():
    """tools/safegcd_model.py: the limb-exact model of the kernels' Fp inversion (f_inv in zkp_coop.hip) against pow(x, -1, p);
    its constants are the ones tools/gen_constants.py writes into zkp_constants28.h"""
    import safegcd_model as sgm
    assert sgm.selftest(n=300)
    hdr = open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_constants28.h")).read()
    assert "#define ZKP30_PINV %du" % sgm.PINV30 in hdr
    assert "#define ZKP30_P_LIMBS %s" % ", ".join(str(x) for x in sgm.PL) in hdr


def test_run_time_k_miller_program_matches_the_model():
    """round 5: the Miller program whose pair count is a LAUNCH argument (a pair loop inside every iteration: one accumulator, the 63
    squarings shared by all pairs - what checks with more than eight pairs run through) equals the model's multi_miller_loop for
    k = 1, 3, 9 and 11 pairs (an infinity among them), and the unrolled program where both exist"""
    g = m.SplitMix64(2026)
    pts = []
    for _ in range(11):
        a, bq = g.below(1 << 200) + 1, g.below(1 << 200) + 1
        pts.append((m.g1_mul(m.G1_GEN, a), m.g2_mul(m.G2_GEN, bq)))
    pts[4] = (None, pts[4][1])                         # an infinity: the neutral line
    prog = cg.prog_miller_n(True)
    assert sum(1 for st in prog.steps if st["op"] == cg.OP_PLOOP) == sum(1 for st in prog.steps if st["op"] == cg.OP_PENDLOOP) > 5
    for k in (1, 3, 9, 11):
        pairs = pts[:k]
        em = cg.Emu(lines=cg.model_lines(pairs)).run(prog.steps)
        assert em.cursor == cg.n_line_steps()
        assert em.wire_out == m.f12_flat_ints(m.multi_miller_loop([p for p in pairs if p[0] is not None])), k
        if k <= 3:
            assert em.wire_out == cg.Emu(lines=cg.model_lines(pairs)).run(cg.prog_miller(k, True).steps).wire_out
    # the encoded program: the pair loop's line loads carry pair 0 (the kernel adds 6 x the loop counter), the loop end moves the cursor
    hdr, tbl = cg.encode(prog)
    ops = [hdr[i] & 0xff for i in range(0, len(hdr), 4)]
    assert ops.count(cg.OP_PLOOP) == ops.count(cg.OP_PENDLOOP) and all(hdr[i + 1] == 1 for i in range(0, len(hdr), 4) if (hdr[i] & 0xff) == cg.OP_PENDLOOP)
