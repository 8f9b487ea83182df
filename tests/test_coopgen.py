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
    ec = cg.Emu(state=ea.state).run(cg.prog_fexp_c(True).steps)
    return f_true, ec, (em, ea)


def test_emulated_pairing_of_generators_matches_model():
    f_true, ec, (em, ea) = run_pairing([(m.G1_GEN, m.G2_GEN)])
    ml = m.multi_miller_loop([(m.G1_GEN, m.G2_GEN)])
    assert f_true == m.f12_flat_ints(ml)
    assert ec.wire_out == m.f12_flat_ints(m.final_exponentiation(ml))
    assert ec.is_identity is False
    # headroom of the lazy accumulation: columns must stay well inside 63 bits
    assert max(em.max_col, ec.max_col).bit_length() <= 61
    # step census used by DESIGN.md
    assert em.counts["mulacc_steps"] > 100 and ec.counts["mulacc_steps"] > 300


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
    ec = cg.Emu(state=ea.state).run(cg.prog_fexp_c(True).steps)
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


def test_program_encoding_is_consistent():
    for name, mk in cg.PROGRAMS.items():
        b = mk()
        hdr, tbl = cg.encode(b)
        assert len(hdr) % 4 == 0 and hdr[-4] == cg.OP_END
        assert b.peak <= cg.LDS_WIDE_SLOTS, (name, b.peak)     # write_inc asserts the matching constants limit
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


def test_division_step_inversion_model():
    """tools/safegcd_model.py: the limb-exact model of the kernels' Fp inversion (f_inv in zkp_coop.hip) against pow(x, -1, p);
    its constants are the ones tools/gen_constants.py writes into zkp_constants28.h"""
    import safegcd_model as sgm
    assert sgm.selftest(n=300)
    hdr = open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_constants28.h")).read()
    assert "#define ZKP30_PINV %du" % sgm.PINV30 in hdr
    assert "#define ZKP30_P_LIMBS %s" % ", ".join(str(x) for x in sgm.PL) in hdr
