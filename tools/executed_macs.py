#!/usr/bin/env python3
"""EXECUTED 32x32->64-bit multiply-adds per pairing of the cooperative family, counted from what the generators emit - the figure
SURVEY.md 8(d) asks for beside the algorithmic 6,560,700 (which prices the reference-shaped schoolbook tower on 12-limb CIOS):

  * asm blocks (tools/prepasm.py, tools/coopasm.py): the v_mad_i64_i32 / v_mad_u64_u32 instructions of the generated text x the
    lanes that issue them (a wavefront issues for all 64 lanes, idle or not);
  * step programs (tools/coopgen.py): per MULACC step and lane T Karatsuba product blocks (147) + one Montgomery reduction (196),
    walked with the programs' loop counts; 60 executing lanes per wavefront of 5 checks (padding lanes of a step issue too; lanes 60..63
    are switched off since round 5);
  * the compiled kernels of the decompression and the batched inversions: products counted from their source (196 per schoolbook
    product block, 196 per reduction), an estimate marked as such.

`python3 tools/executed_macs.py` prints the breakdown; bench.py prints the total beside the algorithmic count."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import coopasm  # noqa: E402
import coopgen as cg  # noqa: E402
import prepasm  # noqa: E402

# lanes of an interpreter wavefront that execute: 5 checks x 12 lanes.  Lanes 60..63 own nothing; until round 4 they ran every instruction
# on a neighbour's operands (64 lanes issued), since round 5 they leave the kernel at its start (zkp_coop.hip, ZKP_COOP_IDLE_LANES_OFF)
COOP_LANES = 60

MAD = re.compile(r"^\s*v_mad_[iu]64_[iu]32\b")


def macs(lines):
    return sum(1 for l in lines if MAD.match(l))


def program_macs_per_lane(steps):
    """multiply-adds one lane issues for a step program (loops walked)"""
    total, mult, stack = 0, 1, []
    for st in steps:
        if st["op"] == cg.OP_LOOP:
            stack.append(mult)
            mult *= st["n"]
        elif st["op"] == cg.OP_ENDLOOP:
            mult = stack.pop()
        elif st["op"] == cg.OP_MULACC:
            total += mult * (st["T"] * 147 + 196)
    return total


def per_pairing():
    out = {}
    dbl, add = macs(prepasm.generate().lines), macs(prepasm.generate_add().lines)
    out["k_prep_lines<true>"] = 2 * (63 * dbl + 5 * add)                 # two lanes per pair
    out["k_coop miller1"] = program_macs_per_lane(cg.prog_miller(1, False).steps) * COOP_LANES / 5
    out["k_coop fexp_a"] = program_macs_per_lane(cg.prog_fexp_a(False).steps) * COOP_LANES / 5
    ksq_body = macs(coopasm.generate_ksq().lines)
    prog_c = prog_c_deep = ksq = kdec_a = kdec_b = inv = 0
    for st in cg.fexp_c_plan():
        if st[0] == "prog":
            m = program_macs_per_lane(st[1].steps) * COOP_LANES / 5
            if st[1].peak > cg.LDS_SLOTS:                                 # the 36-slot ("deep") LDS configuration: k_coop<36,24>
                prog_c_deep += m
            else:
                prog_c += m
        elif st[0] == cg.PLAN_KSQ:
            ksq += st[3] * ksq_body * 4                                   # four lanes per check
        elif st[0] == cg.PLAN_KDEC_A:
            # per snapshot, two lanes: two Fp2 squarings (2 x (196 + 196)), the zero test's reduction (196), |D|^2 (2 x 196 + 196)
            kdec_a += st[2] * 2 * (2 * 392 + 196 + 588)
        elif st[0] == cg.PLAN_KDEC_B:
            # conj(D) / |D|^2 (392), N / D (2 x 196 + 196), t under one reduction (5 x 196 + 196)
            kdec_b += st[2] * 2 * (392 + 588 + 1176)
        elif st[0] == cg.PLAN_INV:
            inv += st[3] * 3 * 392                                        # Montgomery's trick: three products per value (+ one shared inversion per 32)
    inv += 3 * 392                                                        # the single inversion of Fp12::invert
    out["k_coop<36,24> hard-part step programs"] = prog_c_deep
    out["k_coop<24,34> phase-C step programs"] = prog_c
    out["k_ksq"] = ksq
    out["k_kdec_a (estimate from the source)"] = kdec_a
    out["k_kdec_b (estimate from the source)"] = kdec_b
    out["k_batch_inv (estimate from the source)"] = inv
    out["total"] = sum(out.values())
    return out


def secondary():
    """executed multiply-adds per unit of the other BASELINE configs: a three-pair check of config 4 (three line streams, the miller3 program,
    one final exponentiation), one G1 / G2 point of config 5's is_valid (asm steps of tools/validasm.py; G1 one lane per point with
    127 doublings + 16 additions, G2 two lanes per point with 63 + 5; the curve-equation test, endomorphism and comparison are not
    counted: < 1 %), and multi_miller_loop() of one pair (the upstream-shaped line steps + miller1)"""
    import validasm
    base = per_pairing()
    fexp = sum(v for k, v in base.items() if k not in ("total", "k_prep_lines<true>", "k_coop miller1"))
    out = {}
    out["config4_three_pair_check"] = 3 * base["k_prep_lines<true>"] + program_macs_per_lane(cg.prog_miller(3, False).steps) * COOP_LANES / 5 + fexp
    out["g1_is_valid_point"] = 127 * macs(validasm.g1_dbl().lines) + 16 * macs(validasm.g1_madd().lines)
    out["g2_is_valid_point"] = 2 * (63 * macs(validasm.g2_dbl3().lines) + 5 * macs(validasm.g2_madd3().lines))
    jac = 2 * (63 * macs(prepasm.generate_jac().lines) + 5 * macs(prepasm.generate_jac_add().lines))
    out["multi_miller_loop_pair"] = jac + program_macs_per_lane(cg.prog_miller(1, True).steps) * COOP_LANES / 5
    out["final_exponentiation"] = fexp
    return out


if __name__ == "__main__":
    r = per_pairing()
    for k, v in r.items():
        print("%-52s %12.0f" % (k, v))
    print("algorithmic (SURVEY 8d): 6560700; executed / algorithmic = %.3f" % (r["total"] / 6560700))
    for k, v in secondary().items():
        print("%-52s %12.0f" % (k, v))
