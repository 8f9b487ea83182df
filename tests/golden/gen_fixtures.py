#!/usr/bin/env python3
"""Generate tests/golden/model_vectors.json from the independent big-int model
(tests/golden/bls12_381_model.py).  These are the BUILD'S OWN vectors (the reference has no pairing
code and cannot be built here); they pin the C oracle and, through it, the HIP path.
Deterministic: fixed SplitMix64 seeds.  Run: python3 tests/golden/gen_fixtures.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import bls12_381_model as m  # noqa: E402

P = m.P
hx = m.hex_fp


def f2_sqrt(a):
    if a == m.F2_ZERO:
        return a
    n = (a[0] * a[0] + a[1] * a[1]) % P
    s = m.fp_sqrt(n)
    if s is None:
        return None
    inv2 = m.fp_inv(2)
    for ss in (s, (-s) % P):
        x0 = m.fp_sqrt((a[0] + ss) * inv2 % P)
        if x0 is not None and x0 != 0:
            x1 = a[1] * m.fp_inv(2 * x0 % P) % P
            r = (x0, x1)
            if m.f2_sqr(r) == a:
                return r
    return None


def rnd_fp(g):
    return g.below(P)


def rnd_f2(g):
    return (rnd_fp(g), rnd_fp(g))


def rnd_f12(g):
    return [rnd_f2(g) for _ in range(6)]


def flat12(a):
    return [hx(v) for v in m.f12_flat_ints(a)]


def flat6(a):
    return [hx(v) for c in a for v in c]


def flat2(a):
    return [hx(a[0]), hx(a[1])]


def g1hex(p):
    return [hx(p[0]), hx(p[1])]


def g2hex(q):
    return [hx(q[0][0]), hx(q[0][1]), hx(q[1][0]), hx(q[1][1])]


def main():
    out = {}
    g = m.SplitMix64(0xF1E1D)
    # ---- Fp
    fpv = []
    for _ in range(6):
        a, b = rnd_fp(g), rnd_fp(g)
        s = m.fp_sqrt(a * a % P)
        fpv.append({"a": hx(a), "b": hx(b), "add": hx((a + b) % P), "sub": hx((a - b) % P), "mul": hx(a * b % P),
                    "neg": hx(-a % P), "inv": hx(m.fp_inv(a)), "sqrt_of_a2": hx(s), "is_residue": m.fp_sqrt(a) is not None})
    fpv.append({"a": hx(P - 1), "b": hx(P - 1), "add": hx((2 * P - 2) % P), "sub": hx(0), "mul": hx(1), "neg": hx(1),
                "inv": hx(P - 1), "sqrt_of_a2": hx(m.fp_sqrt(1)), "is_residue": m.fp_sqrt(P - 1) is not None})
    out["fp"] = fpv
    # ---- Fp2
    v = []
    for _ in range(4):
        a, b = rnd_f2(g), rnd_f2(g)
        v.append({"a": flat2(a), "b": flat2(b), "mul": flat2(m.f2_mul(a, b)), "square": flat2(m.f2_sqr(a)),
                  "inv": flat2(m.f2_inv(a)), "mul_by_nonresidue": flat2(m.f2_mul_xi(a)), "conj": flat2(m.f2_conj(a))})
    out["fp2"] = v
    # ---- Fp6
    v = []
    for _ in range(3):
        a = (rnd_f2(g), rnd_f2(g), rnd_f2(g))
        b = (rnd_f2(g), rnd_f2(g), rnd_f2(g))
        c0, c1 = rnd_f2(g), rnd_f2(g)
        v.append({"a": flat6(a), "b": flat6(b), "mul": flat6(m.f6_mul(a, b)), "square": flat6(m.f6_mul(a, a)),
                  "inv": flat6(m.f6_inv(a)), "mul_by_nonresidue": flat6(m.f6_mul_by_v(a)),
                  "c0": flat2(c0), "c1": flat2(c1),
                  "mul_by_1": flat6(m.f6_mul(a, (m.F2_ZERO, c1, m.F2_ZERO))),
                  "mul_by_01": flat6(m.f6_mul(a, (c0, c1, m.F2_ZERO))),
                  "frobenius_true": flat6(m.f6_frob_true(a)),
                  "frobenius_refcompat": flat6(m.f6_frob_refcompat(a))})
    out["fp6"] = v
    # ---- Fp12
    v = []
    for _ in range(3):
        a, b = rnd_f12(g), rnd_f12(g)
        c0, c1, c4 = rnd_f2(g), rnd_f2(g), rnd_f2(g)
        fr = m.f12_frob(a)
        assert fr == m.f12_pow(a, P), "true Frobenius must equal x^p"
        v.append({"a": flat12(a), "b": flat12(b), "mul": flat12(m.f12_mul(a, b)), "square": flat12(m.f12_sqr(a)),
                  "inv": flat12(m.f12_inv(a)), "conj": flat12(m.f12_conj(a)), "frobenius_true": flat12(fr),
                  "c0": flat2(c0), "c1": flat2(c1), "c4": flat2(c4),
                  "mul_by_014": flat12(m.f12_mul(a, m._sparse_014(c0, c1, c4)))})
    # cyclotomic element: easy part of final exp
    a = rnd_f12(g)
    t = m.f12_mul(m.f12_conj(a), m.f12_inv(a))
    t = m.f12_mul(m.f12_frob(m.f12_frob(t)), t)
    out["fp12_cyclotomic"] = {"a": flat12(t), "square": flat12(m.f12_sqr(t)), "exp_x_conj": flat12(m.cyclotomic_exp(t))}
    out["fp12"] = v
    # ---- groups
    grp = {"g1_mul": [], "g2_mul": []}
    for k in (1, 2, 3, 5, 0xD201000000010000, g.below(m.R_ORDER), g.below(m.R_ORDER), m.R_ORDER - 1):
        grp["g1_mul"].append({"k": hex(k), "p": g1hex(m.g1_mul(m.G1_GEN, k))})
        grp["g2_mul"].append({"k": hex(k), "p": g2hex(m.g2_mul(m.G2_GEN, k))})
    # on-curve points outside the prime-order subgroup, and off-curve points
    bad1, bad2 = [], []
    x = 3
    while len(bad1) < 3:
        y = m.fp_sqrt((x * x * x + 4) % P)
        if y is not None:
            pt = (x, y)
            assert m.g1_on_curve(pt)
            bad1.append({"p": g1hex(pt), "status": 0 if m.g1_torsion_free(pt) else 2})
        x += 1
    bad1.append({"p": g1hex((m.G1_GEN[0], (m.G1_GEN[1] + 1) % P)), "status": 1})
    xx = (5, 1)
    while len(bad2) < 3:
        y = f2_sqrt(m.f2_add(m.f2_mul(m.f2_sqr(xx), xx), (4, 4)))
        if y is not None:
            pt = (xx, y)
            assert m.g2_on_curve(pt)
            bad2.append({"p": g2hex(pt), "status": 0 if m.g2_torsion_free(pt) else 2})
        xx = ((xx[0] + 1) % P, xx[1])
    bad2.append({"p": g2hex((m.G2_GEN[0], m.f2_add(m.G2_GEN[1], (1, 0)))), "status": 1})
    grp["g1_validity"], grp["g2_validity"] = bad1, bad2
    grp["g2_psi_of_gen"] = g2hex(m.g2_psi(m.G2_GEN))
    out["groups"] = grp
    # ---- pairing
    pr = {}
    ml = m.multi_miller_loop([(m.G1_GEN, m.G2_GEN)])
    e = m.final_exponentiation(ml)
    assert e == m.final_exponentiation_direct(ml), "chain must equal f^(3(p^12-1)/r)"
    assert m.f12_pow(e, m.R_ORDER) == m.f12_one() and e != m.f12_one()
    assert m.final_exponentiation(m.miller_affine(m.G1_GEN, m.G2_GEN)) == e, "affine-slope Miller agrees after final exp"
    pr["gen"] = {"miller": flat12(ml), "gt": flat12(e),
                 "gt_sha256_le": hashlib.sha256(b"".join(v.to_bytes(48, "little") for v in m.f12_flat_ints(e))).hexdigest()}
    cases = []
    for _ in range(3):
        a, b = 1 + g.below(m.R_ORDER - 1), 1 + g.below(m.R_ORDER - 1)
        p1, q = m.g1_mul(m.G1_GEN, a), m.g2_mul(m.G2_GEN, b)
        mlr = m.multi_miller_loop([(p1, q)])
        gt = m.final_exponentiation(mlr)
        assert gt == m.f12_pow(e, a * b % m.R_ORDER), "bilinearity"
        cases.append({"a": hex(a), "b": hex(b), "g1": g1hex(p1), "g2": g2hex(q), "miller": flat12(mlr), "gt": flat12(gt)})
    pr["random"] = cases
    # 3-pair product check (Groth16-shaped): a1 b1 + a2 b2 + a3 b3 = 0 mod r
    a1, b1, a2, b2, b3 = [1 + g.below(m.R_ORDER - 1) for _ in range(5)]
    a3 = (-(a1 * b1 + a2 * b2) * pow(b3, -1, m.R_ORDER)) % m.R_ORDER
    pairs = [(m.g1_mul(m.G1_GEN, a1), m.g2_mul(m.G2_GEN, b1)), (m.g1_mul(m.G1_GEN, a2), m.g2_mul(m.G2_GEN, b2)),
             (m.g1_mul(m.G1_GEN, a3), m.g2_mul(m.G2_GEN, b3))]
    mm = m.multi_miller_loop(pairs)
    assert m.final_exponentiation(mm) == m.f12_one()
    pr["multi3"] = {"g1": [g1hex(p1) for p1, _ in pairs], "g2": [g2hex(q) for _, q in pairs], "miller": flat12(mm),
                    "gt": flat12(m.final_exponentiation(mm))}
    # a failing variant: perturb the last G1 point
    pairs_bad = pairs[:2] + [(m.g1_mul(m.G1_GEN, (a3 + 1) % m.R_ORDER), pairs[2][1])]
    mb = m.multi_miller_loop(pairs_bad)
    pr["multi3_bad"] = {"g1": [g1hex(p1) for p1, _ in pairs_bad], "g2": [g2hex(q) for _, q in pairs_bad],
                        "miller": flat12(mb), "gt": flat12(m.final_exponentiation(mb))}
    out["pairing"] = pr
    # ---- synthetic-input generator pin (seed of SURVEY 8d)
    s = m.SplitMix64(0x5EEDB15381)
    out["splitmix64"] = {"seed": hex(0x5EEDB15381), "first8": [hex(s.next()) for _ in range(8)]}
    with open(os.path.join(HERE, "model_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote model_vectors.json", os.path.getsize(os.path.join(HERE, "model_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
