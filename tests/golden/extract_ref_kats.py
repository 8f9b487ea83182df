#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA (limb arrays and one hex string held by its
#[cfg(test)] modules) into tests/golden/ref_kats.json.  Only numbers are taken - no reference code.

Run in the build container (needs /root/reference); the JSON is committed so that the tests never
read /root/reference at run time.   Sources (file:line of the arrays):
  src/fp.rs:577-588     sqrt(300855555557) expected Debug string; 72057594037927816 non-residue
  src/g1.rs:262-341     G1 generator a, 2a;  second point b, 2b (declared, never asserted upstream)
  src/g2.rs:348-398     2*G2gen;  src/g2.rs:400-443 point that must NOT be torsion free
  src/fp6.rs:561-734    operands a,b,c of Fp6 test_arithmetic
  src/fp12.rs:413-762   operands a,b,c of Fp12 test_arithmetic
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ARR = re.compile(r"from_raw_unchecked\(\[\s*((?:0x[0-9a-fA-F_]+|\d+)\s*,\s*){5}(?:0x[0-9a-fA-F_]+|\d+)\s*,?\s*\]\)")
NUM = re.compile(r"0x[0-9a-fA-F_]+|\d+")


def arrays(path, lo, hi):
    with open(path) as f:
        text = "".join(f.readlines()[lo - 1:hi])
    out = []
    for mt in ARR.finditer(text):
        limbs = [int(x.replace("_", ""), 0) for x in NUM.findall(mt.group(0)[len("from_raw_unchecked(["):])]
        assert len(limbs) == 6
        out.append(sum(l << (64 * i) for i, l in enumerate(limbs)))
    return out


def hx(v):
    return "0x%096x" % v


def main():
    k = {}
    with open(os.path.join(REF, "fp.rs")) as f:
        fp_lines = f.readlines()
    txt = "".join(fp_lines[575:590])
    k["fp_sqrt"] = {
        "input": int(re.search(r"from_raw_unchecked\(\[(\d+), 0, 0, 0, 0, 0\]\)\s*\.sqrt\(\)\s*\.unwrap", txt).group(1)),
        "expected_debug": re.search(r'"(0x[0-9a-f]{96})"', txt).group(1),
        "non_residue": int(re.findall(r"from_raw_unchecked\(\[(\d+), 0, 0, 0, 0, 0\]\)", txt)[1]),
    }
    a = arrays(os.path.join(REF, "g1.rs"), 260, 345)
    assert len(a) == 8
    k["g1_double"] = {"a": [hx(a[0]), hx(a[1])], "a_double": [hx(a[2]), hx(a[3])],
                      "b": [hx(a[4]), hx(a[5])], "b_double": [hx(a[6]), hx(a[7])]}
    a = arrays(os.path.join(REF, "g2.rs"), 346, 399)
    assert len(a) == 4
    k["g2_gen_double"] = [hx(v) for v in a]
    a = arrays(os.path.join(REF, "g2.rs"), 400, 444)
    assert len(a) == 4
    k["g2_not_torsion_free"] = [hx(v) for v in a]
    a = arrays(os.path.join(REF, "g2.rs"), 276, 346)
    assert len(a) == 4
    k["g2_generator_in_test"] = [hx(v) for v in a]
    a = arrays(os.path.join(REF, "fp6.rs"), 561, 736)
    assert len(a) == 18, len(a)
    k["fp6_arith"] = {"a": [hx(v) for v in a[0:6]], "b": [hx(v) for v in a[6:12]], "c": [hx(v) for v in a[12:18]]}
    a = arrays(os.path.join(REF, "fp12.rs"), 413, 764)
    assert len(a) == 36, len(a)
    k["fp12_arith"] = {"a": [hx(v) for v in a[0:12]], "b": [hx(v) for v in a[12:24]], "c": [hx(v) for v in a[24:36]]}
    with open(os.path.join(HERE, "ref_kats.json"), "w") as f:
        json.dump(k, f, indent=1)
    print("wrote ref_kats.json")


if __name__ == "__main__":
    main()
