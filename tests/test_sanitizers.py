"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU restatement (oracle/), both arithmetic modes, and over the PRODUCT's
host planners (zkvm_pairings_amd/csrc/zkp_plan.hpp) at the sizes the C ABI admits (sanitizers run on the CPU
build; the GPU pool offers none)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("oracle sanitize run ok") == 2
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_host_planners_under_asan_and_ubsan_at_the_abi_maxima(tmp_path):
    """the product's own host arithmetic - chunk / group / part / slice plans, workspace sizes, launch counts, the 32-bit offsets inside
    k_prep_lines - is one header (csrc/zkp_plan.hpp) that zkp_coop.hip / zkp_pairings.hip include; tests/plan_check.cpp walks it with gcc
    -fsanitize=address,undefined over n in {1 .. 2^31 - 1} x k in {1 .. 65535} x every chunk / pipeline / group knob the clamps admit
    and asserts, in overflow-checked 64-bit arithmetic, that every count fits the type the kernel receives it in (round 4's 32-bit
    scratch offset at n >= 2^27 was found by reading, not by a test)"""
    exe = str(tmp_path / "plan_check")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                         "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "plan_check.cpp")], capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stdout[-3000:] + cc.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "plan_check ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    # the product includes the very header the test walked
    for src in ("zkp_coop.hip", "zkp_pairings.hip"):
        with open(os.path.join(ROOT, "zkvm_pairings_amd", "csrc", src)) as f:
            assert '#include "zkp_plan.hpp"' in f.read(), src


def test_slow_mode_build_equals_the_fast_build():
    """liborc_slow.so (-DORC_SLOW: canonical integers, schoolbook product + long division per Fp::mul, the shape of the
    reference's src/fp.rs:416-434) gives the same pairings as the Montgomery build"""
    import numpy as np
    import oracle_lib as o
    g1 = np.stack([o.g1_generator()] * 3)
    g2 = np.stack([o.g2_generator()] * 3)
    ks = np.array([[3, 0, 0, 0], [5, 7, 0, 0], [11, 1, 2, 3]], dtype=np.uint64)
    g1, g2 = o.g1_mul_batch(g1, ks), o.g2_mul_batch(g2, ks[::-1].copy())
    assert np.array_equal(o.pairing_batch(g1, g2), o.pairing_batch_slow(g1, g2))
