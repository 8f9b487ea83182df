"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU restatement (oracle/), both arithmetic modes (SURVEY.md
section 5: sanitizers run on the CPU build; the GPU pool offers none)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("oracle sanitize run ok") == 2
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


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
