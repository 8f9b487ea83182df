"""CPU gate for the boundary: the C-ABI library loads and exports every symbol include/zkp_pairings.h
declares (no compute calls: there is no GPU here), and refuses to pretend when no device exists."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "zkp_pairings.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from zkvm_pairings_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libzkp_pairings.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding table lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.zkp_abi_version() == 2


def test_rust_and_c_bindings_list_the_same_symbols():
    """the (never compiled) Rust -sys crate declares every entry point of the header, nothing else"""
    with open(os.path.join(ROOT, "integration", "rust", "src", "lib.rs")) as f:
        rust = sorted(set(re.findall(r"pub fn (zkp_[a-z0-9_]+)\s*\(", f.read())))
    assert rust == _declared_symbols()


def test_gt_identity_and_strerror():
    from zkvm_pairings_amd import PairingEngine, _lib
    one = PairingEngine.gt_identity()
    assert one[0] == 1 and not one[1:].any()          # Fp12::one(), reference src/fp12.rs:87-89
    assert _lib.load().zkp_strerror(-2) == b"no usable HIP device"


def test_no_silent_cpu_fallback():
    """Without a GPU the product must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zkvm_pairings_amd import PairingEngine, _lib
    with pytest.raises(_lib.ZkpError) as ei:
        PairingEngine(0)
    assert ei.value.status == -2


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "zkvm_pairings_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    txt = f.read()
                assert "oracle_lib" not in txt and "liborc" not in txt and "bls12_381_oracle" not in txt, fn


def test_splitmix_vectorised_matches_model(model_vectors):
    from zkvm_pairings_amd import synthetic
    v = model_vectors["splitmix64"]
    got = synthetic.splitmix64(int(v["seed"], 16), 8)
    assert [hex(int(x)) for x in got] == v["first8"]
    s = synthetic.scalars(synthetic.SEED, 5)
    assert s.shape == (5, 4) and all(0 < synthetic.scalar_to_int(r) < synthetic.R_ORDER for r in s)
    # offset continues the same stream
    assert np.array_equal(synthetic.scalars(synthetic.SEED, 5)[2:], synthetic.scalars(synthetic.SEED, 3, offset=2))
