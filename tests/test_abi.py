"""CPU gate for the boundary: the C-ABI library loads and exports every symbol include/zkp_pairings.h
declares (no compute calls: there is no GPU here), and refuses to pretend when no device exists."""
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
    assert lib.zkp_abi_version() == 4


def test_rust_and_c_bindings_list_the_same_symbols():
    """the (never compiled) Rust -sys crate declares every entry point of the header, nothing else"""
    with open(os.path.join(ROOT, "integration", "rust", "src", "lib.rs")) as f:
        rust = sorted(set(re.findall(r"pub fn (zkp_[a-z0-9_]+)\s*\(", f.read())))
    assert rust == _declared_symbols()


def _split_params(txt):
    txt = txt.strip()
    return [] if txt in ("", "void") else [p.strip() for p in txt.split(",")]


def _c_signatures():
    """name -> (return kind, [parameter kinds]) from the header; kinds: ptr / size / int / void"""
    with open(os.path.join(ROOT, "include", "zkp_pairings.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)

    def kind(t):
        if "*" in t:
            return "ptr"
        if "size_t" in t:
            return "size"
        if re.search(r"\b(int|unsigned|uint32_t)\b", t):
            return "int"
        assert t.strip() == "void", t
        return "void"

    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(zkp_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        out[name] = (kind(ret), [kind(p) for p in _split_params(params)])
    return out


def _rust_signatures():
    with open(os.path.join(ROOT, "integration", "rust", "src", "lib.rs")) as f:
        text = re.sub(r"//[^\n]*", "", f.read())

    def kind(t):
        t = t.strip()
        if t.startswith("*"):
            return "ptr"
        if t == "usize":
            return "size"
        assert t in ("c_int", "c_uint", "u32", "i32"), t
        return "int"

    out = {}
    for name, params, ret in re.findall(r"pub fn (zkp_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text):
        ps = [kind(p.split(":", 1)[1]) for p in _split_params(params)]
        out[name] = ("void" if not ret.strip() else kind(ret), ps)
    return out


def test_rust_and_ctypes_signatures_match_the_header():
    """arity and the kind of every parameter (pointer / size_t / int) of the Rust `extern "C"` block and of the ctypes table
    against the prototypes of include/zkp_pairings.h - not just the names: a binding with a swapped or missing size argument
    corrupts memory at the first call, and the Rust side cannot be compiled here"""
    import ctypes as ct
    from zkvm_pairings_amd import _lib
    c = _c_signatures()
    assert sorted(c) == _declared_symbols()
    rust = _rust_signatures()
    assert sorted(rust) == sorted(c)
    for name, sig in c.items():
        assert rust[name] == sig, (name, "rust", rust[name], "header", sig)

    def ckind(t):
        if t is None:
            return "void"
        if t is ct.c_size_t:
            return "size"
        if t in (ct.c_int, ct.c_uint, ct.c_uint32):
            return "int"
        assert t in (ct.c_void_p, ct.c_char_p) or issubclass(t, ct._Pointer), t
        return "ptr"

    for name, (res, args) in _lib.SIGNATURES.items():
        assert (ckind(res), [ckind(x) for x in args]) == c[name], (name, "ctypes")


def test_plain_c_consumers_compile_and_link(tmp_path):
    """integration/c/*.c against the header and the library (build + link only: they need a GPU to run; -m gpu runs them)"""
    import glob
    import subprocess
    libdir = os.path.join(ROOT, "zkvm_pairings_amd")
    for src in sorted(glob.glob(os.path.join(ROOT, "integration", "c", "*.c"))):
        exe = str(tmp_path / os.path.basename(src)[:-2])
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", libdir, "-lzkp_pairings",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])


def test_library_links_rccl_for_the_one_collective():
    """the AND-reduce lives behind the C ABI - the library itself depends on librccl"""
    import subprocess
    out = subprocess.run(["readelf", "-d", os.path.join(ROOT, "zkvm_pairings_amd", "libzkp_pairings.so")], capture_output=True, text=True).stdout
    assert "librccl.so" in out


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


def test_wrappers_enter_the_collective_before_raising_a_local_argument_error():
    """a rank whose OWN arguments are wrong (sizes that do not match, k that does not divide n) must still enter the collective its
    peers wait in - through the same entry point, with arguments the library refuses, so that it takes part with flag 0 / the zero
    record - and raise afterwards.  Checked on a recording stand-in for the library: no GPU, no communicator needed."""
    from zkvm_pairings_amd import PairingEngine

    class Recorder:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            def fn(*a):
                self.calls.append((name, a))
                return -1
            return fn

    eng = object.__new__(PairingEngine)
    eng._lib, eng._h, eng.device = Recorder(), None, 0
    g1, g2 = np.zeros((4, 12), dtype=np.uint64), np.zeros((3, 24), dtype=np.uint64)
    with pytest.raises(ValueError):
        eng.pairing_check_allreduce(g1, g2, 1)
    with pytest.raises(ValueError):
        eng.pairing_check_allreduce(g1, np.zeros((4, 24), dtype=np.uint64), 3)          # k does not divide n
    with pytest.raises(ValueError):
        eng.points_check_allreduce(np.zeros((4, 96), dtype=np.uint8), np.zeros((3, 192), dtype=np.uint8), 1)
    with pytest.raises(ValueError):
        eng.pairing_product_check_allgather(g1, g2)
    names = [c[0] for c in eng._lib.calls]
    assert names == ["zkp_pairing_check_batch_allreduce", "zkp_pairing_check_batch_allreduce", "zkp_points_check_batch_allreduce",
                     "zkp_pairing_product_check_allgather"]
    # ... with no points and ONE check / pair: the library's own ZKP_ERR_ARG path, which joins with flag 0 / the zero record
    for name, a in eng._lib.calls:
        ints = [x for x in a if isinstance(x, int)]
        assert ints[:1] == [1], (name, a)
