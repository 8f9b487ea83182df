"""BASELINE.json configs 3, 4 and 5 at their stated sizes on one MI355X (run with -m gpu), through the C ABI.

At 2^20 elements the CPU oracle cannot check everything in seconds, so every test combines
  * a size-independent property that covers EVERY element (cancelling 2-pair checks, flags known by construction,
    statuses known by construction, one kernel family against the other),
  * a seeded sample compared bit for bit with the oracle.
tools/soak.py runs the same checks stand-alone and prints the digests kept under profiles/."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import bls12_381_model as m
import oracle_lib as o

pytestmark = pytest.mark.gpu
NTHREADS = 16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from zkvm_pairings_amd import PairingEngine
    e = PairingEngine(0)
    yield e
    e.close()


def test_config3_full_batch_2p20(eng):
    """config 3 on one GPU: 2^20 random (G1,G2) pairs through pairing() + the Gt::identity() check (the bench step)."""
    import torch
    from zkvm_pairings_amd import configs
    r = configs.run_config3(eng, 1 << 20, sample=1 << 12, prefix=1 << 16)
    assert r["flags_all_zero"] and r["all_ok"] == 0                   # random pairings are never the identity
    assert r["cancel_all_one"] and r["cancel_all_ok"] == 1            # e(P,Q) e(-P,Q) == 1 for EVERY element
    assert r["infinity_all_one"]                                      # every infinity flag set: AND flag true
    assert r["families_sha256_equal"], (r["sha256_prefix_coop"], r["sha256_prefix_thread"])
    g1, g2, got = r["sample_g1"], r["sample_g2"], r["sample_gt"]
    want = o.pairing_batch(g1, g2, nthreads=NTHREADS)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()
    assert np.array_equal(got, want)
    torch.cuda.empty_cache()


def test_config4_groth16_checks_2p18(eng):
    """config 4: 2^18 three-pair checks with one shared final exponentiation each; a3 solved so that the product is one,
    a seeded 1/1024 of the checks perturbed: every flag equals the constructed expectation, 64 vs the oracle."""
    from zkvm_pairings_amd import configs
    r = configs.run_config4(eng, 1 << 18)
    assert r["n_bad"] > 100 and r["flags_equal_expectation"] and r["all_ok"] == 0
    assert r["good_prefix_all_ok"] == 1
    g1, g2 = r["sample_g1"], r["sample_g2"]
    assert np.array_equal(r["sample_ok"], o.pairing_check_batch(g1, g2, 64, 3))
    assert np.array_equal(r["sample_ok"], r["sample_expect"])


def test_config5_raw_points_2p20(eng):
    """config 5 on one GPU: 2^20 uncompressed G1 and G2 byte strings -> decode -> is_valid with seeded fractions of
    off-curve, wrong-subgroup, non-canonical and infinity encodings; then the pairing check on the valid ones."""
    from zkvm_pairings_amd import configs
    r = configs.run_config5(eng, 1 << 20)
    for which in ("g1", "g2"):
        assert r[which + "_decode_status_equal"] and r[which + "_valid_status_equal"], which
        assert min(r[which + "_class_counts"].values()) > 50, r[which + "_class_counts"]
        pts, inf, st = r[which + "_sample"]
        fn = o.g1_is_valid if which == "g1" else o.g2_is_valid
        assert [fn(p, int(i)) for p, i in zip(pts, inf)] == st.tolist()
    assert r["pairing_checks_all_one"]
    # round 4: the same flow as ONE call (zkp_points_check_batch / _dev): raw bytes -> decode -> is_valid -> pairing check on 2^20
    # two-pair checks built from the same byte strings; every status byte and every ok byte equals the expectation the three-call
    # flow above established, the resident flavour gives the same bytes, a batch of good checks sets the AND flag
    assert r["one_call_status_equal"] and r["one_call_ok_equal"] and r["one_call_dev_equal"], r["one_call_sha256"]
    assert r["one_call_all_ok"] == 0 and r["one_call_n_ok"] > (1 << 19) and r["one_call_good_subset_all_ok"]


def test_points_check_small_cases_vs_oracle(eng):
    """zkp_points_check_batch on hand-made byte strings: every point class in either position, k = 1 and k = 3, against the
    oracle's from_bytes / is_valid / pairing check; empty batch; an infinity pairs to the identity"""
    from zkvm_pairings_amd import configs, synthetic
    n = 42
    raw1, cls1 = configs.raw_points(eng, n, 1, 77)
    raw2, cls2 = configs.raw_points(eng, n, 2, 78)
    pool1, pc1 = configs.raw_points(eng, 4096, 1, 5)
    pool2, pc2 = configs.raw_points(eng, 4096, 2, 6)
    for c in range(1, len(configs.CLASSES)):          # every class on either side, whatever the seed of the 42 drew
        raw1[c - 1] = pool1[np.flatnonzero(pc1 == c)[0]]
        raw2[5 + c] = pool2[np.flatnonzero(pc2 == c)[0]]

    def oracle_status(raw, which):
        nfp = 2 if which == 1 else 4
        out = []
        for row in raw:
            flags = row[0] & 0xE0
            if flags & 0xA0:
                out.append((2, None, 0))
                continue
            if flags & 0x40:
                rest = row.copy()
                rest[0] &= 0x1F
                out.append((2, None, 0) if rest.any() else (0, None, 1))
                continue
            fes = []
            body = row.copy()
            for e in range(nfp):
                fe = o.fp_from_bytes_be(bytes(body[48 * e: 48 * e + 48]))
                fes.append(fe)
            if any(f is None for f in fes):
                out.append((1, None, 0))
                continue
            pt = np.concatenate(fes if which == 1 else [fes[1], fes[0], fes[3], fes[2]])
            v = (o.g1_is_valid if which == 1 else o.g2_is_valid)(pt, 0)
            out.append((v + 2 if v else 0, pt, 0))
        return out

    o1, o2 = oracle_status(raw1, 1), oracle_status(raw2, 2)
    for k in (1, 3):
        s1, s2, ok, allok = eng.points_check(raw1, raw2, k)
        assert s1.tolist() == [x[0] for x in o1] and s2.tolist() == [x[0] for x in o2], k
        want = []
        for c in range(n // k):
            idx = range(c * k, c * k + k)
            if any(o1[i][0] or o2[i][0] for i in idx):
                want.append(0)
                continue
            live = [i for i in idx if not (o1[i][2] or o2[i][2])]
            if not live:
                want.append(1)
                continue
            g1 = np.stack([o1[i][1] for i in live])
            g2 = np.stack([o2[i][1] for i in live])
            want.append(int(o.pairing_check_batch(g1, g2, 1, len(live))[0]))
        assert ok.tolist() == want and allok == all(want), k
    s1, s2, ok, allok = eng.points_check(raw1[:0], raw2[:0], 1)
    assert s1.size == 0 and ok.size == 0 and allok
    # cancelling pair from the generators: ok, AND flag true
    P = synthetic.G1_GENERATOR.reshape(1, 12)
    B1 = np.concatenate([configs.to_bytes(P, 1), configs.to_bytes(configs.negate_g1(eng, P), 1)])
    B2 = np.repeat(configs.to_bytes(synthetic.G2_GENERATOR.reshape(1, 24), 2), 2, axis=0)
    s1, s2, ok, allok = eng.points_check(B1, B2, 2)
    assert not s1.any() and not s2.any() and ok.tolist() == [1] and allok


def test_new_entry_points_reject_bad_arguments(eng):
    """status codes instead of crashes: null pointers, mismatched sizes, a collective without a communicator"""
    import ctypes
    from zkvm_pairings_amd import _lib
    lib, h = eng._lib, eng._h
    b = np.zeros(96, dtype=np.uint8)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    allok = ctypes.c_int(7)
    assert lib.zkp_points_check_batch(h, None, None, 1, 1, None, None, None, ctypes.byref(allok)) == -1
    assert lib.zkp_points_check_batch(h, None, None, 0, 1, None, None, None, ctypes.byref(allok)) == 0 and allok.value == 1
    assert lib.zkp_points_check_batch(None, p(b), p(b), 1, 1, None, None, None, None) == -1
    assert lib.zkp_g1_decode_batch_dev(h, None, 4, None, None, None, None) == -1
    assert lib.zkp_and_allreduce_dev(h, None, None) == -1
    assert lib.zkp_comm_init_rank(h, 0, 0, b"\0" * 128) == -1 and lib.zkp_comm_init_rank(h, 2, 2, b"\0" * 128) == -1
    assert lib.zkp_comm_unique_id(None) == -1
    g1 = np.zeros(12, dtype=np.uint64)
    g2 = np.zeros(24, dtype=np.uint64)
    assert lib.zkp_pairing_check_batch_allreduce(h, p(g1), p(g2), None, None, 1, 1, None, ctypes.byref(allok)) == -6      # ZKP_ERR_COMM
    assert b"communicator" in lib.zkp_last_error(h)
    assert lib.zkp_strerror(-6) == b"RCCL error / no communicator"
    with pytest.raises(ValueError):
        eng.points_check(np.zeros((3, 96), dtype=np.uint8), np.zeros((2, 192), dtype=np.uint8), 1)
    with pytest.raises(ValueError):
        eng.points_check(np.zeros((3, 96), dtype=np.uint8), np.zeros((3, 192), dtype=np.uint8), 2)
    with pytest.raises(_lib.ZkpError):
        eng.tower_op(21, np.zeros((1, 72), dtype=np.uint64))


def test_codec_on_resident_tensors(eng):
    """zkp_g1/g2_decode_batch_dev / encode_batch_dev: the same bytes and points as the host-pointer codec, aligned and unaligned views"""
    import torch
    from zkvm_pairings_amd import configs
    for which in (1, 2):
        raw, cls = configs.raw_points(eng, 3000, which, 1234 + which)
        pts, inf, st = eng.decode_points(raw, which)
        t = torch.from_numpy(raw).cuda()
        dp, di, ds = eng.decode_points_dev(t, which)
        assert np.array_equal(dp.cpu().numpy().view(np.uint64), pts) and np.array_equal(di.cpu().numpy(), inf) and np.array_equal(ds.cpu().numpy(), st)
        # an unaligned view of the same bytes takes the byte-wise kernel: same result
        pad = torch.empty(raw.size + 3, dtype=torch.uint8, device="cuda")
        pad[3:] = t.reshape(-1)
        up, ui, us = eng.decode_points_dev(pad[3:], which)
        assert torch.equal(up, dp) and torch.equal(ui, di) and torch.equal(us, ds)
        good = np.flatnonzero(st == 0)
        enc = eng.encode_points_dev(dp[torch.from_numpy(good).cuda()].contiguous(), which, di[torch.from_numpy(good).cuda()].contiguous())
        assert np.array_equal(enc.cpu().numpy(), raw[good])
        assert eng.encode_points(pts[good], which, inf[good]) == raw[good].tobytes()


def test_rccl_and_reduce_behind_the_c_abi(tmp_path):
    """integration/c/zkp_comm.c: a plain-C rank runs its block of a sharded check and the path's one collective through
    include/zkp_pairings.h alone (zkp_comm_unique_id / zkp_comm_init_rank / zkp_pairing_check_batch_allreduce /
    zkp_pairing_product_check_allgather) - here as a ONE-rank communicator, which is what one GPU allows (RCCL refuses two
    ranks on one device); and the same entry points from Python on device tensors"""
    exe = str(tmp_path / "zkp_comm")
    libdir = os.path.join(ROOT, "zkvm_pairings_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "c", "zkp_comm.c"),
                           "-L", libdir, "-lzkp_pairings", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "1", "0", str(tmp_path / "comm.id"), "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "rank 0 of 1: RCCL AND-reduce behind the C ABI ok" in out.stdout
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic, configs
eng = z.PairingEngine(0)
assert eng.comm_info() == (0, 0)
eng.comm_init_rank(1, 0, z.PairingEngine.comm_unique_id())
assert eng.comm_info() == (1, 0)
g1, g2, _, _ = synthetic.random_pairs(eng, 64, seed=3)
G1 = np.stack([g1, configs.negate_g1(eng, g1)], axis=1).reshape(-1, 12)
G2 = np.repeat(g2, 2, axis=0)
ok, allok = eng.pairing_check_allreduce(G1, G2, 2)
assert ok.all() and allok is True
ok, allok = eng.pairing_check_allreduce(g1, g2, 1)
assert not ok.any() and allok is False
t1, t2 = torch.from_numpy(G1.view(np.int64)).cuda(), torch.from_numpy(G2.view(np.int64)).cuda()
ok, flag = eng.pairing_check_allreduce(t1, t2, 2)
torch.cuda.synchronize()
assert bool(ok.all().item()) and int(flag.item()) == 1
f = torch.zeros(1, dtype=torch.int32, device="cuda"); eng.and_allreduce(f); assert int(f.item()) == 0
B1, B2 = configs.to_bytes(G1, 1), configs.to_bytes(G2, 2)
s1, s2, okb, allb = eng.points_check_allreduce(B1, B2, 2)
assert not s1.any() and not s2.any() and okb.all() and allb is True
B1[3, 5] ^= 1
s1, s2, okb, allb = eng.points_check_allreduce(B1, B2, 2)
assert s1[3] == 3 and okb[1] == 0 and okb.sum() == len(okb) - 1 and allb is False
gt, one = eng.pairing_product_check_allgather(G1, G2)
assert one and np.array_equal(gt, eng.gt_identity())
gt2, one2 = eng.pairing_product_check(g1, g2)
gt3, one3 = eng.pairing_product_check_allgather(g1, g2)
assert np.array_equal(gt2, gt3) and one2 == one3 == False
# a rank whose own part fails still joins the collective (no peer hangs) with a flag / record that fails the global check, and returns its error
import ctypes
lib, h = eng._lib, eng._h
vp = lambda a: ctypes.c_void_p(a.ctypes.data)
allok = ctypes.c_int(7)
assert lib.zkp_pairing_check_batch_allreduce(h, None, vp(G2), None, None, 4, 2, None, ctypes.byref(allok)) == -1 and allok.value == 0
bad = G1.copy(); bad[0, 5] = 2**64 - 1                       # limbs >= p: with validation on the local call reports ZKP_ERR_NONCANONICAL
assert lib.zkp_set_validate(h, 1) == 0
allok = ctypes.c_int(7)
assert lib.zkp_pairing_check_batch_allreduce(h, vp(bad), vp(G2), None, None, len(G1) // 2, 2, None, ctypes.byref(allok)) == -4 and allok.value == 0
one = ctypes.c_int(7)
assert lib.zkp_pairing_product_check_allgather(h, None, vp(G2), None, None, 8, None, ctypes.byref(one)) == -1 and one.value == 0
assert b"null points" in lib.zkp_last_error(h)
one = ctypes.c_int(7)
assert lib.zkp_pairing_product_check_allgather(h, vp(bad), vp(G2), None, None, len(G1), None, ctypes.byref(one)) == -4 and one.value == 0
assert b"limbs" in lib.zkp_last_error(h) and lib.zkp_set_validate(h, 0) == 0
dflag = torch.full((1,), 7, dtype=torch.int32, device="cuda")
rc = lib.zkp_pairing_check_batch_allreduce_dev(h, None, ctypes.c_void_p(t2.data_ptr()), None, None, 4, 2, None, ctypes.c_void_p(dflag.data_ptr()), None)
torch.cuda.synchronize()
assert rc == -1 and int(dflag.item()) == 0
gt, one = eng.pairing_product_check_allgather(G1, G2)          # and the engine still works afterwards
assert one and np.array_equal(gt, eng.gt_identity())
eng.comm_destroy(); assert eng.comm_info() == (0, 0)
eng.close(); print("ABI COMM OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ABI COMM OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_bench_abi_collective_probe_on_one_rank():
    """bench.py's probe of the C-ABI collective (what every rank of an N > 1 run executes behind the timed region), rehearsed on
    one rank: communicator of one, 8 all-reduces, one zkp_pairing_check_batch_allreduce_dev, the outcome in the JSON line"""
    import json
    env = dict(os.environ, ZKP_BENCH_FORCE_ABI_PROBE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--pairs", "32768", "--bare"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    probe = line["abi_collective"]
    assert probe["ok"] is True and probe["ranks_ok"] == 1 and probe["check_flag"] == 0 and probe["and_of_ones"] == 1 and probe["allreduce_ms"] > 0


def test_pairing_over_several_contexts_from_c(tmp_path):
    """integration/c/zkp_multi.c: zkp_pairing_batch_multi / zkp_pairing_check_batch_multi with three contexts (all on
    device 0 of a one-GPU box) against the single-context calls, from plain C."""
    exe = str(tmp_path / "zkp_multi")
    libdir = os.path.join(ROOT, "zkvm_pairings_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "c", "zkp_multi.c"),
                           "-L", libdir, "-lzkp_pairings", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "3", "1500"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "C ABI multi-context ok" in out.stdout
    out = subprocess.run([exe, "2", "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr


def test_multi_context_entry_points_from_python(eng):
    from zkvm_pairings_amd import PairingEngine, multi, synthetic
    others = [PairingEngine(0), PairingEngine(0)]
    try:
        n = 777
        g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=99)
        inf1 = np.zeros(n, dtype=np.uint8)
        inf1[[0, 300, 776]] = 1
        gt, ok, allok = multi.pairing_multi([eng] + others, g1, g2, inf1, None)
        assert np.array_equal(gt, eng.pairing(g1, g2, inf1, None))
        assert np.array_equal(ok, inf1) and not allok
        ok3, all3 = multi.pairing_check_multi([eng] + others, g1[:774], g2[:774], 3)
        ok1, all1 = eng.pairing_check(g1[:774], g2[:774], 3)
        assert np.array_equal(ok3, ok1) and all3 == all1
    finally:
        for e in others:
            e.close()


def test_dev_calls_on_two_streams_do_not_race(eng):
    """two back-to-back zkp_pairing_batch_dev calls of ONE context on two different streams: the second waits (event) for
    the first one's use of the shared workspace; results equal the serial ones."""
    import torch
    from zkvm_pairings_amd import synthetic
    dev = torch.device("cuda", 0)
    n = 1 << 15
    a1, a2, _, _ = synthetic.random_pairs(eng, n, seed=5, device_tensors=True)
    b1, b2, _, _ = synthetic.random_pairs(eng, n, seed=6, device_tensors=True)
    ref_a, ref_b = eng.pairing(a1, a2), eng.pairing(b1, b2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_a = eng.pairing(a1, a2)
        with torch.cuda.stream(s2):
            got_b = eng.pairing(b1, b2)
        torch.cuda.synchronize()
        assert torch.equal(got_a, ref_a) and torch.equal(got_b, ref_b)


def test_validation_mode_on_device_pointers(eng):
    import torch
    from zkvm_pairings_amd import synthetic
    dev = torch.device("cuda", 0)
    g1, g2, _, _ = synthetic.random_pairs(eng, 64, seed=8, device_tensors=True)
    eng.set_validate(True)
    try:
        eng.pairing(g1, g2)
        assert eng.take_validation_status() is False
        bad = g1.clone()
        bad[5, :6] = torch.from_numpy(o.to_limbs(m.P).view(np.int64)).to(dev)   # x = p: not canonical
        eng.pairing(bad, g2)
        assert eng.take_validation_status() is True
        assert eng.take_validation_status() is False                              # reading clears the word
    finally:
        eng.set_validate(False)


def test_two_ranks_sharded_check_equals_single_rank(tmp_path):
    """two torchrun ranks sharing cuda:0 (gloo for the single collective of the path): the sharded AND flag, the
    sharded product check and bench.py's N = 2 line against the single-rank results"""
    script = os.path.join(ROOT, "tests", "dist_two_ranks.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ZKP_BENCH_SHARE_GPU="1", ZKP_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", script], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "TWO RANKS OK" in out.stdout
    # bench.py's own N = 2 path (contiguous shards of one seeded batch, max-over-ranks timing, the MIN all-reduce)
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs", "65536",
                          "--cpu-sample", "512"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["pairs_per_gpu"] == 32768
    assert line["config"]["gt_sample_bit_exact"] is True and "all-reduce(MIN)" in line["config"]["workload"] and "gloo" in line["config"]["workload"]
    # the N > 1 line carries what a bad scaling curve would be diagnosed from: every rank's own kernel time, collective time, wall time
    detail = line["ranks_detail"]["per_rank"]
    assert [d["rank"] for d in detail] == [0, 1] and all(d["pairs"] == 32768 and d["kernel_ms_alone"] > 0 and d["collective_ms_alone"] >= 0
                                                         and d["step_wall_ms"] > 0 for d in detail)
    # the same line from a PLAIN start (no torchrun around it): bench.py launches its two ranks itself, as a child process, before it
    # touches torch or HIP, and relays rank 0's line and the ranks' exit code
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs", "65536",
                            "--cpu-sample", "512"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                           env={k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert plain.returncode == 0, plain.stdout[-2000:] + plain.stderr[-2000:]
    lines = [ln for ln in plain.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # ONE JSON line on stdout, everything else went to stderr
    pl = json.loads(lines[0])
    assert pl["n_gpus"] == 2 and pl["config"]["ranks"] == 2 and pl["config"]["gt_sample_bit_exact"] is True
    assert pl["config"]["gt_fingerprint_rank0"] == line["config"]["gt_fingerprint_rank0"]
    # and the guard that stays: a WORLD_SIZE that does not match --gpus
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def test_bench_step_through_the_abi_collective_equals_the_torch_one():
    """`bench.py --collective abi`: the timed step is ONE zkp_pairing_gt_check_batch_allreduce_dev per rank on the library's own RCCL
    communicator (one rank here: all one GPU allows).  Same inputs, same build: Gt fingerprint, flag and parity equal the torch line's."""
    import json

    def run(*extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "32768", "--cpu-sample", "256",
                              "--no-secondary"] + list(extra), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])

    t, a = run(), run("--collective", "abi")
    assert t["config"]["collective"] == "torch" and a["config"]["collective"] == "abi"
    assert "zkp_pairing_gt_check_batch_allreduce_dev" in a["config"]["workload"] and a["abi_collective"]["ok"] is True
    for key in ("gt_fingerprint_rank0", "all_ok_flag", "gt_sample_bit_exact", "pairs_per_gpu", "ranks"):
        assert t["config"][key] == a["config"][key], key
    assert a["config"]["gt_sample_bit_exact"] is True and a["value"] > 0
    # the N = 1 line measures the ceiling of the strong-scaling curve instead of quoting it
    sb = t["strong_scaling_bound"]
    assert [x["n_gpus"] for x in sb["shards"]] == [2, 4, 8] and all(0 < x["rate_vs_full"] < 1.5 and x["pairs"] == 32768 // x["n_gpus"] for x in sb["shards"])


def test_rccl_all_reduce_min_on_one_rank():
    """the path's only collective through RCCL itself (backend "nccl" on ROCm), as far as one GPU allows: a single-rank
    process group, all_reduce(MIN) of the int32 AND flag and the all_gather of a 576-byte Fp12 value on the device"""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from zkvm_pairings_amd import dist as zd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for v in (0, 1):
    f = torch.tensor([v], dtype=torch.int32, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    assert int(f.item()) == v
part = torch.arange(72, dtype=torch.int64, device=dev)
out = [torch.empty_like(part)]
dist.all_gather(out, part)
assert torch.equal(out[0], part)
assert zd.max_over_ranks(1.5, dev) == 1.5
dist.barrier(); dist.destroy_process_group(); print("RCCL ONE RANK OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "RCCL ONE RANK OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_points_check_skips_the_pairing_of_invalid_checks(eng):
    """checks with an invalid point never enter the Miller loop (validate first, then use - reference src/g1.rs:49-62).  Round 6: the number
    of checks that do stays on the device - zkp_points_check_batch_dev launches the pairing phase for the worst case and its kernels
    read the count (no host read-back).  A 2^18 batch of two-pair checks (P, Q), (-P, Q) - each is the identity when its points are
    valid - with every second check / scattered checks carrying an invalid point: status bytes, ok bytes and the AND flag equal the
    expectation by construction and the bytes of the separate decode / is_valid / pairing-check calls, on both kernel families.
    (Timing lives in bench.py: secondary_workloads.config5_points_check.half_invalid_ratio.)"""
    import torch
    from zkvm_pairings_amd import configs, synthetic
    dev = torch.device("cuda", 0)
    n = 1 << 18
    g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=31, device_tensors=True)
    neg = torch.from_numpy(configs.negate_g1(eng, g1.cpu().numpy().view(np.uint64)).view(np.int64)).to(dev)
    G1 = torch.stack([g1, neg], dim=1).reshape(2 * n, 12).contiguous()
    G2 = torch.stack([g2, g2], dim=1).reshape(2 * n, 24).contiguous()
    b1, b2 = eng.encode_points_dev(G1, 1), eng.encode_points_dev(G2, 2)
    st1 = torch.empty(2 * n, dtype=torch.uint8, device=dev)
    st2 = torch.empty_like(st1)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)

    def run(a):
        eng.points_check(a, b2, 2, st1, st2, ok, flag)
        torch.cuda.synchronize()
        return st1.cpu().numpy().copy(), st2.cpu().numpy().copy(), ok.cpu().numpy().copy(), int(flag.item())

    s1, s2, okb, fl = run(b1)                                    # every point valid: every check the identity
    assert not s1.any() and not s2.any() and okb.all() and fl == 1
    half = b1.clone()
    half[::4, 95] ^= 1                                           # the first point of every second check: off the curve
    s1, s2, okb, fl = run(half)
    assert (s1[::4] == 3).all() and not s1.reshape(-1, 4)[:, 1:].any() and not s2.any()
    assert np.array_equal(okb, np.tile(np.array([0, 1], dtype=np.uint8), n // 2)) and fl == 0
    mixed = b1.clone()
    mixed[5::7, 0] |= 0x80                                       # a malformed flag byte here and there (decode status 2)
    s1, s2, okb, fl = run(mixed)
    bad = np.zeros(2 * n, dtype=bool)
    bad[5::7] = True
    assert np.array_equal(s1, np.where(bad, 2, 0).astype(np.uint8)) and not s2.any()
    want_ok = (~bad.reshape(n, 2).any(axis=1)).astype(np.uint8)
    assert np.array_equal(okb, want_ok) and fl == 0 and 30000 < int((okb == 0).sum()) < n
    sha = hashlib.sha256(s1.tobytes() + s2.tobytes() + okb.tobytes()).hexdigest()
    # the same bytes from the separate calls (decode, is_valid, pairing check over ALL checks with the invalid points flagged)
    p1, i1, d1 = eng.decode_points_dev(mixed, 1)
    p2, i2, d2 = eng.decode_points_dev(b2, 2)
    v1, v2 = eng.g1_is_valid(p1, i1), eng.g2_is_valid(p2, i2)
    comb = lambda d, v: torch.where(d != 0, d, torch.where(v != 0, v + 2, torch.zeros_like(v)))
    c1, c2 = comb(d1, v1), comb(d2, v2)
    okc, _ = eng.pairing_check(p1, p2, 2, ((i1 != 0) | (c1 != 0)).to(torch.uint8), ((i2 != 0) | (c2 != 0)).to(torch.uint8))
    okc = okc.bool() & ~((c1 != 0) | (c2 != 0)).reshape(n, 2).any(dim=1)
    assert sha == hashlib.sha256(c1.cpu().numpy().tobytes() + c2.cpu().numpy().tobytes() + okc.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()
    # the thread family has no count-reading kernels: it runs every check and fails the invalid ones afterwards - the same bytes
    eng.set_kernel("thread")
    try:
        m = 4096
        eng.points_check(mixed[: 2 * m], b2[: 2 * m], 2, st1[: 2 * m], st2[: 2 * m], ok[:m], flag)
        torch.cuda.synchronize()
        assert np.array_equal(st1[: 2 * m].cpu().numpy(), s1[: 2 * m]) and np.array_equal(ok[:m].cpu().numpy(), okb[:m]) and int(flag.item()) == 0
    finally:
        eng.set_kernel("auto")


def test_points_check_device_count_across_chunks_groups_and_super_chunks(monkeypatch):
    """the device-resident count (NDev) with every offset the plans produce: small chunks over three pipelines, two super-chunks, phase C
    in parts, and checks of 1, 2, 9 and 20 pairs (unrolled program, run-time-k program, groups joined by f12mul) - a scattered third of
    the checks carries an invalid point, the rest cancel to the identity; status bytes, ok bytes and the flag equal the expectation by
    construction, and an all-invalid and an all-valid batch behave"""
    import torch
    from zkvm_pairings_amd import PairingEngine, configs, synthetic
    monkeypatch.setenv("ZKP_COOP_CHUNK", "320")
    monkeypatch.setenv("ZKP_COOP_STREAMS", "3")
    monkeypatch.setenv("ZKP_COOP_SUPER", "640")
    monkeypatch.setenv("ZKP_COOP_C_SPLIT", "2")
    monkeypatch.setenv("ZKP_COOP_MAX_STREAM", "16")
    e = PairingEngine(0)
    dev = torch.device("cuda", 0)
    try:
        rng = np.random.default_rng(66)
        for k, n in ((1, 1003), (2, 1003), (9, 700), (20, 333)):
            h = (k + 1) // 2                                           # pairs (P, Q), (-P, Q), ... : an even k cancels; odd k: the last pair has P = infinity
            g1, g2, _, _ = synthetic.random_pairs(e, n * h, seed=1000 + k)
            neg = configs.negate_g1(e, g1)
            P = np.stack([g1, neg], axis=1).reshape(n, 2 * h, 12)[:, :k].reshape(n * k, 12)
            Q = np.repeat(g2, 2, axis=0).reshape(n, 2 * h, 24)[:, :k].reshape(n * k, 24)
            b1 = np.frombuffer(e.encode_points(P, 1), dtype=np.uint8).reshape(n * k, 96).copy()
            b2 = np.frombuffer(e.encode_points(Q, 2), dtype=np.uint8).reshape(n * k, 192).copy()
            if k % 2:                                                  # the unpaired last point of a check: the infinity encoding
                b1[k - 1::k] = 0
                b1[k - 1::k, 0] = 0x40
            bad_checks = rng.random(n) < 0.33
            bad_checks[[0, n - 1]] = [True, False]
            which = rng.integers(0, k, n)
            idx = np.flatnonzero(bad_checks) * k + which[bad_checks]
            b1[idx, 95] ^= 1                                           # off the curve (an infinity encoding becomes malformed: status 2)
            t1, t2 = torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)
            st1 = torch.empty(n * k, dtype=torch.uint8, device=dev)
            st2 = torch.empty_like(st1)
            ok = torch.empty(n, dtype=torch.uint8, device=dev)
            flag = torch.empty(1, dtype=torch.int32, device=dev)
            e.points_check(t1, t2, k, st1, st2, ok, flag)
            torch.cuda.synchronize()
            s1 = st1.cpu().numpy()
            assert (s1[idx] != 0).all() and int((s1 != 0).sum()) == len(idx) and not st2.any(), k
            assert np.array_equal(ok.cpu().numpy(), (~bad_checks).astype(np.uint8)) and int(flag.item()) == 0, k
            hs1, hs2, hok, hall = e.points_check(b1, b2, k)            # the host flavour: the same bytes
            assert np.array_equal(hs1, s1) and np.array_equal(hok, ok.cpu().numpy()) and not hall
            good = np.flatnonzero(~bad_checks)
            sel = (good[:, None] * k + np.arange(k)[None, :]).reshape(-1)
            _, _, okg, allg = e.points_check(b1[sel], b2[sel], k)      # every check valid: the count equals the batch
            assert okg.all() and allg, k
            badc = np.flatnonzero(bad_checks)
            sel = (badc[:, None] * k + np.arange(k)[None, :]).reshape(-1)
            _, _, okb, allb = e.points_check(b1[sel], b2[sel], k)      # no check valid: the count is zero, every pairing launch is empty
            assert not okb.any() and not allb, k
    finally:
        e.close()


def test_points_check_is_asynchronous_and_capturable(eng):
    """round 6: zkp_points_check_batch_dev no longer reads anything back - the call can be captured into a hipGraph
    (a host synchronisation or an allocation inside the capture would fail it) and the replayed graph gives the bytes of the plain call,
    also after the inputs behind the captured pointers have changed"""
    import torch
    from zkvm_pairings_amd import synthetic
    dev = torch.device("cuda", 0)
    n = 4096
    g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=77, device_tensors=True)
    b1, b2 = eng.encode_points_dev(g1, 1), eng.encode_points_dev(g2, 2)
    b1[3::5, 95] ^= 1
    st1 = torch.empty(n, dtype=torch.uint8, device=dev)
    st2 = torch.empty_like(st1)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    eng.points_check(b1, b2, 1, st1, st2, ok, flag)            # plain call: the workspaces reach their size
    torch.cuda.synchronize()
    want = (st1.clone(), st2.clone(), ok.clone(), flag.clone())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.points_check(b1, b2, 1, st1, st2, ok, flag)
    for t in (st1, st2, ok):
        t.fill_(9)
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(want, (st1, st2, ok, flag)))
    b1[3::5, 95] ^= 1                                           # every point valid now: the replay sees the new bytes
    graph.replay()
    torch.cuda.synchronize()
    assert not st1.any() and not st2.any() and int(flag.item()) == 0 and not ok.any()      # random pairs: valid, never the identity
    # and the context still serves ordinary calls on other streams afterwards
    eng.points_check(b1, b2, 1, st1, st2, ok, flag)
    torch.cuda.synchronize()
    assert not st1.any() and int(flag.item()) == 0


def test_pairing_call_captured_into_a_hipgraph_equals_the_plain_call(eng):
    """the launch sequence of a small batch captured into a hipGraph and replayed gives the plain path's bytes at
    n = 1, 5, 320, 4096 (k = 1) and n = 1, 320 (k = 3).  (Whether replaying it is FASTER is bench.py's business - batch_sweep.rows[].hipgraph:
    launch gaps are 0.7 % of a single pairing, profiles/r06/v62_trace_n1.txt.)"""
    import torch
    from zkvm_pairings_amd import synthetic
    dev = torch.device("cuda", 0)
    g1, g2, _, _ = synthetic.random_pairs(eng, 4096, seed=4242, device_tensors=True)
    for k, sizes in ((1, (1, 5, 320, 4096)), (3, (1, 320))):
        for n in sizes:
            a, b = g1[: n * k], g2[: n * k]
            gt = torch.empty((n, 72), dtype=torch.int64, device=dev)
            ok = torch.empty(n, dtype=torch.uint8, device=dev)
            flag = torch.empty(1, dtype=torch.int32, device=dev)
            eng.pairing_gt_check(a, b, k, gt, ok, flag)
            torch.cuda.synchronize()
            want = (gt.clone(), ok.clone(), flag.clone())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eng.pairing_gt_check(a, b, k, gt, ok, flag)
            gt.zero_(), ok.fill_(9), flag.fill_(7)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(gt, want[0]) and torch.equal(ok, want[1]) and torch.equal(flag, want[2]), (k, n)
            if n == 1 and k == 1:
                got = gt.cpu().numpy().view(np.uint64)
                assert np.array_equal(got, o.pairing_batch(a.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64), nthreads=1))
            del graph


def test_checks_with_more_than_eight_pairs_share_their_squarings():
    """round 5: a check of 9..16 pairs runs through ONE accumulator (the run-time-k Miller program), more pairs in groups of 16.  The Miller
    values equal those of the rounds-1-4 flow (groups of eight joined by f12mul: ZKP_COOP_NO_STREAM=1) for k = 8 .. 96 - and for groups of
    64 (ZKP_COOP_MAX_STREAM).  (What the shared squarings save is measured by bench.py: secondary_workloads.k9_shared_squarings_ratio.)"""
    import json

    def go(env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_kpairs.py"), str(1 << 18)], capture_output=True, text=True, timeout=900,
                           cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    new, old, wide = go(dict(os.environ)), go(dict(os.environ, ZKP_COOP_NO_STREAM="1")), go(dict(os.environ, ZKP_COOP_MAX_STREAM="64"))
    assert new["streaming"] and not old["streaming"]
    for k in (8, 9, 12, 16, 24, 32, 48, 64, 96):
        assert new["k%d" % k]["sha256"] == old["k%d" % k]["sha256"] == wide["k%d" % k]["sha256"], k
