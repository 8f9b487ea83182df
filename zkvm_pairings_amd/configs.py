"""BASELINE.json configs 3, 4 and 5 as runnable workloads (inputs per SURVEY.md 8d) with the size-independent checks that
cover EVERY element; the callers (tests/test_gpu_configs.py, tools/soak.py) add the oracle comparison of the returned
samples.  Nothing here touches oracle/: expectations are known by construction."""
import hashlib

import numpy as np

from . import synthetic

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = synthetic.R_ORDER


def _limbs(v, n=6):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def _int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def negate_g1(eng, g1):
    """(n,12) host points -> (x, p - y) through the engine's field negation (Fp::neg, reference src/fp.rs:383-405)"""
    out = g1.copy()
    out[:, 6:] = eng.fp_op("neg", np.ascontiguousarray(g1[:, 6:]))
    return out


# ------------------------------------------------------------------------------------------------- config 3
def run_config3(eng, n, sample=1 << 12, prefix=1 << 16, seed=synthetic.SEED):
    import torch
    dev = torch.device("cuda", eng.device)
    r = {}
    g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=seed, device_tensors=True)
    out_gt = torch.empty((n, 72), dtype=torch.int64, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    eng.pairing_gt_check(g1, g2, 1, out_gt, ok, flag)
    r["flags_all_zero"] = not bool(ok.any().item())
    r["all_ok"] = int(flag.item())
    idx = torch.arange(0, n, max(1, n // sample), device=dev)[:sample]
    r["sample_g1"] = g1[idx].cpu().numpy().view(np.uint64)
    r["sample_g2"] = g2[idx].cpu().numpy().view(np.uint64)
    r["sample_gt"] = out_gt[idx].cpu().numpy().view(np.uint64)
    r["sha256_all_gt"] = hashlib.sha256(out_gt.cpu().numpy().tobytes()).hexdigest()
    # the two kernel families on a prefix: every Gt equal
    pre = min(prefix, n)
    r["sha256_prefix_coop"] = hashlib.sha256(out_gt[:pre].cpu().numpy().tobytes()).hexdigest()
    eng.set_kernel("thread")
    try:
        thr = eng.pairing(g1[:pre].contiguous(), g2[:pre].contiguous())
        r["sha256_prefix_thread"] = hashlib.sha256(thr.cpu().numpy().tobytes()).hexdigest()
    finally:
        eng.set_kernel("auto")
    r["families_sha256_equal"] = r["sha256_prefix_coop"] == r["sha256_prefix_thread"]
    del out_gt, thr
    # every infinity flag set -> every pairing is the identity, AND flag true
    ones = torch.ones(n, dtype=torch.uint8, device=dev)
    eng.pairing_gt_check(g1, g2, 1, None, ok, flag, inf1=ones)
    r["infinity_all_one"] = bool(ok.all().item()) and int(flag.item()) == 1
    # e(P,Q) e(-P,Q) == 1 through one shared final exponentiation, for EVERY element
    h1 = g1.cpu().numpy().view(np.uint64)
    neg = torch.from_numpy(negate_g1(eng, h1).view(np.int64)).to(dev)
    G1 = torch.stack([g1, neg], dim=1).reshape(2 * n, 12).contiguous()
    del neg
    G2 = torch.stack([g2, g2], dim=1).reshape(2 * n, 24).contiguous()
    ok2, all2 = eng.pairing_check(G1, G2, 2)
    r["cancel_all_one"] = bool(ok2.all().item())
    r["cancel_all_ok"] = int(all2.item())
    return r


# ------------------------------------------------------------------------------------------------- config 4
def groth16_checks(eng, n_checks, seed=0x4B1D, bad_one_in=1024):
    """3-pair checks with a1 b1 + a2 b2 + a3 b3 = 0 (mod r): a3 = -(a1 b1 + a2 b2) / b3; a seeded 1 / bad_one_in of the
    checks gets a3 + 1 instead.  -> (g1 (3n,12), g2 (3n,24), expect (n,) uint8)"""
    a = synthetic.scalars(seed, 3 * n_checks).reshape(n_checks, 3, 4)
    b = synthetic.scalars(seed ^ 0xB5, 3 * n_checks).reshape(n_checks, 3, 4)
    bad = (synthetic.splitmix64(seed ^ 0xBAD, n_checks) % np.uint64(bad_one_in)) == 0 if bad_one_in else np.zeros(n_checks, dtype=bool)
    sh = [64 * i for i in range(4)]
    ai = [[sum(int(x) << s for x, s in zip(row, sh)) for row in a[:, j]] for j in range(2)]
    bi = [[sum(int(x) << s for x, s in zip(row, sh)) for row in b[:, j]] for j in range(3)]
    for c in range(n_checks):
        a3 = (-(ai[0][c] * bi[0][c] + ai[1][c] * bi[1][c]) * pow(bi[2][c], -1, R)) % R
        if bad[c]:
            a3 = (a3 + 1) % R
        a[c, 2] = [(a3 >> s) & 0xFFFFFFFFFFFFFFFF for s in sh]
    g1, _ = eng.g1_mul(synthetic.G1_GENERATOR, a.reshape(-1, 4))
    g2, _ = eng.g2_mul(synthetic.G2_GENERATOR, b.reshape(-1, 4))
    return g1, g2, (~bad).astype(np.uint8)


def run_config4(eng, n_checks, seed=0x4B1D):
    r = {}
    g1, g2, expect = groth16_checks(eng, n_checks, seed)
    ok, allok = eng.pairing_check(g1, g2, 3)
    r["n_bad"] = int((expect == 0).sum())
    r["flags_equal_expectation"] = bool(np.array_equal(ok, expect))
    r["all_ok"] = int(allok)
    first_bad = int(np.flatnonzero(expect == 0)[0]) if r["n_bad"] else n_checks
    m = min(first_bad, 4096)
    r["good_prefix_all_ok"] = int(eng.pairing_check(g1[:3 * m], g2[:3 * m], 3)[1]) if m else 1
    sel = np.concatenate([np.arange(min(48, n_checks)), np.flatnonzero(expect == 0)[:16]])[:64]
    if sel.size < 64:
        sel = np.concatenate([sel, np.arange(48, 48 + 64 - sel.size)])
    pick = (3 * sel[:, None] + np.arange(3)[None, :]).reshape(-1)
    r["sample_g1"], r["sample_g2"] = g1[pick], g2[pick]
    r["sample_ok"], r["sample_expect"] = ok[sel], expect[sel]
    r["sha256_flags"] = hashlib.sha256(ok.tobytes()).hexdigest()
    return r


# ------------------------------------------------------------------------------------------------- config 5
def _fp_sqrt(a):
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def _fp2_sqrt(a):
    a0, a1 = a
    if a1 == 0:
        s = _fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = _fp_sqrt((-a0) % P)
        return (0, s) if s is not None else None
    n = _fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    half = pow(2, -1, P)
    for cand in ((a0 + n) * half % P, (a0 - n) * half % P):
        x0 = _fp_sqrt(cand)
        if x0:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if ((x0 * x0 - x1 * x1) % P, 2 * x0 * x1 % P) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def curve_point_outside_subgroup(which):
    """one point of E(Fp): y^2 = x^3 + 4 (which = 1) or E'(Fp2): y^2 = x^3 + 4 (1 + u) (which = 2) found by trial x; a random
    curve point lies in the prime-order subgroup with probability 1 / cofactor"""
    x = 1
    while True:
        x += 1
        if which == 1:
            y = _fp_sqrt((x * x * x + 4) % P)
            if y is not None:
                return np.concatenate([_limbs(x), _limbs(y)])
        else:
            X = (x, 1)
            x2 = ((X[0] * X[0] - X[1] * X[1]) % P, 2 * X[0] * X[1] % P)
            x3 = ((x2[0] * X[0] - x2[1] * X[1]) % P, (x2[0] * X[1] + x2[1] * X[0]) % P)
            y = _fp2_sqrt(((x3[0] + 4) % P, (x3[1] + 4) % P))
            if y is not None:
                return np.concatenate([_limbs(X[0]), _limbs(X[1]), _limbs(y[0]), _limbs(y[1])])


def to_bytes(pts, which):
    """(n, 12 | 24) canonical limbs -> (n, 96 | 192) uncompressed big-endian bytes (G2: c1 before c0)"""
    n = pts.shape[0]
    fe = pts.reshape(n, -1, 6)
    if which == 2:
        fe = fe[:, [1, 0, 3, 2], :]
    return np.ascontiguousarray(fe[:, :, ::-1]).byteswap().view(np.uint8).reshape(n, -1)


CLASSES = ("valid", "off_curve", "wrong_subgroup", "non_canonical", "infinity", "bad_infinity", "compressed_flag")
# (decode status, decoded infinity, is_valid status) per class; a point that fails to decode comes out as (0, 0): not on the curve
EXPECT = {"valid": (0, 0, 0), "off_curve": (0, 0, 1), "wrong_subgroup": (0, 0, 2), "non_canonical": (1, 0, 1), "infinity": (0, 1, 0),
          "bad_infinity": (2, 0, 1), "compressed_flag": (2, 0, 1)}


def raw_points(eng, n, which, seed):
    """n uncompressed byte strings with seeded fractions of every class -> (bytes (n, 96 | 192), class index (n,))"""
    sel = (synthetic.splitmix64(seed ^ 0xC1A55, n) % np.uint64(1024)).astype(np.int64)
    cls = np.zeros(n, dtype=np.int64)
    for lo, hi, c in ((0, 8, 1), (8, 12, 2), (12, 16, 3), (16, 20, 4), (20, 22, 5), (22, 24, 6)):
        cls[(sel >= lo) & (sel < hi)] = c
    k = synthetic.scalars(seed, n)
    gen = synthetic.G1_GENERATOR if which == 1 else synthetic.G2_GENERATOR
    pts, _ = (eng.g1_mul if which == 1 else eng.g2_mul)(gen, k)
    wrong = np.flatnonzero(cls == 2)
    if wrong.size:
        t = curve_point_outside_subgroup(which)
        w, winf = (eng.g1_mul if which == 1 else eng.g2_mul)(t, synthetic.scalars(seed ^ 0x5B, wrong.size))
        assert not winf.any()
        pts[wrong] = w
    ncol = 6 if which == 1 else 12
    off = np.flatnonzero(cls == 1)
    if off.size:        # y + 1 (first y coordinate): leaves the curve
        y = eng.fp_op("add", np.ascontiguousarray(pts[off, ncol:ncol + 6]), np.tile(_limbs(1), (off.size, 1)))
        pts[off, ncol:ncol + 6] = y
    raw = to_bytes(pts, which).copy()
    nc = np.flatnonzero(cls == 3)
    if nc.size:         # first encoded coordinate := p + (low 16 bits): >= p, < 2^381 (no flag bit)
        low = pts[nc, 0] & np.uint64(0xFFFF)
        for j, i in enumerate(nc):
            raw[i, :48] = np.frombuffer((P + int(low[j])).to_bytes(48, "big"), dtype=np.uint8)
    width = raw.shape[1]
    raw[cls == 4] = 0
    raw[cls == 4, 0] = 0x40
    raw[cls == 5, 0] |= 0x40          # infinity flag on a finite point's bytes
    raw[cls == 6, 0] |= 0x80
    assert width == (96 if which == 1 else 192)
    return raw, cls


def run_config5(eng, n, seed=0x5EED5):
    r = {}
    decoded, raws = {}, {}
    for which, name in ((1, "g1"), (2, "g2")):
        raw, cls = raw_points(eng, n, which, seed + which)
        pts, inf, st = eng.decode_points(raw, which)
        exp_dec = np.array([EXPECT[c][0] for c in CLASSES], dtype=np.uint8)[cls]
        exp_inf = np.array([EXPECT[c][1] for c in CLASSES], dtype=np.uint8)[cls]
        exp_val = np.array([EXPECT[c][2] for c in CLASSES], dtype=np.uint8)[cls]
        r[name + "_decode_status_equal"] = bool(np.array_equal(st, exp_dec) and np.array_equal(inf, exp_inf))
        val = (eng.g1_is_valid if which == 1 else eng.g2_is_valid)(pts, inf)
        r[name + "_valid_status_equal"] = bool(np.array_equal(val, exp_val))
        r[name + "_class_counts"] = {CLASSES[c]: int((cls == c).sum()) for c in range(len(CLASSES))}
        pick = np.concatenate([np.flatnonzero(cls == c)[:96] for c in range(len(CLASSES))] + [np.arange(0, n, max(1, n // 3400))])[:4096]
        r[name + "_sample"] = (pts[pick], inf[pick], val[pick])
        r[name + "_sha256_status"] = hashlib.sha256(val.tobytes()).hexdigest()
        decoded[which] = (pts, inf, val, st, cls)
        raws[which] = raw
    # ---- the same through ONE call (round 4, zkp_points_check_batch): check c = the two pairs (P_c, Q_c), (-P_c, Q_c) given as raw
    # bytes.  Expected by construction: the status byte of every point (decode status, else is_valid status + 2) and
    # ok[c] = both points of c valid (an infinity pairs to the identity), for EVERY check.
    comb = lambda st, val: np.where(st != 0, st, np.where(val != 0, val + 2, 0)).astype(np.uint8)
    pts1, inf1, val1, dec1, cls1 = decoded[1]
    pts2, inf2, val2, dec2, cls2 = decoded[2]
    raw1 = raws[1]
    neg1 = raw1.copy()
    dec_ok = (dec1 == 0) & (inf1 == 0)
    neg1[dec_ok] = to_bytes(negate_g1(eng, np.ascontiguousarray(pts1[dec_ok])), 1)
    B1 = np.stack([raw1, neg1], axis=1).reshape(2 * n, 96)
    B2 = np.repeat(raws[2], 2, axis=0)
    want1, want2 = np.repeat(comb(dec1, val1), 2), np.repeat(comb(dec2, val2), 2)
    want_ok = ((comb(dec1, val1) == 0) & (comb(dec2, val2) == 0)).astype(np.uint8)
    s1, s2, okb, allok = eng.points_check(B1, B2, 2)
    r["one_call_status_equal"] = bool(np.array_equal(s1, want1) and np.array_equal(s2, want2))
    r["one_call_ok_equal"] = bool(np.array_equal(okb, want_ok))
    r["one_call_all_ok"] = int(allok)
    r["one_call_n_ok"] = int(want_ok.sum())
    r["one_call_sha256"] = hashlib.sha256(s1.tobytes() + s2.tobytes() + okb.tobytes()).hexdigest()
    # device-resident flavour on the same bytes: identical bytes out; and a batch of good checks only: AND flag true
    import torch
    dev = torch.device("cuda", eng.device)
    t1, t2 = torch.from_numpy(B1).to(dev), torch.from_numpy(B2).to(dev)
    d1 = torch.empty(2 * n, dtype=torch.uint8, device=dev)
    d2 = torch.empty(2 * n, dtype=torch.uint8, device=dev)
    dok = torch.empty(n, dtype=torch.uint8, device=dev)
    dflag = torch.empty(1, dtype=torch.int32, device=dev)
    eng.points_check(t1, t2, 2, d1, d2, dok, dflag)
    r["one_call_dev_equal"] = bool(np.array_equal(d1.cpu().numpy(), s1) and np.array_equal(d2.cpu().numpy(), s2) and
                                   np.array_equal(dok.cpu().numpy(), okb) and int(dflag.item()) == int(allok))
    goodc = np.flatnonzero(want_ok)[: 1 << 14]
    sel = (2 * goodc[:, None] + np.arange(2)[None, :]).reshape(-1)
    _, _, okg, allg = eng.points_check(B1[sel], B2[sel], 2)
    r["one_call_good_subset_all_ok"] = bool(okg.all() and allg)
    del t1, t2, d1, d2, dok
    # the pairing leg on points that passed both checks: e(P,Q) e(-P,Q) == 1 on a 2^16 subset
    good = np.flatnonzero((decoded[1][4] == 0) & (decoded[2][4] == 0))[: 1 << 16]
    p1, p2 = decoded[1][0][good], decoded[2][0][good]
    G1 = np.stack([p1, negate_g1(eng, p1)], axis=1).reshape(-1, 12)
    G2 = np.stack([p2, p2], axis=1).reshape(-1, 24)
    ok, allok = eng.pairing_check(G1, G2, 2)
    r["pairing_checks_all_one"] = bool(ok.all() and allok)
    return r
