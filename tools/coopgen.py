#!/usr/bin/env python3
"""Microcode generator + bit-accurate emulator for the lane-cooperative kernel family.

One "check" (k pairs, one shared Fp12 accumulator) is processed by a GROUP of 12 lanes of a
wavefront; lane j of the group owns output coefficient j of the Fp12 value being produced
(tower order c0.c0.c0, c0.c0.c1, ..., c1.c2.c1 == reference src/fp12.rs:13-16).  All Fp values of a
group live in LDS slots.  A program is a list of steps; every step is executed by all lanes in
lockstep with per-lane operands taken from tables:

  MULACC  r = sum_t (+-)(A1 +- A2) * (B1 +- B2)  -> Montgomery reduce -> dst = alpha*r + beta*E
          (A*, B*, E are slots; one reduction per output coefficient = lazy reduction)
  LIN     dst = sum_t coef_t * slot_t  (|coef| <= 8), then one-pass weak limb normalisation
  GLOAD   dst = global[...]   (line coefficients stream / state buffer / wire format)
  GSTORE  global[...] = slot  (state buffer / wire format, optional Gt==identity compare)
  LOOP n / ENDLOOP            (single level)

This file (1) builds the programs from the tower formulas (symbolically: linear forms over slots,
bilinear term lists, ksi = 1+u mixing), (2) emulates them with exactly the integer arithmetic the
HIP interpreter performs (14 signed 28-bit limbs, 64-bit columns, R = 2^392), asserting that no
column leaves 63 bits and no limb leaves 31 bits, and (3) writes zkvm_pairings_amd/csrc/zkp_coop_prog.inc.
The emulator is checked against tests/golden/bls12_381_model.py in tests/test_coopgen.py.
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bls12_381_model as M  # noqa: E402

P = M.P
NL, W = 14, 28
RADIX = 1 << (NL * W)
RMOD = RADIX % P
RINV = pow(RADIX, -1, P)
PINV28 = (-pow(P, -1, 1 << W)) % (1 << W)
P_LIMBS = [(P >> (W * i)) & ((1 << W) - 1) for i in range(NL)]
def _balanced(v):
    out, carry = [], 0
    for i in range(NL):
        d = ((v >> (W * i)) & ((1 << W) - 1)) + carry
        if i < NL - 1 and d >= (1 << (W - 1)):
            d -= 1 << W
            carry = 1
        else:
            carry = 0
        out.append(d)
    return out


P_BAL = _balanced(P)                      # balanced limbs of p (|limb| <= 2^27), used by the value renormalisation
VRED_SHIFT_IN, VRED_SHIFT_OUT = 9, 24
VRED_C = round((1 << (VRED_SHIFT_IN + VRED_SHIFT_OUT)) / P_BAL[NL - 1])
LDS_SLOTS = 24         # slots every program must fit: 16 wavefronts of 5 groups share a CU's 160 KB of LDS
# ... or, for programs that reference at most LDS_WIDE_CONSTS constants, LDS_WIDE_SLOTS slots: the constants region
# (64 B per constant per wavefront) shrinks by as much as the five groups' extra slots take - same 10,096 B per wavefront
LDS_WIDE_SLOTS, LDS_WIDE_CONSTS = 30, 4
# ... or "deep": 36 slots + the first 24 constants, 13,056 B per wavefront - twelve wavefronts (the register bound of three per SIMD) still
# fit the 160 KB of a CU.  The hard part's programs: two Fp12 values + the companion sums / differences of one of them.
LDS_DEEP_SLOTS, LDS_DEEP_CONSTS = 36, 24
G = 12                 # lanes per group
NSLOT = 64             # group-local slots are 0..NSLOT-1; constants are NSLOT..127
CONST_BASE = 64

OP_END, OP_MULACC, OP_LIN, OP_GLOAD, OP_GSTORE, OP_LOOP, OP_ENDLOOP = 0, 1, 2, 3, 4, 5, 6
# round 5: a loop over the PAIRS of a check (count = the launch's k, a run-time value): the body sees the pair number in its line loads
OP_PLOOP, OP_PENDLOOP = 7, 8
PAIR_VAR = -1          # pair index of a K_LINE load inside a pair loop: the loop's counter
# GLOAD / GSTORE address kinds
K_LINE, K_STATE, K_WIRE, K_WIRE2 = 0, 1, 2, 3   # K_WIRE2: wire record of check + chk_off (tree reductions)


# ------------------------------------------------------------------------------------------ constants region
def _const_table():
    """name -> true field value (NOT Montgomery); index in the table = slot - CONST_BASE."""
    g = [M.f2_pow(M.XI, i * (P - 1) // 6) for i in range(6)]
    t = [("ZERO", 0), ("ONE", 1), ("RAW_R2", None), ("RAW_ONE", None)]
    # Frobenius^j coefficients gamma_{j,i} = gamma_i^(1 + p + ... + p^(j-1)) for w^i, j = 1, 2, 3
    for j in (1, 2, 3):
        for i in range(1, 6):
            c = M.F2_ONE
            for e in range(j):
                x = g[i]
                for _ in range(e):
                    x = M.f2_conj(x)  # Frobenius on Fp2 = conjugation
                c = M.f2_mul(c, x)
            t.append(("F%d_%d_0" % (j, i), c[0]))
            t.append(("F%d_%d_1" % (j, i), c[1]))
    return t


CONSTS = _const_table()
# RAW_R2 / RAW_ONE are "raw" constants for Montgomery conversion: their LIMBS hold R^2 mod p resp. 1
CONST_SLOT = {name: CONST_BASE + i for i, (name, _) in enumerate(CONSTS)}
N_CONST = len(CONSTS)
assert CONST_BASE + N_CONST <= 128
ZERO = CONST_SLOT["ZERO"]


# ------------------------------------------------------------------------------------------ symbolic layer
class Lin(dict):
    """linear form over slots: {slot: int coef}"""

    @staticmethod
    def of(slot, c=1):
        return Lin({slot: c})

    def __add__(self, o):
        r = Lin(self)
        for k, v in o.items():
            r[k] = r.get(k, 0) + v
            if r[k] == 0:
                del r[k]
        return r

    def __neg__(self):
        return Lin({k: -v for k, v in self.items()})

    def __sub__(self, o):
        return self + (-o)

    def scale(self, c):
        return Lin({k: v * c for k, v in self.items()}) if c else Lin()

    def key(self):
        return tuple(sorted(self.items()))


class Bil(list):
    """bilinear value: list of (LinA, LinB, coef)"""

    def __add__(self, o):
        return Bil(list(self) + list(o))

    def __neg__(self):
        return Bil([(a, b, -c) for a, b, c in self])

    def __sub__(self, o):
        return self + (-o)

    def scale(self, k):
        return Bil([(a, b, c * k) for a, b, c in self])


def lin2(s0, s1):
    return (Lin.of(s0), Lin.of(s1))


def l2_add(x, y):
    return (x[0] + y[0], x[1] + y[1])


def l2_sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def l2_neg(x):
    return (-x[0], -x[1])


def l2_xi(x):
    return (x[0] - x[1], x[0] + x[1])


def l2_conj(x):
    return (x[0], -x[1])


def b2_mul(x, y):
    """(x0 + x1 u)(y0 + y1 u) with Lin components -> (Bil, Bil)"""
    return (Bil([(x[0], y[0], 1), (x[1], y[1], -1)]), Bil([(x[0], y[1], 1), (x[1], y[0], 1)]))


def b2_sqr(x):
    """(x0 + x1 u)^2 = (x0 + x1)(x0 - x1) + (2 x0) x1 u : 2 products"""
    return (Bil([(x[0] + x[1], x[0] - x[1], 1)]), Bil([(x[0].scale(2), x[1], 1)]))


def b2_add(x, y):
    return (x[0] + y[0], x[1] + y[1])


def b2_sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def b2_xi(x):
    return (x[0] - x[1], x[0] + x[1])


def b2_scale(x, k):
    return (x[0].scale(k), x[1].scale(k))


def b6_mul(a, b):
    """a, b: 3-tuples of Lin-pairs -> 3-tuple of Bil-pairs (schoolbook over Fp2, v^3 = xi)"""
    m = [[b2_mul(a[i], b[j]) for j in range(3)] for i in range(3)]
    c0 = b2_add(m[0][0], b2_xi(b2_add(m[1][2], m[2][1])))
    c1 = b2_add(b2_add(m[0][1], m[1][0]), b2_xi(m[2][2]))
    c2 = b2_add(b2_add(m[0][2], m[1][1]), m[2][0])
    return (c0, c1, c2)


def b6_add(x, y):
    return tuple(b2_add(p, q) for p, q in zip(x, y))


def b6_mul_v(x):
    return (b2_xi(x[2]), x[0], x[1])


def l6_add(x, y):
    return tuple(l2_add(p, q) for p, q in zip(x, y))


def l6_mul_v(x):
    return (l2_xi(x[2]), x[0], x[1])


def l6_neg(x):
    return tuple(l2_neg(p) for p in x)


class V12:
    """an Fp12 value resident in 12 consecutive... (not necessarily) slots, with a sign per coefficient
    (conjugation and negation are free views)."""

    def __init__(self, slots, signs=None, sd=None):
        self.slots = list(slots)
        self.signs = list(signs) if signs else [1] * 12
        # optional companion slots kept by runs of cyclotomic squarings: sd[2j] holds x0 + x1 and sd[2j+1] holds
        # x0 - x1 of the j-th Fp2 coefficient (x0, x1) = slots (2j, 2j+1); only valid while every sign is +
        self.sd = list(sd) if sd else None

    def lin(self, i):
        return Lin.of(self.slots[i], self.signs[i])

    def fp2(self, j):  # j-th Fp2 coefficient in tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2)
        return (self.lin(2 * j), self.lin(2 * j + 1))

    def fp6(self, h):
        return tuple(self.fp2(3 * h + j) for j in range(3))

    def conj(self):
        return V12(self.slots, [s if i < 6 else -s for i, s in enumerate(self.signs)])


def flatten12(c0, c1):
    """two Fp6 (3-tuples of pairs) -> list of 12"""
    out = []
    for h in (c0, c1):
        for p in h:
            out += [p[0], p[1]]
    return out


def merge_terms(bil):
    """group by A form (summing B), then drop zero terms"""
    by_a = {}
    for a, b, c in bil:
        if not a or not b or c == 0:
            continue
        ka = a.key()
        # normalise the sign of A so that A and -A merge
        first = ka[0][1]
        if first < 0:
            a, c = -a, -c
            ka = a.key()
        if ka in by_a:
            by_a[ka] = (a, by_a[ka][1] + b.scale(c))
        else:
            by_a[ka] = (a, b.scale(c))
    out = []
    for a, b in by_a.values():
        if b:
            out.append((a, b))
    # second pass: merge equal B forms (sum the A's)
    by_b = {}
    for a, b in out:
        kb = b.key()
        sgn = 1
        if kb[0][1] < 0:
            b, sgn = -b, -1
            kb = b.key()
        if kb in by_b:
            by_b[kb] = (by_b[kb][0] + a.scale(sgn), b)
        else:
            by_b[kb] = (a.scale(sgn), b)
    return [(a, b) for a, b in by_b.values() if a and b]


# round 5: ZKP_GEN_SHARE_FETCHES=1 regenerates the programs with the term order that shares operand fetches (measured: no gain; default off)
SHARE_FETCHES = int(os.environ.get("ZKP_GEN_SHARE_FETCHES", "0"))


def fetch_records(lanes):
    """LDS records the twelve lanes of a check fetch in one step, a record counted once per term position however many lanes read it
    (lanes that read one address are served by one bank access): what the operand fetches cost in LDS bank reads"""
    T = max(len(l["terms"]) for l in lanes)
    n = 0
    for t in range(T):
        recs = set()
        for l in lanes:
            if t < len(l["terms"]):
                a1, a2, _, b1, b2, _, _, _ = l["terms"][t]
                recs.update((("a", a1), ("b", b1)))
                if a2 != ZERO:
                    recs.add(("a", a2))
                if b2 != ZERO:
                    recs.add(("b", b2))
        n += len(recs)
    return n


_SHARE_CACHE = {}


def share_operand_fetches(lanes, iters=12000):
    """Reorder every lane's terms - inside the classes the sort above made (negated / second-operand terms keep their positions'
    class, so the step's wave-uniform term flags do not change) - so that at each term position the lanes of a check read as FEW
    distinct LDS records as possible: the sum of a step's products does not depend on their order, but what the operand fetches
    cost (power: DESIGN section 4, round 5) does - lanes that read one address share one bank access.  The typical gain: the two
    lanes of an Fp2 coefficient take a b from the SAME record at the same position (a_re b_re beside a_im b_re, then -a_im b_im beside
    a_re b_im) instead of each its own.  Deterministic hill climbing over swaps of two terms of a lane (seeded by nothing but the
    step: the same step always gets the same order; identical steps of different programs are solved once)."""
    import random
    key = lambda x: (bool(x[6]), x[1] != ZERO, x[4] != ZERO)
    sig = tuple(tuple(l["terms"]) for l in lanes)
    if sig in _SHARE_CACHE:
        for l, terms in zip(lanes, _SHARE_CACHE[sig]):
            l["terms"] = list(terms)
        return fetch_records(lanes)

    def recs_of(term):
        a1, a2, _, b1, b2, _, _, _ = term
        r = {("a", a1), ("b", b1)}
        if a2 != ZERO:
            r.add(("a", a2))
        if b2 != ZERO:
            r.add(("b", b2))
        return r

    T = max(len(l["terms"]) for l in lanes)
    cnt = [collections.Counter() for _ in range(T)]
    for l in lanes:
        for t, tm in enumerate(l["terms"]):
            cnt[t].update(recs_of(tm))
    cand = []
    for li, l in enumerate(lanes):
        cl = collections.defaultdict(list)
        for t, tm in enumerate(l["terms"]):
            cl[key(tm)].append(t)
        cand += [(li, pos) for pos in cl.values() if len(pos) > 1]
    rng = random.Random(0x5EED)
    for _ in range(iters if cand else 0):
        li, pos = cand[rng.randrange(len(cand))]
        t1, t2 = rng.sample(pos, 2)
        terms = lanes[li]["terms"]
        r1, r2 = recs_of(terms[t1]), recs_of(terms[t2])
        out1, out2 = r1 - r2, r2 - r1          # records that leave position t1 / t2 (and arrive at the other)
        d = 0
        for r in out1:
            d += (cnt[t2][r] == 0) - (cnt[t1][r] == 1)
        for r in out2:
            d += (cnt[t1][r] == 0) - (cnt[t2][r] == 1)
        if d <= 0:
            for r in out1:
                cnt[t1][r] -= 1
                cnt[t2][r] += 1
            for r in out2:
                cnt[t2][r] -= 1
                cnt[t1][r] += 1
            terms[t1], terms[t2] = terms[t2], terms[t1]
    _SHARE_CACHE[sig] = tuple(tuple(l["terms"]) for l in lanes)
    return fetch_records(lanes)


def encode_form(f):
    """Lin with <=2 slots and coefficients in {+-1} (or one slot with +-2) -> (s1, s2, sub, neg[, doubled]) or None"""
    items = sorted(f.items())
    if len(items) == 1:
        s, c = items[0]
        if c in (1, -1):
            return (s, ZERO, 0, c < 0)
        if c in (2, -2):
            return (s, ZERO, 0, c < 0, True)      # 2 x: the operand is doubled by a shift (A side only)
        return None
    if len(items) == 2:
        (s1, c1), (s2, c2) = items
        if abs(c1) != 1 or abs(c2) != 1:
            return None
        if c1 > 0 and c2 > 0:
            return (s1, s2, 0, False)
        if c1 > 0 and c2 < 0:
            return (s1, s2, 1, False)
        if c1 < 0 and c2 > 0:
            return (s2, s1, 1, False)
        return (s1, s2, 0, True)
    return None


# ------------------------------------------------------------------------------------------ program builder
class Seg:
    """one step program of a multi-kernel plan (what encode() and the emulator need of a Builder)"""

    def __init__(self, steps, peak):
        self.steps, self.peak = steps, peak


class Builder:
    K_REDUCED = 1.05      # |value| / p after a Montgomery reduction of a budget-respecting accumulation
    K_VRED = 0.51         # after the value renormalisation
    K_INPUT = 1.1         # anything a program hands to another through the state buffer (asserted at the store;
                          # spills inside a program keep their own bound)
    K_MAX = 8.0           # LIN results above this get the renormalisation

    def __init__(self):
        self.steps = []
        self.free = list(range(NSLOT - 1, -1, -1))
        self.peak = 0
        self.K = {}           # static worst-case |value|/p per group-local slot
        self.L = {}           # limb bound of a slot in units of 2^27 (1 unless it holds a stored sum of two values)
        self.plan = []        # finished segments of a multi-kernel program: ("prog", Seg) / kernel steps (see cut)

    def kof(self, slot):
        return 1.0 if slot >= CONST_BASE else self.K.get(slot, self.K_INPUT)

    def lof(self, slot):
        return 1 if slot >= CONST_BASE else self.L.get(slot, 1)

    # ---- slots
    def alloc(self, n=1):
        assert len(self.free) >= n, "out of LDS slots"
        r = [self.free.pop() for _ in range(n)]
        self.peak = max(self.peak, NSLOT - len(self.free))
        return r

    def release(self, slots):
        for s in slots:
            assert s < NSLOT and s not in self.free
            self.free.append(s)

    def alloc12(self):
        return V12(self.alloc(12))

    def cut(self, kernel_steps):
        """end the current step program (every LDS slot must be free: nothing survives between launches but the
        per-check state buffer), queue the given non-interpreter kernel steps, start a new step program"""
        assert len(self.free) == NSLOT, "values still LDS-resident at a program boundary"
        self.plan.append(("prog", Seg(self.steps, self.peak)))
        self.plan += list(kernel_steps)
        self.steps, self.peak, self.K, self.L = [], 0, {}, {}

    def finish_plan(self):
        self.plan.append(("prog", Seg(self.steps, self.peak)))
        return self.plan

    # ---- raw steps
    def lin(self, lanes):
        """lanes: list (<=12) of (dst, Lin) ; coefficient magnitudes <= 8, <=4 terms"""
        assert 0 < len(lanes) <= G
        nt = max(len(f) for _, f in lanes)
        assert 1 <= nt <= 4, nt
        ent = []
        kout = []
        for dst, f in lanes:
            terms = sorted(f.items())
            assert all(-8 <= c <= 8 and c != 0 for _, c in terms)
            ent.append((dst, terms))
            kout.append(sum(abs(c) * self.kof(s) for s, c in terms))
        vred_needed = max(kout) > self.K_MAX
        assert max(kout) < 14, "LIN result too large even for the renormalisation"
        assert all(self.lof(s) == 1 for _, terms in ent for s, _ in terms)
        for (dst, _), k in zip(ent, kout):
            self.L[dst] = 1
            self.K[dst] = self.K_VRED if vred_needed else k
        self.steps.append({"op": OP_LIN, "nt": nt, "lanes": ent, "vred": vred_needed})

    def mulacc(self, outs, subst=None, sd=None):
        """outs: list (<=12) of dict(dst=slot, bil=Bil or list of (Lin,Lin) merged, alpha=1, beta=0, e=ZERO).
        Forms that do not fit the on-the-fly encoding are materialised by LIN pre-steps.
        subst: {form key: Lin} - forms that already exist as stored values (companion sums/differences);
        sd: per lane a companion slot (or None): lane 2j also stores r_2j + r_2j+1, lane 2j+1 stores r_2j - r_2j+1."""
        assert 0 < len(outs) <= G
        merged = []
        for o in outs:
            terms = merge_terms(o["bil"])
            merged.append(terms)
        # materialise awkward forms
        temps = {}
        pre = []

        def fit(f):
            if subst:
                k = f.key()
                if k in subst:
                    return subst[k]
                k = (-f).key()
                if k in subst:
                    return -subst[k]
            e = encode_form(f)
            if e is not None:
                return f
            k = f.key()
            sgn = 1
            if k[0][1] < 0:
                k = (-f).key()
                sgn = -1
            if k not in temps:
                t = self.alloc(1)[0]
                temps[k] = t
                pre.append((t, f.scale(sgn)))
            return Lin.of(temps[k], sgn)

        fitted = [[(fit(a), fit(b)) for a, b in terms] for terms in merged]
        for i in range(0, len(pre), G):
            self.lin(pre[i:i + G])
        lanes = []
        for o, terms in zip(outs, fitted):
            enc = []
            for a, b in terms:
                ea, eb = encode_form(a), encode_form(b)
                assert ea and eb
                da, db = len(ea) > 4, len(eb) > 4
                if db and not da:          # the doubling shift exists on the A side only
                    ea, eb, da, db = eb, ea, True, False
                a1, a2, asub, b1, b2, bsub = ea[0], ea[1], ea[2], eb[0], eb[1], eb[2]
                if db:                     # both doubled: B keeps the two-slot form x + x
                    b2 = b1
                neg = bool(ea[3]) ^ bool(eb[3])
                # a sign is free when a form is a difference: -(x - y) = (y - x)
                if neg and asub:
                    a1, a2, neg = a2, a1, False
                elif neg and bsub:
                    b1, b2, neg = b2, b1, False
                enc.append((a1, a2, asub, b1, b2, bsub, neg, da))
            # order every lane's terms alike (plain terms first, negated / two-slot ones last) so that more term
            # positions are uniformly free of negations and second operands across the 12 lanes
            enc.sort(key=lambda x: (bool(x[6]), x[1] != ZERO, x[4] != ZERO))
            lanes.append({"dst": o["dst"], "terms": enc, "alpha": o.get("alpha", 1), "beta": o.get("beta", 0), "e": o.get("e", ZERO)})
        if SHARE_FETCHES:
            share_operand_fetches(lanes)
        T = max(len(l["terms"]) for l in lanes)
        assert T >= 1
        # static worst case, valid for EVERY input: every LDS-resident value has |limb| <= 2^27 (+16) except raw
        # wire limbs (< 2^28); a two-slot form doubles the bound.  sum_t La*Lb*14*2^54 (+ 14*2^56 from m*p)
        # must stay below 2^63  =>  sum_t La*Lb <= 30.
        for l in lanes:
            budget = 0
            for (a1, a2, asub, b1, b2, bsub, neg, da) in l["terms"]:
                la = (self.lof(a1) + (self.lof(a2) if a2 != ZERO else 0)) * (2 if da else 1)
                lb = self.lof(b1) + (self.lof(b2) if b2 != ZERO else 0)
                budget += la * lb
            assert budget <= 30, "column budget exceeded: %d" % budget
        epi = [(l["alpha"], l["beta"]) != (1, 0) for l in lanes]
        assert all(epi) or not any(epi), "epilogue must be step-uniform"
        # value bounds: sum_t Ka*Kb / 2^11 + 1 must stay small; two-slot forms add their slots' bounds
        for l in lanes:
            tot = 0.0
            for (a1, a2, asub, b1, b2, bsub, neg, da) in l["terms"]:
                ka = (self.kof(a1) + (self.kof(a2) if a2 != ZERO else 0)) * (2 if da else 1)
                kb = self.kof(b1) + (self.kof(b2) if b2 != ZERO else 0)
                tot += ka * kb
            assert tot / 2048 + 1 <= 1.3, "value budget exceeded: %.1f" % tot
            if epi[0]:
                assert abs(l["alpha"]) * 1.3 + abs(l["beta"]) * self.kof(l["e"]) < 14
                # limbs of alpha r + beta e - q p before the single weak normalisation: (|alpha| + |beta| + |q|) 2^27 < 2^31
                assert abs(l["alpha"]) + abs(l["beta"]) + abs(l["alpha"]) * 1.3 + abs(l["beta"]) * self.kof(l["e"]) + 1 < 15
        for l in lanes:
            self.K[l["dst"]] = self.K_VRED if epi[0] else self.K_REDUCED
            self.L[l["dst"]] = 1
        if sd:
            assert len(sd) == len(lanes) == G and all(x is not None for x in sd)
            for i, l in enumerate(lanes):
                self.K[sd[i]] = self.K[lanes[i]["dst"]] + self.K[lanes[i ^ 1]["dst"]]
                self.L[sd[i]] = 2
        self.steps.append({"op": OP_MULACC, "T": T, "lanes": lanes, "epi": bool(epi[0]), "sd": list(sd) if sd else None})
        self.release(list(temps.values()))
        return T

    def gload(self, kind, lanes, advance=0, kbound=None):
        """lanes: list of (dst, index).  K_LINE: index = coefficient (0..5) of pair `pair` at the stream
        cursor; K_STATE: index = state element; K_WIRE: index = Fp index in the wire record."""
        for dst, _ in lanes:
            self.L[dst] = 1
            self.K[dst] = kbound if kbound is not None else (self.K_REDUCED if kind != K_STATE else self.K_INPUT)
        self.steps.append({"op": OP_GLOAD, "kind": kind, "lanes": list(lanes), "advance": advance})

    def gstore(self, kind, lanes, check_identity=False, spill=False):
        if kind == K_STATE and not spill:
            assert all(self.kof(s) <= self.K_INPUT for s, _ in lanes), "inter-program state must be reduced values"
        self.steps.append({"op": OP_GSTORE, "kind": kind, "lanes": list(lanes), "check": check_identity})

    def loop(self, n):
        self.steps.append({"op": OP_LOOP, "n": n})

    def endloop(self):
        self.steps.append({"op": OP_ENDLOOP})

    def ploop(self):
        self.steps.append({"op": OP_PLOOP})

    def pendloop(self, advance=0):
        """end of a pair loop; advance: how far the line-stream cursor moves when the loop has run out"""
        self.steps.append({"op": OP_PENDLOOP, "advance": advance})

    # ---- tower macro-ops (lane j produces coefficient j).  dst is a V12 or a slot list; the result
    # is returned as a fresh V12 over those slots (all signs +).
    @staticmethod
    def _slots(dst):
        return dst.slots if isinstance(dst, V12) else list(dst)

    def fp12_mul(self, dst, a, b, sd_out=None):
        """a b.  merge_terms groups the 144 monomials by a's coefficients, so every two-slot operand form is a sum / difference
        inside ONE Fp2 coefficient of b: when b carries companions (b.sd) every form of the step is a stored value.
        sd_out: companions of the result (may be b.sd's slots: operands are read before anything is stored)."""
        d = self._slots(dst)
        c0 = b6_add(b6_mul(a.fp6(0), b.fp6(0)), b6_mul_v(b6_mul(a.fp6(1), b.fp6(1))))
        c1 = b6_add(b6_mul(a.fp6(0), b.fp6(1)), b6_mul(a.fp6(1), b.fp6(0)))
        self.mulacc([{"dst": d[i], "bil": bl} for i, bl in enumerate(flatten12(c0, c1))], subst=self._sd_subst(b), sd=sd_out)
        return V12(d, sd=sd_out)

    def _sd_subst(self, a):
        """companion slots of a -> substitution table for its sum / difference forms"""
        if not a.sd:
            return None
        assert all(sg == 1 for sg in a.signs)
        subst = {}
        for j in range(6):
            x0, x1 = a.lin(2 * j), a.lin(2 * j + 1)
            subst[(x0 + x1).key()] = Lin.of(a.sd[2 * j])
            subst[(x0 - x1).key()] = Lin.of(a.sd[2 * j + 1])
        return subst

    def fp12_sqr(self, dst, a, sd_out=None):
        """a^2 in ONE accumulation step.  In the basis 1, w, .., w^5 over Fp2 (w^6 = xi; g_j = the Fp2 coefficient of
        w^j) the square is sum_{i<=j} (2 - [i=j]) g_i g_j w^(i+j): every output coefficient sums three or four Fp2
        products (squares as (x0+x1)(x0-x1), 2 x0 x1), i.e. 6-7 Fp products per lane, all operand forms fit the
        on-the-fly encoding.  The reference's complex squaring (src/fp12.rs:173-184) needs 6 products per lane but two
        extra LIN steps in this machine (its operand forms have three slots), ~10 % slower here; same value."""
        d = self._slots(dst)
        tw = [0, 3, 1, 4, 2, 5]        # tower index (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) of g_0 .. g_5
        g = [a.fp2(tw[j]) for j in range(6)]
        flat = [None] * 12
        for k in range(6):
            acc0, acc1 = Bil(), Bil()
            for i in range(6):
                for j in range(i, 6):
                    if (i + j) % 6 != k:
                        continue
                    p = b2_sqr(g[i]) if i == j else b2_mul(b2_scale(g[i], 2), g[j])
                    if i + j >= 6:
                        p = b2_xi(p)
                    acc0, acc1 = acc0 + p[0], acc1 + p[1]
            flat[2 * tw[k]], flat[2 * tw[k] + 1] = acc0, acc1
        self.mulacc([{"dst": d[i], "bil": bl} for i, bl in enumerate(flat)], subst=self._sd_subst(a), sd=sd_out)
        return V12(d, sd=sd_out)

    def fp12_mul_by_014(self, dst, a, l0, l1, l4, sd_out=None):
        """a * ((l0 + l1 v) + (l4 v) w)  (reference src/fp12.rs:99-111); l* are Lin pairs"""
        d = self._slots(dst)
        z = (Lin(), Lin())
        b0 = (l0, l1, z)
        b1 = (z, l4, z)
        # sparse operand on the A side of every term so that merge_terms groups by its 6 coefficients
        c0 = b6_add(b6_mul(b0, a.fp6(0)), b6_mul_v(b6_mul(b1, a.fp6(1))))
        c1 = b6_add(b6_mul(b1, a.fp6(0)), b6_mul(b0, a.fp6(1)))
        self.mulacc([{"dst": d[i], "bil": bl} for i, bl in enumerate(flatten12(c0, c1))], subst=self._sd_subst(a), sd=sd_out)
        return V12(d, sd=sd_out)

    def cyclotomic_sqr(self, dst, a, sd_out=None):
        """Granger-Scott squaring; one MULACC (T = 3) with the 3t +- 2z combination in the epilogue.
        sd_out: 12 slots that receive the companion sums/differences of the result (x0 + x1, x0 - x1 per Fp2
        coefficient); when the input carries them (a.sd) every operand form of the step is a stored value."""
        d = self._slots(dst)
        z0, z4, z3, z2, z1, z5 = (a.fp2(j) for j in range(6))

        def fp4(x, y):
            o0 = b2_add(b2_xi(b2_sqr(y)), b2_sqr(x))
            o1 = b2_mul((x[0].scale(2), x[1].scale(2)), y)
            return o0, o1

        t0, t1 = fp4(z0, z1)
        u0, u1 = fp4(z2, z3)
        w0, w1 = fp4(z4, z5)
        # z0' = 3 t0 - 2 z0 ; z4' = 3 u0 - 2 z4 ; z3' = 3 w0 - 2 z3 ; z2' = 3 xi w1 + 2 z2 ; z1' = 3 t1 + 2 z1 ; z5' = 3 u1 + 2 z5
        res = [(t0, -2), (u0, -2), (w0, -2), (b2_xi(w1), 2), (t1, 2), (u1, 2)]
        outs = []
        for j, (val, beta) in enumerate(res):
            for c in range(2):
                i = 2 * j + c
                outs.append({"dst": d[i], "bil": val[c], "alpha": 3, "beta": beta * a.signs[i], "e": a.slots[i]})
        self.mulacc(outs, subst=self._sd_subst(a), sd=sd_out)
        return V12(d, sd=sd_out)

    def frobenius(self, dst, a, power):
        """TRUE Frobenius^power (power in 1..3): coefficient of w^i is conjugated `power` times and
        multiplied by the constant F{power}_i.  Tower index j <-> w-power: (0,2,4,1,3,5)[j]."""
        d = self._slots(dst)
        wp = (0, 2, 4, 1, 3, 5)
        one = Lin.of(CONST_SLOT["ONE"])
        outs = []
        for j in range(6):
            x = a.fp2(j)
            if power & 1:
                x = l2_conj(x)
            if wp[j] == 0:
                val = (Bil([(x[0], one, 1)]), Bil([(x[1], one, 1)]))
            else:
                k = lin2(CONST_SLOT["F%d_%d_0" % (power, wp[j])], CONST_SLOT["F%d_%d_1" % (power, wp[j])])
                val = b2_mul(x, k)
            outs.append({"dst": d[2 * j], "bil": val[0]})
            outs.append({"dst": d[2 * j + 1], "bil": val[1]})
        self.mulacc(outs)
        return V12(d)

    def copy12(self, dst, a):
        d = self._slots(dst)
        self.lin([(d[i], a.lin(i)) for i in range(12)])
        return V12(d)

    # ---- spills to the per-check state buffer (12 elements starting at `elem`)
    def spill(self, v, elem):
        self.gstore(K_STATE, [(v.slots[i], elem + i) for i in range(12)], spill=True)
        kb = max(self.kof(s) for s in v.slots)
        self.release(v.slots)
        return (elem, list(v.signs), kb)

    def fill(self, handle):
        elem, signs, kb = handle
        s = self.alloc(12)
        self.gload(K_STATE, [(s[i], elem + i) for i in range(12)], kbound=kb)
        return V12(s, signs)


# ------------------------------------------------------------------------------------------ programs
def miller_bits():
    """iteration structure of the Miller loop over |x| >> 1 below its leading one: list of bools (add step?)"""
    bits = bin(M.BLS_X >> 1)[3:]
    return [b == "1" for b in bits]


def n_line_steps():
    return len(miller_bits()) + sum(miller_bits()) + 1   # 62 doublings + 5 additions + final doubling = 68


def runs(flags):
    """[False, False, True, False] -> [(2, True), (1, False)] : (number of plain iterations, then-has-add)"""
    out = []
    n = 0
    for f in flags:
        n += 1
        if f:
            out.append((n, True))
            n = 0
    if n:
        out.append((n, False))
    return out


def emit_line_mul(b, f, k):
    """for each of the k pairs: load the next line (6 Fp, already scaled by P) and multiply f by it"""
    tmp6 = b.alloc(6)
    for j in range(k):
        b.gload(K_LINE, [(tmp6[c], (j, c)) for c in range(6)], advance=1 if j == k - 1 else 0)
        l0 = lin2(tmp6[0], tmp6[1])
        l1 = lin2(tmp6[2], tmp6[3])
        l4 = lin2(tmp6[4], tmp6[5])
        f = b.fp12_mul_by_014(f, f, l0, l1, l4, sd_out=f.sd)
    b.release(tmp6)
    return f


def emit_line_mul_n(b, f):
    """the same for a check of ANY number of pairs: a pair loop (count = the launch's k) around ONE load + product; the
    accumulator and its companions are rewritten in place, so the body is the same for every pair"""
    tmp6 = b.alloc(6)
    b.ploop()
    b.gload(K_LINE, [(tmp6[c], (PAIR_VAR, c)) for c in range(6)])
    g = b.fp12_mul_by_014(f, f, lin2(tmp6[0], tmp6[1]), lin2(tmp6[2], tmp6[3]), lin2(tmp6[4], tmp6[5]), sd_out=f.sd)
    assert g.slots == f.slots and all(sg == 1 for sg in getattr(g, "signs", [1] * 12)), "the pair loop's body must map the accumulator to itself"
    g.sd = f.sd
    b.pendloop(advance=1)
    b.release(tmp6)
    return g


def prog_miller_n(to_wire):
    """multi_miller_loop of one check of k pairs, k a RUN-TIME value (round 5; the launch's k, 9..64 in practice): ONE accumulator,
    every iteration squares it once and multiplies the lines of all k pairs in (prog_miller unrolls the pairs and exists for k <= 8;
    more pairs used to run as groups of eight with their own 63 squarings each, joined by f12mul)"""
    b = Builder()
    f = b.alloc12()
    one = CONST_SLOT["ONE"]
    b.lin([(f.slots[i], Lin.of(one if i == 0 else ZERO)) for i in range(12)])
    f.sd = b.alloc(12)
    b.lin([(f.sd[i], Lin.of(one if i < 2 else ZERO)) for i in range(12)])
    for n_plain, has_add in runs(miller_bits()):
        body_plain = n_plain - 1 if has_add else n_plain
        if body_plain > 0:
            if body_plain > 1:
                b.loop(body_plain)
            f = emit_line_mul_n(b, f)
            f = b.fp12_sqr(f, f, sd_out=f.sd)
            if body_plain > 1:
                b.endloop()
        if has_add:
            f = emit_line_mul_n(b, f)
            f = emit_line_mul_n(b, f)
            f = b.fp12_sqr(f, f, sd_out=f.sd)
    f = emit_line_mul_n(b, f)
    b.release(f.sd)
    finish_output(b, f.conj(), to_wire, ST_F)
    return b


def prog_miller(k, to_wire):
    """multi_miller_loop of one check of k pairs from the precomputed line stream.
    Line stream order per step: (c2, c1*xP, c0*yP) == the (c0, c1, c4) arguments of mul_by_014."""
    b = Builder()
    f = b.alloc12()
    one = CONST_SLOT["ONE"]
    b.lin([(f.slots[i], Lin.of(one if i == 0 else ZERO)) for i in range(12)])
    # the accumulator keeps companion slots (x0 + x1, x0 - x1 of every Fp2 coefficient), rewritten by every step's
    # epilogue: the steps' operand forms are then stored or shifted values.  30 slots: a "wide" program (few constants)
    f.sd = b.alloc(12)
    b.lin([(f.sd[i], Lin.of(one if i < 2 else ZERO)) for i in range(12)])
    for n_plain, has_add in runs(miller_bits()):
        # iterations: [dbl-line, square] * (n_plain - 1), then [dbl-line, (add-line), square]
        body_plain = n_plain - 1 if has_add else n_plain
        if body_plain > 0:
            if body_plain > 1:
                b.loop(body_plain)
            f = emit_line_mul(b, f, k)
            f = b.fp12_sqr(f, f, sd_out=f.sd)
            if body_plain > 1:
                b.endloop()
        if has_add:
            f = emit_line_mul(b, f, k)
            f = emit_line_mul(b, f, k)
            f = b.fp12_sqr(f, f, sd_out=f.sd)
    f = emit_line_mul(b, f, k)
    b.release(f.sd)
    finish_output(b, f.conj(), to_wire, ST_F)
    return b


def prog_f12mul(to_wire):
    """ST_F <- ST_F * ST_G (or the canonical wire record of the product): joins the Miller values of the groups of
    at most four pairs a check with k > 4 pairs is split into (prod conj(f_i) = conj(prod f_i))."""
    b = Builder()
    f = load_input(b, False, ST_F)
    g = load_input(b, False, ST_G)
    h = b.fp12_mul(f, f, g)
    if to_wire:
        raw1 = Lin.of(CONST_SLOT["RAW_ONE"])
        b.mulacc([{"dst": g.slots[i], "bil": Bil([(h.lin(i), raw1, 1)])} for i in range(12)])
        b.gstore(K_WIRE, [(g.slots[i], i) for i in range(12)])
    else:
        b.gstore(K_STATE, [(h.slots[i], ST_F + i) for i in range(12)])
    return b


def prog_f12mul_pairs():
    """wire[check] <- wire[check] * wire[check + chk_off]: one level of the tree that multiplies n Fp12 values
    (product of Miller values before ONE shared final exponentiation)."""
    b = Builder()
    f = load_input(b, True)
    g = load_input(b, True, wire_kind=K_WIRE2)
    h = b.fp12_mul(f, f, g)
    raw1 = Lin.of(CONST_SLOT["RAW_ONE"])
    b.mulacc([{"dst": g.slots[i], "bil": Bil([(h.lin(i), raw1, 1)])} for i in range(12)])
    b.gstore(K_WIRE, [(g.slots[i], i) for i in range(12)])
    return b


def finish_output(b, v, to_wire, state_base, check_identity=False):
    t = b.alloc(12)
    if to_wire:
        # leave Montgomery form: reduce(v * 1); the store canonicalises and packs
        raw1 = Lin.of(CONST_SLOT["RAW_ONE"])
        b.mulacc([{"dst": t[i], "bil": Bil([(v.lin(i), raw1, 1)])} for i in range(12)])
        b.gstore(K_WIRE, [(t[i], i) for i in range(12)], check_identity)
    else:
        b.lin([(t[i], v.lin(i)) for i in range(12)])
        b.gstore(K_STATE, [(t[i], state_base + i) for i in range(12)])
    b.release(t)


def load_input(b, from_wire, state_base=0, wire_kind=K_WIRE):
    f = b.alloc12()
    if from_wire:
        b.gload(wire_kind, [(f.slots[i], i) for i in range(12)])
        r2 = Lin.of(CONST_SLOT["RAW_R2"])
        b.mulacc([{"dst": f.slots[i], "bil": Bil([(f.lin(i), r2, 1)])} for i in range(12)])
    else:
        b.gload(K_STATE, [(f.slots[i], state_base + i) for i in range(12)])
    return f


# state buffer layout (elements of 16 dwords each, per check)
ST_F = 0        # 12: Miller output (Montgomery Fp28)
ST_C = 12       # 6 : Fp6-inverse cofactors C0..C2
ST_NC = 18      # 2 : N = (N0, N1) in Fp2
ST_N = 20       # 1 : n = N0^2 + N1^2 in Fp  (input of the batched inversion kernel)
ST_NINV = 21    # 1 : n^-1                      (its output)
ST_TI = 22      # 6 : t^-1 (Fp6), parked while conj(f)^2 is formed
ST_SPILL = 28   # 8 x 12 spill areas used by the hard part
ST_G = ST_SPILL  # 12: Miller value of a later group of pairs (k > 4), multiplied into ST_F before the final exponentiation
ST_SNAP = ST_SPILL + 8 * 12   # 6 x 12: snapshots of a compressed squaring run (k_ksq writes z2..z5, k_kdec_b adds z0, z1)
ST_KN = ST_SNAP + 6 * 12      # 6 : |D|^2 of the six decompressions (input planes of the batched inversion)
ST_KNINV = ST_KN + 6          # 6 : their inverses
ST_SIZE = ST_KNINV + 6
# plan step kinds of a multi-kernel program (the final exponentiation's phase C)
PLAN_PROG, PLAN_KSQ, PLAN_KDEC_A, PLAN_INV, PLAN_KDEC_B = 0, 1, 2, 3, 4


def prog_fexp_a(from_wire):
    """first half of the final exponentiation: everything of Fp12::invert (reference src/fp12.rs:186-190,
    src/fp6.rs:291-309, src/fp2.rs:278-296) down to the single Fp inversion."""
    b = Builder()
    f = load_input(b, from_wire, ST_F)
    if from_wire:
        b.gstore(K_STATE, [(f.slots[i], ST_F + i) for i in range(12)])
    a0, a1 = f.fp6(0), f.fp6(1)
    # t = a0^2 - v a1^2  (Fp6): lanes 0-5 give a0^2, lanes 6-11 give a1^2
    sq = b.alloc(12)
    b.mulacc([{"dst": sq[i], "bil": bl} for i, bl in enumerate(flatten12(b6_mul(a0, a0), b6_mul(a1, a1)))])
    b.release(f.slots)
    S0 = tuple(lin2(sq[2 * j], sq[2 * j + 1]) for j in range(3))
    S1 = tuple(lin2(sq[6 + 2 * j], sq[6 + 2 * j + 1]) for j in range(3))
    vS1 = l6_mul_v(S1)
    t6 = b.alloc(6)
    b.lin([(t6[2 * j + c], l2_sub(S0[j], vS1[j])[c]) for j in range(3) for c in range(2)])
    b.release(sq)
    t = tuple(lin2(t6[2 * j], t6[2 * j + 1]) for j in range(3))
    # cofactors C0 = t0^2 - xi t1 t2 ; C1 = xi t2^2 - t0 t1 ; C2 = t1^2 - t0 t2
    C0 = b2_sub(b2_mul(t[0], t[0]), b2_xi(b2_mul(t[1], t[2])))
    C1 = b2_sub(b2_xi(b2_mul(t[2], t[2])), b2_mul(t[0], t[1]))
    C2 = b2_sub(b2_mul(t[1], t[1]), b2_mul(t[0], t[2]))
    cs = b.alloc(6)
    b.mulacc([{"dst": cs[2 * j + c], "bil": (C0, C1, C2)[j][c]} for j in range(3) for c in range(2)])
    C = tuple(lin2(cs[2 * j], cs[2 * j + 1]) for j in range(3))
    # N = xi (t1 C2 + t2 C1) + t0 C0   (Fp2)
    Nb = b2_add(b2_xi(b2_add(b2_mul(t[1], C[2]), b2_mul(t[2], C[1]))), b2_mul(t[0], C[0]))
    ns = b.alloc(2)
    b.mulacc([{"dst": ns[c], "bil": Nb[c]} for c in range(2)])
    # n = N0^2 + N1^2
    n1 = b.alloc(1)
    b.mulacc([{"dst": n1[0], "bil": Bil([(Lin.of(ns[0]), Lin.of(ns[0]), 1), (Lin.of(ns[1]), Lin.of(ns[1]), 1)])}])
    b.gstore(K_STATE, [(cs[i], ST_C + i) for i in range(6)] + [(ns[i], ST_NC + i) for i in range(2)] + [(n1[0], ST_N)])
    return b


def sqr_run(b, v, n):
    """n cyclotomic squarings in place on the interpreter (timing programs only; the x-power chains run compressed
    squarings, cyc_exp).  The run keeps 12 companion slots (x0 + x1, x0 - x1 of every Fp2 coefficient, written by
    each squaring's epilogue): after the first squaring every operand form of the step is a stored value or a
    shifted one, so the product loop fetches no second operands."""
    if n <= 0:
        return v
    sd = b.alloc(12)
    v = b.cyclotomic_sqr(v, v, sd_out=sd)
    if n > 1:
        if n > 2:
            b.loop(n - 1)
        v = b.cyclotomic_sqr(v, v, sd_out=sd)
        if n > 2:
            b.endloop()
    b.release(sd)
    return V12(v.slots)


X_BITS = [i for i in range(64) if (M.BLS_X >> i) & 1]       # 16, 48, 57, 60, 62, 63
KSQ_NSQ = X_BITS[-1]                                         # squarings of one x-power run
KSQ_MASK = sum(1 << (e - 1) for e in X_BITS)                 # snapshot after squaring number e (loop index e - 1)
KSQ_COMPANIONS = int(os.environ.get("ZKP_GEN_KSQ_COMPANIONS", "1"))  # companions for the squarings / products behind a compressed run (36 slots)
KSQ_TOP_CHAIN = int(os.environ.get("ZKP_GEN_KSQ_TOP_CHAIN", "1"))   # the bits above the compressed run by the 7 * 15 chain (0: binary method)
KSQ_SPLIT = int(os.environ.get("ZKP_GEN_KSQ_SPLIT", "3"))    # set bits of |x| the compressed run covers (6: all of them - rounds 2-3)


def cyc_exp(b, a, park, half=False):
    """-> (a^x for the curve's NEGATIVE x, i.e. conj(a^|x|), in freshly allocated slots; spill handle of a).  a (cyclotomic
    subgroup; an LDS-resident V12, or the spill handle of a value that is already parked - then `park` is ignored) goes to
    the state buffer, ONE compressed squaring run (k_ksq) squares it 63 times and keeps the six powers a^(2^e), e the set
    bits of |x|, the decompression kernels complete them, and the step program multiplies them: a^|x| = prod_e a^(2^e).
    half: the exponent is x / 2 (|x| is even: one squaring less, the snapshots one squaring earlier)."""
    if isinstance(a, V12):
        if any(s != 1 for s in a.signs):
            a = b.copy12(a, a)            # the kernels read raw records: a conjugated view must be materialised
        assert max(b.kof(s) for s in a.slots) <= Builder.K_INPUT
        ha = b.spill(a, park)
    else:
        ha = a
        park, signs, kb = ha
        assert all(s == 1 for s in signs) and kb <= Builder.K_INPUT
    bits = [e - 1 for e in X_BITS] if half else X_BITS
    assert bits[0] >= 1
    # the compressed run covers the first KSQ_SPLIT set bits (16, 48, 57); the squarings above its last snapshot (six, with a
    # product after the third, fifth and sixth) are Granger-Scott squarings of the step program on the decompressed value: three
    # snapshots + decompressions instead of six per run
    n = min(KSQ_SPLIT, len(bits))
    b.cut([(PLAN_KSQ, park, ST_SNAP, bits[n - 1], sum(1 << (e - 1) for e in bits[:n])), (PLAN_KDEC_A, ST_SNAP, n, ST_KN),
           (PLAN_INV, ST_KN, ST_KNINV, n), (PLAN_KDEC_B, ST_SNAP, n, ST_KNINV)])
    snap = lambda k: (ST_SNAP + 12 * k, [1] * 12, Builder.K_REDUCED)
    r = b.fill(snap(n - 1))
    top = sum(1 << (e - bits[n - 1]) for e in bits[n - 1:])
    if top == 105 and KSQ_TOP_CHAIN:
        # d = a^(2^57); the bits above it are d^105, and 105 = 7 * 15 = (2^3 - 1)(2^4 - 1): with the free inverse of the cyclotomic
        # subgroup (the conjugate) that is 7 squarings + 2 products, d^7 = d^8 conj(d), (d^7)^15 = (d^7)^16 conj(d^7), instead of
        # the 6 + 3 of the binary method - a product costs 2.6 Granger-Scott squarings on this machine
        # With KSQ_COMPANIONS (36 slots: the "deep" LDS configuration) the running square and the running product carry their
        # companion sums / differences: every operand form of the squarings and of the products is then a stored value.
        def pow2k_times_conj(x, k):
            if KSQ_COMPANIONS:
                ysd = x.sd if x.sd else b.alloc(12)          # x's companions become y's (read as operands, then overwritten)
                y = b.cyclotomic_sqr(b.alloc(12), x, sd_out=ysd)
                x = V12(x.slots, x.signs)
                for _ in range(k - 1):
                    y = b.cyclotomic_sqr(y, y, sd_out=y.sd)
                y = b.fp12_mul(y, x.conj(), y, sd_out=y.sd)
            else:
                y = b.cyclotomic_sqr(b.alloc(12), x)
                for _ in range(k - 1):
                    y = b.cyclotomic_sqr(y, y)
                y = b.fp12_mul(y, y, x.conj())
            b.release(x.slots)
            return y
        r = pow2k_times_conj(pow2k_times_conj(r, 3), 4)
    elif n < len(bits):
        sq = None
        for e in range(bits[n - 1] + 1, bits[-1] + 1):
            sq = b.cyclotomic_sqr(b.alloc(12), r) if sq is None else b.cyclotomic_sqr(sq, sq)
            if e in bits:
                r = b.fp12_mul(r, r, sq)
        b.release(sq.slots)
    for k in range(n - 2, -1, -1):
        s = b.fill(snap(k))
        r = b.fp12_mul(r, s, r, sd_out=r.sd) if r.sd else b.fp12_mul(r, r, s)
        b.release(s.slots)
    if r.sd:
        b.release(r.sd)
        r = V12(r.slots, r.signs)
    return r.conj(), ha


def prog_fexp_c(to_wire=True):
    """second half: finish the inversion, easy part, hard part (x-chain), output Gt.  At most TWO Fp12
    values are LDS-resident at any time - plus, behind a compressed squaring run, the companion sums / differences of one of
    them (36 slots: the "deep" LDS configuration, still twelve wavefronts per CU); everything else is parked in the
    per-check state buffer (a spill or fill moves 768 B per check).  Returns a PLAN (Builder.cut): the step program is
    cut at every x-power chain, whose 63 squarings run as one compressed squaring kernel (cyc_exp)."""
    b = Builder()
    st = b.alloc(9)
    b.gload(K_STATE, [(st[i], ST_C + i) for i in range(6)] + [(st[6 + i], ST_NC + i) for i in range(2)] + [(st[8], ST_NINV)])
    C = tuple(lin2(st[2 * j], st[2 * j + 1]) for j in range(3))
    N = lin2(st[6], st[7])
    ninv = Lin.of(st[8])
    # N^-1 = (N0 ninv, -N1 ninv)
    ni = b.alloc(2)
    b.mulacc([{"dst": ni[0], "bil": Bil([(N[0], ninv, 1)])}, {"dst": ni[1], "bil": Bil([(N[1], ninv, -1)])}])
    NI = lin2(ni[0], ni[1])
    # t^-1 = (C0, C1, C2) * N^-1   (Fp6), parked while conj(f)^2 is formed
    ti = b.alloc(6)
    b.mulacc([{"dst": ti[2 * j + c], "bil": b2_mul(C[j], NI)[c]} for j in range(3) for c in range(2)])
    b.gstore(K_STATE, [(ti[i], ST_TI + i) for i in range(6)])
    b.release(st)
    b.release(ni)
    b.release(ti)
    # f^(p^6-1) = conj(f) / f = conj(f)^2 * t^-1   because f^-1 = conj(f) * t^-1 with t = f conj(f) in Fp6
    f = V12(b.alloc(12))
    b.gload(K_STATE, [(f.slots[i], ST_F + i) for i in range(12)])
    c2 = b.fp12_sqr(f, f.conj())
    ti = b.alloc(6)
    b.gload(K_STATE, [(ti[i], ST_TI + i) for i in range(6)])
    TI = tuple(lin2(ti[2 * j], ti[2 * j + 1]) for j in range(3))
    u = V12(c2.slots)
    b.mulacc([{"dst": u.slots[i], "bil": bl} for i, bl in enumerate(flatten12(b6_mul(TI, c2.fp6(0)), b6_mul(TI, c2.fp6(1))))])
    b.release(ti)
    # easy part continued: t2 = frob^2(u) * u
    t2 = b.frobenius(b.alloc(12), u, 2)
    t2 = b.fp12_mul(t2, t2, u)
    # hard part: r^((x-1)^2 (x+p) (x^2+p^2-1) + 3) = r^(3 (p^4-p^2+1)/r) (Hayashida-Hayasaka-Teruya, ePrint 2020/875) - the same
    # power as the upstream-shaped chain of SURVEY.md S6 (the oracle's; tests/test_coopgen.py holds both against the model), with
    # five x-power chains, 7 products and one Granger-Scott squaring outside them instead of 10 and 2.  x is negative: a^x
    # is conj(a^|x|), and inverses in the cyclotomic subgroup are conjugates (free sign views).
    SP = [ST_SPILL + 12 * i for i in range(8)]
    r = t2                                           # resident: r (and u's slots, free for reuse)
    sq = b.cyclotomic_sqr(u, r)                      # r^2                         resident: r, r^2
    hr = b.spill(r, SP[0])
    a1, hsq = cyc_exp(b, sq, SP[1], half=True)       # (r^2)^(x/2) = r^x
    rr = b.fill(hr)
    a1 = b.fp12_mul(a1, a1, rr.conj())               # r^(x-1)
    b.release(rr.slots)
    a2, h1 = cyc_exp(b, a1, SP[2])                   # r^((x-1) x)                 parked: r^(x-1)
    a1 = b.fill(h1)
    a1 = b.fp12_mul(a1, a1.conj(), a2)               # r^((x-1)^2)
    b.release(a2.slots)
    a2, h1 = cyc_exp(b, a1, SP[3])                   # r^((x-1)^2 x)               parked: r^((x-1)^2)
    a1 = b.fill(h1)
    a1 = b.frobenius(a1, a1, 1)                      # r^((x-1)^2 p)
    a1 = b.fp12_mul(a1, a1, a2)                      # c = r^((x-1)^2 (x+p))
    b.release(a2.slots)
    hc = b.spill(a1, SP[4])
    rr = b.fill(hr)
    s2 = b.fill(hsq)
    rr = b.fp12_mul(rr, rr, s2)                      # r^3
    b.release(s2.slots)
    h3 = b.spill(rr, SP[0])
    a0, _ = cyc_exp(b, hc, None)                     # c^x
    a2, _ = cyc_exp(b, a0, SP[5])                    # c^(x^2)
    a1 = b.fill(hc)
    a2 = b.fp12_mul(a2, a2, a1.conj())               # c^(x^2 - 1)
    a1 = b.frobenius(a1, a1, 2)                      # c^(p^2)
    a2 = b.fp12_mul(a2, a2, a1)                      # c^(x^2 + p^2 - 1)
    b.release(a1.slots)
    rr = b.fill(h3)
    t3 = b.fp12_mul(rr, rr, a2)
    b.release(a2.slots)
    finish_output(b, t3, to_wire, ST_F, check_identity=True)
    return b.finish_plan()


_FEXP_C_PLAN = None


def fexp_c_plan():
    """the final exponentiation's phase C as a plan: [("prog", Seg) | (PLAN_KSQ, elem_in, elem_snap, nsq, mask) |
    (PLAN_KDEC_A, elem_snap, count, elem_n) | (PLAN_INV, elem_n, elem_ninv, count) | (PLAN_KDEC_B, elem_snap, count, elem_ninv)]"""
    global _FEXP_C_PLAN
    if _FEXP_C_PLAN is None:
        _FEXP_C_PLAN = prog_fexp_c(True)
    return _FEXP_C_PLAN


def fexp_c_segments():
    return [st[1] for st in fexp_c_plan() if st[0] == "prog"]


# ------------------------------------------------------------------------------------------ emulator (bit-accurate)
def to_limbs_balanced(v):
    """true integer v (|v| < 2^391) -> 14 balanced limbs"""
    neg = v < 0
    v = abs(v)
    out = []
    carry = 0
    for i in range(NL):
        d = ((v >> (W * i)) & ((1 << W) - 1)) + carry
        if i < NL - 1 and d >= (1 << (W - 1)):
            d -= 1 << W
            carry = 1
        else:
            carry = 0
        out.append(d)
    if neg:
        out = [-x for x in out]
    return out


def limbs_value(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def mont(v):
    return to_limbs_balanced(v % P * RMOD % P)


def from_mont(l):
    return limbs_value(l) * RINV % P


def lo28s(v):
    x = v & ((1 << W) - 1)
    return x - (1 << W) if x >= (1 << (W - 1)) else x


def acc_reduce(col):
    """exactly zkp28::acc_reduce"""
    col = list(col) + [0] * (2 * NL - len(col))
    for i in range(NL):
        m = ((col[i] & 0xFFFFFFFF) * PINV28) & ((1 << W) - 1)
        for j in range(NL):
            col[i + j] += m * P_LIMBS[j]
            assert abs(col[i + j]) < (1 << 63), "column overflow in reduction"
        assert col[i] & ((1 << W) - 1) == 0
        col[i + 1] += col[i] >> W
    # carry-propagate the signed 64-bit columns c[14..26] into balanced 28-bit limbs (the top limb keeps the rest)
    out = []
    v = col[NL]
    for k in range(NL - 1):
        assert abs(v) < (1 << 63) - (1 << W)
        t = v + (1 << (W - 1))
        out.append((t & ((1 << W) - 1)) - (1 << (W - 1)))
        v = col[NL + k + 1] + (t >> W)
    assert abs(v) < (1 << 31)
    out.append(v)
    assert limbs_value(out) == sum(col[NL + k] << (W * k) for k in range(NL - 1))
    return out


def weak_norm(x):
    x = list(x)
    c = []
    for i in range(NL - 1):
        assert abs(x[i]) < (1 << 31) - (1 << 27), "limb overflow before weak_norm"
        ci = (x[i] + (1 << (W - 1))) >> W
        c.append(ci)
        x[i] -= ci << W
    for i in range(1, NL):
        x[i] += c[i - 1]
        assert abs(x[i]) < (1 << 31)
    return x


def vred(x):
    """value renormalisation, exactly as the kernel does it: q = round(top / p_top) from the top limb,
    x -= q * p limb-wise (balanced p), weak_norm.  Input: weakly normalised limbs, |value| < ~14 p."""
    t = x[NL - 1]
    q = ((t >> VRED_SHIFT_IN) * VRED_C + (1 << (VRED_SHIFT_OUT - 1))) >> VRED_SHIFT_OUT
    assert abs(q) <= 14, "value too large for vred (q = %d)" % q
    y = [a - q * b for a, b in zip(x, P_BAL)]
    y = weak_norm(y)
    assert abs(limbs_value(y)) < 0.51 * P
    return y


def epilogue(r, e, alpha, beta):
    """alpha r + beta e, renormalised, exactly as the kernel's MULACC epilogue: q from the top limb of the
    combination, x = alpha r + beta e - q p limb-wise, ONE weak_norm"""
    t = alpha * r[NL - 1] + beta * e[NL - 1]
    q = ((t >> VRED_SHIFT_IN) * VRED_C + (1 << (VRED_SHIFT_OUT - 1))) >> VRED_SHIFT_OUT
    assert abs(q) <= 14, "value too large for the epilogue (q = %d)" % q
    y = weak_norm([alpha * a + beta * b - q * c for a, b, c in zip(r, e, P_BAL)])
    assert abs(limbs_value(y)) < 0.51 * P
    return y


def canonical_from_reduced(l):
    v = limbs_value(l)
    assert -P < v < 2 * P, "value out of canonicalisation range"
    return v % P


def const_limbs(name, val):
    if name == "RAW_R2":
        return to_limbs_balanced(RMOD * RMOD % P)
    if name == "RAW_ONE":
        return [1] + [0] * (NL - 1)
    return mont(val)


class Emu:
    """one lane-group.  slots hold limb vectors.  `lines`: list of steps, each a list (per pair) of 6
    true field values (c2, c1*xP, c0*yP); `state`/`wire_in`: dict/list of true values."""

    def __init__(self, lines=None, state=None, wire_in=None, wire_in2=None):
        self.slot = {}
        for name, val in CONSTS:
            self.slot[CONST_SLOT[name]] = const_limbs(name, val)
        self.lines = lines or []
        self.cursor = 0
        self.state = dict(state or {})     # element -> limb vector
        self.wire_in = wire_in
        self.wire_in2 = wire_in2
        self.wire_out = None
        self.is_identity = None
        self.max_col = 0
        self.counts = {"mulacc_steps": 0, "products": 0, "lin_steps": 0, "P_blocks": 0}

    def form(self, s1, s2, sub):
        a, b = self.slot[s1], self.slot[s2]
        return [x - y for x, y in zip(a, b)] if sub else [x + y for x, y in zip(a, b)]

    def run(self, steps):
        pc = 0
        loop_start, loop_left = None, 0
        ploop_start, ploop_left, pair_var = None, 0, 0
        while pc < len(steps):
            st = steps[pc]
            op = st["op"]
            if op == OP_LOOP:
                loop_start, loop_left = pc + 1, st["n"]
            elif op == OP_ENDLOOP:
                loop_left -= 1
                if loop_left > 0:
                    pc = loop_start
                    continue
            elif op == OP_PLOOP:       # count = the launch's k = the pairs of a line-stream step
                ploop_start, ploop_left, pair_var = pc + 1, len(self.lines[self.cursor]), 0
            elif op == OP_PENDLOOP:
                ploop_left -= 1
                if ploop_left > 0:
                    pair_var += 1
                    pc = ploop_start
                    continue
                pair_var = 0
                self.cursor += st.get("advance", 0)
            elif op == OP_MULACC:
                self.counts["mulacc_steps"] += 1
                self.counts["P_blocks"] += st["T"]
                res = []
                for ln in st["lanes"]:
                    col = [0] * (2 * NL - 1)
                    for (a1, a2, asub, b1, b2, bsub, neg, da) in ln["terms"]:
                        a = self.form(a1, a2, asub)
                        b = self.form(b1, b2, bsub)
                        if neg:
                            a = [-x for x in a]
                        if da:
                            a = [2 * x for x in a]
                        assert all(abs(x) < (1 << 31) for x in a + b)
                        for i in range(NL):
                            if a[i]:
                                for j in range(NL):
                                    col[i + j] += a[i] * b[j]
                        self.counts["products"] += 1
                    mx = max(abs(c) for c in col)
                    self.max_col = max(self.max_col, mx)
                    assert mx < (1 << 62), "column overflow in accumulation (%d bits)" % mx.bit_length()
                    r = acc_reduce(col)
                    e = self.slot[ln["e"]]
                    if st["epi"]:
                        out = epilogue(r, e, ln["alpha"], ln["beta"])
                    else:
                        assert (ln["alpha"], ln["beta"]) == (1, 0)
                        out = r
                    res.append((ln["dst"], out))
                for d, v in res:
                    self.slot[d] = v
                if st.get("sd"):
                    for i, sl in enumerate(st["sd"]):
                        mine, other = res[i][1], res[i ^ 1][1]
                        self.slot[sl] = [x + y for x, y in zip(mine, other)] if i % 2 == 0 else [y - x for x, y in zip(mine, other)]
            elif op == OP_LIN:
                self.counts["lin_steps"] += 1
                res = []
                for dst, terms in st["lanes"]:
                    acc = [0] * NL
                    for s, c in terms:
                        acc = [x + c * y for x, y in zip(acc, self.slot[s])]
                    acc = weak_norm(acc)
                    res.append((dst, vred(acc) if st.get("vred", True) else acc))
                for d, v in res:
                    self.slot[d] = v
            elif op == OP_GLOAD:
                for dst, idx in st["lanes"]:
                    if st["kind"] == K_LINE:
                        pair, c = idx
                        self.slot[dst] = mont(self.lines[self.cursor][pair_var if pair == PAIR_VAR else pair][c])
                    elif st["kind"] == K_STATE:
                        self.slot[dst] = list(self.state[idx])
                    else:
                        v = (self.wire_in2 if st["kind"] == K_WIRE2 else self.wire_in)[idx]
                        assert 0 <= v < P
                        self.slot[dst] = [(v >> (W * i)) & ((1 << W) - 1) for i in range(NL)]
                self.cursor += st.get("advance", 0)
            elif op == OP_GSTORE:
                if st["kind"] == K_STATE:
                    for src, idx in st["lanes"]:
                        self.state[idx] = list(self.slot[src])
                else:
                    out = {}
                    for src, idx in st["lanes"]:
                        out[idx] = canonical_from_reduced(self.slot[src])
                    self.wire_out = [out[i] for i in range(12)]
                    if st["check"]:
                        self.is_identity = self.wire_out == [1] + [0] * 11
            pc += 1
        return self


# ---- limb-exact models of the non-interpreter kernels of a plan (zkp_coop.hip k_ksq, k_kdec_a, k_kdec_b) -------------
def mont_mul(pairs):
    """reduce(sum of limb-vector products): zkp28::acc_mul + acc_reduce == mont_mul_ps (same columns, same m_i)"""
    col = [0] * (2 * NL - 1)
    for a, b in pairs:
        assert all(abs(x) < (1 << 31) for x in a) and all(abs(x) < (1 << 31) for x in b)
        for i in range(NL):
            if a[i]:
                for j in range(NL):
                    col[i + j] += a[i] * b[j]
    assert max(abs(c) for c in col) < (1 << 62), "column overflow in accumulation"
    return acc_reduce(col)


def sq_combine(t, x, neg):
    """zkp_coop.hip sq_combine: 3 t + 2 sgn x - q p as one exact carry chain"""
    sx = [-v for v in x] if neg else list(x)
    top = 3 * t[NL - 1] + 2 * sx[NL - 1]
    assert abs(top) < (1 << 31)
    q = ((top >> VRED_SHIFT_IN) * VRED_C + (1 << (VRED_SHIFT_OUT - 1))) >> VRED_SHIFT_OUT
    assert abs(q) <= 14
    out, v = [], 0
    for i in range(NL):
        assert abs(t[i]) < (1 << 31) and abs(sx[i]) < (1 << 31)
        v += 3 * t[i] + 2 * sx[i] - q * P_BAL[i]
        assert abs(v) < (1 << 62)
        if i < NL - 1:
            u = v + (1 << (W - 1))
            out.append((u & ((1 << W) - 1)) - (1 << (W - 1)))
            v = u >> W
        else:
            assert abs(v) < (1 << 31)
            out.append(v)
    assert abs(limbs_value(out)) < 0.51 * P
    return out


KS_TU, KS_TV = (3, 1), (2, 5)       # tower positions of (u, v) = (z2, z3) for lanes 0, 1 and (z4, z5) for lanes 2, 3


def emu_ksq(state, elem_in, elem_snap, nsq, mask):
    """k_ksq on one check: four lanes, lane r computes A23, B23, A45, B45; state is updated in place"""
    vadd = lambda a, b: [x + y for x, y in zip(a, b)]
    vsub = lambda a, b: [x - y for x, y in zip(a, b)]

    def form(r, mine, other):
        (mr, mi), (o_r, oi) = mine, other
        if r & 1:
            return [list(mr), list(mi), list(o_r), list(oi)]
        xr, xi = vadd(mr, o_r), vadd(mi, oi)
        mine_is_v = r in (0, 3)
        return [xr, xi, vsub(xr, mi if mine_is_v else oi), vadd(xi, mr if mine_is_v else o_r)]

    pairv = []
    for pr in range(2):
        u = (state[elem_in + 2 * KS_TU[pr]], state[elem_in + 2 * KS_TU[pr] + 1])
        v = (state[elem_in + 2 * KS_TV[pr]], state[elem_in + 2 * KS_TV[pr] + 1])
        pairv.append((u, v))
    mine = [pairv[0][1], pairv[0][0], pairv[1][0], pairv[1][1]]
    other = [pairv[0][0], pairv[0][1], pairv[1][1], pairv[1][0]]
    snap = elem_snap
    for it in range(nsq):
        prod = []
        for r in range(4):
            xr, xi, yr, yi = form(r, mine[r], other[r])
            sim = mont_mul([(xr, yi), (xi, yr)])
            nxi = [-x for x in xi]
            sre = mont_mul([(xr, yr), (nxi, yi)])
            prod.append((sre, sim))
        new = []
        for r in range(4):
            ao, bo = prod[(r & 2) ^ 2], prod[((r & 2) ^ 2) | 1]
            if not (r & 1):
                tr = [a - 2 * b + c for a, b, c in zip(ao[0], bo[0], bo[1])]
                ti = [a - b - 2 * c for a, b, c in zip(ao[1], bo[0], bo[1])]
            elif r == 1:
                tr = [2 * (b - c) for b, c in zip(bo[0], bo[1])]
                ti = [2 * (b + c) for b, c in zip(bo[0], bo[1])]
            else:
                tr, ti = [2 * b for b in bo[0]], [2 * c for c in bo[1]]
            neg = not (r & 1)
            new.append((sq_combine(tr, mine[r][0], neg), sq_combine(ti, mine[r][1], neg)))
        mine = new
        other = [new[r ^ 1] for r in range(4)]
        if (mask >> it) & 1:
            for r in (0, 2):
                pr = r >> 1
                tm, to = (KS_TV[pr], KS_TU[pr]) if r == 0 else (KS_TU[pr], KS_TV[pr])
                state[snap + 2 * tm], state[snap + 2 * tm + 1] = list(mine[r][0]), list(mine[r][1])
                state[snap + 2 * to], state[snap + 2 * to + 1] = list(other[r][0]), list(other[r][1])
            snap += 12
    return state


def _c_lin(a, b, sb):
    return weak_norm([x + sb * y for x, y in zip(a, b)])


def c2_add(a, b):
    return (_c_lin(a[0], b[0], 1), _c_lin(a[1], b[1], 1))


def c2_sub(a, b):
    return (_c_lin(a[0], b[0], -1), _c_lin(a[1], b[1], -1))


def c2_dbl(a):
    return c2_add(a, a)


def c2_sqr(a):
    """zkp_coop.hip c_sqr on a lane pair: (a0 + a1)(a0 - a1), (2 a0) a1"""
    s = [x + y for x, y in zip(a[0], a[1])]
    d = [x - y for x, y in zip(a[0], a[1])]
    return (mont_mul([(s, d)]), mont_mul([([2 * x for x in a[0]], a[1])]))


def c2_mul(a, b):
    """zkp_coop.hip c_mul_q on a lane pair"""
    return (mont_mul([(a[0], b[0]), ([-x for x in a[1]], b[1])]), mont_mul([(a[0], b[1]), (a[1], b[0])]))


def c2_xi(a):
    """xi a formed the way the kernels do: lane 0 mine - other, lane 1 other + mine"""
    return (_c_lin(a[0], a[1], -1), _c_lin(a[0], a[1], 1))


def c2_is_zero(a):
    return limbs_value(a[0]) % P == 0 and limbs_value(a[1]) % P == 0


def _snap_fp2(state, base, pos):
    return (state[base + 2 * pos], state[base + 2 * pos + 1])


def emu_kdec_a(state, elem_snap, count, elem_n):
    for sn in range(count):
        base = elem_snap + 12 * sn
        z2, z3, z4, z5 = (_snap_fp2(state, base, pos) for pos in (3, 2, 1, 5))
        z2_zero = c2_is_zero(z2)
        na = c2_xi(c2_sqr(z5))
        s4 = c2_sqr(z4)
        na = c2_sub(c2_add(na, c2_add(c2_dbl(s4), s4)), c2_dbl(z3))
        nb = c2_dbl(c2_mul(z4, z5))
        N, D = (nb, z3) if z2_zero else (na, z2)
        n = mont_mul([(D[0], D[0]), (D[1], D[1])])
        if not z2_zero:
            n = _c_lin(n, n, 1)
            n = _c_lin(n, n, 1)
        state[base + 8], state[base + 9] = N            # (the kernel tags the record's padding with z2_zero: D itself is not stored)
        state[elem_n + sn] = n
    return state


def emu_inv(state, elem_n, elem_ninv, count):
    """value-exact stand-in for k_batch_inv (its limbs depend on the batching); zero gives zero"""
    for j in range(count):
        v = from_mont(state[elem_n + j])
        state[elem_ninv + j] = mont(M.fp_inv(v)) if v else [0] * NL
    return state


def emu_kdec_b(state, elem_snap, count, elem_ninv):
    one = (const_limbs("ONE", 1), [0] * NL)
    for sn in range(count):
        base = elem_snap + 12 * sn
        N = (state[base + 8], state[base + 9])
        z2, z3 = _snap_fp2(state, base, 3), _snap_fp2(state, base, 2)
        D = z3 if c2_is_zero(z2) else z2                 # k_kdec_a's tag in N's record: which of the two the denominator is
        ninv = state[elem_ninv + sn]
        dinv = (mont_mul([(D[0], ninv)]), mont_mul([([-x for x in D[1]], ninv)]))
        z1 = c2_mul(N, dinv)
        z2, z3, z4, z5 = (_snap_fp2(state, base, pos) for pos in (3, 2, 1, 5))
        # t = 2 z1^2 + z2 z5 - 3 z3 z4 under ONE reduction per coefficient (zkp_coop.hip k_kdec_b): five products each
        neg = lambda v: [-x for x in v]
        k3 = lambda v: [-3 * x for x in v]
        t = (mont_mul([([2 * (x + y) for x, y in zip(z1[0], z1[1])], [x - y for x, y in zip(z1[0], z1[1])]),
                       (z2[0], z5[0]), (neg(z2[1]), z5[1]), (k3(z3[0]), z4[0]), (neg(k3(z3[1])), z4[1])]),
             mont_mul([([4 * x for x in z1[0]], z1[1]),
                       (z2[0], z5[1]), (z2[1], z5[0]), (k3(z3[0]), z4[1]), (k3(z3[1]), z4[0])]))
        xt = c2_xi(t)
        z0 = (vred(_c_lin(xt[0], one[0], 1)), vred(xt[1]))
        state[base + 0], state[base + 1] = z0
        state[base + 8], state[base + 9] = z1
    return state


def run_plan(plan, state, wire_in=None):
    """emulate a multi-kernel plan on one check; returns the Emu of the last step program (state, wire_out, ...)"""
    state = dict(state)
    em = None
    stats = {"max_col": 0, "mulacc_steps": 0}
    for st in plan:
        if st[0] == "prog":
            em = Emu(state=state, wire_in=wire_in).run(st[1].steps)
            state = em.state
            stats["max_col"] = max(stats["max_col"], em.max_col)
            stats["mulacc_steps"] += em.counts["mulacc_steps"]
        elif st[0] == PLAN_KSQ:
            emu_ksq(state, *st[1:])
        elif st[0] == PLAN_KDEC_A:
            emu_kdec_a(state, *st[1:])
        elif st[0] == PLAN_INV:
            emu_inv(state, *st[1:])
        elif st[0] == PLAN_KDEC_B:
            emu_kdec_b(state, *st[1:])
        else:
            raise ValueError(st[0])
    em.plan_stats = stats
    return em


def model_lines(pairs):
    """line stream for a check: list over the 68 steps of [per pair (c2, c1*xP, c0*yP)]; pairs with an
    infinity get the neutral line (1, 0, 0)."""
    k = len(pairs)
    rs = [None if (p is None or q is None) else (q[0], q[1], M.F2_ONE) for p, q in pairs]
    out = []

    def emit(step_fn):
        row = []
        for i, (p1, q) in enumerate(pairs):
            if rs[i] is None:
                row.append([1, 0, 0, 0, 0, 0])
                continue
            rs[i], c = step_fn(i)
            c0 = M.f2_muls(c[0], p1[1])
            c1 = M.f2_muls(c[1], p1[0])
            row.append([c[2][0], c[2][1], c1[0], c1[1], c0[0], c0[1]])
        out.append(row)

    for has_add in miller_bits():
        emit(lambda i: M.doubling_step(rs[i]))
        if has_add:
            emit(lambda i: M.addition_step(rs[i], pairs[i][1]))
    emit(lambda i: M.doubling_step(rs[i]))
    assert len(out) == n_line_steps()
    return out


# ------------------------------------------------------------------------------------------ binary encoding
def encode(builder):
    """-> (hdr words, table words).  hdr: 4 words per step: op | arg0<<8 , arg1, table offset, arg2."""
    hdr, tbl = [], []

    def lanes_pad(lst, filler):
        return list(lst) + [filler] * (G - len(lst))

    for st in builder.steps:
        op = st["op"]
        off = len(tbl)
        if op == OP_MULACC:
            T = st["T"]
            for t in range(T):
                for ln in lanes_pad(st["lanes"], None):
                    if ln is None or t >= len(ln["terms"]):
                        w = ZERO | (ZERO << 7) | (ZERO << 14) | (ZERO << 21)
                    else:
                        a1, a2, asub, b1, b2, bsub, neg, da = ln["terms"][t]
                        w = a1 | (a2 << 7) | (b1 << 14) | (b2 << 21) | (int(asub) << 28) | (int(bsub) << 29) | (int(neg) << 30) | (int(da) << 31)
                    tbl.append(w)
            sd = st.get("sd")
            for li, ln in enumerate(lanes_pad(st["lanes"], None)):
                if ln is None:
                    tbl.append(0)
                else:
                    assert -8 <= ln["alpha"] <= 7 and -8 <= ln["beta"] <= 7
                    ew = ln["dst"] | (1 << 7) | ((ln["alpha"] & 15) << 8) | ((ln["beta"] & 15) << 12) | (ln["e"] << 16)
                    if sd:
                        assert sd[li] < 64
                        ew |= (sd[li] << 23) | (1 << 29)
                    tbl.append(ew)
            no_a2 = no_b2 = 0
            for t in range(T):
                ts_ = [ln["terms"][t] for ln in st["lanes"] if t < len(ln["terms"])]
                if all(x[1] == ZERO and not x[2] for x in ts_):
                    no_a2 |= 1 << t
                if all(x[4] == ZERO and not x[5] for x in ts_):
                    no_b2 |= 1 << t
            no_neg = has_da = 0
            for t in range(T):
                if not any(ln["terms"][t][6] for ln in st["lanes"] if t < len(ln["terms"])):
                    no_neg |= 1 << t
                if any(ln["terms"][t][7] for ln in st["lanes"] if t < len(ln["terms"])):
                    has_da |= 1 << t
            # word 1: bit 0 epilogue, bit 1 companion (sum/difference) store, bits 4..15 term has no negation,
            # bits 16..27 term has a doubled A operand in some lane
            hdr += [op | (T << 8), int(st["epi"]) | (int(bool(sd)) << 1) | (no_neg << 4) | (has_da << 16), off, no_a2 | (no_b2 << 12)]
        elif op == OP_LIN:
            nt = st["nt"]
            for t in range(nt):
                for ln in lanes_pad(st["lanes"], None):
                    if ln is None or t >= len(ln[1]):
                        tbl.append(ZERO)
                    else:
                        s, c = ln[1][t]
                        tbl.append(s | ((c & 0xFF) << 8))
            for ln in lanes_pad(st["lanes"], None):
                tbl.append(0 if ln is None else (ln[0] | (1 << 7)))
            hdr += [op | (nt << 8), int(st.get("vred", True)), off, 0]
        elif op in (OP_GLOAD, OP_GSTORE):
            for ln in lanes_pad(st["lanes"], None):
                if ln is None:
                    tbl.append(0)
                else:
                    slot, idx = ln
                    if st["kind"] == K_LINE:      # inside a pair loop the kernel adds 6 x the loop's counter
                        idx = (0 if idx[0] == PAIR_VAR else idx[0]) * 6 + idx[1]
                    tbl.append(slot | (1 << 7) | (idx << 8))
            arg = st.get("advance", 0) if op == OP_GLOAD else int(st.get("check", False))
            hdr += [op | (st["kind"] << 8), arg, off, 0]
        elif op == OP_LOOP:
            hdr += [op, st["n"], 0, 0]
        elif op == OP_ENDLOOP:
            hdr += [op, 0, 0, 0]
        elif op == OP_PLOOP:
            hdr += [op, 0, 0, 0]
        elif op == OP_PENDLOOP:
            hdr += [op, st.get("advance", 0), 0, 0]
    hdr += [OP_END, 0, 0, 0]
    return hdr, tbl


# ---- direct hooks for the tower primitives (zkp_tower_op_batch): wire records in, wire record out -------------------------
TOWER_OPS = ("fp2_mul", "fp2_sqr", "fp6_mul", "fp6_sqr", "fp12_mul", "fp12_sqr", "fp12_014", "fp12_frob", "fp12_conj", "cyc_sqr")


def prog_tower(op):
    """one tower operation on wire records (72 u64; smaller tower elements occupy the leading coefficients, the rest
    of the result is zero): a = record of the check, b = record of check + chk_off.  Reference formulas: Fp2 mul / square
    src/fp2.rs:171-209, Fp6 mul / square src/fp6.rs:188-288, Fp12 mul / square / mul_by_014 / conjugate src/fp12.rs:99-210,
    TRUE Frobenius (src/fp12.rs:143-170 is built on the wrong Fp6 map), Granger-Scott cyclotomic squaring."""
    b = Builder()
    a = load_input(b, True)
    bb = load_input(b, True, wire_kind=K_WIRE2) if op in ("fp2_mul", "fp6_mul", "fp12_mul", "fp12_014") else None
    t = bb.slots if bb else b.alloc(12)          # 24 slots in all: results go to a's slots, the wire conversion to t

    def out_wire(v):
        raw1 = Lin.of(CONST_SLOT["RAW_ONE"])
        b.mulacc([{"dst": t[i], "bil": Bil([(v.lin(i), raw1, 1)])} for i in range(12)])
        b.gstore(K_WIRE, [(t[i], i) for i in range(12)])
        return b

    if op == "fp12_conj":
        return out_wire(a.conj())
    if op == "fp12_frob":
        r = b.frobenius(t, a, 1)
        t = a.slots
    elif op == "fp12_mul":
        r = b.fp12_mul(a, a, bb)
    elif op == "fp12_sqr":
        r = b.fp12_sqr(t, a)
        t = a.slots
    elif op == "fp12_014":
        r = b.fp12_mul_by_014(a, a, bb.fp2(0), bb.fp2(1), bb.fp2(2))
    elif op == "cyc_sqr":
        r = b.cyclotomic_sqr(t, a)
        t = a.slots
    else:
        if op == "fp2_mul":
            parts = list(b2_mul(a.fp2(0), bb.fp2(0)))
        elif op == "fp2_sqr":
            parts = list(b2_sqr(a.fp2(0)))
        else:
            x = a.fp6(0)
            parts = [c for pr in b6_mul(x, bb.fp6(0) if op == "fp6_mul" else x) for c in pr]
        d = a.slots
        b.mulacc([{"dst": d[i], "bil": bl} for i, bl in enumerate(parts)])
        b.lin([(d[i], Lin.of(ZERO)) for i in range(len(parts), 12)])
        r = V12(d)
    return out_wire(r)


# ---- the rest of SURVEY 8(a)'s tower functions as direct hooks (round 4): sparse Fp6 products, the nonresidue maps, Fp2 x Fp, and
# the three inversions.  An inversion is a plan of three steps: program A (wire -> state: the norm chain down to ONE Fp value in
# ST_N), the batched inversion kernel (ST_N -> ST_NINV; 0 gives 0), program B (state -> wire).  A non-invertible input (zero)
# therefore gives the zero record where the reference returns None (src/fp2.rs:278-296, src/fp6.rs:291-309, src/fp12.rs:186-190).
TOWER_OPS2 = ("fp2_nr", "fp2_mulfp", "fp6_by1", "fp6_by01", "fp6_nr")


def _tower_out_wire(b, v, t):
    raw1 = Lin.of(CONST_SLOT["RAW_ONE"])
    b.mulacc([{"dst": t[i], "bil": Bil([(v.lin(i), raw1, 1)])} for i in range(12)])
    b.gstore(K_WIRE, [(t[i], i) for i in range(12)])
    return b


def prog_tower2(op):
    """Fp2::mul_by_nonresidue src/fp2.rs:161-168, Mul<&Fp> for Fp2 :95-102 (b = the Fp value in the first 6 u64 of its record),
    Fp6::mul_by_1 src/fp6.rs:102-108 (b = c1), Fp6::mul_by_01 :110-127 (b = c0 | c1), Fp6::mul_by_nonresidue :130-141"""
    b = Builder()
    a = load_input(b, True)
    bb = load_input(b, True, wire_kind=K_WIRE2) if op in ("fp2_mulfp", "fp6_by1", "fp6_by01") else None
    t = bb.slots if bb else b.alloc(12)
    d = a.slots
    if op == "fp2_nr":
        parts = list(l2_xi(a.fp2(0)))
    elif op == "fp6_nr":
        parts = [c for pr in l6_mul_v(a.fp6(0)) for c in pr]
    else:
        parts = None
    if parts is not None:
        b.lin([(t[i], parts[i] if i < len(parts) else Lin.of(ZERO)) for i in range(12)])
        return _tower_out_wire(b, V12(t), d)
    if op == "fp2_mulfp":
        s = bb.lin(0)
        bil = [Bil([(a.lin(0), s, 1)]), Bil([(a.lin(1), s, 1)])]
    else:
        x = a.fp6(0)
        z = (Lin(), Lin())
        y = (z, bb.fp2(0), z) if op == "fp6_by1" else (bb.fp2(0), bb.fp2(1), z)
        bil = [c for pr in b6_mul(y, x) for c in pr]       # the sparse operand on the A side: merge_terms drops its zero forms
    b.mulacc([{"dst": d[i], "bil": bl} for i, bl in enumerate(bil)])
    b.lin([(d[i], Lin.of(ZERO)) for i in range(len(bil), 12)])
    return _tower_out_wire(b, V12(d), t)


def _fp6_inv_a(b, t):
    """t: 3-tuple of Lin pairs (an Fp6 value in slots).  Cofactors C (ST_C), N in Fp2 (ST_NC), n = |N|^2 (ST_N) - the chain of
    prog_fexp_a behind its t = a0^2 - v a1^2"""
    C0 = b2_sub(b2_mul(t[0], t[0]), b2_xi(b2_mul(t[1], t[2])))
    C1 = b2_sub(b2_xi(b2_mul(t[2], t[2])), b2_mul(t[0], t[1]))
    C2 = b2_sub(b2_mul(t[1], t[1]), b2_mul(t[0], t[2]))
    cs = b.alloc(6)
    b.mulacc([{"dst": cs[2 * j + c], "bil": (C0, C1, C2)[j][c]} for j in range(3) for c in range(2)])
    C = tuple(lin2(cs[2 * j], cs[2 * j + 1]) for j in range(3))
    Nb = b2_add(b2_xi(b2_add(b2_mul(t[1], C[2]), b2_mul(t[2], C[1]))), b2_mul(t[0], C[0]))
    ns = b.alloc(2)
    b.mulacc([{"dst": ns[c], "bil": Nb[c]} for c in range(2)])
    n1 = b.alloc(1)
    b.mulacc([{"dst": n1[0], "bil": Bil([(Lin.of(ns[0]), Lin.of(ns[0]), 1), (Lin.of(ns[1]), Lin.of(ns[1]), 1)])}])
    b.gstore(K_STATE, [(cs[i], ST_C + i) for i in range(6)] + [(ns[i], ST_NC + i) for i in range(2)] + [(n1[0], ST_N)])


def _fp6_inv_b(b):
    """-> slots of t^-1 = C N^-1 from ST_C, ST_NC, ST_NINV (the opening of prog_fexp_c)"""
    st = b.alloc(9)
    b.gload(K_STATE, [(st[i], ST_C + i) for i in range(6)] + [(st[6 + i], ST_NC + i) for i in range(2)] + [(st[8], ST_NINV)])
    C = tuple(lin2(st[2 * j], st[2 * j + 1]) for j in range(3))
    N = lin2(st[6], st[7])
    ninv = Lin.of(st[8])
    ni = b.alloc(2)
    b.mulacc([{"dst": ni[0], "bil": Bil([(N[0], ninv, 1)])}, {"dst": ni[1], "bil": Bil([(N[1], ninv, -1)])}])
    NI = lin2(ni[0], ni[1])
    ti = b.alloc(6)
    b.mulacc([{"dst": ti[2 * j + c], "bil": b2_mul(C[j], NI)[c]} for j in range(3) for c in range(2)])
    b.release(st)
    b.release(ni)
    return ti


def prog_tower_inv_a(which):
    """program A of Fp2::invert / Fp6::invert (Fp12::invert uses fexp_a_wire as it stands)"""
    b = Builder()
    a = load_input(b, True)
    if which == "fp2":
        n1 = b.alloc(1)
        b.mulacc([{"dst": n1[0], "bil": Bil([(a.lin(0), a.lin(0), 1), (a.lin(1), a.lin(1), 1)])}])
        b.gstore(K_STATE, [(a.slots[i], ST_NC + i) for i in range(2)] + [(n1[0], ST_N)])
    else:
        _fp6_inv_a(b, a.fp6(0))
    return b


def prog_tower_inv_b(which):
    """program B: fp2: (a0 ninv, -a1 ninv); fp6: C N^-1; fp12: conj(f) t^-1 = (a0 t^-1, -a1 t^-1) with f from ST_F"""
    b = Builder()
    if which == "fp2":
        st = b.alloc(3)
        b.gload(K_STATE, [(st[i], ST_NC + i) for i in range(2)] + [(st[2], ST_NINV)])
        r = b.alloc(12)
        b.mulacc([{"dst": r[0], "bil": Bil([(Lin.of(st[0]), Lin.of(st[2]), 1)])}, {"dst": r[1], "bil": Bil([(Lin.of(st[1]), Lin.of(st[2]), -1)])}])
        b.lin([(r[i], Lin.of(ZERO)) for i in range(2, 12)])
        b.release(st)
        return _tower_out_wire(b, V12(r), b.alloc(12))
    ti = _fp6_inv_b(b)
    if which == "fp6":
        r = b.alloc(12)
        b.lin([(r[i], Lin.of(ti[i]) if i < 6 else Lin.of(ZERO)) for i in range(12)])
        b.release(ti)
        return _tower_out_wire(b, V12(r), b.alloc(12))
    TI = tuple(lin2(ti[2 * j], ti[2 * j + 1]) for j in range(3))
    f = V12(b.alloc(12))
    b.gload(K_STATE, [(f.slots[i], ST_F + i) for i in range(12)])
    fc = f.conj()
    # in place: every lane reads f's slots before the step stores (one MULACC step)
    b.mulacc([{"dst": f.slots[i], "bil": bl} for i, bl in enumerate(flatten12(b6_mul(TI, fc.fp6(0)), b6_mul(TI, fc.fp6(1))))])
    b.release(ti)
    return _tower_out_wire(b, V12(f.slots), b.alloc(12))


def prog_tower_to_state(base=0):
    """wire record -> Montgomery limbs in state elements base..base+11 (0: input of a compressed squaring run; ST_SNAP:
    a snapshot whose z0, z1 the decompression kernels are to recover)"""
    b = Builder()
    f = load_input(b, True)
    b.gstore(K_STATE, [(f.slots[i], base + i) for i in range(12)])
    return b


def prog_tower_from_snap():
    """first snapshot area of the state buffer -> canonical wire record"""
    b = Builder()
    f = V12(b.alloc(12))
    b.gload(K_STATE, [(f.slots[i], ST_SNAP + i) for i in range(12)], kbound=Builder.K_REDUCED)
    finish_output(b, f, True, 0)
    return b


def prog_timing(T, epi, nloop=400, lin=False, forms="ab"):
    """synthetic timing program (not a meaningful computation): LOOP nloop { MULACC with T two-term products
    per lane (+ epilogue) } or { LIN with 3 terms }.  forms: "ab" both operands are two-slot forms, "b" only the B operand
    (the shape of the Fp12 products of the final exponentiation), "s" one-slot operands (the shape of the Miller loop's steps)"""
    b = Builder()
    v = b.alloc(12)
    w = b.alloc(12)
    b.lin([(v[i], Lin.of(CONST_SLOT["ONE"])) for i in range(12)])
    b.lin([(w[i], Lin.of(CONST_SLOT["F1_1_0"])) for i in range(12)])
    b.loop(nloop)
    if lin:
        b.lin([(v[i], Lin({v[i]: 1, w[(i + 1) % 12]: 1, w[(i + 5) % 12]: -1})) for i in range(12)])
    else:
        outs = []
        for i in range(12):
            fa = lambda t: Lin({v[(i + t) % 12]: 1, w[(i + t + 3) % 12]: 1}) if forms == "ab" else Lin({v[(i + t) % 12]: 1})
            fb = lambda t: Lin({w[(i + 2 * t) % 12]: 1, v[(i + t + 7) % 12]: -1 if (forms == "ab" or (i + t) % 2) else 1}) if forms in ("ab", "b") else Lin({w[(i + 2 * t) % 12]: 1})
            bil = Bil([(fa(t), fb(t), 1) for t in range(T)])
            o = {"dst": v[i], "bil": bil}
            if epi:
                o.update(alpha=3, beta=-2, e=v[i])
            outs.append(o)
        b.steps_before = len(b.steps)
        # bypass merge_terms so that exactly T two-term products are emitted
        lanes = []
        for o in outs:
            enc = []
            for a_, b_, c_ in o["bil"]:
                ea, eb = encode_form(a_), encode_form(b_)
                enc.append((ea[0], ea[1], ea[2], eb[0], eb[1], eb[2], bool(ea[3]) ^ bool(eb[3]), False))
            lanes.append({"dst": o["dst"], "terms": enc, "alpha": o.get("alpha", 1), "beta": o.get("beta", 0), "e": o.get("e", ZERO)})
        b.steps.append({"op": OP_MULACC, "T": T, "lanes": lanes, "epi": bool(epi), "sd": None})
    b.endloop()
    b.gstore(K_STATE, [(v[i], i) for i in range(12)], spill=True)   # timing only: values are meaningless
    return b


def prog_timing_cyc(companions, nloop=400):
    """timing only: nloop cyclotomic squarings in place, with or without the companion slots of a squaring run
    (the values are the state buffer's leftovers: meaningless, but the instruction stream is the real one)"""
    b = Builder()
    v = V12(b.alloc(12))
    b.gload(K_STATE, [(v.slots[i], i) for i in range(12)], kbound=Builder.K_VRED)
    sd = b.alloc(12) if companions else None
    if companions:
        v = b.cyclotomic_sqr(v, v, sd_out=sd)
    b.loop(nloop)
    v = b.cyclotomic_sqr(v, v, sd_out=sd)
    b.endloop()
    b.gstore(K_STATE, [(v.slots[i], i) for i in range(12)], spill=True)
    return b


def prog_timing_fill(nloop=400):
    """timing only: nloop x (spill 12 records to the state buffer, fill 12 records from another area)"""
    b = Builder()
    v = b.alloc(12)
    b.gload(K_STATE, [(v[i], i) for i in range(12)])
    b.loop(nloop)
    b.gstore(K_STATE, [(v[i], ST_SPILL + i) for i in range(12)], spill=True)
    b.gload(K_STATE, [(v[i], ST_SPILL + 12 + i) for i in range(12)])
    b.endloop()
    b.gstore(K_STATE, [(v[i], i) for i in range(12)], spill=True)
    return b


PROGRAMS = {
    "miller1_state": lambda: prog_miller(1, False),
    "miller1_wire": lambda: prog_miller(1, True),
    "miller2_state": lambda: prog_miller(2, False),
    "miller2_wire": lambda: prog_miller(2, True),
    "miller3_state": lambda: prog_miller(3, False),
    "miller3_wire": lambda: prog_miller(3, True),
    "miller4_state": lambda: prog_miller(4, False),
    "miller4_wire": lambda: prog_miller(4, True),
    "miller5_state": lambda: prog_miller(5, False),
    "miller5_wire": lambda: prog_miller(5, True),
    "miller6_state": lambda: prog_miller(6, False),
    "miller6_wire": lambda: prog_miller(6, True),
    "miller7_state": lambda: prog_miller(7, False),
    "miller7_wire": lambda: prog_miller(7, True),
    "miller8_state": lambda: prog_miller(8, False),
    "miller8_wire": lambda: prog_miller(8, True),
    "f12mul_pairs": prog_f12mul_pairs,
    "f12mul_state": lambda: prog_f12mul(False),
    "f12mul_wire": lambda: prog_f12mul(True),
    "fexp_a_state": lambda: prog_fexp_a(False),
    "fexp_a_wire": lambda: prog_fexp_a(True),
    "time_t1": lambda: prog_timing(1, False),
    "time_t3": lambda: prog_timing(3, False),
    "time_t3e": lambda: prog_timing(3, True),
    "time_t6": lambda: prog_timing(6, False),
    "time_t12": lambda: prog_timing(12, False),
    "time_lin": lambda: prog_timing(0, False, lin=True),
    "time_t6s": lambda: prog_timing(6, False, forms="s"),
    "time_t12s": lambda: prog_timing(12, False, forms="s"),
    "time_t12b": lambda: prog_timing(12, False, forms="b"),
    "time_cyc": lambda: prog_timing_cyc(False),
    "time_cycsd": lambda: prog_timing_cyc(True),
    "time_fill": prog_timing_fill,
}


for _op in TOWER_OPS:
    PROGRAMS["tw_" + _op] = (lambda op=_op: prog_tower(op))
for _op in TOWER_OPS2:
    PROGRAMS["tw_" + _op] = (lambda op=_op: prog_tower2(op))
for _w in ("fp2", "fp6"):
    PROGRAMS["tw_%s_inv_a" % _w] = (lambda w=_w: prog_tower_inv_a(w))
for _w in ("fp2", "fp6", "fp12"):
    PROGRAMS["tw_%s_inv_b" % _w] = (lambda w=_w: prog_tower_inv_b(w))
PROGRAMS["millern_state"] = lambda: prog_miller_n(False)
PROGRAMS["millern_wire"] = lambda: prog_miller_n(True)
PROGRAMS["tw_to_state"] = prog_tower_to_state
PROGRAMS["tw_to_snap"] = lambda: prog_tower_to_state(ST_SNAP)
PROGRAMS["tw_from_snap"] = prog_tower_from_snap
for _i in range(len(fexp_c_segments())):
    PROGRAMS["fexp_c%d" % _i] = (lambda i=_i: fexp_c_segments()[i])


def lds_config(peak, nconst):
    """0: LDS_SLOTS slots + all constants; 1: LDS_WIDE_SLOTS slots + LDS_WIDE_CONSTS constants (same LDS bytes);
    2: LDS_DEEP_SLOTS slots + LDS_DEEP_CONSTS constants"""
    if peak <= LDS_SLOTS:
        return 0
    if peak <= LDS_WIDE_SLOTS and nconst <= LDS_WIDE_CONSTS:
        return 1
    assert peak <= LDS_DEEP_SLOTS and nconst <= LDS_DEEP_CONSTS, (peak, nconst)
    return 2


def write_inc(path):
    lines = ["// GENERATED by tools/coopgen.py - do not edit.  Step programs of the lane-cooperative kernels.",
             "#pragma once", "#include <stdint.h>",
             "#define ZKP_COOP_G %d" % G, "#define ZKP_COOP_NSLOT_MAX %d" % NSLOT, "#define ZKP_COOP_NSLOT %d" % LDS_SLOTS,
             "#define ZKP_COOP_NCONST %d" % N_CONST, "#define ZKP_COOP_WIDE_NSLOT %d" % LDS_WIDE_SLOTS, "#define ZKP_COOP_WIDE_NCONST %d" % LDS_WIDE_CONSTS,
             "#define ZKP_COOP_DEEP_NSLOT %d" % LDS_DEEP_SLOTS, "#define ZKP_COOP_DEEP_NCONST %d" % LDS_DEEP_CONSTS,
             "#define ZKP_COOP_ST_SIZE %d" % ST_SIZE,
             "#define ZKP_COOP_ST_G %d" % ST_G, "#define ZKP_COOP_ST_N %d" % ST_N, "#define ZKP_COOP_ST_NINV %d" % ST_NINV,
             "#define ZKP_COOP_ST_SNAP %d" % ST_SNAP, "#define ZKP_COOP_ST_KN %d" % ST_KN, "#define ZKP_COOP_ST_KNINV %d" % ST_KNINV,
             "#define ZKP_COOP_NLINES %d" % n_line_steps(),
             "#define ZKP_COOP_VRED_C %d" % VRED_C, "#define ZKP_COOP_VRED_SHIFT_IN %d" % VRED_SHIFT_IN,
             "#define ZKP_COOP_VRED_SHIFT_OUT %d" % VRED_SHIFT_OUT,
             "#define ZKP_COOP_P_BAL %s" % ", ".join(str(x) for x in P_BAL)]
    # constants region: Montgomery limbs (balanced), 16 dwords per constant
    rows = [const_limbs(name, val) for name, val in CONSTS]
    lines.append("static const int32_t ZKP_COOP_CONSTS[%d][16] = {" % N_CONST)
    for r in rows:
        lines.append("  {" + ", ".join(str(x) for x in r + [0, 0]) + "},")
    lines.append("};")
    names = sorted(PROGRAMS)
    lines.append("enum { " + ", ".join("ZKP_PROG_%s = %d" % (n.upper(), i) for i, n in enumerate(names)) + ", ZKP_PROG_COUNT = %d };" % len(names))
    meta = []
    for n in names:
        b = PROGRAMS[n]()
        hdr, tbl = encode(b)
        lines.append("static const uint32_t ZKP_PROG_%s_HDR[%d] = {%s};" % (n.upper(), len(hdr), ",".join(str(x) for x in hdr)))
        lines.append("static const uint32_t ZKP_PROG_%s_TBL[%d] = {%s};" % (n.upper(), max(1, len(tbl)), ",".join(str(x) for x in tbl) if tbl else "0"))
        used = [CONST_BASE]
        for st in b.steps:
            if st["op"] == OP_MULACC:
                for ln in st["lanes"]:
                    for tm in ln["terms"]:
                        used += [tm[0], tm[1], tm[3], tm[4]]
                    used.append(ln["e"])
            elif st["op"] == OP_LIN:
                for _, terms in st["lanes"]:
                    used += [sl for sl, _ in terms]
        nconst = max(x for x in used if x >= CONST_BASE) - CONST_BASE + 1
        wide = lds_config(b.peak, nconst)
        meta.append((n, len(hdr), len(tbl), b.peak, nconst, wide))
    lines.append("// wide: 1 = the program runs with ZKP_COOP_WIDE_NSLOT slots and ZKP_COOP_WIDE_NCONST constants instead of ZKP_COOP_NSLOT / ZKP_COOP_NCONST, 2 = with ZKP_COOP_DEEP_NSLOT / ZKP_COOP_DEEP_NCONST")
    lines.append("struct ZkpProgDesc { const uint32_t* hdr; uint32_t n_hdr; const uint32_t* tbl; uint32_t n_tbl; uint32_t nslot; uint32_t nconst; uint32_t wide; };")
    lines.append("static const ZkpProgDesc ZKP_PROGS[ZKP_PROG_COUNT] = {")
    for n, nh, nt, peak, nconst, wide in meta:
        lines.append("  {ZKP_PROG_%s_HDR, %d, ZKP_PROG_%s_TBL, %d, %d, %d, %d}," % (n.upper(), nh, n.upper(), max(1, nt), peak, nconst, wide))
    lines.append("};")
    # phase C of the final exponentiation: step programs alternating with the compressed squaring runs and the decompression
    lines.append("enum { ZKP_PLAN_PROG = %d, ZKP_PLAN_KSQ = %d, ZKP_PLAN_KDEC_A = %d, ZKP_PLAN_INV = %d, ZKP_PLAN_KDEC_B = %d };"
                 % (PLAN_PROG, PLAN_KSQ, PLAN_KDEC_A, PLAN_INV, PLAN_KDEC_B))
    lines.append("// PROG: a = program; KSQ: a = input element, b = first snapshot element, c = squarings, mask = snapshot bits;")
    lines.append("// KDEC_A: a = snapshots, b = count, c = |D|^2 planes; INV: a = input planes, b = output planes, c = count; KDEC_B: a, b, c = inverse planes")
    lines.append("struct ZkpPlanStep { uint32_t kind, a, b, c; uint64_t mask; };")
    rows, seg = [], 0
    for st in fexp_c_plan():
        if st[0] == "prog":
            rows.append("{ZKP_PLAN_PROG, ZKP_PROG_FEXP_C%d, 0, 0, 0}" % seg)
            seg += 1
        elif st[0] == PLAN_KSQ:
            rows.append("{ZKP_PLAN_KSQ, %d, %d, %d, 0x%xull}" % st[1:])
        else:
            rows.append("{%d, %d, %d, %d, 0}" % st)
    lines.append("#define ZKP_FEXP_C_PLAN_LEN %d" % len(rows))
    lines.append("static const ZkpPlanStep ZKP_FEXP_C_PLAN[ZKP_FEXP_C_PLAN_LEN] = {" + ", ".join(rows) + "};")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return meta


if __name__ == "__main__":
    out = os.path.join(ROOT, "zkvm_pairings_amd", "csrc", "zkp_coop_prog.inc")
    for n, nh, nt, peak, nconst, wide in write_inc(out):
        print("%-16s steps=%4d table_words=%6d peak_slots=%d consts=%d%s" % (n, nh // 4, nt, peak, nconst, " (wide)" if wide else ""))
    print("wrote", out)
