#!/usr/bin/env python3
"""The Jacobian doubling / mixed-addition steps of the subgroup checks (G1Affine::is_valid / G2Affine::is_valid, reference
src/g1.rs:49-62, 95-115, src/g2.rs:57-69, 109-170) as hand-allocated gfx950 asm blocks -> csrc/zkp_valid_steps.inc.

Round 3 left k_g1_valid28 / k_g2_valid28 as compiled by-value C++ (schoolbook 196-multiply-add blocks, a full reduction and a value
renormalisation per product, 2 waves per SIMD): 0.37 / 0.46 of the integer roofline.  Here a step is ONE register allocation on the
machinery of tools/coopasm.py (Karatsuba product block `kterm`, fold + Montgomery reduction + limb extraction `tail`) and
tools/prepasm.py (lane-pair Fp2 forms):

  doubling (a = 0), both curves:   A = X^2, B = Y^2, Z' = (2Y) Z, S = 4 X B, M = 3A,
                                   X' = M^2 - 2S,   Y' = M (S - X') - 8 B^2   [Y' is ONE lazy accumulation: one reduction]
     = dbl-2009-l with D = 2((X+B)^2 - A - C) replaced by 4 X B, so that C = B^2 is never a value of its own: 7 products and
     6 reductions in Fp (G1), 10 product blocks and 6 reductions per lane in Fp2 (G2) - the compiled step had 7 + 7 / 9 + 7 on
     schoolbook blocks, plus three value renormalisations.  Constant factors sit on operands (S = (2X)(2 Y^2) comes out of its
     reduction as 4 X Y^2), so the doubling renormalises nothing; in Fp the squares are 105-multiply-add blocks.
  mixed addition (Z2 = 1; madd-2004-hmv shape, no doublings of intermediate values):
                                   ZZ = Z^2, U2 = qx ZZ, S2 = qy Z ZZ, H = U2 - X, r = S2 - Y, Z' = Z H, HH = H^2, HHH = H HH,
                                   V = X HH, X' = r^2 - HHH - 2V, Y' = r (V - X') - Y HHH   [Y': one reduction]

NO exceptional case is handled here - and none needs to be: infinity (Z = 0), P + P, P - P and a point of order two all send Z'
to 0 (Z' = 2YZ resp. Z' = Z H), and Z = 0 is absorbing under both steps, so the chain's final Z is 0 mod p exactly when an
exceptional case occurred somewhere.  The kernels mark those points and the generic compiled kernels (every case handled, round 3)
redo them; a point of the prime-order subgroup never takes that route (the chain's scalars are below r).

G1: one lane per point, Fp values, five 14-register value blocks + the product set = 162 VGPRs -> 3 waves per SIMD; the affine
point (and one parked value in the addition) live in LDS.  G2: two lanes per point (lane parity = Fp2 coefficient, as
k_prep_lines), the register layout of tools/prepasm.py, 2 waves per SIMD.
Gate: tools/asmemu.py against big-integer formulas with the limb / value bounds of the next step (tests/test_validasm.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import coopasm  # noqa: E402
import prepasm  # noqa: E402
from coopasm import NL, kterm, tail, p_balanced, VRED_C, VRED_SHIFT_IN, VRED_SHIFT_OUT  # noqa: E402


# ---------------------------------------------------------------------------------------------------------------- G1: Fp, one lane per point
class FpStep:
    def __init__(self, vb=6):
        self.lines = []
        v = vb
        assert v % 2 == 0
        self.vb = vb
        self.X = v; v += 14
        self.Y = v; v += 14
        self.Z = v; v += 14
        self.V0 = v; v += 14
        self.V1 = v; v += 14
        self.D = v; v += 14
        self.ACC = {}
        for k in range(27):
            if k == 13:
                continue
            self.ACC[k] = v; v += 2
        self.MID = {}
        for k in range(13):
            self.MID[k] = v; v += 2
        self.vend = v
        assert self.vend <= 168, self.vend          # three waves per SIMD
        # temporaries: the renormalisation's live in accumulator registers (never used while an accumulation is in flight); the LDS
        # address lives in the difference block, which is free BETWEEN the products of a lazy accumulation too
        self.vq, self.vt, self.vlds = self.ACC[0], self.ACC[0] + 1, self.D + 13
        self.vc = self.D                       # carries of the one-pass normalisation
        s = 36
        self.sb = s
        self.sC = s; s += 2                    # carry-out sink of the multiply-adds
        self.sPB = s; s += NL                  # balanced limbs of p
        self.send = s
        self.budget = 0

    e = coopasm.Asm.e
    mad = coopasm.Asm.mad
    op2 = prepasm.Prep.op2
    add = prepasm.Prep.add
    sub = prepasm.Prep.sub
    neg = prepasm.Prep.neg
    shl = prepasm.Prep.shl
    times3 = prepasm.Prep.times3
    mov = prepasm.Prep.mov
    norm = prepasm.Prep.norm
    vred = prepasm.Prep.vred
    prod = prepasm.Prep.prod

    def prologue(self):
        for i, v in enumerate(p_balanced()):
            self.e("s_mov_b32 s%d, 0x%x" % (self.sPB + i, v & 0xffffffff))

    def shl_add(self, d, a, k, b):
        """d = (a << k) + b, limb-wise"""
        for i in range(NL):
            self.e("v_lshl_add_u32 v%d, v%d, %d, v%d" % (d + i, a + i, k, b + i))

    def mul(self, dst, a, b, la, lb):
        self.prod(a, b, True, la, lb)
        tail(self, dst)

    def sqr(self, dst, a, la, P1=None, P2=None):
        """dst = a^2 (or s a^2 with P1 = s a, P2 = 2 s a; la then bounds the larger factor's limbs) - 105 multiply-adds, fold-free reduction"""
        assert la * la <= 30 and la <= 7, "column budget / int32 of the doubled limbs"
        coopasm.ksqr_plain(self, a, P1, P2)
        tail(self, dst, fold=False)

    def sqr_acc(self, a, P1, P2, la):
        """the running lazy accumulation += s a^2 (P1 = s a, P2 = 2 s a; la bounds |2 s| in limb units); P2 is destroyed"""
        self.budget += la
        assert self.budget <= 30 and la <= 7, "column budget"
        coopasm.ksqr_k(self, a, P1, P2, False)

    def lds_addr(self):
        self.e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % self.vlds)
        self.e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (self.vlds, self.vlds))
        self.e("v_lshlrev_b32 v%d, 4, v%d" % (self.vlds, self.vlds))

    def lds_read(self, dst, slot, wait=True):
        """LDS slot (limb quad q of slot v at (v * 4 + q) * 1024 + lane * 16: the layout of valid_park) -> 14 registers.
        wait=False: the caller places lds_wait() in front of the first use (the transfer then runs under the instructions between)"""
        self.lds_addr()
        for q in range(3):
            self.e("ds_read_b128 v[%d:%d], v%d offset:%d" % (dst + 4 * q, dst + 4 * q + 3, self.vlds, (slot * 4 + q) * 1024))
        self.e("ds_read_b64 v[%d:%d], v%d offset:%d" % (dst + 12, dst + 13, self.vlds, (slot * 4 + 3) * 1024))
        if wait:
            self.lds_wait()

    def lds_write(self, src, slot, wait=True):
        """wait=False: the caller guarantees an lds_wait() in front of the next write to the source registers"""
        self.lds_addr()
        for q in range(3):
            self.e("ds_write_b128 v%d, v[%d:%d] offset:%d" % (self.vlds, src + 4 * q, src + 4 * q + 3, (slot * 4 + q) * 1024))
        self.e("ds_write_b64 v%d, v[%d:%d] offset:%d" % (self.vlds, src + 12, src + 13, (slot * 4 + 3) * 1024))
        if wait:
            self.lds_wait()    # the source registers are about to be reused

    def lds_wait(self):
        self.e("s_waitcnt lgkmcnt(0)")


QX, QY, PARK = 0, 1, 2      # LDS slots of the G1 kernel


def g1_dbl(vb=6):
    """the doubling on (X, Y, Z) with every constant factor moved onto an operand, so that no value needs a renormalisation:
        Z' = (2Y) Z,  B2 = 2 Y^2,  A = X^2,  S = (2X) B2 = 4 X Y^2,  M = 3A,  X' = M^2 - 2S,  Y' = M (S - X') - 2 B2^2
    Squares are 105-multiply-add blocks (ksqr_plain) under a fold-free reduction; Y' is one lazy accumulation of a Karatsuba product
    and a Karatsuba-layout square.  X' leaves with limbs of 3 units and |value| <= 2.2 p, Y' and Z' as reduced products."""
    g = FpStep(vb)
    X, Y, Z, V0, V1, D = g.X, g.Y, g.Z, g.V0, g.V1, g.D
    g.prologue()
    g.shl(V1, Y, 1)
    g.mul(Z, V1, Z, 2, 1)                 # Z' = (2Y) Z
    g.shl(D, Y, 2)
    g.sqr(V1, Y, 2, P1=V1, P2=D)          # B2 = 2 Y^2  (P1 = 2Y, P2 = 4Y)
    g.sqr(V0, X, 3)                       # A (X enters with limbs of up to 3 units)
    g.shl(Y, X, 1)
    g.mul(Y, Y, V1, 6, 1)                 # S = (2X) B2
    g.times3(V0, V0)                      # M = 3A: limbs 3 units, |value| <= 3.2 p
    g.sqr(X, V0, 3)                       # M^2
    g.neg(D, Y)
    g.shl_add(X, D, 1, X)                 # X' = M^2 - 2S
    g.sub(Y, Y, X)                        # S - X': limbs 4 units, |value| <= 3.2 p
    g.prod(V0, Y, True, 3, 4)
    g.neg(Y, V1)
    g.shl(Y, Y, 1)                        # P1 = -2 B2
    g.shl(V0, Y, 1)                       # P2 = -4 B2
    g.sqr_acc(V1, Y, V0, 4)               # - 2 B2^2 in the Karatsuba layout
    tail(g, Y)                            # Y'
    return g


def g1_madd(vb=6):
    """(X, Y, Z) += (qx, qy), the affine point in LDS slots QX, QY; one value parked in slot PARK.  X' leaves normalised
    (limbs of one unit, |value| <= 3.3 p)."""
    g = FpStep(vb)
    X, Y, Z, V0, V1, D = g.X, g.Y, g.Z, g.V0, g.V1, g.D
    g.prologue()
    g.lds_write(Y, PARK)                  # Y is not needed before r = S2 - Y: its block holds the LDS operands meanwhile
    g.sqr(V0, Z, 1)                       # ZZ
    g.mul(V1, Z, V0, 1, 1)                # ZZZ
    g.lds_read(Y, QX)
    g.mul(V0, Y, V0, 1, 1)                # U2 = qx ZZ
    g.sub(V0, V0, X)                      # H (limbs 4 units)
    g.lds_read(Y, QY)
    g.mul(V1, Y, V1, 1, 1)                # S2 = qy ZZZ
    g.lds_read(Y, PARK)
    g.sub(V1, V1, Y)                      # r (limbs 2)
    g.mul(Z, Z, V0, 1, 4)                 # Z' = Z H
    g.lds_write(Z, PARK)                  # done: its block is the temporary of the rest
    g.sqr(Z, V0, 4)                       # HH
    g.mul(V0, V0, Z, 4, 1)                # HHH
    g.mul(Z, X, Z, 3, 1)                  # V = X HH
    g.sqr(X, V1, 2)                       # r^2
    g.sub(X, X, V0)
    g.neg(D, Z)
    g.shl_add(X, D, 1, X)                 # X' = r^2 - HHH - 2V: limbs 4 units, |value| <= 3.3 p
    g.norm(X)
    g.sub(Z, Z, X)                        # V - X' (limbs 2)
    g.prod(V1, Z, True, 2, 2)
    g.neg(Y, Y)
    g.prod(Y, V0, False, 1, 1)
    tail(g, Y)                            # Y' = r (V - X') - Y HHH
    g.lds_read(Z, PARK)
    return g


# ---------------------------------------------------------------------------------------------------------------- G2: Fp2 on a lane pair
def g2_dbl(vb=6):
    """the same doubling on a lane pair.  M = 3 X^2 comes straight out of the squaring's reduction (the factor sits on one operand
    form), S = (4X) Y^2 out of its product, so X' = M^2 - 2S needs one carry pass (its squaring's operand forms double the limbs)
    and no value renormalisation: X' leaves with limbs of one unit and |value| <= 2.2 p."""
    g = prepasm.Prep(vb)
    X, Y, Z = g.X, g.Y, g.W
    V0, V1, V2, _ = g.V
    T0, T1, T2 = g.T
    g.prologue()
    g.shl(V1, Y, 1)
    g.mul(Z, V1, Z, 2, 1)                 # Z' = (2Y) Z
    g.sqr_forms(X)                        # T1 = x, T0 = y of X^2's coefficient
    g.times3(T0, T0)
    g.prod(T1, T0, True, 2, 6)
    tail(g, V0)                           # M = 3 X^2, a reduced value
    g.sqr(V1, Y)                          # B
    g.shl(Y, X, 2)
    g.mul(Y, Y, V1, 4, 1)                 # S = (4X) B
    g.sqr(X, V0)                          # M^2
    g.neg(T2, Y)
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (X + i, T2 + i, X + i))      # X' = M^2 - 2S (limbs 3 units)
    g.norm(X)
    g.sub(Y, Y, X)                        # S - X' (limbs 2 units, |value| <= 3.3 p)
    g.mul_acc(V0, Y, True, 1, 2)          # M (S - X'): two product blocks
    g.sqr_forms(V1)                       # T1 = x, T0 = y of B^2's coefficient
    g.norm(T1)                            # -8 x y = (-4 norm(x)) (2 y): limbs 4 and 4 units
    g.shl(T1, T1, 2)
    g.neg(T1, T1)
    g.shl(T0, T0, 1)
    g.prod(T1, T0, False, 4, 4)
    tail(g, Y)                            # Y' = M (S - X') - 8 B^2
    return g


def g2_madd(vb=6):
    """(X, Y, Z) += (qx, qy): this lane's coefficient of the affine point in LDS values 0, 1 of the Prep park"""
    g = prepasm.Prep(vb)
    X, Y, Z = g.X, g.Y, g.W
    V0, V1, V2, _ = g.V
    T0, T1, T2 = g.T
    g.prologue()
    g.sqr(V0, Z)                          # ZZ
    g.mul(V1, Z, V0, 1, 1)                # ZZZ
    g.park_read(V2, 0)
    g.e("s_waitcnt lgkmcnt(0)")
    g.mul(V0, V2, V0, 1, 1)               # U2 = qx ZZ
    g.sub(V0, V0, X)                      # H
    g.park_read(V2, 1)
    g.e("s_waitcnt lgkmcnt(0)")
    g.mul(V1, V2, V1, 1, 1)               # S2 = qy ZZZ
    g.sub(V1, V1, Y)                      # r
    g.mul(Z, Z, V0, 1, 2)                 # Z' = Z H
    g.sqr(V2, V0, 2)                      # HH
    g.mul(V0, V0, V2, 2, 1)               # HHH
    g.mul(V2, X, V2, 1, 1)                # V = X HH
    g.sqr(X, V1, 2)                       # r^2
    g.sub(X, X, V0)
    g.neg(T2, V2)
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (X + i, T2 + i, X + i))      # X' = r^2 - HHH - 2V
    g.norm(X)
    g.vred(X)
    g.sub(V2, V2, X)                      # V - X'
    g.mul_acc(V1, V2, True, 2, 2)
    g.neg(Y, Y)
    g.mul_acc(Y, V0, False, 1, 1)
    tail(g, Y)                            # Y' = r (V - X') - Y HHH
    return g


# ---------------------------------------------------------------------------------------------------------------- G2 at three waves per SIMD
class Fp2Step3(FpStep):
    """Fp2 on a lane pair in the register budget of three waves per SIMD: five 14-register value blocks R0..R4 + the Karatsuba product
    set (162 VGPRs).  An Fp2 product needs four operand blocks (a, b and the pair partner's a', b'), so between steps only Y stays
    in registers (R1): X lives in LDS slot 0, Z in slot 1, slot 2 is the addition's parking place (12 KB per wavefront, twelve
    wavefronts per CU); the affine point of the additions comes from a scratch buffer in global memory (five loads per chain)."""

    def __init__(self, vb=6):
        super().__init__(vb)
        self.R = [self.X, self.Y, self.Z, self.V0, self.V1]
        s = self.send
        self.sEX = s; s += 2
        self.sM0 = s; s += 2
        self.sM1 = s; s += 2
        self.sQ = s; s += 2                   # running pointer of the scratch loads
        self.send = s

    def prologue(self):
        super().prologue()
        self.e("s_mov_b64 s[%d:%d], exec" % (self.sEX, self.sEX + 1))
        for (sr, m) in ((self.sM0, 0x55555555), (self.sM1, 0xaaaaaaaa)):
            self.e("s_mov_b32 s%d, 0x%x" % (sr, m))
            self.e("s_mov_b32 s%d, 0x%x" % (sr + 1, m))
            self.e("s_and_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (sr, sr + 1, sr, sr + 1, self.sEX, self.sEX + 1))

    def all_lanes(self):
        self.e("s_mov_b64 exec, s[%d:%d]" % (self.sEX, self.sEX + 1))

    def lanes(self, c):
        m = self.sM1 if c else self.sM0
        self.e("s_mov_b64 exec, s[%d:%d]" % (m, m + 1))

    def swap(self, d, a):
        self.e("s_nop 1")
        for i in range(NL):
            self.e("v_mov_b32_dpp v%d, v%d %s" % (d + i, a + i, prepasm.QP))

    def forms(self, a, x, y):
        """operand forms of the squaring of a on this lane: c = 0: x = a' + a, y = a - a';  c = 1: x = 2 a', y = a"""
        self.swap(y, a)
        self.lanes(0)
        self.add(x, y, a)
        self.sub(y, a, y)
        self.lanes(1)
        self.add(x, y, y)
        self.mov(y, a)
        self.all_lanes()

    def mul_acc(self, a, b, ra, rb, first, la, lb, restore, b_in_flight=False):
        """the lazy accumulation (+)= coefficient c of a b: c = 0: a b - a' b', c = 1: a b' + a' b.  ra, rb: blocks for the partner's
        values.  On c = 1 lanes b and b' change places for the products (v_swap); restore puts b back.
        b_in_flight: b is still arriving from LDS - the wait sits behind the first operand's exchange"""
        self.swap(ra, a)
        if b_in_flight:
            self.lds_wait()
        self.swap(rb, b)
        self.lanes(1)
        for i in range(NL):
            self.e("v_swap_b32 v%d, v%d" % (b + i, rb + i))
        self.lanes(0)
        self.neg(ra, ra)
        self.all_lanes()
        self.prod(a, b, first, la, lb)
        self.prod(ra, rb, False, la, lb)
        if restore:
            self.lanes(1)
            for i in range(NL):
                self.e("v_swap_b32 v%d, v%d" % (b + i, rb + i))
            self.all_lanes()

    def gload(self, dst, v):
        """value v (0: qx, 1: qy) of this lane from the scratch buffer: quad q at qbase + ((v * 4 + q) * lanes + lane) * 16"""
        self.e("s_mul_i32 s%d, %%[qstride], %d" % (self.sQ, 4 * v))
        self.e("s_mul_hi_u32 s%d, %%[qstride], %d" % (self.sQ + 1, 4 * v))
        self.e("s_add_u32 s%d, s%d, %%[qlo]" % (self.sQ, self.sQ))
        self.e("s_addc_u32 s%d, s%d, %%[qhi]" % (self.sQ + 1, self.sQ + 1))
        for q in range(4):
            if q < 3:
                self.e("global_load_dwordx4 v[%d:%d], %%[qoff], s[%d:%d]" % (dst + 4 * q, dst + 4 * q + 3, self.sQ, self.sQ + 1))
                self.e("s_add_u32 s%d, s%d, %%[qstride]" % (self.sQ, self.sQ))
                self.e("s_addc_u32 s%d, s%d, 0" % (self.sQ + 1, self.sQ + 1))
            else:
                self.e("global_load_dwordx2 v[%d:%d], %%[qoff], s[%d:%d]" % (dst + 12, dst + 13, self.sQ, self.sQ + 1))
        self.e("s_waitcnt vmcnt(0)")


LX, LZ, LT = 0, 1, 2        # LDS slots of the three-wave G2 kernel: X, Z, the addition's parking place


def g2_dbl3(vb=6):
    """g2_dbl's doubling in five register blocks; in / out: Y in R1, X in LDS slot LX, Z in slot LZ"""
    g = Fp2Step3(vb)
    R0, R1, R2, R3, R4 = g.R
    D = g.D
    g.prologue()
    g.lds_read(R2, LZ, wait=False)        # Z arrives under the forming of 2Y and its exchange
    g.shl(R3, R1, 1)
    g.mul_acc(R3, R2, R4, R0, True, 2, 1, False, b_in_flight=True)
    tail(g, R2)                           # Z' = (2Y) Z
    g.lds_write(R2, LZ, wait=False)       # (R2 is next written by M's reduction, behind the wait for X)
    g.forms(R1, R4, R3)
    g.lds_read(R1, LX, wait=False)        # Y is dead behind its operand forms: X arrives under B's product and reduction
    g.prod(R4, R3, True, 2, 2)
    tail(g, R0)                           # B = Y^2
    g.lds_wait()
    g.forms(R1, R4, R3)
    g.times3(R3, R3)
    g.prod(R4, R3, True, 2, 6)
    tail(g, R2)                           # M = 3 X^2
    g.shl(R1, R1, 2)
    g.mul_acc(R1, R0, R3, R4, True, 4, 1, True)
    tail(g, R1)                           # S = (4X) B   (B restored)
    g.forms(R2, R4, R3)
    g.prod(R4, R3, True, 2, 2)
    tail(g, R3)                           # M^2
    g.neg(D, R1)
    g.shl_add(R3, D, 1, R3)               # X' = M^2 - 2S
    g.norm(R3)
    g.sub(R1, R1, R3)                     # S - X'
    g.lds_write(R3, LX)
    g.mul_acc(R2, R1, R3, R4, True, 1, 2, False)
    g.forms(R0, R4, R3)
    g.norm(R4)
    g.shl(R4, R4, 2)
    g.neg(R4, R4)
    g.shl(R3, R3, 1)
    g.prod(R4, R3, False, 4, 4)
    tail(g, R1)                           # Y' = M (S - X') - 8 B^2
    return g


def g2_madd3(vb=6):
    """g2_madd's mixed addition in five register blocks; the affine point comes from the scratch buffer"""
    g = Fp2Step3(vb)
    R0, R1, R2, R3, R4 = g.R
    D = g.D
    g.prologue()
    g.lds_write(R1, LT)                   # Y parked
    g.lds_read(R2, LZ)
    g.forms(R2, R4, R3)
    g.prod(R4, R3, True, 2, 2)
    tail(g, R0)                           # ZZ
    g.mul_acc(R2, R0, R3, R4, True, 1, 1, True)
    tail(g, R3)                           # ZZZ  (ZZ restored)
    g.gload(R1, 0)
    g.mul_acc(R1, R0, R2, R4, True, 1, 1, False)
    tail(g, R0)                           # U2 = qx ZZ
    g.lds_read(R1, LX)
    g.sub(R0, R0, R1)                     # H
    g.gload(R1, 1)
    g.mul_acc(R1, R3, R2, R4, True, 1, 1, False)
    tail(g, R3)                           # S2 = qy ZZZ
    g.lds_read(R1, LT)
    g.sub(R3, R3, R1)                     # r
    g.lds_read(R1, LZ)
    g.mul_acc(R1, R0, R2, R4, True, 1, 2, True)
    tail(g, R1)                           # Z' = Z H   (H restored)
    g.lds_write(R1, LZ)
    g.forms(R0, R2, R1)
    g.prod(R2, R1, True, 4, 4)
    tail(g, R1)                           # HH
    g.mul_acc(R0, R1, R2, R4, True, 2, 1, True)
    tail(g, R0)                           # HHH  (HH restored)
    g.lds_read(R2, LX)                    # X
    g.lds_write(R0, LX)                   # HHH parked in X's slot
    g.mul_acc(R2, R1, R4, R0, True, 1, 1, False)
    tail(g, R1)                           # V = X HH
    g.forms(R3, R2, R0)
    g.prod(R2, R0, True, 4, 4)
    tail(g, R2)                           # r^2
    g.lds_read(R0, LX)                    # HHH
    g.sub(R2, R2, R0)
    g.neg(D, R1)
    g.shl_add(R2, D, 1, R2)               # X' = r^2 - HHH - 2V
    g.norm(R2)
    g.vred(R2)
    g.sub(R1, R1, R2)                     # V - X'
    g.lds_write(R2, LX)
    g.mul_acc(R3, R1, R2, R4, True, 2, 2, False)
    g.lds_read(R3, LT)                    # Y
    g.neg(R3, R3)
    g.mul_acc(R3, R0, R2, R4, False, 1, 1, False)
    tail(g, R1)                           # Y' = r (V - X') - Y HHH
    return g


def _emit(f, name, g, comment):
    n = sum(1 for l in g.lines if not l.endswith(":"))
    f.write("// %s: %d instructions; VGPRs v%d..v%d, SGPRs s%d..s%d.\n" % (comment, n, g.vb, g.vend - 1, g.sb, g.send - 1))
    f.write("#define %s \\\n" % name)
    for l in g.lines:
        f.write('    "%s\\n\\t" \\\n' % l)
    f.write('    ""\n')
    return n


def write_inc(path, vb=6):
    g1d, g1a, g2d, g2a = g1_dbl(vb), g1_madd(vb), g2_dbl(vb), g2_madd(vb)
    with open(path, "w") as f:
        f.write("// GENERATED by tools/validasm.py - do not edit.  Jacobian doubling / mixed addition of the subgroup checks as inline-asm blocks.\n")
        f.write("#pragma once\n")
        counts = [_emit(f, "ZKP_G1_DBL_ASM", g1d, "G1 doubling (one lane per point)"),
                  _emit(f, "ZKP_G1_MADD_ASM", g1a, "G1 mixed addition (affine point in LDS slots 0, 1; slot 2 parks a value)"),
                  _emit(f, "ZKP_G2_DBL_ASM", g2d, "G2 doubling (two lanes per point)"),
                  _emit(f, "ZKP_G2_MADD_ASM", g2a, "G2 mixed addition (affine point in LDS values 0, 1)")]
        for nm, g, z in (("G1", g1d, g1d.Z), ("G2", g2d, g2d.W)):
            io = []
            for arg, base in (("x", g.X), ("y", g.Y), ("z", z)):
                io += ['"+{v%d}"((%s)[%d])' % (base + i, arg, i) for i in range(NL)]
            f.write("// in/out: the Jacobian point (this lane's coefficient for G2), reduced values\n")
            f.write("#define ZKP_%s_STEP_IO(x, y, z) " % nm + ", ".join(io) + "\n")
            first = z + 14
            f.write("#define ZKP_%s_STEP_CLOBBERS " % nm + ", ".join('"v%d"' % v for v in range(first, g.vend)) + ", "
                    + ", ".join('"s%d"' % s for s in range(g.sb, g.send)) + ', "vcc", "scc", "memory"\n')
        f.write("#define ZKP_G1_STEP_VGPR_END %d\n#define ZKP_G2_STEP_VGPR_END %d\n" % (g1d.vend, g2d.vend))
        # the three-wave G2 variant: Y in registers (in / out), X and Z in LDS slots 0 and 1, slot 2 parks; the affine point comes from
        # a scratch buffer: [qlo]:[qhi] its base (two SGPR inputs), [qstride] the bytes of one limb-quad plane, [qoff] the lane's offset
        g3d, g3a = g2_dbl3(vb), g2_madd3(vb)
        counts += [_emit(f, "ZKP_G2W3_DBL_ASM", g3d, "G2 doubling, three waves per SIMD (Y in registers; X, Z in LDS slots 0, 1)"),
                   _emit(f, "ZKP_G2W3_MADD_ASM", g3a, "G2 mixed addition, three waves per SIMD (affine point from the scratch buffer; LDS slot 2 parks)")]
        ybase = g3d.R[1]
        f.write("#define ZKP_G2W3_STEP_IO(y) " + ", ".join('"+{v%d}"((y)[%d])' % (ybase + i, i) for i in range(NL)) + "\n")
        f.write("#define ZKP_G2W3_STEP_CLOBBERS " + ", ".join('"v%d"' % v for v in range(g3d.vb, g3d.vend) if not ybase <= v < ybase + NL) + ", "
                + ", ".join('"s%d"' % s for s in range(g3d.sb, g3d.send)) + ', "vcc", "scc", "memory"\n')
    return counts, (g1d, g1a, g2d, g2a)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "zkvm_pairings_amd", "csrc", "zkp_valid_steps.inc")
    counts, gs = write_inc(path)
    print("wrote %s: G1 dbl %d, madd %d (v%d..v%d); G2 dbl %d, madd %d (v%d..v%d); G2 at three waves dbl %d, madd %d instructions"
          % (path, counts[0], counts[1], gs[0].vb, gs[0].vend - 1, counts[2], counts[3], gs[2].vb, gs[2].vend - 1, counts[4], counts[5]))
