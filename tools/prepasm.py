#!/usr/bin/env python3
"""The homogeneous-projective doubling step of k_prep_lines<true> (zkp_coop.hip dbl_step_cln) as ONE hand-allocated gfx950
asm block -> csrc/zkp_prep_dbl.inc.

Why: the compiled step calls eleven by-value field routines; per lane and step it issues 4,508 multiply-adds and ~3,100 other
instructions (argument moves, operand forms through v_cndmask, a renormalisation per scaled product), and Karatsuba product
blocks cost the callers their registers.  Here the whole step has ONE register allocation:

  * six 14-register value blocks (X, Y, W pinned in/out + three), three 16-register temporaries (the last two registers of
    each stay zero: a line record is stored from them as four dwordx4), the Karatsuba product set of tools/coopasm.py
    (kterm / tail: 80 accumulator registers + 14 differences) - 232 VGPRs, two waves per SIMD as before;
  * lane parity c = Fp2 coefficient (two lanes per pair, as in the compiled kernel); the lane roles are EXEC parity masks and
    v_cndmask on a parity VCC, the partner's coefficient comes by DPP quad_perm [1,0,3,2] (always under full EXEC);
  * every product is kterm (147 multiply-adds), every reduction tail(): 13 products + 10 reductions per step
    = 3,871 multiply-adds; scalings sit on operands, so only E = 3 xi C and the stored c2 coefficient are renormalised.

Formulas (the same values as dbl_step_cln; Costello-Lange-Naehrig doubling on (X : Y : W = 2Z), scaled by 4, b' = 4 xi):
    B = Y^2, C = W^2, H2 = (Y + W)^2 - B - C, E = 3 xi C, F = 3 E,
    X' = ((X + Y)^2 - X^2 - B)(B - F),  Y' = (B + F)^2 - 12 E^2,  W' = B (4 H2)
    line: c2 = 2 (B - E) [record 0 + c], c1 xP = (-6 X^2) xP [record 2 + c], c0 yP = H2 yP [record 4 + c]
Reference anchors: G2Projective::double / the doubling line of the Miller loop the reference never wrote (src/g2.rs:210-242 is
its Jacobian doubling; src/pairings.rs is empty); gate: tools/asmemu.py against big-integer formulas (tests/test_prepasm.py).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import coopasm  # noqa: E402
from coopasm import NL, kterm, tail, vreg, p_balanced, VRED_C, VRED_SHIFT_IN, VRED_SHIFT_OUT  # noqa: E402

QP = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"


class Prep:
    def __init__(self, vb=6):
        self.lines = []
        v = vb
        assert v % 2 == 0
        self.vb = vb
        self.X = v; v += 14
        self.Y = v; v += 14
        self.W = v; v += 14
        self.V = []
        for _ in range(4):
            self.V.append(v); v += 14
        self.T = []
        for _ in range(3):
            self.T.append(v); v += 16          # 14 limbs + two registers that stay zero (record padding)
        self.D = v; v += 14
        self.ACC = {}
        for k in range(27):
            if k == 13:
                continue
            self.ACC[k] = v; v += 2
        self.MID = {}
        for k in range(13):
            self.MID[k] = v; v += 2
        self.vlds = v; v += 1                  # lane * 16: the lane's column of the LDS park
        self.voff = v; v += 1                  # store offset of the current record
        self.vq = v; v += 1                    # renormalisation quotient
        self.vt = v; v += 1
        self.vc = self.D                       # carries of the one-pass normalisation: the difference block is free between products
        self.vend = v
        assert self.vend <= 256, self.vend
        s = 36
        self.sb = s
        self.sC = s; s += 2                    # carry-out sink of the multiply-adds
        self.sEX = s; s += 2                   # EXEC at entry
        self.sM0 = s; s += 2                   # lanes with c = 0
        self.sM1 = s; s += 2                   # lanes with c = 1
        self.sPB = s; s += NL                  # balanced limbs of p
        self.send = s

    e = coopasm.Asm.e
    mad = coopasm.Asm.mad

    # ---- EXEC / lane roles
    def all_lanes(self):
        self.e("s_mov_b64 exec, s[%d:%d]" % (self.sEX, self.sEX + 1))

    def lanes(self, c):
        m = self.sM1 if c else self.sM0
        self.e("s_mov_b64 exec, s[%d:%d]" % (m, m + 1))

    # ---- limb-wise operations (all lanes unless said otherwise)
    def op2(self, op, d, a, b):
        for i in range(NL):
            self.e("%s v%d, v%d, v%d" % (op, d + i, a + i, b + i))

    def add(self, d, a, b):
        self.op2("v_add_u32", d, a, b)

    def sub(self, d, a, b):
        self.op2("v_sub_u32", d, a, b)

    def neg(self, d, a):
        for i in range(NL):
            self.e("v_sub_u32 v%d, 0, v%d" % (d + i, a + i))

    def shl(self, d, a, k):
        for i in range(NL):
            self.e("v_lshlrev_b32 v%d, %d, v%d" % (d + i, k, a + i))

    def times3(self, d, a):
        for i in range(NL):
            self.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (d + i, a + i, a + i))

    def mov(self, d, a):
        for i in range(NL):
            self.e("v_mov_b32 v%d, v%d" % (d + i, a + i))

    def swap(self, d, a):
        """the pair partner's coefficient (full EXEC; two wait states behind the last write of a)"""
        self.e("s_nop 1")
        for i in range(NL):
            self.e("v_mov_b32_dpp v%d, v%d %s" % (d + i, a + i, QP))

    def norm(self, x):
        """zkp_fp28.hpp weak_norm: one carry pass, |limb| <= 2^27 afterwards (+ the incoming carry)"""
        for i in range(NL - 1):
            self.e("v_add_u32 v%d, 0x8000000, v%d" % (self.vc + i, x + i))
            self.e("v_ashrrev_i32 v%d, 28, v%d" % (self.vc + i, self.vc + i))
            self.e("v_bfe_i32 v%d, v%d, 0, 28" % (x + i, x + i))
        for i in range(1, NL):
            self.e("v_add_u32 v%d, v%d, v%d" % (x + i, x + i, self.vc + i - 1))

    def vred(self, x):
        """zkp_coop.hip vred: q = round(value / p) from the top limb, x -= q p, one carry pass: |value| <= 0.51 p"""
        q, t = self.vq, self.vt
        self.e("v_ashrrev_i32 v%d, %d, v%d" % (q, VRED_SHIFT_IN, x + NL - 1))
        self.e("v_mul_i32_i24 v%d, 0x%x, v%d" % (q, VRED_C, q))
        self.e("v_add_u32 v%d, 0x%x, v%d" % (q, 1 << (VRED_SHIFT_OUT - 1), q))
        self.e("v_ashrrev_i32 v%d, %d, v%d" % (q, VRED_SHIFT_OUT, q))
        for i in range(NL):
            self.e("v_mul_lo_u32 v%d, v%d, s%d" % (t, q, self.sPB + i))
            self.e("v_sub_u32 v%d, v%d, v%d" % (x + i, x + i, t))
        self.norm(x)

    # ---- bounds: a product block takes operands whose limbs are at most L (2^27 + 16): the Karatsuba differences a_hi - a_lo
    # need 2 L 2^27 < 2^31 (L <= 7), and the columns of one lazy accumulation sum(La Lb) <= 30 (zkp_fp28.hpp)
    def prod(self, A, B, first, la, lb):
        assert la <= 7 and lb <= 7, "Karatsuba difference would leave int32"
        self.budget = (0 if first else self.budget) + la * lb
        assert self.budget <= 30, "column budget"
        kterm(self, A, B, first)

    # ---- products (coefficient c of an Fp2 product on a lane pair)
    def sqr_forms(self, a, k=None):
        """T0 = y, T1 = x of the squaring of a: c = 0: (a' + a)(a - a'), c = 1: (2 a') a  (' = the partner's coefficient);
        k: scale of x (a power of two as a shift count) - None: plain"""
        T0, T1 = self.T[0], self.T[1]
        self.swap(T0, a)
        self.lanes(0)
        self.add(T1, T0, a)
        self.sub(T0, a, T0)
        self.lanes(1)
        self.add(T1, T0, T0)
        self.mov(T0, a)
        self.all_lanes()

    def sqr(self, dst, a, la=1):
        self.sqr_forms(a)
        self.prod(self.T[1], self.T[0], True, 2 * la, 2 * la)
        tail(self, dst)

    def mul_acc(self, a, b, first, la, lb):
        """accumulate coefficient c of a b into the product set (two product blocks); b is read before the first of them"""
        T0, T1, T2 = self.T
        # T0 = a' (negated on c = 0), T2 = c ? b' : b, T1 = c ? b : b'
        self.swap(T0, a)
        self.swap(T1, b)
        for i in range(NL):
            self.e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (T2 + i, b + i, T1 + i))
        for i in range(NL):
            self.e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (T1 + i, T1 + i, b + i))
        self.lanes(0)
        self.neg(T0, T0)
        self.all_lanes()
        self.prod(a, T2, first, la, lb)
        self.prod(T0, T1, False, la, lb)

    def mul(self, dst, a, b, la, lb):
        """a may be overwritten by dst, and so may b (it is read before the first product)"""
        self.mul_acc(a, b, True, la, lb)
        tail(self, dst)

    def prologue(self):
        self.e("s_mov_b64 s[%d:%d], exec" % (self.sEX, self.sEX + 1))
        for (sr, m) in ((self.sM0, 0x55555555), (self.sM1, 0xaaaaaaaa)):
            self.e("s_mov_b32 s%d, 0x%x" % (sr, m))
            self.e("s_mov_b32 s%d, 0x%x" % (sr + 1, m))
            # the lane-role masks stay INSIDE the entry EXEC: a caller with lanes switched off never has them switched on here
            self.e("s_and_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (sr, sr + 1, sr, sr + 1, self.sEX, self.sEX + 1))
        self.e("s_mov_b64 vcc, s[%d:%d]" % (self.sM1, self.sM1 + 1))
        for i, v in enumerate(p_balanced()):
            self.e("s_mov_b32 s%d, 0x%x" % (self.sPB + i, v & 0xffffffff))
        self.e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % self.vlds)
        self.e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (self.vlds, self.vlds))
        self.e("v_lshlrev_b32 v%d, 4, v%d" % (self.vlds, self.vlds))

    def park_read(self, dst, v):
        """parked value v (0: xP, 1: yP) of the lane's pair -> 14 registers"""
        for q in range(3):
            self.e("ds_read_b128 v[%d:%d], v%d offset:%d" % (dst + 4 * q, dst + 4 * q + 3, self.vlds, (v * 4 + q) * 1024))
        self.e("ds_read_b64 v[%d:%d], v%d offset:%d" % (dst + 12, dst + 13, self.vlds, (v * 4 + 3) * 1024))

    def fmul(self, dst, a, v, la):
        """dst = a * (parked Fp value v): both lanes multiply their coefficient by the same Fp value"""
        self.park_read(self.T[0], v)
        self.e("s_waitcnt lgkmcnt(0)")
        self.prod(a, self.T[0], True, la, 1)
        tail(self, dst)

    def store(self, src, rec):
        """the 16-register temporary src (its last two registers are zero) -> line record rec + c of this step"""
        assert src in self.T
        self.e("v_mov_b32 v%d, 0" % (src + 14))
        self.e("v_mov_b32 v%d, 0" % (src + 15))
        if rec:
            self.e("s_mul_i32 s%d, %%[estride], %d" % (self.sC, rec // 2))
            self.e("v_add_u32 v%d, s%d, %%[voff]" % (self.voff, self.sC))
        else:
            self.e("v_mov_b32 v%d, %%[voff]" % self.voff)
        self.e("s_mov_b64 exec, %[smask]")
        for q in range(4):
            self.e("global_store_dwordx4 v%d, v[%d:%d], %%[base] offset:%d" % (self.voff, src + 4 * q, src + 4 * q + 3, 16 * q))
        self.all_lanes()


def generate(vb=6):
    g = Prep(vb)
    X, Y, W = g.X, g.Y, g.W
    V0, V1, V2, _ = g.V
    T0, T1, T2 = g.T
    g.prologue()
    # 1-3: B = Y^2, C = W^2, H2 = (Y + W)^2 - B - C
    g.sqr(V0, Y)
    g.sqr(V1, W)
    g.add(T2, Y, W)
    g.sqr(V2, T2, 2)
    g.sub(V2, V2, V0)
    g.sub(V2, V2, V1)
    # 4: c0 yP = H2 yP -> record 4 + c
    g.fmul(T2, V2, 1, 3)
    g.store(T2, 4)
    # 5: W' = (2 B)(2 H2): the factor 4 split over both operands (limbs 2 and 6 units: the Karatsuba differences stay in int32)
    g.shl(V2, V2, 1)
    g.shl(W, V0, 1)
    g.mul(W, W, V2, 2, 6)
    # 6: E = 3 xi C (renormalised): xi C = (C - C') on c = 0, (C' + C) on c = 1
    g.swap(T0, V1)
    g.lanes(0)
    g.sub(V1, V1, T0)
    g.lanes(1)
    g.add(V1, T0, V1)
    g.all_lanes()
    g.times3(V1, V1)
    g.vred(V1)
    # 7: c2 = 2 (B - E), renormalised -> record 0 + c
    g.sub(T2, V0, V1)
    g.shl(T2, T2, 1)
    g.vred(T2)
    g.store(T2, 0)
    # 8-9: X2 = X^2; c1 xP = (-6 X2) xP -> record 2 + c
    g.sqr(V2, X)
    g.times3(T2, V2)
    g.shl(T2, T2, 1)
    g.neg(T2, T2)
    g.fmul(T2, T2, 0, 6)
    g.store(T2, 2)
    # 10: XY2 = (X + Y)^2 - X2 - B   (into the X block: X and Y are dead behind the sum)
    g.add(T2, X, Y)
    g.sqr(X, T2, 2)
    g.sub(X, X, V2)
    g.sub(X, X, V0)
    # 11: F = 3 E; X' = XY2 (B - F)
    g.times3(V2, V1)
    g.sub(Y, V0, V2)                 # B - F (the Y block is free)
    g.mul(X, X, Y, 3, 4)
    # 12: Y' = (B + F)^2 - 12 E^2 under one reduction: s = norm(B + F)
    g.add(V2, V0, V2)
    g.norm(V2)
    g.sqr_forms(V2)                  # T1 = x_s, T0 = y_s
    g.prod(T1, T0, True, 2, 2)
    g.sqr_forms(V1)                  # T1 = x_e, T0 = y_e
    g.norm(T1)                       # -12 x_e y_e = (-4 norm(x_e)) (3 y_e): limbs 4 and 6 units
    g.shl(T1, T1, 2)
    g.neg(T1, T1)
    g.times3(T0, T0)
    g.prod(T1, T0, False, 4, 6)
    tail(g, Y)
    g.e("s_waitcnt vmcnt(0)")
    return g


def generate_add(vb=6):
    """the mixed addition T + Q on the same coordinates (zkp_coop.hip add_step_cln; Aranha et al. eq. (13), (14) with every quantity
    doubled: W = 2 Z), Q = (qx, qy) parked in LDS (values 2, 3: this lane's coefficient):
        th = 2 Y - qy W, la = 2 X - qx W (renormalised), line: c2 = th qx - la qy, c1 xP = -th xP, c0 yP = la yP,
        C = th^2, D = la^2, E = la D, F = W C, G = 2 X D, H = E + F - 2 G, X' = la H, Y' = th (G - H) - 2 Y E, W' = 2 W E
    Thirteen reductions (c2 and Y' are sums of two products under one reduction), seven value blocks."""
    g = Prep(vb)
    X, Y, W = g.X, g.Y, g.W
    V0, V1, V2, V3 = g.V
    T0, T1, T2 = g.T
    QX, QY = 2, 3
    g.prologue()
    # th = 2 Y - qy W -> V1, la = 2 X - qx W -> V2
    for (dst, q, src) in ((V1, QY, Y), (V2, QX, X)):
        g.park_read(V0, q)
        g.e("s_waitcnt lgkmcnt(0)")
        g.mul(dst, V0, W, 1, 1)
        g.shl(T2, src, 1)
        g.sub(dst, T2, dst)
        g.vred(dst)
    # c2 = th qx - la qy under one reduction (V0 still holds qx) -> record 0 + c
    g.mul_acc(V1, V0, True, 1, 1)
    g.park_read(V0, QY)
    g.e("s_waitcnt lgkmcnt(0)")
    g.neg(V0, V0)
    g.mul_acc(V2, V0, False, 1, 1)
    tail(g, T2)
    g.vred(T2)
    g.store(T2, 0)
    # c1 xP = -th xP -> record 2 + c; c0 yP = la yP -> record 4 + c
    g.neg(T2, V1)
    g.fmul(T2, T2, 0, 1)
    g.store(T2, 2)
    g.fmul(T2, V2, 1, 1)
    g.store(T2, 4)
    # D = la^2 -> V0; G = (2 X) D -> X; E = la D -> V0
    g.sqr(V0, V2)
    g.shl(X, X, 1)
    g.mul(X, X, V0, 2, 1)
    g.mul(V0, V2, V0, 1, 1)
    # C = th^2 -> V3; F = W C -> V3
    g.sqr(V3, V1)
    g.mul(V3, W, V3, 1, 1)
    # H = E + F - 2 G -> V3 (renormalised); GH = G - H -> X (renormalised)
    g.add(V3, V3, V0)
    g.shl(T2, X, 1)
    g.sub(V3, V3, T2)
    g.vred(V3)
    g.sub(X, X, V3)
    g.vred(X)
    # Y' = th GH + (-2 Y) E under one reduction
    g.mul_acc(V1, X, True, 1, 1)
    g.shl(Y, Y, 1)
    g.neg(Y, Y)
    g.mul_acc(Y, V0, False, 2, 1)
    tail(g, Y)
    # X' = la H; W' = (2 W) E
    g.mul(X, V2, V3, 1, 1)
    g.shl(W, W, 1)
    g.mul(W, W, V0, 2, 1)
    g.e("s_waitcnt vmcnt(0)")
    return g


def generate_jac(vb=6):
    """The doubling step of k_prep_lines<false> - the lines zkp_multi_miller_loop_batch must return the value of - as one block:
    the point in Jacobian coordinates as ePrint 2010/354 Alg. 26 walks it (zkp_coop.hip dbl_step; reference src/g2.rs:210-242 is the
    same doubling) and the upstream-shaped line coefficients, every VALUE mod p the one Alg. 26 produces (the Miller value is
    compared bit for bit with the oracle's), the route to it the cheapest this machine has:
        B = Y^2, zz = Z^2, Z' = (Y + Z)^2 - B - zz,  M = 3 X^2 (straight out of the squaring's reduction),  S = (4X) B,
        X' = M^2 - 2S,  Y' = M (S - X') - 8 B^2 (one lazy accumulation),
        c0 yP = (2 Z' zz) yP  [record 4 + c],  c1 xP = (-2 M zz) xP  [record 2 + c],  c2 = 2 X M - 4 B = 6 X^3 - 4 Y^2  [record 0 + c]
    (Alg. 26's tmp6 = (X + tmp4)^2 - tmp0 - tmp5 is 6 X^3, its tmp3 is S).  18 product blocks and 12 reductions per lane against the
    16 + 13 of the literal algorithm on schoolbook blocks; X' and Z' leave normalised (one carry pass), nothing is renormalised
    but the stored c2."""
    g = Prep(vb)
    X, Y, W = g.X, g.Y, g.W
    V0, V1, V2, V3 = g.V
    T0, T1, T2 = g.T
    g.prologue()
    g.sqr(V0, Y)                          # B
    g.sqr(V1, W)                          # zz
    g.add(T2, Y, W)
    g.sqr(V2, T2, 2)
    g.sub(V2, V2, V0)
    g.sub(V2, V2, V1)
    g.norm(V2)                            # Z' = 2 Y Z: limbs of one unit, |value| <= 2.2 p
    # c0 yP = (2 Z' zz) yP -> record 4 + c
    g.mul(V3, V2, V1, 1, 1)
    g.shl(V3, V3, 1)
    g.fmul(T2, V3, 1, 2)
    g.store(T2, 4)
    # M = 3 X^2
    g.sqr_forms(X)
    g.times3(T0, T0)
    g.prod(T1, T0, True, 2, 6)
    tail(g, V3)
    # c1 xP = (-2 M zz) xP -> record 2 + c
    g.mul(T2, V3, V1, 1, 1)
    g.shl(T2, T2, 1)
    g.neg(T2, T2)
    g.fmul(T2, T2, 0, 2)
    g.store(T2, 2)
    # c2 = 2 X M - 4 B, renormalised -> record 0 + c   (zz is dead: its block holds 2X, then -B)
    g.shl(V1, X, 1)
    g.mul(T2, V1, V3, 2, 1)
    g.neg(V1, V0)
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 2, v%d" % (T2 + i, V1 + i, T2 + i))
    g.vred(T2)
    g.store(T2, 0)
    # S = (4X) B
    g.shl(V1, X, 2)
    g.mul(V1, V1, V0, 4, 1)
    # X' = M^2 - 2S
    g.sqr(X, V3)
    g.neg(T2, V1)
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (X + i, T2 + i, X + i))
    g.norm(X)
    # Y' = M (S - X') - 8 B^2
    g.sub(V1, V1, X)
    g.mul_acc(V3, V1, True, 1, 2)
    g.sqr_forms(V0)
    g.norm(T1)
    g.shl(T1, T1, 2)
    g.neg(T1, T1)
    g.shl(T0, T0, 1)
    g.prod(T1, T0, False, 4, 4)
    tail(g, Y)
    g.mov(W, V2)
    g.e("s_waitcnt vmcnt(0)")
    return g


def generate_jac_add(vb=6):
    """The mixed-addition step of k_prep_lines<false>: ePrint 2010/354 Alg. 27 in VALUES (zkp_coop.hip add_step), Q = (qx, qy) parked
    in LDS values 2, 3:
        zz = Z^2, U2 = qx zz, S2' = (2 qy) Z zz, H = U2 - X, r = S2' - 2Y  [Alg. 27's t2, t6],  Z' = (2Z) H  [its nz],
        HH = H^2, t5 = (2H)(2HH) = 4 H^3, t7 = (2X)(2HH) = 4 X H^2,  X' = r^2 - t5 - 2 t7,  Y' = r (t7 - X') - (2Y) t5,
        c0 yP = (2 Z') yP [record 4 + c], c1 xP = (-2 r) xP [record 2 + c], c2 = r (2 qx) - (2 qy) Z' [record 0 + c, one reduction]
    25 product blocks and 13 reductions per lane; X' leaves renormalised, Y' and Z' as reduced products."""
    g = Prep(vb)
    X, Y, W = g.X, g.Y, g.W
    V0, V1, V2, V3 = g.V
    T0, T1, T2 = g.T
    QX, QY = 2, 3
    g.prologue()
    g.sqr(V0, W)                          # zz
    g.mul(V1, W, V0, 1, 1)                # Z zz
    g.park_read(V2, QX)
    g.e("s_waitcnt lgkmcnt(0)")
    g.mul(V0, V2, V0, 1, 1)               # U2
    g.sub(V0, V0, X)                      # H (limbs 2 units)
    g.park_read(V2, QY)
    g.e("s_waitcnt lgkmcnt(0)")
    g.shl(V2, V2, 1)
    g.mul(V1, V2, V1, 2, 1)               # S2' = (2 qy) Z zz
    g.shl(T2, Y, 1)
    g.sub(V1, V1, T2)                     # r = S2' - 2Y (limbs 3 units, |value| <= 3.2 p)
    g.norm(V1)
    g.shl(W, W, 1)
    g.mul(W, W, V0, 2, 2)                 # Z' = (2Z) H
    # lines: c0 yP, c1 xP, c2
    g.shl(T2, W, 1)
    g.fmul(T2, T2, 1, 2)
    g.store(T2, 4)
    g.shl(T2, V1, 1)
    g.neg(T2, T2)
    g.fmul(T2, T2, 0, 2)
    g.store(T2, 2)
    g.park_read(V3, QX)
    g.e("s_waitcnt lgkmcnt(0)")
    g.shl(V3, V3, 1)
    g.mul_acc(V1, V3, True, 1, 2)         # r (2 qx)
    g.park_read(V3, QY)
    g.e("s_waitcnt lgkmcnt(0)")
    g.shl(V3, V3, 1)
    g.neg(V3, V3)
    g.mul_acc(V3, W, False, 2, 1)         # - (2 qy) Z'
    tail(g, T2)
    g.vred(T2)
    g.store(T2, 0)
    # the point
    g.sqr(V2, V0, 2)                      # HH
    g.shl(V2, V2, 1)                      # 2 HH
    g.shl(V0, V0, 1)                      # 2 H (limbs 4 units)
    g.mul(V0, V0, V2, 4, 2)               # t5 = 4 H^3
    g.shl(V3, X, 1)
    g.mul(V2, V3, V2, 2, 2)               # t7 = 4 X H^2
    g.sqr(X, V1)                          # r^2
    g.sub(X, X, V0)
    g.neg(T2, V2)
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (X + i, T2 + i, X + i))      # X' = r^2 - t5 - 2 t7
    g.norm(X)
    g.vred(X)
    g.sub(V2, V2, X)                      # t7 - X'
    g.mul_acc(V1, V2, True, 1, 2)
    g.shl(Y, Y, 1)
    g.neg(Y, Y)
    g.mul_acc(Y, V0, False, 2, 1)
    tail(g, Y)                            # Y' = r (t7 - X') - (2Y) t5
    g.e("s_waitcnt vmcnt(0)")
    return g


def write_inc(path, vb=6):
    g = generate(vb)
    ga = generate_add(vb)
    n = sum(1 for l in g.lines if not l.endswith(":"))
    io = []
    for nm, base in (("x", g.X), ("y", g.Y), ("w", g.W)):
        io += ['"+{v%d}"((%s)[%d])' % (base + i, nm, i) for i in range(NL)]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/prepasm.py - do not edit.  The doubling step of k_prep_lines<true> as one inline-asm block.\n")
        f.write("// %d instructions; VGPRs v%d..v%d, SGPRs s%d..s%d, vcc.\n" % (n, g.vb, g.vend - 1, g.sb, g.send - 1))
        f.write("#pragma once\n")
        f.write("#define ZKP_PREP_DBL_ASM \\\n")
        for l in g.lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("// in/out: this lane's coefficient of X, Y, W (reduced values)\n")
        f.write("#define ZKP_PREP_DBL_IO(x, y, w) " + ", ".join(io) + "\n")
        f.write("// the mixed addition step (T + Q, Q parked in LDS): %d instructions, the same operands and clobbers\n"
                % sum(1 for l in ga.lines if not l.endswith(":")))
        f.write("#define ZKP_PREP_ADD_ASM \\\n")
        for l in ga.lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        gj = generate_jac(vb)
        f.write("// the Jacobian doubling step with the upstream-shaped lines (k_prep_lines<false>): %d instructions, the same operands and clobbers\n"
                % sum(1 for l in gj.lines if not l.endswith(":")))
        f.write("#define ZKP_PREP_JAC_DBL_ASM \\\n")
        for l in gj.lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        gja = generate_jac_add(vb)
        f.write("// ... and its mixed addition step (Alg. 27 in values): %d instructions\n" % sum(1 for l in gja.lines if not l.endswith(":")))
        f.write("#define ZKP_PREP_JAC_ADD_ASM \\\n")
        for l in gja.lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define ZKP_PREP_DBL_CLOBBERS " + ", ".join('"v%d"' % v for v in range(g.V[0], g.vend)) + ", "
                + ", ".join('"s%d"' % s for s in range(g.sb, g.send)) + ', "vcc", "scc", "memory"\n')
    return g, n


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "zkvm_pairings_amd", "csrc", "zkp_prep_dbl.inc")
    g, n = write_inc(path)
    print("wrote %s: %d instructions, v%d..v%d" % (path, n, g.vb, g.vend - 1))
