#!/usr/bin/env python3
"""Hand-scheduled gfx950 code of the step interpreter's MULACC step (k_coop, zkp_coop.hip) -> csrc/zkp_coop_mulacc.inc.

The C++ term loop of k_coop costs 201 VALU instructions per term for 147 multiply-adds (28 operand copies, address
arithmetic, 14 Karatsuba half sums) and its per-step part 432 for 196 (80 accumulator clears, a column-serial reduction);
every attempt to get a copy-free loop out of the compiler ended in spills or a lost wavefront.  This
generator emits the step as ONE inline-asm block with pinned registers:

  * two operand register sets (terms alternate), loaded straight by ds_read_b128 - no copies;
  * LDS addresses come RESOLVED from a per-lane table (built by the host at start-up from the generated step tables:
    zkp_coop.hip coop_init), so a term has no address arithmetic at all;
  * the first term's multiply-adds take the inline constant 0 as addend - the 40 accumulator pairs are never cleared;
  * sign and doubling of the A operand are one v_xad_u32 / v_lshlrev_b32 per limb, only in terms that need them
    (wave-uniform header bits);
  * the Montgomery reduction runs row by row, the twelve trailing multiply-adds of row i interleaved behind the
    m-computation of row i + 1, so the dependent chain (mul_lo, and, mad, shift, add) never stalls the wavefront;
  * limb extraction is one v_bfe_i32 per limb + the 64-bit carry.

The arithmetic is EXACTLY acc_mul_k / acc_fold / acc_reduce of zkp_fp28.hpp (the same columns mod 2^64, the same m_i, the
same balanced limbs), so tools/coopgen.py's emulator stays the gate and `ZKP_COOP_ASM=0` builds the C++ loop as the A/B baseline.

Reference anchors (what the step computes): Fp12::mul_by_014 src/fp12.rs:99-111, Fp12::square :173-184, Fp12 Mul :193-210.
"""
import os
import sys

NL, NH = 14, 7


class Asm:
    def __init__(self, vb):
        self.lines = []
        self.vb = vb
        v = vb
        self.E = v; v += 3            # resolved table entry of the next term: x = A1 address, y = B1 address, z = A2 | B2 << 16
        self.vz = v; v += 1           # B2 address of a term with two second operands (its A2 is the one that is prefetched)
        self.F = v; v += 2            # per-step flag words: F1 = neg bits | doubling bits << 12, F2 = a2-sign bits | b2-sign bits << 12
        self.vm = v; v += 1           # sign mask / shift count / address temporary
        self.vc = v; v += 1           # -mask
        assert v % 2 == 0
        self.D = v; v += 14           # Karatsuba differences / prefetched second operand / tail temporaries
        self.A = [0, 0]
        self.B = [0, 0]
        self.A[0] = v; v += 14
        self.B[0] = v; v += 14
        self.A[1] = v; v += 14
        self.B[1] = v; v += 14
        self.ACC = {}
        for k in range(27):
            if k == 13:
                continue
            self.ACC[k] = v; v += 2
        self.MID = {}
        for k in range(13):
            self.MID[k] = v; v += 2
        self.vend = v
        # scalars (clobbered)
        self.sb = 36
        s = self.sb
        self.sRT = s; s += 2
        self.sT = s; s += 2           # loop counter (+ pad: 64-bit scalar operands are even-aligned)
        self.sX = s; s += 2           # temporaries
        self.sN = s; s += 2           # number of the next term (+ pad)
        self.sC = s; s += 2           # carry-out sink of the multiply-adds
        self.sFR = s; s += 2          # address of the step's flag / store row (row T of the resolved table)
        self.sEX = s; s += 2          # EXEC at entry
        self.send = s

    def e(self, s):
        self.lines.append(s)

    def mad(self, dst, a, b, add, signed=True):
        op = "v_mad_i64_i32" if signed else "v_mad_u64_u32"
        addt = "0" if add is None else "v[%d:%d]" % (add, add + 1)
        self.e("%s v[%d:%d], s[%d:%d], %s, %s, %s" % (op, dst, dst + 1, self.sC, self.sC + 1, a, b, addt))


def vreg(i):
    return "v%d" % i


def ds_read_rec(g, dst, addr):
    """record (four planes of limb quads) at LDS byte address in VGPR addr -> 14 registers from dst"""
    g.e("ds_read_b128 v[%d:%d], %s" % (dst, dst + 3, addr))
    g.e("ds_read_b128 v[%d:%d], %s offset:%%[ps1]" % (dst + 4, dst + 7, addr))
    g.e("ds_read_b128 v[%d:%d], %s offset:%%[ps2]" % (dst + 8, dst + 11, addr))
    g.e("ds_read_b64 v[%d:%d], %s offset:%%[ps3]" % (dst + 12, dst + 13, addr))


def second_fetch(g, L, nbit):
    """request the second operand of the term whose header bits are (h3 >> nbit) from the entry in E: its A2 record if it has
    one (the B2 address is then parked in vz), else its B2 record, into D.  Called where D is free: in the prologue and behind
    a term's middle product."""
    g.e("s_lshr_b32 s%d, %%[h3], s%d" % (g.sX, nbit))
    g.e("s_and_b32 s%d, s%d, 0x1001" % (g.sX + 1, g.sX))
    g.e("s_cmp_eq_u32 s%d, 0x1001" % (g.sX + 1))
    g.e("s_cbranch_scc1 %s_nosf" % L)
    g.e("s_bitcmp1_b32 s%d, 0" % g.sX)
    g.e("s_cbranch_scc1 %s_sfb" % L)
    g.e("v_and_b32 v%d, 0xffff, v%d" % (g.vm, g.E + 2))
    g.e("v_lshrrev_b32 v%d, 16, v%d" % (g.vz, g.E + 2))
    g.e("s_branch %s_sfgo" % L)
    g.e("%s_sfb:" % L)
    g.e("v_lshrrev_b32 v%d, 16, v%d" % (g.vm, g.E + 2))
    g.e("%s_sfgo:" % L)
    ds_read_rec(g, g.D, vreg(g.vm))
    g.e("%s_nosf:" % L)


def add_signed(g, X, bit):
    """X[i] += +-D[i]: the lane's sign is bit `bit` (a scalar register) of flag word F2"""
    g.e("v_bfe_i32 v%d, v%d, s%d, 1" % (g.vm, g.F + 1, bit))
    g.e("v_sub_u32 v%d, 0, v%d" % (g.vc, g.vm))
    for i in range(NL):
        g.e("v_xad_u32 v%d, v%d, v%d, v%d" % (X + i, g.D + i, g.vm, X + i))
    for i in range(NL):
        g.e("v_add_u32 v%d, v%d, v%d" % (X + i, X + i, g.vc))


def term(g, p, first, tag):
    """one term on operand set p; its operands (and its A2, or else B2, record in D) were requested during the previous term"""
    A, B, D = g.A[p], g.B[p], g.D
    L = ".Lm%s_%%=" % tag
    g.e("s_waitcnt lgkmcnt(0)")
    # ---- operand forming (wave-uniform header bits; st = term number)
    # second operands: h3 bit st = no A2, bit 12 + st = no B2
    g.e("s_lshr_b32 s%d, %%[h3], s%d" % (g.sX, g.sT))
    g.e("s_and_b32 s%d, s%d, 0x1001" % (g.sX + 1, g.sX))
    g.e("s_cmp_eq_u32 s%d, 0x1001" % (g.sX + 1))
    g.e("s_cbranch_scc1 %s_nosec" % L)
    g.e("s_bitcmp1_b32 s%d, 0" % g.sX)
    g.e("s_cbranch_scc1 %s_noa2" % L)
    add_signed(g, A, g.sT)
    g.e("s_bitcmp1_b32 s%d, 12" % g.sX)
    g.e("s_cbranch_scc1 %s_nosec" % L)
    ds_read_rec(g, D, vreg(g.vz))          # both second operands: B2 on demand
    g.e("s_waitcnt lgkmcnt(0)")
    g.e("%s_noa2:" % L)
    g.e("s_add_u32 s%d, s%d, 12" % (g.sX + 1, g.sT))
    add_signed(g, B, g.sX + 1)
    g.e("%s_nosec:" % L)
    # sign of the product (h1 bit 4 + st set = no lane negates) and doubled A operand (h1 bit 16 + st = some lane doubles)
    g.e("s_lshr_b32 s%d, %%[h1], s%d" % (g.sX, g.sT))
    g.e("s_bitcmp1_b32 s%d, 4" % g.sX)
    g.e("s_cbranch_scc1 %s_noneg" % L)
    g.e("v_bfe_i32 v%d, v%d, s%d, 1" % (g.vm, g.F, g.sT))
    g.e("v_sub_u32 v%d, 0, v%d" % (g.vc, g.vm))
    for i in range(NL):
        g.e("v_xad_u32 v%d, v%d, v%d, v%d" % (A + i, A + i, g.vm, g.vc))
    g.e("%s_noneg:" % L)
    g.e("s_bitcmp1_b32 s%d, 16" % g.sX)
    g.e("s_cbranch_scc0 %s_noda" % L)
    g.e("s_add_u32 s%d, s%d, 12" % (g.sX + 1, g.sT))
    g.e("v_bfe_u32 v%d, v%d, s%d, 1" % (g.vm, g.F, g.sX + 1))
    for i in range(NL):
        g.e("v_lshlrev_b32 v%d, v%d, v%d" % (A + i, g.vm, A + i))
    g.e("%s_noda:" % L)
    # ---- first operands of the next term into the other set
    g.e("s_add_u32 s%d, s%d, 1" % (g.sN, g.sT))
    g.e("s_cmp_ge_u32 s%d, %%[T]" % g.sN)
    g.e("s_cbranch_scc1 %s_nopre" % L)
    g.e("s_waitcnt vmcnt(0)")
    ds_read_rec(g, g.A[1 - p], vreg(g.E))
    ds_read_rec(g, g.B[1 - p], vreg(g.E + 1))
    g.e("%s_nopre:" % L)
    # ---- Karatsuba differences: D[i] = a_hi - a_lo, D[7 + i] = b_lo - b_hi
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + i, A + NH + i, A + i))
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + NH + i, B + i, B + NH + i))
    # ---- 147 multiply-adds: lo -> columns 0..12, hi -> 14..26, middle -> its own 13 columns.  The middle product goes first:
    # behind it D is free again and takes the next term's second operand while the other 98 multiply-adds run.
    touched = set()

    def acc(reg_key, dst, a, b):
        add = dst
        if first and reg_key not in touched:
            add = None
            touched.add(reg_key)
        g.mad(dst, vreg(a), vreg(b), add)

    for j in range(NH):
        for i in range(NH):
            acc(("m", i + j), g.MID[i + j], D + i, D + NH + j)
    g.e("s_cmp_ge_u32 s%d, %%[T]" % g.sN)
    g.e("s_cbranch_scc1 %s_nopre2" % L)
    second_fetch(g, L, g.sN)
    g.e("s_add_u32 s%d, s%d, %%[row]" % (g.sRT, g.sRT))
    g.e("s_addc_u32 s%d, s%d, 0" % (g.sRT + 1, g.sRT + 1))
    g.e("global_load_dwordx3 v[%d:%d], %%[lane16], s[%d:%d] offset:%%[row]" % (g.E, g.E + 2, g.sRT, g.sRT + 1))
    g.e("%s_nopre2:" % L)
    for j in range(NH):
        for i in range(NH):
            acc(("a", i + j), g.ACC[i + j], A + i, B + j)
            acc(("a", NL + i + j), g.ACC[NL + i + j], A + NH + i, B + NH + j)


def tail(g, out, fold=True):
    """fold, Montgomery reduction, limb extraction -> 14 registers from out.
    fold=False: the accumulation holds plain (schoolbook) columns only - ksqr_plain: columns 0..26 as they stand, column 13 in MID[6]"""
    D = g.D
    col = dict(g.ACC)
    col[13] = g.MID[6]
    if fold:
        # fold: mid[k] += lo[k] + hi[k]; column 7 + k += mid[k]  (column 13 has no product of its own: it IS mid[6])
        for k in range(13):
            g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (g.MID[k], g.MID[k] + 1, g.MID[k], g.MID[k] + 1, g.ACC[k], g.ACC[k] + 1))
        for k in range(13):
            g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (g.MID[k], g.MID[k] + 1, g.MID[k], g.MID[k] + 1, g.ACC[NL + k], g.ACC[NL + k] + 1))
        for k in range(13):
            if k == 6:
                continue
            c = col[NH + k]
            g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (c, c + 1, c, c + 1, g.MID[k], g.MID[k] + 1))
    # reduction, row i: m = (low word of column i * PINV) mod 2^28; column i + j += m p_j; column i + 1 += column i >> 28.
    # Emission order per row: m, the j = 0 and j = 1 products, the carry - then the twelve remaining products of the
    # PREVIOUS row fill the latency of the next row's m.
    mreg = [D + 0, D + 1]
    tmp = D + 2   # pair (even-aligned: D is even)
    assert tmp % 2 == 0

    def rest(i):
        """the products j = 2..13 of row i as a list of instructions"""
        r = []
        for j in range(2, NL):
            c = col[i + j]
            r.append("v_mad_u64_u32 v[%d:%d], s[%d:%d], v%d, %%[p%d], v[%d:%d]" % (c, c + 1, g.sC, g.sC + 1, mreg[i & 1], j, c, c + 1))
        return r

    for i in range(NL):
        m = mreg[i & 1]
        c0, c1 = col[i], col[i + 1]
        # the dependent chain of row i (mul_lo, and, mad, shift, add) with the previous row's products in its gaps
        R = rest(i - 1) if i > 0 else []
        fill = [R[0:3], R[3:6], R[6:9], R[9:12]] if R else [[], [], [], []]
        g.e("v_mul_lo_u32 v%d, v%d, %%[pinv]" % (m, c0))
        for x in fill[0]:
            g.e(x)
        g.e("v_and_b32 v%d, 0xfffffff, v%d" % (m, m))
        for x in fill[1]:
            g.e(x)
        g.e("v_mad_u64_u32 v[%d:%d], s[%d:%d], v%d, %%[p0], v[%d:%d]" % (c0, c0 + 1, g.sC, g.sC + 1, m, c0, c0 + 1))
        g.e("v_mad_u64_u32 v[%d:%d], s[%d:%d], v%d, %%[p1], v[%d:%d]" % (c1, c1 + 1, g.sC, g.sC + 1, m, c1, c1 + 1))
        for x in fill[2]:
            g.e(x)
        g.e("v_ashrrev_i64 v[%d:%d], 28, v[%d:%d]" % (tmp, tmp + 1, c0, c0 + 1))
        for x in fill[3]:
            g.e(x)
        g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (c1, c1 + 1, c1, c1 + 1, tmp, tmp + 1))
    for x in rest(NL - 1):
        g.e(x)
    # limb extraction: t = column 14; limb k = sign-extended low 28 bits of t; t = column 15 + k + ((t + 2^27) >> 28)
    k27 = D + 4
    g.e("v_mov_b32 v%d, 0x8000000" % k27)
    g.e("v_mov_b32 v%d, 0" % (k27 + 1))
    t = col[NL]
    for k in range(NL - 1):
        g.e("v_bfe_i32 v%d, v%d, 0, 28" % (out + k, t))
        g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (tmp, tmp + 1, t, t + 1, k27, k27 + 1))
        g.e("v_ashrrev_i64 v[%d:%d], 28, v[%d:%d]" % (tmp, tmp + 1, tmp, tmp + 1))
        if k + 1 < NL - 1:
            nxt = col[NL + k + 1]
            g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (nxt, nxt + 1, nxt, nxt + 1, tmp, tmp + 1))
            t = nxt
        else:
            g.e("v_mov_b32 v%d, v%d" % (out + NL - 1, tmp))    # column 27 is empty: the last carry is the top limb


SETPRIO = int(os.environ.get("ZKP_GEN_SETPRIO", "0"))


def generate(vb=8):
    g = Asm(vb)
    out = g.A[0]
    # prologue: per-step flag words (row T of the resolved table), entry of term 0, operands of term 0, entry of term 1
    g.e("s_mov_b64 s[%d:%d], %%[rt]" % (g.sRT, g.sRT + 1))
    g.e("s_mul_i32 s%d, %%[T], %%[row]" % g.sX)       # row: bytes per table row (64 lanes x 16 B x wavefronts per workgroup)
    g.e("s_add_u32 s%d, s%d, s%d" % (g.sX, g.sRT, g.sX))
    g.e("s_addc_u32 s%d, s%d, 0" % (g.sX + 1, g.sRT + 1))
    g.e("s_mov_b64 s[%d:%d], s[%d:%d]" % (g.sFR, g.sFR + 1, g.sX, g.sX + 1))
    g.e("global_load_dwordx3 v[%d:%d], %%[lane16], s[%d:%d]" % (g.E, g.E + 2, g.sRT, g.sRT + 1))
    g.e("global_load_dwordx2 v[%d:%d], %%[lane16], s[%d:%d]" % (g.F, g.F + 1, g.sX, g.sX + 1))
    g.e("s_mov_b32 s%d, 0" % g.sT)
    g.e("s_waitcnt vmcnt(0)")        # the flag words too: the first term's operand forming reads them
    ds_read_rec(g, g.A[0], vreg(g.E))
    ds_read_rec(g, g.B[0], vreg(g.E + 1))
    second_fetch(g, ".Lmp_%=", g.sT)
    g.e("global_load_dwordx3 v[%d:%d], %%[lane16], s[%d:%d] offset:%%[row]" % (g.E, g.E + 2, g.sRT, g.sRT + 1))
    term(g, 0, True, "a")
    g.e("s_add_u32 s%d, s%d, 1" % (g.sT, g.sT))
    g.e("s_cmp_ge_u32 s%d, %%[T]" % g.sT)
    g.e("s_cbranch_scc1 .Lmtail_%=")
    g.e(".Lmloop_%=:")
    term(g, 1, False, "b")
    g.e("s_add_u32 s%d, s%d, 1" % (g.sT, g.sT))
    g.e("s_cmp_ge_u32 s%d, %%[T]" % g.sT)
    g.e("s_cbranch_scc1 .Lmtail_%=")
    term(g, 0, False, "c")
    g.e("s_add_u32 s%d, s%d, 1" % (g.sT, g.sT))
    g.e("s_cmp_lt_u32 s%d, %%[T]" % g.sT)
    g.e("s_cbranch_scc1 .Lmloop_%=")
    g.e(".Lmtail_%=:")
    if SETPRIO:
        # experiment (round 5): a wavefront in its reduction tail / store section issues ahead of wavefronts in their term loops
        g.e("s_setprio %d" % SETPRIO)
    g.e("s_waitcnt vmcnt(0)")
    # the step's store words (z, w of the flag row; the flag words themselves are dead): LDS byte address of the result slot and of
    # its companion slot, -1 where the lane stores nothing - requested here, they arrive behind the reduction
    g.e("global_load_dwordx2 v[%d:%d], %%[lane16], s[%d:%d] offset:8" % (g.F, g.F + 1, g.sFR, g.sFR + 1))
    tail(g, out)
    g.e("s_waitcnt vmcnt(0)")        # nothing of this block may still be in flight when the compiler's code resumes
    # ---- round 4: the result store and the companion store of a step WITHOUT epilogue happen here (nost = 0), under the caller's
    # active lanes: the compiled code behind the block (lane context, 14 DPP moves + 42 ALU of the companion forms, the copies out
    # of the pinned registers) runs for the rare epilogue steps only
    g.e("s_cmp_lg_u32 %[nost], 0")
    g.e("s_cbranch_scc1 .Lmdone_%=")
    g.e("s_mov_b64 s[%d:%d], exec" % (g.sEX, g.sEX + 1))
    g.e("v_cmp_ne_u32_e32 vcc, -1, v%d" % g.F)
    g.e("s_and_b64 vcc, vcc, %[act]")
    g.e("s_and_b64 exec, vcc, s[%d:%d]" % (g.sEX, g.sEX + 1))

    def st_rec(src, addr):
        g.e("ds_write_b128 v%d, v[%d:%d]" % (addr, src, src + 3))
        g.e("ds_write_b128 v%d, v[%d:%d] offset:%%[ps1]" % (addr, src + 4, src + 7))
        g.e("ds_write_b128 v%d, v[%d:%d] offset:%%[ps2]" % (addr, src + 8, src + 11))
        g.e("ds_write_b64 v%d, v[%d:%d] offset:%%[ps3]" % (addr, src + 12, src + 13))

    st_rec(out, g.F)
    g.e("s_mov_b64 exec, s[%d:%d]" % (g.sEX, g.sEX + 1))
    g.e("s_bitcmp1_b32 %[h1], 1")
    g.e("s_cbranch_scc0 .Lmdone_%=")
    # companion slot: lanes 2j, 2j + 1 hold one Fp2 coefficient (x0, x1); the even lane keeps x0 + x1, the odd lane x0 - x1 =
    # partner + s r with s r = r on even lanes, -r = (r xor -1) + 1 on odd lanes; the partner's r arrives inside the addition (DPP)
    T = g.B[0]
    g.e("v_bfe_u32 v%d, %%[lane16], 4, 1" % g.vm)
    g.e("v_sub_u32 v%d, 0, v%d" % (g.vc, g.vm))
    for i in range(NL):
        g.e("v_xad_u32 v%d, v%d, v%d, v%d" % (T + i, out + i, g.vc, g.vm))
    g.e("s_nop 1")
    for i in range(NL):
        g.e("v_add_u32_dpp v%d, v%d, v%d quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (T + i, out + i, T + i))
    g.e("v_cmp_ne_u32_e32 vcc, -1, v%d" % (g.F + 1))
    g.e("s_and_b64 vcc, vcc, %[act]")
    g.e("s_and_b64 exec, vcc, s[%d:%d]" % (g.sEX, g.sEX + 1))
    st_rec(T, g.F + 1)
    g.e("s_mov_b64 exec, s[%d:%d]" % (g.sEX, g.sEX + 1))
    g.e(".Lmdone_%=:")
    g.e("s_waitcnt lgkmcnt(0)")
    if SETPRIO:
        g.e("s_setprio 0")
    return g, out


# ------------------------------------------------------------------------------------------------------------------
# k_ksq: one Fp2 product per lane and compressed squaring, (xr + xi u)(yr + yi u) = (xr yr - xi yi) + (xr yi + xi yr) u, each
# coefficient ONE lazy accumulation of two Karatsuba products and one reduction - the MULACC step's term and tail on operands that
# sit in registers (no LDS, no tables).  It replaces two product-scanning multiplies (mont_mul_ps, 2 x 602 multiply-adds): 4 x 147 +
# 2 x 196 = 980 multiply-adds, the same limbs (the columns are the same integers mod 2^64).
class Fp2Mul:
    def __init__(self, vb):
        self.lines = []
        self.vb = vb
        v = vb
        assert v % 2 == 0
        self.XR = v; v += 14          # in: xr; out: the real part
        self.XI = v; v += 14          # in: xi; out: dead (holds -xi)
        self.YR = v; v += 14          # in
        self.YI = v; v += 14          # in
        self.S = v; v += 14           # out: the imaginary part
        self.D = v; v += 14           # Karatsuba differences / tail temporaries
        self.ACC = {}
        for k in range(27):
            if k == 13:
                continue
            self.ACC[k] = v; v += 2
        self.MID = {}
        for k in range(13):
            self.MID[k] = v; v += 2
        self.vend = v
        self.sb = 36
        self.sC = 36
        self.send = 38

    e = Asm.e
    mad = Asm.mad


def kterm(g, A, B, first):
    """acc (+)= A * B, 14-limb operands in registers, one level of Karatsuba (147 multiply-adds + 14 subtractions)"""
    D = g.D
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + i, A + NH + i, A + i))
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + NH + i, B + i, B + NH + i))
    touched = set()

    def acc(reg_key, dst, a, b):
        add = dst
        if first and reg_key not in touched:
            add = None
            touched.add(reg_key)
        g.mad(dst, vreg(a), vreg(b), add)

    for j in range(NH):
        for i in range(NH):
            acc(("a", i + j), g.ACC[i + j], A + i, B + j)
            acc(("a", NL + i + j), g.ACC[NL + i + j], A + NH + i, B + NH + j)
            acc(("m", i + j), g.MID[i + j], D + i, D + NH + j)


def ksqr_plain(g, A, P1=None, P2=None):
    """acc = sum_i P1[i] A[i] w^2i + sum_{i<j} P2[i] A[j] w^(i+j): a square in 105 multiply-adds instead of the 147 of a Karatsuba
    product.  Default P1 = A and P2 = 2A (formed in D): acc = A^2; a scaled square s A^2 takes P1 = s A, P2 = 2 s A.  Plain columns
    (column 13 in MID[6]): the accumulation holds nothing else and is reduced by tail(fold=False) - no fold of Karatsuba halves."""
    if P2 is None:
        assert P1 is None
        P1, P2 = A, g.D
        for i in range(NL):
            g.e("v_lshlrev_b32 v%d, 1, v%d" % (P2 + i, A + i))
    col = dict(g.ACC)
    col[13] = g.MID[6]
    touched = set()
    for i in range(NL):
        for j in range(i, NL):
            k = i + j
            add = col[k] if k in touched else None
            touched.add(k)
            g.mad(col[k], vreg((P1 if i == j else P2) + i), vreg(A + j), add)


def ksqr_k(g, A, P1, P2, first):
    """the same value, sum_i P1[i] A[i] w^2i + sum_{i<j} P2[i] A[j] w^(i+j) with P1 = s A and P2 = 2 s A, accumulated in the
    Karatsuba layout of kterm (low halves -> columns 0..12, high halves -> 14..26, the subtractive middle term
    (a_hi - a_lo)(b_lo - b_hi) = -s (a_hi - a_lo)^2 -> MID) so that it can share a lazy accumulation - and its fold - with kterm
    products: 84 multiply-adds + 21 subtractions.  P2 is destroyed (its low half receives the middle term's doubled differences)."""
    D = g.D
    touched = set()

    def acc(key, dst, a, b):
        add = dst
        if first and key not in touched:
            add = None
            touched.add(key)
        g.mad(dst, vreg(a), vreg(b), add)

    for h, cols in ((0, 0), (NH, NL)):
        for i in range(NH):
            for j in range(i, NH):
                acc(("a", cols + i + j), g.ACC[cols + i + j], (P1 if i == j else P2) + h + i, A + h + j)
    # middle term: dq = a_hi - a_lo (D[0..6]), dp1 = P1_lo - P1_hi = -s dq (D[7..13]), dp2 = P2_lo - P2_hi (in place, P2's low half)
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + i, A + NH + i, A + i))
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (D + NH + i, P1 + i, P1 + NH + i))
    for i in range(NH):
        g.e("v_sub_u32 v%d, v%d, v%d" % (P2 + i, P2 + i, P2 + NH + i))
    for i in range(NH):
        for j in range(i, NH):
            acc(("m", i + j), g.MID[i + j], (D + NH + i) if i == j else (P2 + i), D + j)


# ---------------------------------------------------------------------------------------------------------------
# k_ksq: one compressed squaring per loop iteration.  The block continues the Fp2 product above with everything that
# follows it in the iteration except the rare snapshot store and the operand forms of the next product (C++):
#   * the parked copy of the lane's old coefficient is requested behind the last product block (into the dead Y registers);
#   * A lanes multiply X by -Y (the C++ forms negate Y for free), so their product arrives as -A: with -A the A-lane
#     combination -t = 2 B_r - A_r - B_i (and its imaginary twin) is one DPP subtraction / addition and one v_lshl_add_u32 per
#     limb, multiplier -3 in the carry chain; B lanes leave the doubling of their combination to the multiplier (6 instead
#     of 3) and lane 3 needs no combination at all.  (A negated fetch of B would do the same without touching the forms, but
#     v_subrev_u32_dpp does not permute its subtrahend on gfx950 - tools/dbg/dpp_probe.hip - and v_sub_u32_dpp negates the
#     wrong operand.)
#   * the two carry chains (3 t + 2 sgn x - q p, exact, balanced limbs) are three v_mad_i64_i32 + bfe + 64-bit add + shift
#     per limb, interleaved; their results ARE the next product's X registers;
#   * the new coefficient is parked from those registers (seven ds_write_b128: re0..re13, im0..im13 are 28 consecutive
#     registers) and the pair partner's coefficient lands in the Y registers by DPP.
# Same integers as sq_combine / the C++ combinations of zkp_coop.hip (tools/coopgen.py emu_ksq is the model).
P_BLS = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
VRED_C, VRED_SHIFT_IN, VRED_SHIFT_OUT = 80647, 9, 24      # zkp_coop_prog.inc (asserted by write_inc)


def p_balanced():
    out, c = [], 0
    for i in range(NL):
        v = ((P_BLS >> (28 * i)) & 0xfffffff) + c
        c = 0
        if i < NL - 1 and v >= 1 << 27:
            v -= 1 << 28
            c = 1
        out.append(v)
    return out


def generate_ksq(vb=6):
    g = Fp2Mul(vb)
    g.sEX, g.sMA, g.sM1, g.sM3, g.sPB = 38, 40, 42, 44, 46
    g.send = g.sPB + NL
    XR, XI, YR, YI, S, D = g.XR, g.XI, g.YR, g.YI, g.S, g.D
    vaddr, mul, sgn2, nq_r, nq_i = D + 6, D + 8, D + 10, D + 11, D + 12
    acc0 = g.ACC[0]
    T1, T2, U, W = acc0, acc0 + 14, acc0 + 28, acc0 + 42           # 56 of the 78 accumulator registers
    c_r, c_i, u_r, u_i, k27 = acc0 + 56, acc0 + 58, acc0 + 60, acc0 + 62, acc0 + 64
    assert k27 + 2 <= g.vend and all(x % 2 == 0 for x in (c_r, c_i, u_r, u_i, k27))
    QP = "quad_perm:[%s] row_mask:0xf bank_mask:0xf"
    g.e("s_mov_b64 s[%d:%d], exec" % (g.sEX, g.sEX + 1))
    # ---- the product (generate_fp2mul) with the old coefficient requested behind its last product block
    kterm(g, XR, YI, True)
    kterm(g, XI, YR, False)
    tail(g, S)
    for i in range(NL):
        g.e("v_sub_u32 v%d, 0, v%d" % (XI + i, XI + i))
    kterm(g, XR, YR, True)
    kterm(g, XI, YI, False)
    g.e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % vaddr)
    g.e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (vaddr, vaddr))
    g.e("v_lshlrev_b32 v%d, 4, v%d" % (vaddr, vaddr))
    for k in range(7):
        g.e("ds_read_b128 v[%d:%d], v%d offset:%d" % (YR + 4 * k, YR + 4 * k + 3, vaddr, 1024 * k))
    tail(g, XR)
    # ---- scalar constants: lane-role masks, the balanced limbs of p
    for (sr, m) in ((g.sMA, 0x55555555), (g.sM1, 0x22222222)):
        g.e("s_mov_b32 s%d, 0x%x" % (sr, m))
        g.e("s_mov_b32 s%d, 0x%x" % (sr + 1, m))
        # inside the entry EXEC (k_ksq runs full wavefronts; nothing here depends on that)
        g.e("s_and_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (sr, sr + 1, sr, sr + 1, g.sEX, g.sEX + 1))
    for i, v in enumerate(p_balanced()):
        g.e("s_mov_b32 s%d, 0x%x" % (g.sPB + i, v & 0xffffffff))
    g.e("v_mov_b32 v%d, 6" % mul)
    g.e("v_mov_b32 v%d, 2" % sgn2)
    g.e("v_mov_b32 v%d, 0x8000000" % k27)
    g.e("v_mov_b32 v%d, 0" % (k27 + 1))
    # ---- B of the other pair (all lanes; a DPP read needs two wait states behind the write of its source)
    g.e("s_nop 1")
    for i in range(NL):
        g.e(("v_mov_b32_dpp v%d, v%d " + QP) % (T1 + i, XR + i, "3,3,1,1"))
    for i in range(NL):
        g.e(("v_mov_b32_dpp v%d, v%d " + QP) % (T2 + i, S + i, "3,3,1,1"))
    # ---- A lanes: -t = (2 B_r - A_r - B_i) + (2 B_i - A_i + B_r) u, -A from the other pair's even lane (an A lane: enabled)
    g.e("s_mov_b64 exec, s[%d:%d]" % (g.sMA, g.sMA + 1))
    for i in range(NL):
        g.e(("v_sub_u32_dpp v%d, v%d, v%d " + QP) % (U + i, XR + i, T2 + i, "2,2,0,0"))
        g.e(("v_add_u32_dpp v%d, v%d, v%d " + QP) % (W + i, S + i, T1 + i, "2,2,0,0"))
    for i in range(NL):
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (T1 + i, T1 + i, U + i))
        g.e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (T2 + i, T2 + i, W + i))
    g.e("v_mov_b32 v%d, -3" % mul)
    g.e("v_mov_b32 v%d, -2" % sgn2)
    # ---- lane 1: t/2 = (B_r - B_i) + (B_r + B_i) u;  lane 3: t/2 = B as it stands;  multiplier 6
    g.e("s_mov_b64 exec, s[%d:%d]" % (g.sM1, g.sM1 + 1))
    for i in range(NL):
        g.e("v_sub_u32 v%d, v%d, v%d" % (U + i, T1 + i, T2 + i))
        g.e("v_add_u32 v%d, v%d, v%d" % (T2 + i, T1 + i, T2 + i))
    for i in range(NL):
        g.e("v_mov_b32 v%d, v%d" % (T1 + i, U + i))
    g.e("s_mov_b64 exec, s[%d:%d]" % (g.sEX, g.sEX + 1))
    g.e("s_waitcnt lgkmcnt(0)")          # the old coefficient: re in the YR registers, im in the YI registers
    # ---- q = round((mult t13 + sgn2 x13) / p_top) per coefficient (zkp_coop.hip sq_combine), negated
    for (nq, T, X) in ((nq_r, T1, YR), (nq_i, T2, YI)):
        g.e("v_mul_i32_i24 v%d, v%d, v%d" % (nq, T + NL - 1, mul))
        g.e("v_mad_i32_i24 v%d, v%d, v%d, v%d" % (nq, X + NL - 1, sgn2, nq))
        g.e("v_ashrrev_i32 v%d, %d, v%d" % (nq, VRED_SHIFT_IN, nq))
        g.e("v_mul_i32_i24 v%d, 0x%x, v%d" % (nq, VRED_C, nq))
        g.e("v_add_u32 v%d, 0x%x, v%d" % (nq, 1 << (VRED_SHIFT_OUT - 1), nq))
        g.e("v_ashrrev_i32 v%d, %d, v%d" % (nq, VRED_SHIFT_OUT, nq))
        g.e("v_sub_u32 v%d, 0, v%d" % (nq, nq))
    # ---- the two carry chains, interleaved; results: the new coefficient in XR / XI
    for i in range(NL):
        for (c, u, T, X, nq, OUT) in ((c_r, u_r, T1, YR, nq_r, XR), (c_i, u_i, T2, YI, nq_i, XI)):
            g.mad(c, vreg(T + i), vreg(mul), None if i == 0 else c)
            g.mad(c, vreg(X + i), vreg(sgn2), c)
            g.mad(c, vreg(nq), "s%d" % (g.sPB + i), c)
            if i < NL - 1:
                g.e("v_bfe_i32 v%d, v%d, 0, 28" % (OUT + i, c))
                g.e("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (u, u + 1, c, c + 1, k27, k27 + 1))
                g.e("v_ashrrev_i64 v[%d:%d], 28, v[%d:%d]" % (c, c + 1, u, u + 1))
            else:
                g.e("v_mov_b32 v%d, v%d" % (OUT + i, c))
    # ---- park the new coefficient (the "2 x" of the next squaring), fetch the pair partner's into the Y registers
    for k in range(7):
        g.e("ds_write_b128 v%d, v[%d:%d] offset:%d" % (vaddr, XR + 4 * k, XR + 4 * k + 3, 1024 * k))
    g.e("s_nop 1")
    for i in range(NL):
        g.e(("v_mov_b32_dpp v%d, v%d " + QP) % (YR + i, XR + i, "1,0,3,2"))
    for i in range(NL):
        g.e(("v_mov_b32_dpp v%d, v%d " + QP) % (YI + i, XI + i, "1,0,3,2"))
    g.e("s_waitcnt lgkmcnt(0)")
    return g


def write_ksq(f, vb=6):
    g = generate_ksq(vb)
    n = sum(1 for l in g.lines if not l.endswith(":"))
    f.write("// One compressed squaring of k_ksq behind the operand forms: %d instructions; VGPRs v%d..v%d, SGPRs s%d..s%d.\n"
            % (n, g.vb, g.vend - 1, g.sb, g.send - 1))
    f.write("#define ZKP_KSQ_BODY_ASM \\\n")
    for l in g.lines:
        f.write('    "%s\\n\\t" \\\n' % l)
    f.write('    ""\n')
    f.write("// in: the product's factors X = xr + xi u and Y = yr + yi u (B lanes) or -Y (A lanes); out: the lane's new coefficient (xr, xi), its pair partner's (yr, yi)\n")
    f.write("#define ZKP_KSQ_BODY_IO(xr, xi, yr, yi) " + ", ".join(
        ", ".join('"+{v%d}"((%s)[%d])' % (base + i, nm, i) for i in range(NL))
        for nm, base in (("xr", g.XR), ("xi", g.XI), ("yr", g.YR), ("yi", g.YI))) + "\n")
    f.write("#define ZKP_KSQ_BODY_CLOBBERS " + ", ".join('"v%d"' % v for v in range(g.S, g.vend)) + ", "
            + ", ".join('"s%d"' % x for x in range(g.sb, g.send)) + ', "scc", "memory"\n')
    f.write("#define ZKP_KSQ_BODY_P_BAL " + ", ".join(str(v) for v in p_balanced()) + "\n")
    f.write("#define ZKP_KSQ_BODY_VRED %d, %d, %d\n" % (VRED_C, VRED_SHIFT_IN, VRED_SHIFT_OUT))
    return g


def write_inc(path, vb=8):
    g, out = generate(vb)
    outs = list(range(out, out + NL))
    vclob = [v for v in range(g.vb, g.vend) if v not in outs]
    sclob = list(range(g.sb, g.send))
    with open(path, "w") as f:
        f.write("// GENERATED by tools/coopasm.py - do not edit.  The MULACC step of k_coop as one hand-scheduled inline-asm block.\n")
        f.write("// %d instructions; VGPRs v%d..v%d, SGPRs s%d..s%d.\n" % (sum(1 for l in g.lines if not l.endswith(":")), g.vb, g.vend - 1, g.sb, g.send - 1))
        f.write("#pragma once\n")
        f.write("#define ZKP_MULACC_VGPR_FIRST %d\n#define ZKP_MULACC_VGPR_END %d\n" % (g.vb, g.vend))
        f.write("#define ZKP_MULACC_ASM \\\n")
        for l in g.lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define ZKP_MULACC_OUTS(r) " + ", ".join('"={v%d}"((r)[%d])' % (outs[i], i) for i in range(NL)) + "\n")
        f.write("#define ZKP_MULACC_CLOBBERS " + ", ".join('"v%d"' % v for v in vclob) + ", " + ", ".join('"s%d"' % s for s in sclob) + ', "vcc", "scc", "memory"\n')
        write_ksq(f)
    return g


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "zkvm_pairings_amd", "csrc", "zkp_coop_mulacc.inc")
    g = write_inc(path)
    n = sum(1 for l in g.lines if not l.endswith(":"))
    print("wrote %s: %d instructions, v%d..v%d" % (path, n, g.vb, g.vend - 1))
