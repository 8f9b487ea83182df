#!/usr/bin/env python3
"""A CPU model of the gfx950 instruction subset that tools/coopasm.py emits - the gate for the generated inline-asm blocks.

It executes the text of a generated block (straight-line code: no branches) on a group of lanes: 32-bit VGPRs per lane, SGPRs,
EXEC, an LDS byte image, DPP quad permutations.  Two things a GPU run would only show as wrong numbers are errors here:
a DPP read from a lane that EXEC has switched off, and a read of a register nothing has written.

    emu = Emu(lanes=4, subst={"p0": ..., "pinv": ...})
    emu.v[6][lane] = ...          # operands
    emu.run(lines)
    emu.v[6][lane]                # results (uint32; s32() for the signed value)

tests/test_coopasm.py runs the k_ksq body against tools/coopgen.py's limb-exact model of a compressed squaring.
"""
import re

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


def s24(x):
    x &= 0xffffff
    return x - (1 << 24) if x >> 23 else x


class Emu:
    def __init__(self, lanes=4, subst=None, strict=True):
        self.n = lanes
        self.v = {}                 # register number -> [value per lane]
        self.s = {}
        self.exec = (1 << lanes) - 1
        self.lds = {}               # byte address -> dword (addresses are multiples of 4)
        self.mem = {}               # global memory: byte address -> dword
        self.scc = 0
        self.vcc = 0
        self.subst = dict(subst or {})
        self.strict = strict
        self.count = {}

    # ---- operands
    def _sub(self, text):
        return re.sub(r"%\[(\w+)\]", lambda m: str(self.subst[m.group(1)]), text)

    def rd(self, op, lane, width=1):
        """value of a source operand for one lane (width = number of dwords)"""
        op = op.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            assert hi - lo + 1 == width, op
            return sum(self._rv(lo + k, lane) << (32 * k) for k in range(width))
        m = re.fullmatch(r"v(\d+)", op)
        if m:
            assert width == 1, op
            return self._rv(int(m.group(1)), lane)
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            return sum(self.s[lo + k] << (32 * k) for k in range(hi - lo + 1))
        m = re.fullmatch(r"s(\d+)", op)
        if m:
            v = self.s[int(m.group(1))]
            return v if width == 1 else s32(v) & M64      # a 32-bit scalar source of a 64-bit operand is sign-extended
        v = int(op, 0)
        return v & (M32 if width == 1 else M64)

    def _rv(self, r, lane):
        if r not in self.v or self.v[r][lane] is None:
            if self.strict:
                raise RuntimeError("read of unwritten v%d (lane %d)" % (r, lane))
            return 0
        return self.v[r][lane]

    def wr(self, op, lane, val, width=1):
        op = op.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            assert hi - lo + 1 == width and lo % 2 == 0, "misaligned or mis-sized tuple " + op
            for k in range(width):
                self.v.setdefault(lo + k, [None] * self.n)[lane] = (val >> (32 * k)) & M32
            return
        m = re.fullmatch(r"v(\d+)", op)
        assert m and width == 1, op
        self.v.setdefault(int(m.group(1)), [None] * self.n)[lane] = val & M32

    # ---- execution
    def run(self, lines, max_steps=10_000_000):
        prog = [self._sub(l).strip() for l in lines]
        prog = [l for l in prog if l]
        labels = {l[:-1]: i for i, l in enumerate(prog) if l.endswith(":")}
        pc, n = 0, 0
        while pc < len(prog):
            line = prog[pc]
            pc += 1
            if line.endswith(":"):
                continue
            n += 1
            assert n < max_steps, "runaway loop"
            op = line.split(None, 1)[0]
            if op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1"):
                self.count[op] = self.count.get(op, 0) + 1
                target = line.split(None, 1)[1].strip()
                if op == "s_branch" or (op == "s_cbranch_scc1") == bool(self.scc):
                    pc = labels[target]
                continue
            self.step(line)
        return self

    def step(self, line):
        mods = {}
        m = re.search(r"quad_perm:\[(\d),(\d),(\d),(\d)\]", line)
        if m:
            mods["qp"] = [int(x) for x in m.groups()]
            line = line[:m.start()] + line[m.end():]
        for key in ("row_mask", "bank_mask", "offset"):
            m = re.search(key + r":(0x[0-9a-fA-F]+|\d+)", line)
            if m:
                mods[key] = int(m.group(1), 0)
                line = line[:m.start()] + line[m.end():]
        parts = line.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in self._split(parts[1])] if len(parts) > 1 else []
        self.count[op] = self.count.get(op, 0) + 1
        dpp = op.endswith("_dpp")
        if dpp:
            op = op[:-4]
            assert mods.get("row_mask", 0xf) == 0xf and mods.get("bank_mask", 0xf) == 0xf and "qp" in mods
            if op in ("v_subrev_u32",):
                # measured on gfx950 (tools/dbg/dpp_probe.hip): v_subrev_u32_dpp d, x, zero returns 0 - x of the lane ITSELF
                raise NotImplementedError("v_subrev_u32_dpp does not permute its subtrahend on gfx950")
        fn = getattr(self, "op_" + op, None)
        if fn is None:
            raise NotImplementedError(op)
        if op.startswith("s_") or op.startswith("ds_") or op.startswith("global_"):
            fn(args, mods)
            return
        for lane in range(self.n):
            if not (self.exec >> lane) & 1:
                continue
            src_lane = lane
            if dpp:
                src_lane = (lane & ~3) | mods["qp"][lane & 3]
                if not (self.exec >> src_lane) & 1:
                    raise RuntimeError("DPP read from lane %d, which EXEC has switched off: %s" % (src_lane, line))
            fn(args, lane, src_lane)

    @staticmethod
    def _split(s):
        out, depth, cur = [], 0, ""
        for ch in s:
            if ch == "[":
                depth += 1
            if ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur)
        return out

    # ---- scalar / LDS
    def op_s_nop(self, a, m):
        pass

    def op_s_waitcnt(self, a, m):
        pass

    def op_s_setprio(self, a, m):
        pass

    def op_s_mov_b32(self, a, m):
        self.s[int(a[0][1:])] = int(a[1], 0) & M32 if not a[1].startswith("s") else self.s[int(a[1][1:])]

    def op_s_mov_b64(self, a, m):
        if a[0] == "vcc":
            self.vcc = self.rd(a[1], 0)
            return
        if a[0] == "exec":
            self.exec = self.rd(a[1], 0) & ((1 << self.n) - 1) if a[1] != "-1" else (1 << self.n) - 1
            return
        lo = int(re.match(r"s\[(\d+):", a[0]).group(1))
        val = ((1 << 64) - 1 if self.exec == (1 << self.n) - 1 else self.exec) if a[1] == "exec" else self.rd(a[1], 0)
        self.s[lo], self.s[lo + 1] = val & M32, (val >> 32) & M32

    def _rd64(self, x):
        x = x.strip()
        if x == "vcc":
            return self.vcc
        if x == "exec":
            return self.exec
        return self.rd(x, 0)

    def op_s_and_b64(self, a, m):
        val = self._rd64(a[1]) & self._rd64(a[2])
        d = a[0].strip()
        if d == "vcc":
            self.vcc = val
        elif d == "exec":
            self.exec = val & ((1 << self.n) - 1)
        else:
            lo = int(re.match(r"s\[(\d+):", d).group(1))
            self.s[lo], self.s[lo + 1] = val & M32, (val >> 32) & M32
        self.scc = int(val != 0)

    def op_s_cmp_lg_u32(self, a, m):
        self.scc = int(self._sr(a[0]) != self._sr(a[1]))

    def _sr(self, x):
        x = x.strip()
        return self.s[int(x[1:])] if re.fullmatch(r"s\d+", x) else int(x, 0) & M32

    def _sw(self, x, v):
        self.s[int(x.strip()[1:])] = v & M32

    def op_s_lshr_b32(self, a, m):
        self._sw(a[0], self._sr(a[1]) >> (self._sr(a[2]) & 31))
        self.scc = int(self._sr(a[0]) != 0)

    def op_s_lshl_b32(self, a, m):
        self._sw(a[0], self._sr(a[1]) << (self._sr(a[2]) & 31))
        self.scc = int(self._sr(a[0]) != 0)

    def op_s_and_b32(self, a, m):
        self._sw(a[0], self._sr(a[1]) & self._sr(a[2]))
        self.scc = int(self._sr(a[0]) != 0)

    def op_s_add_u32(self, a, m):
        v = self._sr(a[1]) + self._sr(a[2])
        self._sw(a[0], v)
        self.scc = v >> 32

    def op_s_addc_u32(self, a, m):
        v = self._sr(a[1]) + self._sr(a[2]) + self.scc
        self._sw(a[0], v)
        self.scc = v >> 32

    def op_s_mul_i32(self, a, m):
        self._sw(a[0], s32(self._sr(a[1])) * s32(self._sr(a[2])))

    def op_s_mul_hi_u32(self, a, m):
        self._sw(a[0], (self._sr(a[1]) * self._sr(a[2])) >> 32)

    def op_s_cmp_ge_u32(self, a, m):
        self.scc = int(self._sr(a[0]) >= self._sr(a[1]))

    def op_s_cmp_lt_u32(self, a, m):
        self.scc = int(self._sr(a[0]) < self._sr(a[1]))

    def op_s_cmp_eq_u32(self, a, m):
        self.scc = int(self._sr(a[0]) == self._sr(a[1]))

    def op_s_bitcmp1_b32(self, a, m):
        self.scc = (self._sr(a[0]) >> (self._sr(a[1]) & 31)) & 1

    def _gload(self, a, m, n):
        base = self.rd(a[2], 0) if a[2].strip() != "off" else 0
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = base + self.rd(a[1], lane) + m.get("offset", 0)
                assert addr % 4 == 0
                for k in range(n):
                    if addr + 4 * k not in self.mem:
                        raise RuntimeError("global read of unwritten address 0x%x" % (addr + 4 * k))
                    self.v.setdefault(self._vbase(a[0]) + k, [None] * self.n)[lane] = self.mem[addr + 4 * k]

    @staticmethod
    def _vbase(op):
        mm = re.fullmatch(r"v\[(\d+):(\d+)\]", op.strip())
        return int(mm.group(1)) if mm else int(op.strip()[1:])

    def op_global_store_dwordx4(self, a, m):     # voffset, vdata[4], saddr
        base = self.rd(a[2], 0)
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = base + self.rd(a[0], lane) + m.get("offset", 0)
                assert addr % 16 == 0
                val = self.rd(a[1], lane, 4)
                for k in range(4):
                    self.mem[addr + 4 * k] = (val >> (32 * k)) & M32

    def op_global_load_dwordx4(self, a, m):
        self._gload(a, m, 4)

    def op_global_load_dwordx3(self, a, m):
        self._gload(a, m, 3)

    def op_global_load_dwordx2(self, a, m):
        self._gload(a, m, 2)

    def op_ds_read_b64(self, a, m):
        off = m.get("offset", 0)
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = self.rd(a[1], lane) + off
                assert addr % 8 == 0
                for k in range(2):
                    if addr + 4 * k not in self.lds:
                        raise RuntimeError("LDS read of unwritten address %d" % (addr + 4 * k))
                self.wr(a[0], lane, sum(self.lds[addr + 4 * k] << (32 * k) for k in range(2)), 2)

    def op_ds_read_b128(self, a, m):
        off = m.get("offset", 0)
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = self.rd(a[1], lane) + off
                assert addr % 16 == 0
                for k in range(4):
                    if addr + 4 * k not in self.lds:
                        raise RuntimeError("LDS read of unwritten address %d" % (addr + 4 * k))
                self.wr(a[0], lane, sum(self.lds[addr + 4 * k] << (32 * k) for k in range(4)), 4)

    def op_ds_write_b64(self, a, m):
        off = m.get("offset", 0)
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = self.rd(a[0], lane) + off
                assert addr % 8 == 0
                val = self.rd(a[1], lane, 2)
                for k in range(2):
                    self.lds[addr + 4 * k] = (val >> (32 * k)) & M32

    def op_ds_write_b128(self, a, m):
        off = m.get("offset", 0)
        for lane in range(self.n):
            if (self.exec >> lane) & 1:
                addr = self.rd(a[0], lane) + off
                assert addr % 16 == 0
                val = self.rd(a[1], lane, 4)
                for k in range(4):
                    self.lds[addr + 4 * k] = (val >> (32 * k)) & M32

    # ---- vector
    def op_v_cmp_ne_u32_e32(self, a, l, sl):     # vcc[lane] = src0 != src1 (lanes EXEC has switched off keep their bit clear)
        assert a[0].strip() == "vcc"
        if l == min(i for i in range(self.n) if (self.exec >> i) & 1):
            self.vcc = 0
        if (self.rd(a[1], l) & M32) != (self.rd(a[2], l) & M32):
            self.vcc |= 1 << l

    def op_v_swap_b32(self, a, l, sl):
        x, y = self.rd(a[0], l), self.rd(a[1], l)
        self.wr(a[0], l, y)
        self.wr(a[1], l, x)

    def op_v_mov_b32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[1], sl))

    def op_v_cndmask_b32(self, a, l, sl):        # d = vcc[lane] ? src1 : src0
        assert a[3].strip() == "vcc"
        self.wr(a[0], l, self.rd(a[2], l) if (self.vcc >> l) & 1 else self.rd(a[1], l))

    def op_v_add_u32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[1], sl) + self.rd(a[2], l))

    def op_v_sub_u32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[1], sl) - self.rd(a[2], l))

    def op_v_subrev_u32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[2], l) - self.rd(a[1], sl))

    def op_v_and_b32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[1], sl) & self.rd(a[2], l))

    def op_v_xad_u32(self, a, l, sl):
        self.wr(a[0], l, (self.rd(a[1], l) ^ self.rd(a[2], l)) + self.rd(a[3], l))

    def op_v_lshrrev_b32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[2], l) >> (self.rd(a[1], l) & 31))

    def op_v_bfe_u32(self, a, l, sl):
        x, off, w = self.rd(a[1], l), self.rd(a[2], l) & 31, self.rd(a[3], l) & 31
        self.wr(a[0], l, (x >> off) & ((1 << w) - 1))

    def op_v_lshlrev_b32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[2], l) << (self.rd(a[1], l) & 31))

    def op_v_ashrrev_i32(self, a, l, sl):
        self.wr(a[0], l, s32(self.rd(a[2], l)) >> (self.rd(a[1], l) & 31))

    def op_v_lshl_add_u32(self, a, l, sl):
        self.wr(a[0], l, (self.rd(a[1], l) << (self.rd(a[2], l) & 31)) + self.rd(a[3], l))

    def op_v_bfe_i32(self, a, l, sl):
        x, off, w = self.rd(a[1], l), self.rd(a[2], l) & 31, self.rd(a[3], l) & 31
        f = (x >> off) & ((1 << w) - 1)
        self.wr(a[0], l, f - (1 << w) if w and (f >> (w - 1)) else f)

    def op_v_mul_lo_u32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[1], l) * self.rd(a[2], l))

    def op_v_mul_i32_i24(self, a, l, sl):
        self.wr(a[0], l, s24(self.rd(a[1], l)) * s24(self.rd(a[2], l)))

    def op_v_mad_i32_i24(self, a, l, sl):
        self.wr(a[0], l, s24(self.rd(a[1], l)) * s24(self.rd(a[2], l)) + self.rd(a[3], l))

    def op_v_mad_i64_i32(self, a, l, sl):      # vdst, sdst (carry out, ignored), a, b, c
        self.wr(a[0], l, s32(self.rd(a[2], l)) * s32(self.rd(a[3], l)) + self.rd(a[4], l, 2), 2)

    def op_v_mad_u64_u32(self, a, l, sl):
        self.wr(a[0], l, self.rd(a[2], l) * self.rd(a[3], l) + self.rd(a[4], l, 2), 2)

    def op_v_lshl_add_u64(self, a, l, sl):
        self.wr(a[0], l, (self.rd(a[1], l, 2) << (self.rd(a[2], l) & 63)) + self.rd(a[3], l, 2), 2)

    def op_v_ashrrev_i64(self, a, l, sl):
        self.wr(a[0], l, s64(self.rd(a[2], l, 2)) >> (self.rd(a[1], l) & 63), 2)

    def op_v_mbcnt_lo_u32_b32(self, a, l, sl):
        self.wr(a[0], l, min(l, 32) + self.rd(a[2], l))      # mask -1: the number of lower lanes among lanes 0..31

    def op_v_mbcnt_hi_u32_b32(self, a, l, sl):
        self.wr(a[0], l, max(l - 32, 0) + self.rd(a[2], l))
