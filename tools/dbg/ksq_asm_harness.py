#!/usr/bin/env python3
"""Run the generated k_ksq body (tools/coopasm.py generate_ksq) on one wavefront of the GPU and on tools/asmemu.py, same inputs,
and report the first registers that differ.  Builds a small HIP program with every register of the block as an output.
    python3 tools/dbg/ksq_asm_harness.py           (on the GPU box)"""
import os, random, struct, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tools"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import asmemu, coopasm
import coopgen as cg

NL = 14
g = coopasm.generate_ksq()
regs_out = list(range(g.S, g.vend))
src = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "%(root)s/zkvm_pairings_amd/csrc/zkp_constants28.h"
#include "%(root)s/zkvm_pairings_amd/csrc/zkp_coop_mulacc.inc"
__global__ void __launch_bounds__(64, 3) k(const int* in, int* out) {
    extern __shared__ int4 parked[];
    const int lane = threadIdx.x;
    int x[56];
    for (int i = 0; i < 56; i++) x[i] = in[lane * 84 + i];
    for (int q = 0; q < 7; q++) parked[q * 64 + lane] = make_int4(in[lane * 84 + 56 + 4 * q], in[lane * 84 + 57 + 4 * q], in[lane * 84 + 58 + 4 * q], in[lane * 84 + 59 + 4 * q]);
    constexpr unsigned PL[14] = {ZKP28_P_LIMBS};
    int* const outl = out;
    const unsigned voff = lane * %(tot)d * 4;
    asm volatile(ZKP_KSQ_BODY_ASM
                 %(stores)s
                 "s_waitcnt vmcnt(0)\n\t"
                 : ZKP_KSQ_BODY_IO(x, x + 14, x + 28, x + 42)
                 : [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]), [p6] "s"(PL[6]),
                   [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]), [p12] "s"(PL[12]),
                   [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV), [outp] "s"(outl), [voff] "v"(voff)
                 : ZKP_KSQ_BODY_CLOBBERS);
    for (int q = 0; q < 7; q++) { int4 v = parked[q * 64 + lane]; int* d = out + lane * %(tot)d + 56 + %(nout)d + 4 * q; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
}
int main(int argc, char** argv) {
    std::vector<int> in(64 * 84), out(64 * %(tot)d);
    FILE* f = fopen(argv[1], "rb"); if (!f || fread(in.data(), 4, in.size(), f) != in.size()) return 2; fclose(f);
    int *di, *dout;
    hipMalloc(&di, in.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 7 * 64 * 16, 0, di, dout);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb"); fwrite(out.data(), 4, out.size(), f); fclose(f);
    return 0;
}
''' % {"root": R, "nout": len(regs_out), "tot": 56 + len(regs_out) + 28,
       "stores": " ".join('"global_store_dword %%[voff], v%d, %%[outp] offset:%d\\n\\t"' % (r, 4 * i) for i, r in enumerate(list(range(g.XR, g.XR + 56)) + regs_out))}
os.makedirs("/tmp/ksqh", exist_ok=True)
open("/tmp/ksqh/h.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "/tmp/ksqh/h.hip", "-o", "/tmp/ksqh/h"])

rng = random.Random(7)
lanes = 64
emu = asmemu.Emu(lanes=lanes, subst={**{"p%d" % i: (cg.P >> (28 * i)) & 0xfffffff for i in range(NL)}, "pinv": (-pow(cg.P, -1, 1 << 28)) % (1 << 28)}, strict=False)
words = []
for lane in range(lanes):
    vals = [cg.to_limbs_balanced(rng.randrange(cg.P) - cg.P // 2) for _ in range(4)]
    vals = [cg.vred(list(x)) if abs(cg.limbs_value(x)) >= 0.51 * cg.P else x for x in vals]
    mr, mi, o_r, oi = vals
    r = lane & 3
    if r & 1:
        f = [mr, mi, o_r, oi]
    else:
        xr, xi = [a + b for a, b in zip(mr, o_r)], [a + b for a, b in zip(mi, oi)]
        f = [xr, xi, [b - a for a, b in zip(xr, mi if r == 0 else oi)], [-a - b for a, b in zip(xi, mr if r == 0 else o_r)]]
    flat = [v for part in f for v in part] + list(mr) + list(mi)
    words += flat
    for j, base in enumerate((g.XR, g.XI, g.YR, g.YI)):
        for i in range(NL):
            emu.v.setdefault(base + i, [None] * lanes)[lane] = f[j][i] & asmemu.M32
    for k in range(28):
        emu.lds[(k // 4) * 1024 + 16 * lane + 4 * (k % 4)] = (list(mr) + list(mi))[k] & asmemu.M32
open("/tmp/ksqh/in.bin", "wb").write(struct.pack("<%di" % len(words), *words))
subprocess.check_call(["/tmp/ksqh/h", "/tmp/ksqh/in.bin", "/tmp/ksqh/out.bin"])
tot = 56 + len(regs_out) + 28
out = struct.unpack("<%dI" % (lanes * tot), open("/tmp/ksqh/out.bin", "rb").read())
emu.run(g.lines)
names = [(g.XR + i) for i in range(56)] + regs_out
bad = {}
for lane in range(lanes):
    for j, reg in enumerate(names):
        want = emu.v.get(reg, [None] * lanes)[lane]
        got = out[lane * tot + j]
        if want is not None and want != got:
            bad.setdefault(reg, []).append(lane)
    for k in range(28):
        want = emu.lds[(k // 4) * 1024 + 16 * lane + 4 * (k % 4)]
        if want != out[lane * tot + 56 + len(regs_out) + k]:
            bad.setdefault("lds%d" % k, []).append(lane)
print("registers that differ from the model (register: lanes):")
for reg in sorted(bad, key=lambda x: (isinstance(x, str), x)):
    print("  ", reg, bad[reg][:16], len(bad[reg]))
print("none" if not bad else "%d registers differ" % len(bad))
print("layout: XR %d XI %d YR %d YI %d S %d D %d ACC0 %d" % (g.XR, g.XI, g.YR, g.YI, g.S, g.D, g.ACC[0]))
def G(reg, lane):
    return asmemu.s32(out[lane * tot + names.index(reg)])
def Wn(reg, lane):
    v = emu.v.get(reg, [None] * lanes)[lane]
    return None if v is None else asmemu.s32(v)
for reg in (104, 90, 118, 132):
    print("v%d: got %s" % (reg, [G(reg, l) for l in range(8)]))
    print("      want %s" % [Wn(reg, l) for l in range(8)])
print("S v62 got ", [G(62, l) for l in range(8)])
print("S v62 want", [Wn(62, l) for l in range(8)])
print("v83..86 got", [[G(r, l) for l in range(4)] for r in (83, 84, 85, 86)])
