"""Limb-exact CPU model of the division-step inversion in zkvm_pairings_amd/csrc/zkp_coop.hip (f_inv): int32 limbs of 30 bits,
int64 accumulators, 37 batches of 30 steps (delta = 1 variant of Bernstein-Yang "safegcd").  `python3 tools/safegcd_model.py`
checks it against pow(x, -1, p) on random and edge values."""
import random
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
NL, W = 13, 30
M30 = (1 << W) - 1
NB = 37
PINV30 = pow(P, -1, 1 << W)
def s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x
def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x
def limbs(v):
    out = []
    for i in range(NL - 1):
        out.append(v & M30); v >>= W
    out.append(v); return out
def val(l): return sum(x << (W * i) for i, x in enumerate(l))
PL = limbs(P)
def divsteps30(eta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g = f0 & 0xFFFFFFFF, g0 & 0xFFFFFFFF
    for _ in range(30):
        c1 = 0xFFFFFFFF if s32(eta) < 0 else 0
        c2 = (-(g & 1)) & 0xFFFFFFFF
        x = ((f ^ c1) - c1) & 0xFFFFFFFF; y = ((u ^ c1) - c1) & 0xFFFFFFFF; z = ((v ^ c1) - c1) & 0xFFFFFFFF
        g = (g + (x & c2)) & 0xFFFFFFFF; q = (q + (y & c2)) & 0xFFFFFFFF; r = (r + (z & c2)) & 0xFFFFFFFF
        c1 &= c2
        eta = ((eta ^ c1) - (c1 + 1)) & 0xFFFFFFFF
        f = (f + (g & c1)) & 0xFFFFFFFF; u = (u + (q & c1)) & 0xFFFFFFFF; v = (v + (r & c1)) & 0xFFFFFFFF
        g >>= 1; u = (u << 1) & 0xFFFFFFFF; v = (v << 1) & 0xFFFFFFFF
    return eta, (s32(u), s32(v), s32(q), s32(r))
def update_fg(f, g, t):
    u, v, q, r = t
    cf = s64(u * f[0] + v * g[0]); cg = s64(q * f[0] + r * g[0])
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= W; cg >>= W
    for i in range(1, NL):
        cf = s64(cf + u * f[i] + v * g[i]); cg = s64(cg + q * f[i] + r * g[i])
        f[i - 1] = cf & M30; g[i - 1] = cg & M30
        cf >>= W; cg >>= W
    f[NL - 1] = s32(cf); g[NL - 1] = s32(cg)
    assert f[NL-1] == cf and g[NL-1] == cg
def update_de(d, e, t):
    u, v, q, r = t
    sd = -1 if d[NL - 1] < 0 else 0
    se = -1 if e[NL - 1] < 0 else 0
    md = (u & sd) + (v & se); me = (q & sd) + (r & se)
    cd = s64(u * d[0] + v * e[0]); ce = s64(q * d[0] + r * e[0])
    md -= (PINV30 * (cd & 0xFFFFFFFF) + md) & M30
    me -= (PINV30 * (ce & 0xFFFFFFFF) + me) & M30
    md = s32(md); me = s32(me)
    cd = s64(cd + PL[0] * md); ce = s64(ce + PL[0] * me)
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= W; ce >>= W
    for i in range(1, NL):
        cd = s64(cd + u * d[i] + v * e[i] + PL[i] * md); ce = s64(ce + q * d[i] + r * e[i] + PL[i] * me)
        d[i - 1] = cd & M30; e[i - 1] = ce & M30
        cd >>= W; ce >>= W
    d[NL - 1] = s32(cd); e[NL - 1] = s32(ce)
    assert d[NL-1] == cd and e[NL-1] == ce
def normalize(r, sign):
    # r in (-2p, p); sign < 0 -> negate.  result in [0, p)
    add = -1 if r[NL - 1] < 0 else 0
    neg = -1 if sign < 0 else 0
    c = 0
    for i in range(NL):
        x = r[i] + (PL[i] & add)
        x = (x ^ neg) - neg
        x += c
        if i < NL - 1:
            r[i] = x & M30; c = x >> W
        else:
            r[i] = x
    add = -1 if r[NL - 1] < 0 else 0
    c = 0
    for i in range(NL):
        x = r[i] + (PL[i] & add) + c
        if i < NL - 1:
            r[i] = x & M30; c = x >> W
        else:
            r[i] = x
    return r
def inv(x):
    f = list(PL); g = limbs(x); d = [0] * NL; e = [1] + [0] * (NL - 1)
    eta = -1
    maxd = 0
    for _ in range(NB):
        eta, t = divsteps30(eta, f[0], g[0])
        update_de(d, e, t)
        update_fg(f, g, t)
        assert -2 * P < val(d) < P and -2 * P < val(e) < P, (val(d) / P, val(e) / P)
    assert val(g) == 0 and val(f) in (1, -1)
    r = normalize(d, f[NL - 1])
    return val(r)
def selftest(n=2000, seed=7):
    rng = random.Random(seed)
    for i in range(n):
        x = rng.randrange(1, P)
        assert inv(x) == pow(x, -1, P), i
    for x in (1, 2, 3, P - 1, P - 2, (P + 1) // 2, 1 << 380, (1 << 381) % P):
        assert inv(x) == pow(x, -1, P)
    return True


if __name__ == "__main__":
    selftest()
    print("limb model ok: 2000 random and the edge values agree with pow(x, -1, p)")
    print("PINV30", PINV30, "PL", PL)
