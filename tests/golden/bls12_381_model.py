"""Independent big-integer model of the BLS12-381 tower, groups and optimal-ate pairing.

TEST INFRASTRUCTURE ONLY.  This is the build's own pure-Python model (Python ints, no
Montgomery form, no limb arithmetic).  It exists to (1) generate the committed golden vectors in
tests/golden/*.json and (2) cross-check the C oracle in oracle/ by a second, structurally different
implementation.  It is NOT reference output: the reference (0xWOLAND/zkvm-pairings) cannot be built
here (no Rust toolchain) and its src/pairings.rs is empty.

Representation differs on purpose from the C oracle / HIP code:
  * Fp      : Python int in [0, p)
  * Fp2     : (c0, c1)                       u^2 = -1             (reference src/fp2.rs:10-15)
  * Fp12    : flat list of 6 Fp2 = sum a_i w^i with w^6 = xi = 1+u  (polynomial basis)
              tower order used at the boundary (reference src/fp12.rs:13-16, src/fp6.rs:13-17):
              c0 = (a0, a2, a4), c1 = (a1, a3, a5)   since v = w^2.
  * Fp6     : (c0, c1, c2) of Fp2, v^3 = xi
Multiplication in Fp12 is schoolbook polynomial multiplication mod (w^6 - xi): no Karatsuba, no
sparse tricks, so agreement with the tower formulas of the oracle is a real cross-check.
"""

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_X = 0xD201000000010000  # |x|; the curve parameter is -BLS_X (reference src/common.rs:72)
BLS_X_IS_NEGATIVE = True

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    (
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)
# nontrivial cube root of unity in Fp (reference src/common.rs:83-90)
BETA = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE


# ----------------------------------------------------------------------------- Fp
def fp_inv(a):
    return pow(a, P - 2, P)


def fp_sqrt(a):
    """(p+1)/4 exponent; None if not a residue (reference src/fp.rs:280-300)."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


# ----------------------------------------------------------------------------- Fp2
F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_muls(a, s):
    return (a[0] * s % P, a[1] * s % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_inv(a):
    t = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * t % P, (-a[1] * t) % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


def f2_mul_xi(a):
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


# ----------------------------------------------------------------------------- Fp12 (polynomial basis in w)
def f12_one():
    return [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    acc = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            acc[i + j] = f2_add(acc[i + j], f2_mul(a[i], b[j]))
    out = list(acc[:6])
    for k in range(6, 11):
        out[k - 6] = f2_add(out[k - 6], f2_mul_xi(acc[k]))
    return out


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    """x -> x^(p^6): negates odd powers of w (reference src/fp12.rs:123-125)."""
    return [a[i] if i % 2 == 0 else f2_neg(a[i]) for i in range(6)]


def f12_pow(a, e):
    r = f12_one()
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


# gamma_i = xi^(i (p-1)/6): TRUE Frobenius coefficients
_GAMMA = [f2_pow(XI, i * (P - 1) // 6) for i in range(6)]


def f12_frob(a):
    """TRUE x -> x^p.  (The reference's Fp6::frobenius_map, src/fp6.rs:142-176, is NOT this map.)"""
    return [f2_mul(f2_conj(a[i]), _GAMMA[i]) for i in range(6)]


def f12_inv(a):
    # a = A + B w with A,B in Fp6; use norm to Fp6 then Fp6 inverse, done generically via
    # a^-1 = conj(a) * (a*conj(a))^-1 where a*conj(a) lies in Fp6 (even powers of w only).
    ac = f12_conj(a)
    n = f12_mul(a, ac)  # odd coefficients vanish
    assert n[1] == F2_ZERO and n[3] == F2_ZERO and n[5] == F2_ZERO
    n6 = (n[0], n[2], n[4])
    n6i = f6_inv(n6)
    ninv = [n6i[0], F2_ZERO, n6i[1], F2_ZERO, n6i[2], F2_ZERO]
    return f12_mul(ac, ninv)


def f12_to_tower(a):
    """-> ((c0.c0,c0.c1,c0.c2),(c1.c0,c1.c1,c1.c2)) of Fp2."""
    return ((a[0], a[2], a[4]), (a[1], a[3], a[5]))


def f12_from_tower(t):
    (c0, c1) = t
    return [c0[0], c1[0], c0[1], c1[1], c0[2], c1[2]]


def f12_flat_ints(a):
    """12 Fp ints in boundary order c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1."""
    t = f12_to_tower(a)
    out = []
    for c6 in t:
        for c2 in c6:
            out += [c2[0], c2[1]]
    return out


def f12_from_flat_ints(v):
    t = (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))
    return f12_from_tower(t)


# ----------------------------------------------------------------------------- Fp6 (generic, for tower-level vectors)
def f6_mul(a, b):
    acc = [F2_ZERO] * 5
    for i in range(3):
        for j in range(3):
            acc[i + j] = f2_add(acc[i + j], f2_mul(a[i], b[j]))
    return (
        f2_add(acc[0], f2_mul_xi(acc[3])),
        f2_add(acc[1], f2_mul_xi(acc[4])),
        acc[2],
    )


def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_neg(a):
    return tuple(f2_neg(x) for x in a)


def f6_mul_by_v(a):
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_inv(a):
    # generic: solve via adjugate formulas (same maths as reference src/fp6.rs:291-309, restated)
    c0 = f2_sub(f2_sqr(a[0]), f2_mul_xi(f2_mul(a[1], a[2])))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a[2])), f2_mul(a[0], a[1]))
    c2 = f2_sub(f2_sqr(a[1]), f2_mul(a[0], a[2]))
    t = f2_add(f2_mul_xi(f2_add(f2_mul(a[1], c2), f2_mul(a[2], c1))), f2_mul(a[0], c0))
    ti = f2_inv(t)
    return (f2_mul(ti, c0), f2_mul(ti, c1), f2_mul(ti, c2))


def f6_frob_true(a):
    g1 = f2_pow(XI, (P - 1) // 3)
    g2 = f2_pow(XI, 2 * (P - 1) // 3)
    return (f2_conj(a[0]), f2_mul(f2_conj(a[1]), g1), f2_mul(f2_conj(a[2]), g2))


# constants the reference multiplies by (src/fp6.rs:150-171); both are elements of Fp
REF_FROB6_C1 = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE
REF_FROB6_C2 = 0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC


def f6_frob_refcompat(a):
    """What the reference's Fp6::frobenius_map actually computes (src/fp6.rs:142-176); not x^p."""
    return (f2_conj(a[0]), f2_muls(f2_conj(a[1]), REF_FROB6_C1), f2_muls(f2_conj(a[2]), REF_FROB6_C2))


# ----------------------------------------------------------------------------- G1 / G2 (affine, None = infinity)
def g1_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * fp_inv(2 * y1) % P
    else:
        lam = (y2 - y1) * fp_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def g1_neg(p1):
    return None if p1 is None else (p1[0], (-p1[1]) % P)


def g1_mul(p1, k):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, p1)
    return acc


def g1_on_curve(p1):
    x, y = p1
    return (y * y - x * x * x - 4) % P == 0


def g1_torsion_free(p1):
    """-[X^2]P == (beta*x, y)   (reference src/g1.rs:111-115)."""
    lhs = g1_neg(g1_mul(g1_mul(p1, BLS_X), BLS_X))
    rhs = (p1[0] * BETA % P, p1[1])
    return lhs == rhs


def g2_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return None
        lam = f2_mul(f2_muls(f2_sqr(x1), 3), f2_inv(f2_muls(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_neg(p1):
    return None if p1 is None else (p1[0], f2_neg(p1[1]))


def g2_mul(p1, k):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p1)
    return acc


def g2_on_curve(p1):
    x, y = p1
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), (4, 4))) == F2_ZERO


PSI_X = f2_inv(f2_pow(XI, (P - 1) // 3))
PSI_Y = f2_inv(f2_pow(XI, (P - 1) // 2))


def g2_psi(p1):
    return (f2_mul(f2_conj(p1[0]), PSI_X), f2_mul(f2_conj(p1[1]), PSI_Y))


def g2_torsion_free(p1):
    """psi(P) == -[X]P   (reference src/g2.rs:166-170)."""
    return g2_psi(p1) == g2_neg(g2_mul(p1, BLS_X))


# ----------------------------------------------------------------------------- pairing
def _sparse_014(c0, c1, c4):
    """element with tower coeffs c0.c0=c0, c0.c1=c1, c1.c1=c4  (what Fp12::mul_by_014 multiplies by,
    reference src/fp12.rs:99-111) -> polynomial basis: a0=c0, a2=c1, a3=c4."""
    return [c0, F2_ZERO, c1, c4, F2_ZERO, F2_ZERO]


def doubling_step(r):
    """Jacobian-like doubling with line coefficients (Alg. 26 of ePrint 2010/354 as used by the
    zkcrypto-shaped pairing this tower's API was cut for)."""
    x, y, z = r
    tmp0 = f2_sqr(x)
    tmp1 = f2_sqr(y)
    tmp2 = f2_sqr(tmp1)
    tmp3 = f2_sub(f2_sub(f2_sqr(f2_add(tmp1, x)), tmp0), tmp2)
    tmp3 = f2_add(tmp3, tmp3)
    tmp4 = f2_add(f2_add(tmp0, tmp0), tmp0)
    tmp6 = f2_add(x, tmp4)
    tmp5 = f2_sqr(tmp4)
    zsq = f2_sqr(z)
    nx = f2_sub(f2_sub(tmp5, tmp3), tmp3)
    nz = f2_sub(f2_sub(f2_sqr(f2_add(z, y)), tmp1), zsq)
    ny = f2_mul(f2_sub(tmp3, nx), tmp4)
    t2 = f2_muls(tmp2, 8)
    ny = f2_sub(ny, t2)
    t3 = f2_mul(tmp4, zsq)
    t3 = f2_neg(f2_add(t3, t3))
    t6 = f2_sub(f2_sub(f2_sqr(tmp6), tmp0), tmp5)
    t6 = f2_sub(t6, f2_muls(tmp1, 4))
    t0 = f2_mul(nz, zsq)
    t0 = f2_add(t0, t0)
    return (nx, ny, nz), (t0, t3, t6)


def addition_step(r, q):
    """Mixed addition with line coefficients (Alg. 27 of ePrint 2010/354)."""
    x, y, z = r
    qx, qy = q
    zsq = f2_sqr(z)
    ysq = f2_sqr(qy)
    t0 = f2_mul(zsq, qx)
    t1 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(qy, z)), ysq), zsq), zsq)
    t2 = f2_sub(t0, x)
    t3 = f2_sqr(t2)
    t4 = f2_muls(t3, 4)
    t5 = f2_mul(t4, t2)
    t6 = f2_sub(f2_sub(t1, y), y)
    t9 = f2_mul(t6, qx)
    t7 = f2_mul(t4, x)
    nx = f2_sub(f2_sub(f2_sub(f2_sqr(t6), t5), t7), t7)
    nz = f2_sub(f2_sub(f2_sqr(f2_add(z, t2)), zsq), t3)
    t10 = f2_add(qy, nz)
    t8 = f2_mul(f2_sub(t7, nx), t6)
    t0 = f2_mul(y, t5)
    t0 = f2_add(t0, t0)
    ny = f2_sub(t8, t0)
    t10 = f2_sub(f2_sqr(t10), ysq)
    ztsq = f2_sqr(nz)
    t10 = f2_sub(t10, ztsq)
    t9 = f2_sub(f2_add(t9, t9), t10)
    t10 = f2_add(nz, nz)
    t6 = f2_neg(t6)
    t1 = f2_add(t6, t6)
    return (nx, ny, nz), (t10, t1, t9)


def ell(f, coeffs, p1):
    c0 = f2_muls(coeffs[0], p1[1])
    c1 = f2_muls(coeffs[1], p1[0])
    return f12_mul(f, _sparse_014(coeffs[2], c1, c0))


def multi_miller_loop(pairs):
    """pairs: list of (G1 affine or None, G2 affine or None). Pairs with an infinity contribute 1."""
    live = [(p1, q) for (p1, q) in pairs if p1 is not None and q is not None]
    rs = [(q[0], q[1], F2_ONE) for (_, q) in live]
    f = f12_one()
    bits = bin(BLS_X >> 1)[2:]
    for bit in bits[1:]:
        for i, (p1, q) in enumerate(live):
            rs[i], c = doubling_step(rs[i])
            f = ell(f, c, p1)
        if bit == "1":
            for i, (p1, q) in enumerate(live):
                rs[i], c = addition_step(rs[i], q)
                f = ell(f, c, p1)
        f = f12_sqr(f)
    for i, (p1, q) in enumerate(live):
        rs[i], c = doubling_step(rs[i])
        f = ell(f, c, p1)
    if BLS_X_IS_NEGATIVE:
        f = f12_conj(f)
    return f


def cyclotomic_exp(f):
    """f^|x| then conjugate (x negative); valid in the cyclotomic subgroup."""
    return f12_conj(f12_pow(f, BLS_X))


def final_exponentiation(f):
    """Upstream-shaped chain: returns f^(3 (p^12-1)/r)  (checked against direct exponentiation in
    tests/golden/gen_fixtures.py)."""
    t0 = f
    for _ in range(6):
        t0 = f12_frob(t0)
    t1 = f12_inv(f)
    t2 = f12_mul(t0, t1)
    t1 = t2
    t2 = f12_frob(f12_frob(t2))
    t2 = f12_mul(t2, t1)
    t1 = f12_conj(f12_sqr(t2))
    t3 = cyclotomic_exp(t2)
    t4 = f12_sqr(t3)
    t5 = f12_mul(t1, t3)
    t1 = cyclotomic_exp(t5)
    t0 = cyclotomic_exp(t1)
    t6 = cyclotomic_exp(t0)
    t6 = f12_mul(t6, t4)
    t4 = cyclotomic_exp(t6)
    t5 = f12_conj(t5)
    t4 = f12_mul(t4, f12_mul(t5, t2))
    t5 = f12_conj(t2)
    t1 = f12_mul(t1, t2)
    t1 = f12_frob(f12_frob(f12_frob(t1)))
    t6 = f12_mul(t6, t5)
    t6 = f12_frob(t6)
    t3 = f12_mul(t3, t0)
    t3 = f12_frob(f12_frob(t3))
    t3 = f12_mul(t3, t1)
    t3 = f12_mul(t3, t6)
    return f12_mul(t3, t4)


def final_exponentiation_direct(f):
    return f12_pow(f, 3 * (P**12 - 1) // R_ORDER)


def pairing(p1, q):
    return final_exponentiation(multi_miller_loop([(p1, q)]))


def miller_affine(p1, q):
    """Independent Miller formulation: affine slopes (reference-style G2 double/add, src/g2.rs:81-105,
    210-242) with line l(P) = (lam*x_T - y_T) - lam*x_P * w^2 ... placed as mul_by_014(c0, c1, c4).
    Equal to multi_miller_loop only AFTER final exponentiation (differs by Fp2 factors)."""
    f = f12_one()
    t = q
    bits = bin(BLS_X)[2:]

    def line(lam, tpt):
        c0 = f2_sub(f2_mul(lam, tpt[0]), tpt[1])
        c1 = f2_neg(f2_muls(lam, p1[0]))
        c4 = (p1[1], 0)
        return _sparse_014(c0, c1, c4)

    for bit in bits[1:]:
        lam = f2_mul(f2_muls(f2_sqr(t[0]), 3), f2_inv(f2_muls(t[1], 2)))
        f = f12_mul(f12_sqr(f), line(lam, t))
        t = g2_add(t, t)
        if bit == "1":
            lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
            f = f12_mul(f, line(lam, t))
            t = g2_add(t, q)
    return f12_conj(f)


# ----------------------------------------------------------------------------- misc helpers
def limbs64(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def from_limbs64(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def hex_fp(v):
    """Same text as the reference's Debug for Fp: 0x + 96 hex digits big-endian (src/fp.rs:26-35)."""
    return "0x%096x" % v


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n):
        """uniform-ish integer in [0, n) from 512 random bits (bias < 2^-128)."""
        v = 0
        for _ in range(8):
            v = (v << 64) | self.next()
        return v % n
