"""The pairing pinned to its DEFINITION, independently of the oracle and of tests/golden/bls12_381_model.py.

The reference has no pairing and no pairing vectors, so the oracle's e(P, Q) cannot be compared with a
reference output.  What can be done without any third-party library is to compute the reduced TATE pairing
    t(Q, P) = f_{r,Q}(P)^((p^12 - 1) / r)
straight from its textbook definition with none of the machinery the engine uses:
  * Fp12 = Fp[w] / (w^12 - 2 w^6 + 2) as plain polynomials over Fp (no tower; w^6 = 1 + u, u^2 = -1),
  * Q untwisted into E(Fp12): y^2 = x^3 + 4, affine chord-and-tangent arithmetic with polynomial inversions (extended Euclid),
  * Miller's loop over the 255 bits of the group order r (not over the curve parameter x),
  * one plain square-and-multiply by (p^12 - 1) / r (no easy part / hard part, no Frobenius),
and to check the relation that ties the optimal ate pairing to it (Hess, Smart, Vercauteren, "The Eta Pairing Revisited",
Theorem 1; derivation: f_{T^k,Q}(P) = f_{T,Q}(P)^c with c = sum_i T^(k-1-i) p^i because [T]Q = pi(Q) on G2 and P is
Fp-rational, and f_{T^k,Q} = f_{T^k-1,Q} = f_{r,Q}^((T^k-1)/r)):

    ate(Q, P)^c = t(Q, P)^((T^12 - 1) / r),      T = x = -0xd201000000010000,  ate = f_{T,Q}(P)^((p^12-1)/r)

The engine's pairing is e(P, Q) = ate(Q, P)^3 (the final exponentiation carries the factor 3; T < 0 is the
conjugation of the Miller value), so the test asserts   e^c == t^(3 (T^12 - 1) / r)   in mu_r."""

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X = -0xd201000000010000


# ---- Fp12 = Fp[w] / (w^12 - 2 w^6 + 2): lists of 12 ints, index = power of w
def f_one():
    return [1] + [0] * 11


def f_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f_scal(a, s):
    return [x * s % P for x in a]


def f_mul(a, b):
    c = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                c[i + j] += x * y
    for k in range(22, 11, -1):        # w^12 = 2 w^6 - 2
        v = c[k]
        if v:
            c[k - 6] += 2 * v
            c[k - 12] -= 2 * v
    return [v % P for v in c[:12]]


def f_pow(a, e):
    assert e >= 0
    r = f_one()
    for bit in bin(e)[2:]:
        r = f_mul(r, r)
        if bit == "1":
            r = f_mul(r, a)
    return r


def _poly_trim(a):
    while a and a[-1] == 0:
        a = a[:-1]
    return a


def _poly_divmod(a, b):
    a = a[:]
    q = [0] * max(1, len(a) - len(b) + 1)
    inv = pow(b[-1], -1, P)
    while len(a) >= len(b) and a:
        k = a[-1] * inv % P
        d = len(a) - len(b)
        q[d] = k
        for i, y in enumerate(b):
            a[d + i] = (a[d + i] - k * y) % P
        a = _poly_trim(a)
    return _poly_trim(q), a


def _poly_mul(a, b):
    if not a or not b:
        return []
    c = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            c[i + j] = (c[i + j] + x * y) % P
    return _poly_trim(c)


def _poly_sub(a, b):
    n = max(len(a), len(b))
    a = a + [0] * (n - len(a))
    b = b + [0] * (n - len(b))
    return _poly_trim([(x - y) % P for x, y in zip(a, b)])


def f_inv(a):
    """extended Euclid in Fp[w] against the modulus w^12 - 2 w^6 + 2"""
    mod = [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
    r0, r1 = mod, _poly_trim(a[:])
    assert r1, "zero has no inverse"
    s0, s1 = [], [1]
    while r1:
        q, rem = _poly_divmod(r0, r1)
        r0, r1 = r1, rem
        s0, s1 = s1, _poly_sub(s0, _poly_mul(q, s1))
    assert len(r0) == 1                     # gcd is a constant: the modulus is irreducible
    k = pow(r0[0], -1, P)
    out = [x * k % P for x in s0]
    return out + [0] * (12 - len(out))


def fp2_at(c0, c1, k):
    """(c0 + c1 u) w^k with u = w^6 - 1, k < 6"""
    out = [0] * 12
    out[k] = (c0 - c1) % P
    out[k + 6] = c1 % P
    return out


def from_tower_wire(flat):
    """12 Fp in the wire order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (c_i.c_j = Fp2 coefficient of v^j w^i, v = w^2)"""
    acc = [0] * 12
    for i in range(2):
        for j in range(3):
            c0, c1 = flat[(i * 3 + j) * 2], flat[(i * 3 + j) * 2 + 1]
            acc = f_add(acc, fp2_at(c0, c1, 2 * j + i))
    return acc


def untwist(qx, qy):
    """E'(Fp2): y^2 = x^3 + 4(1+u)  ->  E(Fp12): y^2 = x^3 + 4,  (x', y') -> (x' / w^2, y' / w^3)"""
    w = [0, 1] + [0] * 10
    w2i = f_inv(f_mul(w, w))
    w3i = f_inv(f_mul(f_mul(w, w), w))
    x = f_mul(fp2_at(qx[0], qx[1], 0), w2i)
    y = f_mul(fp2_at(qy[0], qy[1], 0), w3i)
    # on the curve?
    assert f_sub(f_mul(y, y), f_add(f_mul(f_mul(x, x), x), [4] + [0] * 11)) == [0] * 12
    return x, y


def tate(qx, qy, px, py):
    """t(Q, P) = f_{r,Q}(P)^((p^12-1)/r); Q = (qx, qy) in E'(Fp2) as pairs of ints, P = (px, py) in E(Fp).
    Vertical lines are left out: they take values in Fp6 (x of every multiple of the untwisted Q is x'/w^2) and
    vanish under the exponentiation."""
    xq, yq = untwist(qx, qy)
    xt, yt = xq, yq
    f = f_one()
    cx = [px % P] + [0] * 11
    cy = [py % P] + [0] * 11
    bits = bin(R)[3:]
    inf = False
    for n, bit in enumerate(bits):
        # tangent at T
        lam = f_mul(f_scal(f_mul(xt, xt), 3), f_inv(f_scal(yt, 2)))
        line = f_sub(f_sub(cy, yt), f_mul(lam, f_sub(cx, xt)))
        f = f_mul(f_mul(f, f), line)
        x3 = f_sub(f_mul(lam, lam), f_scal(xt, 2))
        yt = f_sub(f_mul(lam, f_sub(xt, x3)), yt)
        xt = x3
        if bit == "1":
            if xt == xq:
                # T = -Q: only at the very last bit ([r-1]Q = -Q); the line through T and Q is the vertical, left out
                assert n == len(bits) - 1 and f_add(yt, yq) == [0] * 12
                inf = True
            else:
                lam = f_mul(f_sub(yt, yq), f_inv(f_sub(xt, xq)))
                line = f_sub(f_sub(cy, yt), f_mul(lam, f_sub(cx, xt)))
                f = f_mul(f, line)
                x3 = f_sub(f_sub(f_mul(lam, lam), xt), xq)
                yt = f_sub(f_mul(lam, f_sub(xt, x3)), yt)
                xt = x3
    assert inf, "[r]Q must be the point at infinity"
    assert (P ** 12 - 1) % R == 0
    return f_pow(f, (P ** 12 - 1) // R)


def ate_relation_exponents():
    """(c mod r, 3 (T^12 - 1)/r mod r)"""
    c = sum(X ** (11 - i) * P ** i for i in range(12))
    assert (X ** 12 - 1) % R == 0
    m = (X ** 12 - 1) // R
    return c % R, 3 * m % R
