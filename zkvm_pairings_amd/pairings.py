"""Host-side mirror of the reference crate's pairing API (the module that is EMPTY upstream:
/root/reference/src/pairings.rs, declared at src/lib.rs:12), backed by the GPU engine.

    pairing(&G1Affine, &G2Affine) -> Gt
    multi_miller_loop(&[(&G1Affine, &G2Affine)]) -> MillerLoopResult
    MillerLoopResult::final_exponentiation() -> Gt
    Gt::identity()

Names, argument meaning and the infinity convention follow the reference's types
(G1Affine{x,y,is_infinity} src/g1.rs:7-11, G2Affine src/g2.rs:8-12, Fp12::one src/fp12.rs:87).
Every call executes on the GPU through the C ABI; there is no host arithmetic here."""
import numpy as np

from . import synthetic
from .engine import PairingEngine

_default = None


def default_engine():
    global _default
    if _default is None:
        _default = PairingEngine(0)
    return _default


def _limbs(v):
    return [(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


class G1Affine:
    """reference src/g1.rs:7-11; identity is (0, 1, infinity) (src/g1.rs:25-31)."""

    def __init__(self, x, y, is_infinity=False):
        self.x, self.y, self.is_infinity = int(x), int(y), bool(is_infinity)

    @classmethod
    def identity(cls):
        return cls(0, 1, True)

    @classmethod
    def generator(cls):
        return cls.from_array(synthetic.G1_GENERATOR)

    @classmethod
    def from_array(cls, a, inf=False):
        a = np.asarray(a, dtype=np.uint64).reshape(12)
        f = lambda r: sum(int(v) << (64 * i) for i, v in enumerate(r))
        return cls(f(a[:6]), f(a[6:]), inf)

    def to_array(self):
        return np.array(_limbs(self.x) + _limbs(self.y), dtype=np.uint64)

    def is_identity(self):
        return self.is_infinity

    def is_valid(self, engine=None):
        """Ok(()) / Err(String) of src/g1.rs:49-62 as None / message."""
        st = (engine or default_engine()).g1_is_valid(self.to_array(), [1 if self.is_infinity else 0])[0]
        return {0: None, 1: "Point is not on curve", 2: "Point is not torsion free"}[int(st)]

    def __mul__(self, k):
        if self.is_infinity:
            return G1Affine.identity()
        out, inf = default_engine().g1_mul(self.to_array(), synthetic.int_to_scalar(int(k) % synthetic.R_ORDER))
        return G1Affine.from_array(out[0], bool(inf[0]))

    def __eq__(self, o):  # src/g1.rs:13-17 compares coordinates only
        return self.x == o.x and self.y == o.y


class G2Affine:
    """reference src/g2.rs:8-12."""

    def __init__(self, x, y, is_infinity=False):
        self.x, self.y, self.is_infinity = (int(x[0]), int(x[1])), (int(y[0]), int(y[1])), bool(is_infinity)

    @classmethod
    def identity(cls):
        return cls((0, 0), (1, 0), True)

    @classmethod
    def generator(cls):
        return cls.from_array(synthetic.G2_GENERATOR)

    @classmethod
    def from_array(cls, a, inf=False):
        a = np.asarray(a, dtype=np.uint64).reshape(24)
        f = lambda r: sum(int(v) << (64 * i) for i, v in enumerate(r))
        return cls((f(a[0:6]), f(a[6:12])), (f(a[12:18]), f(a[18:24])), inf)

    def to_array(self):
        return np.array(_limbs(self.x[0]) + _limbs(self.x[1]) + _limbs(self.y[0]) + _limbs(self.y[1]), dtype=np.uint64)

    def is_identity(self):
        return self.is_infinity

    def is_valid(self, engine=None):
        st = (engine or default_engine()).g2_is_valid(self.to_array(), [1 if self.is_infinity else 0])[0]
        return {0: None, 1: "Point is not on curve", 2: "Point is not torsion free"}[int(st)]

    def __mul__(self, k):
        if self.is_infinity:
            return G2Affine.identity()
        out, inf = default_engine().g2_mul(self.to_array(), synthetic.int_to_scalar(int(k) % synthetic.R_ORDER))
        return G2Affine.from_array(out[0], bool(inf[0]))

    def __eq__(self, o):
        return self.x == o.x and self.y == o.y


class Gt:
    """Newtype over Fp12 (72 canonical u64 limbs); identity == Fp12::one() (src/fp12.rs:87-89)."""

    def __init__(self, limbs):
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(72)

    @classmethod
    def identity(cls):
        return cls(PairingEngine.gt_identity())

    def __eq__(self, o):  # limb equality, src/fp12.rs:46-50
        return bool(np.array_equal(self.limbs, o.limbs))

    def is_identity(self):
        return self == Gt.identity()


class MillerLoopResult:
    def __init__(self, limbs):
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(72)

    def final_exponentiation(self, engine=None):
        return Gt((engine or default_engine()).final_exponentiation(self.limbs)[0])

    def __eq__(self, o):
        return bool(np.array_equal(self.limbs, o.limbs))


def multi_miller_loop(terms, engine=None):
    """terms: sequence of (G1Affine, G2Affine).  Pairs with an identity on either side contribute one."""
    e = engine or default_engine()
    terms = list(terms)
    if not terms:
        return MillerLoopResult(PairingEngine.gt_identity())
    g1 = np.stack([p.to_array() for p, _ in terms])
    g2 = np.stack([q.to_array() for _, q in terms])
    i1 = np.array([p.is_infinity for p, _ in terms], dtype=np.uint8)
    i2 = np.array([q.is_infinity for _, q in terms], dtype=np.uint8)
    return MillerLoopResult(e.multi_miller_loop(g1, g2, len(terms), i1, i2)[0])


def final_exponentiation(ml, engine=None):
    return ml.final_exponentiation(engine)


def pairing(p, q, engine=None):
    e = engine or default_engine()
    out = e.pairing(p.to_array(), q.to_array(), [1 if p.is_infinity else 0], [1 if q.is_infinity else 0])
    return Gt(out[0])
