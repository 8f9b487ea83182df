"""Seeded synthetic inputs: pair i is ([a_i] G1gen, [b_i] G2gen) with scalars from a
SplitMix64 stream.  The scalar multiplications run on the GPU through the engine (zkp_g1_mul_batch /
zkp_g2_mul_batch); nothing here touches oracle/.

The reference's own G1Affine::random / G2Affine::random (src/g1.rs:64-72, src/g2.rs:71-79) return
points that are not on the curve, so they cannot be pairing inputs."""
import numpy as np

SEED = 0x5EEDB15381
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

G1_GENERATOR = np.array([
    0xfb3af00adb22c6bb, 0x6c55e83ff97a1aef, 0xa14e3a3f171bac58, 0xc3688c4f9774b905, 0x2695638c4fa9ac0f, 0x17f1d3a73197d794,
    0x0caa232946c5e7e1, 0xd03cc744a2888ae4, 0x00db18cb2c04b3ed, 0xfcf5e095d5d00af6, 0xa09e30ed741d8ae4, 0x08b3f481e3aaa0f1,
], dtype=np.uint64)  # reference src/common.rs:92-108
G2_GENERATOR = np.array([
    0xd48056c8c121bdb8, 0x0bac0326a805bbef, 0xb4510b647ae3d177, 0xc6e47ad4fa403b02, 0x260805272dc51051, 0x024aa2b2f08f0a91,
    0xe5ac7d055d042b7e, 0x334cf11213945d57, 0xb5da61bbdc7f5049, 0x596bd0d09920b61a, 0x7dacd3a088274f65, 0x13e02b6052719f60,
    0xe193548608b82801, 0x923ac9cc3baca289, 0x6d429a695160d12c, 0xadfd9baa8cbdd3a7, 0x8cc9cdc6da2e351a, 0x0ce5d527727d6e11,
    0xaaa9075ff05f79be, 0x3f370d275cec1da1, 0x267492ab572e99ab, 0xcb3e287e85a763af, 0x32acd2b02bc28b99, 0x0606c4a02ea734cc,
], dtype=np.uint64)  # reference src/common.rs:110-144
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def splitmix64(seed, count, offset=0):
    """count outputs of SplitMix64(seed), starting at output index `offset` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_R_LIMBS = np.array([(R_ORDER >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
_ATTEMPTS = 16


def _below_r(w):
    """(n,4) little-endian uint64 values: 0 < value < r, elementwise"""
    lt = np.zeros(w.shape[0], dtype=bool)
    eq = np.ones(w.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (w[:, i] < _R_LIMBS[i])
        eq &= w[:, i] == _R_LIMBS[i]
    return lt & (w != 0).any(axis=1)


def scalars(seed, n, offset=0):
    """(n,4) uint64 scalars, uniform in [1, r) by rejection: scalar i takes the first of its (at most 16)
    255-bit candidates that lies in [1, r); candidate a of scalar i is words 4 (16 i + a) .. + 3 of the SplitMix64 stream,
    so any slice of the sequence can be generated on its own (offset).  A candidate is rejected with probability 0.095."""
    out = np.zeros((n, 4), dtype=np.uint64)
    step = 1 << 15                                    # small blocks: the temporaries stay cache resident
    lane = np.arange(1, 5, dtype=np.uint64)[None, :]
    for lo in range(0, n, step):
        todo = np.arange(lo, min(n, lo + step), dtype=np.int64)
        for a in range(_ATTEMPTS):
            if todo.size == 0:
                break
            with np.errstate(over="ignore"):
                first = (np.uint64(offset) + todo.astype(np.uint64)) * np.uint64(4 * _ATTEMPTS) + np.uint64(4 * a)
                z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (first[:, None] + lane) * np.uint64(0x9E3779B97F4A7C15)
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                w = z ^ (z >> np.uint64(31))
            w[:, 3] &= np.uint64((1 << 63) - 1)
            good = _below_r(w)
            out[todo[good]] = w[good]
            todo = todo[~good]
        if todo.size:       # probability 4e-17 per scalar
            out[todo] = int_to_scalar(1)
    return out


def scalar_to_int(row):
    return sum(int(x) << (64 * i) for i, x in enumerate(row))


def int_to_scalar(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def random_pairs(engine, n, seed=SEED, offset=0, device_tensors=False):
    """n synthetic pairs.  Returns (g1 (n,12), g2 (n,24), a (n,4), b (n,4)); with device_tensors the
    points are torch int64 tensors resident on the engine's GPU."""
    a = scalars(seed, n, 2 * offset)
    b = scalars(seed ^ 0xB5, n, 2 * offset)
    if device_tensors:
        import torch
        dev = torch.device("cuda", engine.device)
        ta = torch.from_numpy(a.view(np.int64)).to(dev)
        tb = torch.from_numpy(b.view(np.int64)).to(dev)
        g1, _ = engine.g1_mul(G1_GENERATOR, ta)
        g2, _ = engine.g2_mul(G2_GENERATOR, tb)
        return g1, g2, a, b
    g1, _ = engine.g1_mul(G1_GENERATOR, a)
    g2, _ = engine.g2_mul(G2_GENERATOR, b)
    return g1, g2, a, b
