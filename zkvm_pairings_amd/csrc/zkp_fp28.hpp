// zkp_fp28.hpp -- carry-free field core of the lane-cooperative family (gfx950).
//
// Why this representation: on gfx950 v_add_co/v_addc (carry in/out through an SGPR pair) issue at
// the same ~4 cycles/wave/SIMD as v_mad_u64_u32, while a carry-less v_add_u32 costs ~1.3 (measured
// by tools/ubench_valu.hip).  So a 12x32-bit CIOS multiply spends more issue slots on carries than
// on multiplies.  Here an Fp element is 14 signed limbs of 28 bits ("balanced": |limb| <~ 2^27):
//   * a product is 196 v_mad_i64_i32 into 27 signed 64-bit column accumulators - no carries;
//   * several products (a bilinear form) accumulate into the SAME columns and share ONE Montgomery
//     reduction (lazy reduction): sum_t a_t*b_t needs 14*sum_t(La_t*Lb_t)*2^54 < 2^63, i.e.
//     sum_t La_t*Lb_t <= 32 where L is the limb bound in units of 2^27;
//   * add/sub/neg are limb-wise v_add/v_sub with no carry and no modular correction;
//   * Montgomery radix R = 2^392 ~ 2^11 p, so values may drift in (-8p, 8p) and a reduced product of
//     inputs |a|,|b| < 8p is back in (-0.04p, 1.04p): no conditional subtraction anywhere.
// Canonical limbs are produced only at the wire (fp28_to_wire).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zkp_constants28.h"

namespace zkp28 {

constexpr int NL = 14;
constexpr int W = 28;
constexpr int32_t MASK = (1 << W) - 1;

struct Fp28 { int32_t l[NL]; };
struct Acc { int64_t c[2 * NL]; };  // columns 0..26 (+ c[27] as the carry sink)

__device__ __constant__ const int32_t K28_P[NL] = {ZKP28_P_LIMBS};
#define ZKP28_DECL(name, ...) __device__ __constant__ const int32_t K28_##name[14] = {__VA_ARGS__};
ZKP28_CONST_LIST(ZKP28_DECL)
#undef ZKP28_DECL

__device__ __forceinline__ void acc_zero(Acc& a) {
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) a.c[i] = 0;
}

// acc += a * b  (schoolbook, 196 v_mad_i64_i32)
__device__ __forceinline__ void acc_mul(Acc& acc, const int32_t* a, const int32_t* b) {
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) acc.c[i + j] += (int64_t)a[i] * (int64_t)b[j];
}

// The same columns with one level of Karatsuba over the limb halves (a = a_lo + 2^196 a_hi): a_lo b_lo lands in columns
// 0..12 and a_hi b_hi in columns 14..26 of the ordinary accumulator (they do not overlap), the middle product in its
// subtractive form (a_hi - a_lo)(b_lo - b_hi) = a_lo b_hi + a_hi b_lo - a_lo b_lo - a_hi b_hi in 13 extra columns shared by
// every term of a lazy accumulation: 147 multiply-adds and 14 subtractions per product instead of 196.  acc_fold() then adds
// mid + lo + hi at column 7 - additions only (gfx950 has a 64-bit add, v_lshl_add_u64, but no 64-bit subtract).  All of it is
// arithmetic mod 2^64 on columns whose TRUE values fit (the budget of acc_mul), so intermediate wrap-around is harmless and
// the columns, hence every limb downstream, are the very integers acc_mul produces.
constexpr int NH = NL / 2;
struct AccMid { int64_t c[2 * NH - 1]; };
__device__ __forceinline__ void mid_zero(AccMid& m) {
#pragma unroll
    for (int i = 0; i < 2 * NH - 1; i++) m.c[i] = 0;
}
__device__ __forceinline__ void acc_mul_k(Acc& acc, AccMid& mid, const int32_t* a, const int32_t* b) {
    int32_t da[NH], db[NH];
#pragma unroll
    for (int i = 0; i < NH; i++) { da[i] = a[i + NH] - a[i]; db[i] = b[i] - b[i + NH]; }
    // the middle columns (and only they) may pass 2^63 on the way - the sums are taken in uint64_t, where wrap-around is defined
    auto wadd = [](int64_t x, int64_t y) -> int64_t { return (int64_t)((uint64_t)x + (uint64_t)y); };
#pragma unroll
    for (int i = 0; i < NH; i++)
#pragma unroll
        for (int j = 0; j < NH; j++) {
            acc.c[i + j] += (int64_t)a[i] * (int64_t)b[j];
            acc.c[NL + i + j] += (int64_t)a[NH + i] * (int64_t)b[NH + j];
            mid.c[i + j] = wadd(mid.c[i + j], (int64_t)da[i] * (int64_t)db[j]);
        }
}
__device__ __forceinline__ void acc_fold(Acc& acc, AccMid& mid) {
    auto wadd = [](int64_t x, int64_t y) -> int64_t { return (int64_t)((uint64_t)x + (uint64_t)y); };
#pragma unroll
    for (int k = 0; k < 2 * NH - 1; k++) mid.c[k] = wadd(mid.c[k], wadd(acc.c[k], acc.c[NL + k]));
#pragma unroll
    for (int k = 0; k < 2 * NH - 1; k++) acc.c[NH + k] = wadd(acc.c[NH + k], mid.c[k]);
}

// sign-extended low 28 bits: value in [-2^27, 2^27)
__device__ __forceinline__ int32_t lo28s(int64_t v) { return ((int32_t)((uint32_t)v << 4)) >> 4; }

// Montgomery reduction of the accumulated columns (divides by 2^392), then limb extraction.
// Result value = acc * 2^-392 mod p, in (acc/R, acc/R + p).
// The 13 surviving columns c[14..26] are signed 64-bit; a carry chain turns them into balanced 28-bit limbs
// (|limb| <= 2^27, the top limb keeps whatever is left): five instructions per column and no normalisation pass
// afterwards (the three-way lo/mid/top split it replaces needed seven plus a weak_norm).
__device__ __forceinline__ void weak_norm(int32_t* x);
__device__ __forceinline__ void acc_reduce(int32_t* out, Acc& acc) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t m = ((uint32_t)acc.c[i] * ZKP28_PINV) & (uint32_t)MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) acc.c[i + j] += (int64_t)(int32_t)m * (int64_t)K28_P[j];
        acc.c[i + 1] += acc.c[i] >> W;  // low 28 bits of c[i] are now zero
    }
    // (starting the columns at 2^27 to save this chain's rounding addition was measured: +0.7 % time on the 2^20 pass -
    // a column's first multiply-add takes the inline constant 0 as its addend, any other start value costs moves)
    int64_t t = acc.c[NL];
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        t += 1ll << (W - 1);
        out[k] = (int32_t)((uint32_t)t & (uint32_t)MASK) - (1 << (W - 1));
        t = acc.c[NL + k + 1] + (t >> W);       // c[27] is the empty carry sink
    }
    out[NL - 1] = (int32_t)t;
}

// Montgomery product in product-scanning order: out = (a1 b1 [+ a2 b2]) 2^-392 mod p, the SAME limbs acc_mul + acc_reduce
// give (the m_i and the surviving columns are the same integers), but with every operand in registers only ONE 64-bit
// column is live at a time: column k collects its a_i b_(k-i) and m_i p_(k-i), yields m_k (k < 14) or a balanced output
// limb (k >= 14), and its carry is the next column's initial value - no 27-column accumulator (54 VGPRs) and no
// separate carry additions.  For routines whose operands are all in registers (k_ksq, the line precomputation).
template <bool TWO>
__device__ __forceinline__ void mont_mul_ps(int32_t* out, const int32_t* a1, const int32_t* b1, const int32_t* a2, const int32_t* b2) {
    int32_t m[NL];
    int64_t col = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
        const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            col += (int64_t)a1[i] * (int64_t)b1[k - i];
            if (TWO) col += (int64_t)a2[i] * (int64_t)b2[k - i];
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (i != k) col += (int64_t)m[i] * (int64_t)K28_P[k - i];
        if (k < NL) {
            m[k] = (int32_t)(((uint32_t)col * ZKP28_PINV) & (uint32_t)MASK);
            col += (int64_t)m[k] * (int64_t)K28_P[0];
            col >>= W;                                  // the low 28 bits are zero now
        } else {
            const int64_t t = col + (1ll << (W - 1));
            out[k - NL] = (int32_t)((uint32_t)t & (uint32_t)MASK) - (1 << (W - 1));
            col = t >> W;
        }
    }
    out[NL - 1] = (int32_t)col;
}

// one-pass weak normalisation of limb-wise sums: |in| < 2^31  ->  |out| <= 2^27 + 16
__device__ __forceinline__ void weak_norm(int32_t* x) {
    int32_t c[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        c[i] = (x[i] + (1 << (W - 1))) >> W;
        x[i] -= c[i] << W;
    }
#pragma unroll
    for (int i = 1; i < NL; i++) x[i] += c[i - 1];
}

__device__ __forceinline__ void fp28_mul(Fp28& r, const Fp28& a, const Fp28& b) {
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, a.l, b.l);
    acc_reduce(r.l, acc);
}

// ---- wire (6 x u64 canonical) <-> Fp28 Montgomery ------------------------------------------------
__device__ __forceinline__ void fp28_from_wire(Fp28& r, const uint64_t* src) {
    // split the 384-bit integer into 28-bit limbs
    uint64_t w[6];
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = src[i];
    Fp28 t;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = W * i, word = bit >> 6, sh = bit & 63;
        uint64_t v = w[word] >> sh;
        if (sh > 64 - W && word + 1 < 6) v |= w[word + 1] << (64 - sh);
        t.l[i] = (int32_t)(v & (uint64_t)MASK);
    }
    Fp28 r2;
#pragma unroll
    for (int i = 0; i < NL; i++) r2.l[i] = K28_R2[i];
    fp28_mul(r, t, r2);  // a * R^2 / R = a R
}

// Montgomery Fp28 (any representative in (-8p, 8p)) -> canonical 6 x u64 in [0, p)
__device__ __forceinline__ void fp28_to_wire(uint64_t* dst, const Fp28& a) {
    Acc acc;
    acc_zero(acc);
#pragma unroll
    for (int i = 0; i < NL; i++) acc.c[i] = a.l[i];  // a * 1
    int32_t x[NL];
    acc_reduce(x, acc);  // value in (-eps p, p + eps p), balanced limbs
    // exact carry normalisation to unsigned limbs; top limb keeps the sign
    int32_t u[NL], y[NL], z[NL];
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = x[i] + carry;
        if (i < NL - 1) { u[i] = v & MASK; carry = v >> W; } else u[i] = v;
    }
    // y = u - p, z = u + p
    carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = u[i] - K28_P[i] + carry;
        if (i < NL - 1) { y[i] = v & MASK; carry = v >> W; } else y[i] = v;
    }
    carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = u[i] + K28_P[i] + carry;
        if (i < NL - 1) { z[i] = v & MASK; carry = v >> W; } else z[i] = v;
    }
    bool neg = u[NL - 1] < 0;      // value < 0  -> take u + p
    bool ge = y[NL - 1] >= 0;      // value >= p -> take u - p
    uint32_t f[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) f[i] = (uint32_t)(neg ? z[i] : (ge ? y[i] : u[i]));
    // pack 14 x 28 bits into 6 x 64
    uint64_t o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = W * i, word = bit >> 6, sh = bit & 63;
        o[word] |= (uint64_t)f[i] << sh;
        if (sh > 64 - W && word + 1 < 6) o[word + 1] |= (uint64_t)f[i] >> (64 - sh);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) dst[i] = o[i];
}

}  // namespace zkp28
