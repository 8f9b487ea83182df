/*
 * oracle/bls12_381_oracle.c -- see bls12_381_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C11 (+ unsigned __int128, + pthreads for the *_mt helpers).  Field elements are kept in
 * Montgomery form (R = 2^384) inside this file and converted at every exported function, so the
 * observable semantics are those of the reference's canonical representation (src/fp.rs:24,
 * F2 in SURVEY.md): every exported value is the canonical representative in [0,p).
 */
#include "bls12_381_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "orc_constants.h"

/* ORC_SLOW: the "reference-faithful slow mode" of SURVEY.md 8(d) / BASELINE.md 3 - a second build of this file
 * (oracle/_build/liborc_slow.so) in which Fp works the way the reference's host path does (src/fp.rs:352-434): canonical
 * integers, a full 768-bit schoolbook product followed by a long division by p, no Montgomery form anywhere.  Same
 * exported API, same results; only bench.py's cpu_baseline leg times it (on a small sample), labelled as a restatement. */
#ifdef ORC_SLOW
#define KONST(name) ORC_C_##name
#else
#define KONST(name) ORC_M_##name
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

/* =============================================================================== Fp */
static inline fp fp_from_arr(const uint64_t a[6]) { fp r; memcpy(r.l, a, 48); return r; }
static const fp* FP_P(void) { return (const fp*)ORC_P; }

static inline int fp_is_zero(const fp* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }
static inline int fp_eq(const fp* a, const fp* b) { return memcmp(a->l, b->l, 48) == 0; }

/* a >= b as 384-bit integers */
static inline int fp_geq(const fp* a, const fp* b) {
    for (int i = 5; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}

static inline uint64_t sub_limbs(fp* r, const fp* a, const fp* b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}

static inline uint64_t add_limbs(fp* r, const fp* a, const fp* b) {
    uint64_t carry = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a->l[i] + b->l[i] + carry;
        r->l[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return carry;
}

/* (a + b) mod p   -- reference src/fp.rs:352-368 (BigUint add then % p) */
static inline fp fp_add(const fp* a, const fp* b) {
    fp r;
    add_limbs(&r, a, b); /* p < 2^381 so no carry out of 384 bits for reduced inputs */
    if (fp_geq(&r, FP_P())) sub_limbs(&r, &r, FP_P());
    return r;
}

/* -a mod p  -- reference src/fp.rs:383-405 */
static inline fp fp_neg(const fp* a) {
    fp r;
    if (fp_is_zero(a)) return *a;
    sub_limbs(&r, FP_P(), a);
    return r;
}

/* a - b  -- reference src/fp.rs:409-411 computes neg(b) + a; same value for reduced inputs */
static inline fp fp_sub(const fp* a, const fp* b) {
    fp r;
    if (sub_limbs(&r, a, b)) add_limbs(&r, &r, FP_P());
    return r;
}

static inline fp fp_dbl(const fp* a) { return fp_add(a, a); }

#ifdef ORC_SLOW
/* (a * b) % p the long way: 6 x 6 schoolbook product (12 limbs), then Knuth's Algorithm D (TAOCP 4.3.1) dividing the
 * 12-limb product by the 6-limb modulus - the shape of num-bigint's BigUint `*` and `%` in the reference's Fp::mul
 * (src/fp.rs:416-434), minus its heap allocations. */
static fp fp_mul(const fp* a, const fp* b) {
    uint64_t t[13] = {0};
    for (int i = 0; i < 6; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 6; j++) {
            u128 s = (u128)a->l[i] * b->l[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        t[i + 6] = carry;
    }
    /* normalise: p has 381 bits, shift both by 3 so that the divisor's top bit is set */
    uint64_t v[6], u[13];
    for (int i = 5; i >= 0; i--) v[i] = (ORC_P[i] << 3) | (i ? ORC_P[i - 1] >> 61 : 0);
    u[12] = t[11] >> 61;
    for (int i = 11; i >= 0; i--) u[i] = (t[i] << 3) | (i ? t[i - 1] >> 61 : 0);
    for (int j = 6; j >= 0; j--) {
        /* estimate the quotient digit from the top two limbs of the running remainder */
        u128 num = ((u128)u[j + 6] << 64) | u[j + 5];
        u128 qhat = num / v[5], rhat = num % v[5];
        while (qhat >> 64 || (uint64_t)qhat * (u128)v[4] > ((rhat << 64) | u[j + 4])) {
            qhat--;
            rhat += v[5];
            if (rhat >> 64) break;
        }
        /* multiply and subtract */
        uint64_t borrow = 0, carry = 0;
        for (int i = 0; i < 6; i++) {
            u128 p = (u128)(uint64_t)qhat * v[i] + carry;
            carry = (uint64_t)(p >> 64);
            u128 d = (u128)u[i + j] - (uint64_t)p - borrow;
            u[i + j] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        u128 d = (u128)u[j + 6] - carry - borrow;
        u[j + 6] = (uint64_t)d;
        if ((uint64_t)(d >> 64) & 1) { /* the estimate was one too large: add the divisor back */
            uint64_t c = 0;
            for (int i = 0; i < 6; i++) {
                u128 s2 = (u128)u[i + j] + v[i] + c;
                u[i + j] = (uint64_t)s2;
                c = (uint64_t)(s2 >> 64);
            }
            u[j + 6] += c;
        }
    }
    fp r;
    for (int i = 0; i < 6; i++) r.l[i] = (u[i] >> 3) | (u[i + 1] << 61);
    return r;
}
#else
/* Montgomery product a*b*R^-1 mod p (CIOS, 64-bit limbs).  The reference's host Fp::mul
 * (src/fp.rs:416-434) is BigUint (a*b) % p on canonical values; with both operands in Montgomery
 * form this computes the Montgomery form of exactly that value. */
static fp fp_mul(const fp* a, const fp* b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 6; j++) {
            u128 s = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        u128 s = (u128)t[6] + carry;
        t[6] = (uint64_t)s;
        t[7] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * ORC_INV64;
        s = (u128)m * ORC_P[0] + t[0];
        carry = (uint64_t)(s >> 64);
        for (int j = 1; j < 6; j++) {
            s = (u128)m * ORC_P[j] + t[j] + carry;
            t[j - 1] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        s = (u128)t[6] + carry;
        t[5] = (uint64_t)s;
        t[6] = t[7] + (uint64_t)(s >> 64);
    }
    fp r;
    memcpy(r.l, t, 48);
    if (t[6] || fp_geq(&r, FP_P())) sub_limbs(&r, &r, FP_P());
    return r;
}

#endif
static inline fp fp_sqr(const fp* a) { return fp_mul(a, a); } /* src/fp.rs:453-455 */

#ifdef ORC_SLOW
static inline fp fp_to_mont(const fp* a) { return *a; }     /* canonical integers all the way */
static inline fp fp_from_mont(const fp* a) { return *a; }
static inline fp fp_one(void) { fp r = {{1, 0, 0, 0, 0, 0}}; return r; }
#else
static inline fp fp_to_mont(const fp* a) { return fp_mul(a, (const fp*)ORC_R2); }
static inline fp fp_from_mont(const fp* a) {
    fp one = {{1, 0, 0, 0, 0, 0}};
    return fp_mul(a, &one);
}
static inline fp fp_one(void) { return fp_from_arr(ORC_R); }
#endif
static inline fp fp_zero(void) { fp r; memset(&r, 0, sizeof r); return r; }

/* square-and-multiply over all 384 exponent bits, MSB first -- reference src/fp.rs:264-276 */
static fp fp_pow_vartime(const fp* a, const uint64_t e[6]) {
    fp res = fp_one();
    for (int w = 5; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            res = fp_sqr(&res);
            if ((e[w] >> i) & 1) res = fp_mul(&res, a);
        }
    return res;
}

/* a^(p-2); returns 0 when a == 0  -- reference src/fp.rs:307-319 */
static int fp_inv(const fp* a, fp* out) {
    *out = fp_pow_vartime(a, ORC_P_MINUS_2);
    return !fp_is_zero(a);
}

/* a^((p+1)/4), verified by squaring  -- reference src/fp.rs:280-300 */
static int fp_sqrt(const fp* a, fp* out) {
    fp s = fp_pow_vartime(a, ORC_P_PLUS_1_DIV_4);
    fp chk = fp_sqr(&s);
    *out = s;
    return fp_eq(&chk, a);
}

static inline fp fp_load(const uint64_t a[6]) { fp t = fp_from_arr(a); return fp_to_mont(&t); }
static inline void fp_store(uint64_t out[6], const fp* a) { fp t = fp_from_mont(a); memcpy(out, t.l, 48); }

/* =============================================================================== Fp2  (src/fp2.rs) */
static inline fp2 fp2_zero(void) { fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
static inline fp2 fp2_one(void) { fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
static inline int fp2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2* a, const fp2* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline fp2 fp2_add(const fp2* a, const fp2* b) { fp2 r; r.c0 = fp_add(&a->c0, &b->c0); r.c1 = fp_add(&a->c1, &b->c1); return r; } /* :216-218 */
static inline fp2 fp2_sub(const fp2* a, const fp2* b) { fp2 r; r.c0 = fp_sub(&a->c0, &b->c0); r.c1 = fp_sub(&a->c1, &b->c1); return r; } /* :221-223 */
static inline fp2 fp2_neg(const fp2* a) { fp2 r; r.c0 = fp_neg(&a->c0); r.c1 = fp_neg(&a->c1); return r; }                               /* :226-228 */
static inline fp2 fp2_dbl(const fp2* a) { return fp2_add(a, a); }
static inline fp2 fp2_conj(const fp2* a) { fp2 r; r.c0 = a->c0; r.c1 = fp_neg(&a->c1); return r; }                                      /* :153-157 */
/* (c0 - c1) + (c0 + c1) u   -- src/fp2.rs:161-168 */
static inline fp2 fp2_mul_nr(const fp2* a) { fp2 r; r.c0 = fp_sub(&a->c0, &a->c1); r.c1 = fp_add(&a->c0, &a->c1); return r; }
/* schoolbook, 4 Fp mul  -- src/fp2.rs:192-209 */
static fp2 fp2_mul(const fp2* a, const fp2* b) {
    fp t0 = fp_mul(&a->c0, &b->c0), t1 = fp_mul(&a->c1, &b->c1);
    fp t2 = fp_mul(&a->c0, &b->c1), t3 = fp_mul(&a->c1, &b->c0);
    fp2 r;
    r.c0 = fp_sub(&t0, &t1);
    r.c1 = fp_add(&t2, &t3);
    return r;
}
/* (c0+c1)(c0-c1), 2 c0 c1  -- src/fp2.rs:171-189 */
static fp2 fp2_sqr(const fp2* a) {
    fp s = fp_add(&a->c0, &a->c1), d = fp_sub(&a->c0, &a->c1), c = fp_add(&a->c0, &a->c0);
    fp2 r;
    r.c0 = fp_mul(&s, &d);
    r.c1 = fp_mul(&c, &a->c1);
    return r;
}
static inline fp2 fp2_mul_fp(const fp2* a, const fp* s) { fp2 r; r.c0 = fp_mul(&a->c0, s); r.c1 = fp_mul(&a->c1, s); return r; } /* :95-102 */
/* (c0 - c1 u)/(c0^2 + c1^2)  -- src/fp2.rs:278-296 */
static int fp2_inv(const fp2* a, fp2* out) {
    fp s0 = fp_sqr(&a->c0), s1 = fp_sqr(&a->c1), n = fp_add(&s0, &s1), t;
    int ok = fp_inv(&n, &t);
    fp nt = fp_neg(&t);
    out->c0 = fp_mul(&a->c0, &t);
    out->c1 = fp_mul(&a->c1, &nt);
    return ok;
}
static fp2 fp2_pow_vartime(const fp2* a, const uint64_t e[6]) { /* src/fp2.rs:301-313 */
    fp2 res = fp2_one();
    for (int w = 5; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            res = fp2_sqr(&res);
            if ((e[w] >> i) & 1) res = fp2_mul(&res, a);
        }
    return res;
}
/* src/fp2.rs:231-273 */
static int fp2_sqrt(const fp2* a, fp2* out) {
    if (fp2_is_zero(a)) { *out = fp2_zero(); return 1; }
    fp2 a1 = fp2_pow_vartime(a, ORC_P_MINUS_3_DIV_4);
    fp2 a1s = fp2_sqr(&a1);
    fp2 alpha = fp2_mul(&a1s, a);
    fp2 x0 = fp2_mul(&a1, a);
    fp2 one = fp2_one(), m1 = fp2_neg(&one);
    if (fp2_eq(&alpha, &m1)) {
        out->c0 = fp_neg(&x0.c1);
        out->c1 = x0.c0;
        return 1;
    }
    fp2 ap1 = fp2_add(&alpha, &one);
    fp2 pw = fp2_pow_vartime(&ap1, ORC_P_MINUS_1_DIV_2);
    fp2 s = fp2_mul(&pw, &x0);
    fp2 chk = fp2_sqr(&s);
    *out = s;
    return fp2_eq(&chk, a);
}

static inline fp2 fp2_load(const uint64_t a[12]) { fp2 r; r.c0 = fp_load(a); r.c1 = fp_load(a + 6); return r; }
static inline void fp2_store(uint64_t out[12], const fp2* a) { fp_store(out, &a->c0); fp_store(out + 6, &a->c1); }
static inline fp2 fp2_const(const uint64_t c0[6], const uint64_t c1[6]) { fp2 r; r.c0 = fp_from_arr(c0); r.c1 = fp_from_arr(c1); return r; }

/* =============================================================================== Fp6  (src/fp6.rs) */
static inline fp6 fp6_zero(void) { fp6 r; r.c0 = fp2_zero(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
static inline fp6 fp6_one(void) { fp6 r = fp6_zero(); r.c0 = fp2_one(); return r; }
static inline fp6 fp6_add(const fp6* a, const fp6* b) { fp6 r; r.c0 = fp2_add(&a->c0, &b->c0); r.c1 = fp2_add(&a->c1, &b->c1); r.c2 = fp2_add(&a->c2, &b->c2); return r; } /* :321-333 */
static inline fp6 fp6_sub(const fp6* a, const fp6* b) { fp6 r; r.c0 = fp2_sub(&a->c0, &b->c0); r.c1 = fp2_sub(&a->c1, &b->c1); r.c2 = fp2_sub(&a->c2, &b->c2); return r; } /* :357-367 */
static inline fp6 fp6_neg(const fp6* a) { fp6 r; r.c0 = fp2_neg(&a->c0); r.c1 = fp2_neg(&a->c1); r.c2 = fp2_neg(&a->c2); return r; }                                     /* :335-346 */
/* (c2*xi, c0, c1)  -- src/fp6.rs:128-139 */
static inline fp6 fp6_mul_nr(const fp6* a) { fp6 r; r.c0 = fp2_mul_nr(&a->c2); r.c1 = a->c0; r.c2 = a->c1; return r; }

/* full multiplication; the reference expands it to 36 Fp products (mul_interleaved,
 * src/fp6.rs:188-267).  Restated as the same bilinear form over Fp2:
 *   c0 = a0 b0 + xi (a1 b2 + a2 b1), c1 = a0 b1 + a1 b0 + xi a2 b2, c2 = a0 b2 + a1 b1 + a2 b0 */
static fp6 fp6_mul(const fp6* a, const fp6* b) {
    fp2 a0b0 = fp2_mul(&a->c0, &b->c0), a0b1 = fp2_mul(&a->c0, &b->c1), a0b2 = fp2_mul(&a->c0, &b->c2);
    fp2 a1b0 = fp2_mul(&a->c1, &b->c0), a1b1 = fp2_mul(&a->c1, &b->c1), a1b2 = fp2_mul(&a->c1, &b->c2);
    fp2 a2b0 = fp2_mul(&a->c2, &b->c0), a2b1 = fp2_mul(&a->c2, &b->c1), a2b2 = fp2_mul(&a->c2, &b->c2);
    fp2 t = fp2_add(&a1b2, &a2b1);
    fp2 tx = fp2_mul_nr(&t);
    fp6 r;
    r.c0 = fp2_add(&a0b0, &tx);
    t = fp2_add(&a0b1, &a1b0);
    tx = fp2_mul_nr(&a2b2);
    r.c1 = fp2_add(&t, &tx);
    t = fp2_add(&a0b2, &a1b1);
    r.c2 = fp2_add(&t, &a2b0);
    return r;
}
/* CH-SQR2  -- src/fp6.rs:274-288 */
static fp6 fp6_sqr(const fp6* a) {
    fp2 s0 = fp2_sqr(&a->c0);
    fp2 ab = fp2_mul(&a->c0, &a->c1);
    fp2 s1 = fp2_dbl(&ab);
    fp2 t = fp2_sub(&a->c0, &a->c1);
    t = fp2_add(&t, &a->c2);
    fp2 s2 = fp2_sqr(&t);
    fp2 bc = fp2_mul(&a->c1, &a->c2);
    fp2 s3 = fp2_dbl(&bc);
    fp2 s4 = fp2_sqr(&a->c2);
    fp6 r;
    fp2 x = fp2_mul_nr(&s3);
    r.c0 = fp2_add(&x, &s0);
    x = fp2_mul_nr(&s4);
    r.c1 = fp2_add(&x, &s1);
    x = fp2_add(&s1, &s2);
    x = fp2_add(&x, &s3);
    x = fp2_sub(&x, &s0);
    r.c2 = fp2_sub(&x, &s4);
    return r;
}
/* src/fp6.rs:102-108 */
static fp6 fp6_mul_by_1(const fp6* a, const fp2* c1) {
    fp6 r;
    fp2 t = fp2_mul(&a->c2, c1);
    r.c0 = fp2_mul_nr(&t);
    r.c1 = fp2_mul(&a->c0, c1);
    r.c2 = fp2_mul(&a->c1, c1);
    return r;
}
/* src/fp6.rs:110-125 */
static fp6 fp6_mul_by_01(const fp6* a, const fp2* c0, const fp2* c1) {
    fp2 a_a = fp2_mul(&a->c0, c0);
    fp2 b_b = fp2_mul(&a->c1, c1);
    fp2 t = fp2_mul(&a->c2, c1);
    t = fp2_mul_nr(&t);
    fp6 r;
    r.c0 = fp2_add(&t, &a_a);
    fp2 s0 = fp2_add(c0, c1), s1 = fp2_add(&a->c0, &a->c1);
    t = fp2_mul(&s0, &s1);
    t = fp2_sub(&t, &a_a);
    r.c1 = fp2_sub(&t, &b_b);
    t = fp2_mul(&a->c2, c0);
    r.c2 = fp2_add(&t, &b_b);
    return r;
}
/* src/fp6.rs:291-309 */
static int fp6_inv(const fp6* a, fp6* out) {
    fp2 t, c0, c1, c2;
    t = fp2_mul(&a->c1, &a->c2);
    t = fp2_mul_nr(&t);
    c0 = fp2_sqr(&a->c0);
    c0 = fp2_sub(&c0, &t);
    c1 = fp2_sqr(&a->c2);
    c1 = fp2_mul_nr(&c1);
    t = fp2_mul(&a->c0, &a->c1);
    c1 = fp2_sub(&c1, &t);
    c2 = fp2_sqr(&a->c1);
    t = fp2_mul(&a->c0, &a->c2);
    c2 = fp2_sub(&c2, &t);
    fp2 u = fp2_mul(&a->c1, &c2), v = fp2_mul(&a->c2, &c1);
    t = fp2_add(&u, &v);
    t = fp2_mul_nr(&t);
    u = fp2_mul(&a->c0, &c0);
    t = fp2_add(&t, &u);
    fp2 ti;
    int ok = fp2_inv(&t, &ti);
    out->c0 = fp2_mul(&ti, &c0);
    out->c1 = fp2_mul(&ti, &c1);
    out->c2 = fp2_mul(&ti, &c2);
    return ok;
}
/* TRUE Frobenius x -> x^p: conjugate every Fp2 coefficient, c1 *= xi^((p-1)/3), c2 *= xi^(2(p-1)/3).
 * The reference's Fp6::frobenius_map (src/fp6.rs:142-176) uses other constants and is not this map
 * (SURVEY F3); see fp6_frob_refcompat below. */
static fp6 fp6_frob(const fp6* a) {
    fp2 k1 = fp2_const(KONST(FROB6_C1_0), KONST(FROB6_C1_1)), k2 = fp2_const(KONST(FROB6_C2_0), KONST(FROB6_C2_1));
    fp6 r;
    fp2 t;
    r.c0 = fp2_conj(&a->c0);
    t = fp2_conj(&a->c1);
    r.c1 = fp2_mul(&t, &k1);
    t = fp2_conj(&a->c2);
    r.c2 = fp2_mul(&t, &k2);
    return r;
}
/* exactly what src/fp6.rs:142-176 computes: c1 * (omega1, 0), c2 * (omega2, 0) */
static fp6 fp6_frob_refcompat(const fp6* a) {
    fp z = fp_zero();
    fp2 k1, k2, t;
    k1.c0 = fp_from_arr(KONST(REF_FROB6_C1)); k1.c1 = z;
    k2.c0 = fp_from_arr(KONST(REF_FROB6_C2)); k2.c1 = z;
    fp6 r;
    r.c0 = fp2_conj(&a->c0);
    t = fp2_conj(&a->c1);
    r.c1 = fp2_mul(&t, &k1);
    t = fp2_conj(&a->c2);
    r.c2 = fp2_mul(&t, &k2);
    return r;
}
static inline fp6 fp6_load(const uint64_t a[36]) { fp6 r; r.c0 = fp2_load(a); r.c1 = fp2_load(a + 12); r.c2 = fp2_load(a + 24); return r; }
static inline void fp6_store(uint64_t out[36], const fp6* a) { fp2_store(out, &a->c0); fp2_store(out + 12, &a->c1); fp2_store(out + 24, &a->c2); }
static inline int fp6_eq(const fp6* a, const fp6* b) { return fp2_eq(&a->c0, &b->c0) && fp2_eq(&a->c1, &b->c1) && fp2_eq(&a->c2, &b->c2); }

/* =============================================================================== Fp12  (src/fp12.rs) */
static inline fp12 fp12_one(void) { fp12 r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; } /* :87-89 */
static inline int fp12_eq(const fp12* a, const fp12* b) { return fp6_eq(&a->c0, &b->c0) && fp6_eq(&a->c1, &b->c1); }
static inline fp12 fp12_conj(const fp12* a) { fp12 r; r.c0 = a->c0; r.c1 = fp6_neg(&a->c1); return r; } /* :123-125 */
/* Karatsuba over Fp6  -- src/fp12.rs:193-210 */
static fp12 fp12_mul(const fp12* a, const fp12* b) {
    fp6 aa = fp6_mul(&a->c0, &b->c0), bb = fp6_mul(&a->c1, &b->c1);
    fp6 o = fp6_add(&b->c0, &b->c1), s = fp6_add(&a->c1, &a->c0);
    fp6 c1 = fp6_mul(&s, &o);
    c1 = fp6_sub(&c1, &aa);
    c1 = fp6_sub(&c1, &bb);
    fp6 c0 = fp6_mul_nr(&bb);
    c0 = fp6_add(&c0, &aa);
    fp12 r; r.c0 = c0; r.c1 = c1;
    return r;
}
/* src/fp12.rs:173-184 */
static fp12 fp12_sqr(const fp12* a) {
    fp6 ab = fp6_mul(&a->c0, &a->c1);
    fp6 c0c1 = fp6_add(&a->c0, &a->c1);
    fp6 c0 = fp6_mul_nr(&a->c1);
    c0 = fp6_add(&c0, &a->c0);
    c0 = fp6_mul(&c0, &c0c1);
    c0 = fp6_sub(&c0, &ab);
    fp6 c1 = fp6_add(&ab, &ab);
    fp6 abn = fp6_mul_nr(&ab);
    c0 = fp6_sub(&c0, &abn);
    fp12 r; r.c0 = c0; r.c1 = c1;
    return r;
}
/* src/fp12.rs:99-111 */
static fp12 fp12_mul_by_014(const fp12* a, const fp2* c0, const fp2* c1, const fp2* c4) {
    fp6 aa = fp6_mul_by_01(&a->c0, c0, c1);
    fp6 bb = fp6_mul_by_1(&a->c1, c4);
    fp2 o = fp2_add(c1, c4);
    fp6 r1 = fp6_add(&a->c1, &a->c0);
    r1 = fp6_mul_by_01(&r1, c0, &o);
    r1 = fp6_sub(&r1, &aa);
    r1 = fp6_sub(&r1, &bb);
    fp6 r0 = fp6_mul_nr(&bb);
    r0 = fp6_add(&r0, &aa);
    fp12 r; r.c0 = r0; r.c1 = r1;
    return r;
}
/* src/fp12.rs:186-190 */
static int fp12_inv(const fp12* a, fp12* out) {
    fp6 s0 = fp6_sqr(&a->c0), s1 = fp6_sqr(&a->c1);
    s1 = fp6_mul_nr(&s1);
    fp6 n = fp6_sub(&s0, &s1), t;
    int ok = fp6_inv(&n, &t);
    fp6 nt = fp6_neg(&t);
    out->c0 = fp6_mul(&a->c0, &t);
    out->c1 = fp6_mul(&a->c1, &nt);
    return ok;
}
/* TRUE Frobenius.  Shape of src/fp12.rs:143-170 (c1 scaled by gamma = xi^((p-1)/6), constant
 * verified equal to the reference's) on top of the TRUE Fp6 map. */
static fp12 fp12_frob(const fp12* a) {
    fp2 g = fp2_const(KONST(FROB12_C1_0), KONST(FROB12_C1_1));
    fp12 r;
    r.c0 = fp6_frob(&a->c0);
    fp6 t = fp6_frob(&a->c1);
    r.c1.c0 = fp2_mul(&t.c0, &g);
    r.c1.c1 = fp2_mul(&t.c1, &g);
    r.c1.c2 = fp2_mul(&t.c2, &g);
    return r;
}
static fp12 fp12_frob_refcompat(const fp12* a) {
    fp2 g = fp2_const(KONST(FROB12_C1_0), KONST(FROB12_C1_1));
    fp12 r;
    r.c0 = fp6_frob_refcompat(&a->c0);
    fp6 t = fp6_frob_refcompat(&a->c1);
    fp6 g6 = fp6_zero();
    g6.c0 = g; /* Fp6::from(Fp2), src/fp6.rs:29-37 */
    r.c1 = fp6_mul(&t, &g6);
    return r;
}
static inline fp12 fp12_load(const uint64_t a[72]) { fp12 r; r.c0 = fp6_load(a); r.c1 = fp6_load(a + 36); return r; }
static inline void fp12_store(uint64_t out[72], const fp12* a) { fp6_store(out, &a->c0); fp6_store(out + 36, &a->c1); }

/* Granger-Scott squaring in the cyclotomic subgroup (valid only after the easy part). */
static void fp4_square(const fp2* a, const fp2* b, fp2* o0, fp2* o1) {
    fp2 t0 = fp2_sqr(a), t1 = fp2_sqr(b);
    fp2 t2 = fp2_mul_nr(&t1);
    *o0 = fp2_add(&t2, &t0);
    t2 = fp2_add(a, b);
    t2 = fp2_sqr(&t2);
    t2 = fp2_sub(&t2, &t0);
    *o1 = fp2_sub(&t2, &t1);
}
static fp12 fp12_cyclotomic_square(const fp12* f) {
    fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2;
    fp2 t0, t1, t2, t3;
    fp4_square(&z0, &z1, &t0, &t1);
    z0 = fp2_sub(&t0, &z0); z0 = fp2_add(&z0, &z0); z0 = fp2_add(&z0, &t0);
    z1 = fp2_add(&t1, &z1); z1 = fp2_add(&z1, &z1); z1 = fp2_add(&z1, &t1);
    fp4_square(&z2, &z3, &t0, &t1);
    fp4_square(&z4, &z5, &t2, &t3);
    z4 = fp2_sub(&t0, &z4); z4 = fp2_add(&z4, &z4); z4 = fp2_add(&z4, &t0);
    z5 = fp2_add(&t1, &z5); z5 = fp2_add(&z5, &z5); z5 = fp2_add(&z5, &t1);
    t0 = fp2_mul_nr(&t3);
    z2 = fp2_add(&t0, &z2); z2 = fp2_add(&z2, &z2); z2 = fp2_add(&z2, &t0);
    z3 = fp2_sub(&t2, &z3); z3 = fp2_add(&z3, &z3); z3 = fp2_add(&z3, &t2);
    fp12 r;
    r.c0.c0 = z0; r.c0.c1 = z4; r.c0.c2 = z3;
    r.c1.c0 = z2; r.c1.c1 = z1; r.c1.c2 = z5;
    return r;
}
/* f^|x| followed by conjugation (x < 0); MSB-first over the 64 bits of |x| */
static fp12 cyclotomic_exp(const fp12* f) {
    fp12 tmp = fp12_one();
    int found_one = 0;
    for (int i = 63; i >= 0; i--) {
        int bit = (int)((ORC_BLS_X >> i) & 1);
        if (found_one) tmp = fp12_cyclotomic_square(&tmp); else found_one = bit;
        if (bit) tmp = fp12_mul(&tmp, f);
    }
    return fp12_conj(&tmp);
}
static fp12 fp12_pow_u64(const fp12* a, uint64_t e) {
    fp12 res = fp12_one();
    for (int i = 63; i >= 0; i--) {
        res = fp12_sqr(&res);
        if ((e >> i) & 1) res = fp12_mul(&res, a);
    }
    return res;
}

/* =============================================================================== G1 / G2 */
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp2 x, y; int inf; } g2a;

static g1a g1_identity(void) { g1a r; r.x = fp_zero(); r.y = fp_one(); r.inf = 1; return r; }       /* src/g1.rs:25-31 */
static g2a g2_identity(void) { g2a r; r.x = fp2_zero(); r.y = fp2_one(); r.inf = 1; return r; }     /* src/g2.rs:27-33 */

/* src/g1.rs:74-91.  The reference panics (division by zero) when y == 0; here that case returns
 * the identity, like the reference's own G2 double (src/g2.rs:88-91). */
static g1a g1_double(const g1a* p) {
    if (p->inf || fp_is_zero(&p->y)) return g1_identity();
    fp xx = fp_sqr(&p->x), three_xx = fp_add(&xx, &xx);
    three_xx = fp_add(&three_xx, &xx);
    fp two_y = fp_add(&p->y, &p->y), inv;
    fp_inv(&two_y, &inv);
    fp lam = fp_mul(&three_xx, &inv);
    g1a r;
    fp l2 = fp_sqr(&lam), two_x = fp_add(&p->x, &p->x);
    r.x = fp_sub(&l2, &two_x);
    fp d = fp_sub(&p->x, &r.x);
    fp t = fp_mul(&lam, &d);
    r.y = fp_sub(&t, &p->y);
    r.inf = 0;
    return r;
}
/* src/g1.rs:155-187.  P + (-P) panics in the reference; here it returns the identity. */
static g1a g1_add(const g1a* p, const g1a* q) {
    if (p->inf) return *q;
    if (q->inf) return *p;
    if (fp_eq(&p->x, &q->x)) {
        if (fp_eq(&p->y, &q->y)) return g1_double(p);
        return g1_identity();
    }
    fp dy = fp_sub(&q->y, &p->y), dx = fp_sub(&q->x, &p->x), inv;
    fp_inv(&dx, &inv);
    fp lam = fp_mul(&dy, &inv);
    g1a r;
    fp l2 = fp_sqr(&lam);
    r.x = fp_sub(&l2, &p->x);
    r.x = fp_sub(&r.x, &q->x);
    fp d = fp_sub(&p->x, &r.x);
    fp t = fp_mul(&lam, &d);
    r.y = fp_sub(&t, &p->y);
    r.inf = 0;
    return r;
}
static g1a g1_neg(const g1a* p) { g1a r = *p; r.y = fp_neg(&p->y); return r; } /* src/g1.rs:118-128 */
/* MSB-first double-and-add over a 256-bit scalar.  (The reference's G1 Mul, src/g1.rs:130-153,
 * is LSB-first and drops bit 0 -- SURVEY F5; harmless for its only use, mul_by_x with even X.) */
static g1a g1_mul(const g1a* p, const uint64_t k[4]) {
    g1a acc = g1_identity();
    for (int w = 3; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            acc = g1_double(&acc);
            if ((k[w] >> i) & 1) acc = g1_add(&acc, p);
        }
    return acc;
}
static int g1_on_curve(const g1a* p) { /* src/g1.rs:95-101 */
    fp yy = fp_sqr(&p->y), xx = fp_sqr(&p->x), xxx = fp_mul(&xx, &p->x);
    fp rhs = fp_add(&xxx, (const fp*)KONST(B));
    return fp_eq(&yy, &rhs);
}
static int g1_torsion_free(const g1a* p) { /* src/g1.rs:111-115: -[X][X]P == (beta x, y) */
    uint64_t k[4] = {ORC_BLS_X, 0, 0, 0};
    g1a t = g1_mul(p, k);
    t = g1_mul(&t, k);
    t = g1_neg(&t);
    fp bx = fp_mul(&p->x, (const fp*)KONST(BETA));
    if (t.inf) return 0;
    return fp_eq(&t.x, &bx) && fp_eq(&t.y, &p->y);
}

static g2a g2_double(const g2a* p) { /* src/g2.rs:81-105 */
    if (p->inf || fp2_is_zero(&p->y)) return g2_identity();
    fp2 xx = fp2_sqr(&p->x), three_xx = fp2_add(&xx, &xx);
    three_xx = fp2_add(&three_xx, &xx);
    fp2 two_y = fp2_add(&p->y, &p->y), inv;
    fp2_inv(&two_y, &inv);
    fp2 lam = fp2_mul(&three_xx, &inv);
    g2a r;
    fp2 l2 = fp2_sqr(&lam), two_x = fp2_add(&p->x, &p->x);
    r.x = fp2_sub(&l2, &two_x);
    fp2 d = fp2_sub(&p->x, &r.x);
    fp2 t = fp2_mul(&lam, &d);
    r.y = fp2_sub(&t, &p->y);
    r.inf = 0;
    return r;
}
static g2a g2_add(const g2a* p, const g2a* q) { /* src/g2.rs:210-242 */
    if (p->inf) return *q;
    if (q->inf) return *p;
    if (fp2_eq(&p->x, &q->x)) {
        if (fp2_eq(&p->y, &q->y)) return g2_double(p);
        return g2_identity();
    }
    fp2 dy = fp2_sub(&q->y, &p->y), dx = fp2_sub(&q->x, &p->x), inv;
    fp2_inv(&dx, &inv);
    fp2 lam = fp2_mul(&dy, &inv);
    g2a r;
    fp2 l2 = fp2_sqr(&lam);
    r.x = fp2_sub(&l2, &p->x);
    r.x = fp2_sub(&r.x, &q->x);
    fp2 d = fp2_sub(&p->x, &r.x);
    fp2 t = fp2_mul(&lam, &d);
    r.y = fp2_sub(&t, &p->y);
    r.inf = 0;
    return r;
}
static g2a g2_neg(const g2a* p) { g2a r = *p; r.y = fp2_neg(&p->y); return r; } /* src/g2.rs:173-183 */
static g2a g2_mul(const g2a* p, const uint64_t k[4]) {                            /* src/g2.rs:185-208 */
    g2a acc = g2_identity();
    for (int w = 3; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            acc = g2_double(&acc);
            if ((k[w] >> i) & 1) acc = g2_add(&acc, p);
        }
    return acc;
}
static int g2_on_curve(const g2a* p) { /* src/g2.rs:109-120 */
    fp2 yy = fp2_sqr(&p->y), xx = fp2_sqr(&p->x), xxx = fp2_mul(&xx, &p->x);
    fp2 b;
    b.c0 = fp_from_arr(KONST(B)); b.c1 = b.c0;
    fp2 rhs = fp2_add(&xxx, &b);
    return fp2_eq(&yy, &rhs);
}
static g2a g2_psi(const g2a* p) { /* src/g2.rs:126-164 */
    fp2 kx = fp2_const(KONST(PSI_X_0), KONST(PSI_X_1)), ky = fp2_const(KONST(PSI_Y_0), KONST(PSI_Y_1));
    g2a r;
    fp2 t = fp2_conj(&p->x);
    r.x = fp2_mul(&t, &kx);
    t = fp2_conj(&p->y);
    r.y = fp2_mul(&t, &ky);
    r.inf = 0;
    return r;
}
static int g2_torsion_free(const g2a* p) { /* src/g2.rs:166-170: psi(P) == -[X]P */
    uint64_t k[4] = {ORC_BLS_X, 0, 0, 0};
    g2a l = g2_psi(p);
    g2a r = g2_mul(p, k);
    r = g2_neg(&r);
    if (r.inf) return 0;
    return fp2_eq(&l.x, &r.x) && fp2_eq(&l.y, &r.y);
}

static g1a g1_load(const uint64_t a[12], uint8_t inf) { g1a r; r.x = fp_load(a); r.y = fp_load(a + 6); r.inf = inf != 0; return r; }
static void g1_store(uint64_t out[12], uint8_t* inf, const g1a* p) { fp_store(out, &p->x); fp_store(out + 6, &p->y); if (inf) *inf = (uint8_t)p->inf; }
static g2a g2_load(const uint64_t a[24], uint8_t inf) { g2a r; r.x = fp2_load(a); r.y = fp2_load(a + 12); r.inf = inf != 0; return r; }
static void g2_store(uint64_t out[24], uint8_t* inf, const g2a* p) { fp2_store(out, &p->x); fp2_store(out + 12, &p->y); if (inf) *inf = (uint8_t)p->inf; }

/* =============================================================================== pairing (defined by the build) */
typedef struct { fp2 x, y, z; } g2proj;
typedef struct { fp2 c0, c1, c2; } line_t;

/* ePrint 2010/354 Alg. 26 (doubling in Jacobian coordinates with line coefficients) */
static line_t doubling_step(g2proj* r) {
    fp2 tmp0 = fp2_sqr(&r->x);
    fp2 tmp1 = fp2_sqr(&r->y);
    fp2 tmp2 = fp2_sqr(&tmp1);
    fp2 tmp3 = fp2_add(&tmp1, &r->x);
    tmp3 = fp2_sqr(&tmp3);
    tmp3 = fp2_sub(&tmp3, &tmp0);
    tmp3 = fp2_sub(&tmp3, &tmp2);
    tmp3 = fp2_dbl(&tmp3);
    fp2 tmp4 = fp2_add(&tmp0, &tmp0);
    tmp4 = fp2_add(&tmp4, &tmp0);
    fp2 tmp6 = fp2_add(&r->x, &tmp4);
    fp2 tmp5 = fp2_sqr(&tmp4);
    fp2 zsq = fp2_sqr(&r->z);
    fp2 nx = fp2_sub(&tmp5, &tmp3);
    nx = fp2_sub(&nx, &tmp3);
    fp2 nz = fp2_add(&r->z, &r->y);
    nz = fp2_sqr(&nz);
    nz = fp2_sub(&nz, &tmp1);
    nz = fp2_sub(&nz, &zsq);
    fp2 ny = fp2_sub(&tmp3, &nx);
    ny = fp2_mul(&ny, &tmp4);
    tmp2 = fp2_dbl(&tmp2); tmp2 = fp2_dbl(&tmp2); tmp2 = fp2_dbl(&tmp2);
    ny = fp2_sub(&ny, &tmp2);
    tmp3 = fp2_mul(&tmp4, &zsq);
    tmp3 = fp2_dbl(&tmp3);
    tmp3 = fp2_neg(&tmp3);
    tmp6 = fp2_sqr(&tmp6);
    tmp6 = fp2_sub(&tmp6, &tmp0);
    tmp6 = fp2_sub(&tmp6, &tmp5);
    tmp1 = fp2_dbl(&tmp1); tmp1 = fp2_dbl(&tmp1);
    tmp6 = fp2_sub(&tmp6, &tmp1);
    tmp0 = fp2_mul(&nz, &zsq);
    tmp0 = fp2_dbl(&tmp0);
    r->x = nx; r->y = ny; r->z = nz;
    line_t l; l.c0 = tmp0; l.c1 = tmp3; l.c2 = tmp6;
    return l;
}
/* ePrint 2010/354 Alg. 27 (mixed addition with line coefficients) */
static line_t addition_step(g2proj* r, const g2a* q) {
    fp2 zsq = fp2_sqr(&r->z);
    fp2 ysq = fp2_sqr(&q->y);
    fp2 t0 = fp2_mul(&zsq, &q->x);
    fp2 t1 = fp2_add(&q->y, &r->z);
    t1 = fp2_sqr(&t1);
    t1 = fp2_sub(&t1, &ysq);
    t1 = fp2_sub(&t1, &zsq);
    t1 = fp2_mul(&t1, &zsq);
    fp2 t2 = fp2_sub(&t0, &r->x);
    fp2 t3 = fp2_sqr(&t2);
    fp2 t4 = fp2_dbl(&t3);
    t4 = fp2_dbl(&t4);
    fp2 t5 = fp2_mul(&t4, &t2);
    fp2 t6 = fp2_sub(&t1, &r->y);
    t6 = fp2_sub(&t6, &r->y);
    fp2 t9 = fp2_mul(&t6, &q->x);
    fp2 t7 = fp2_mul(&t4, &r->x);
    fp2 nx = fp2_sqr(&t6);
    nx = fp2_sub(&nx, &t5);
    nx = fp2_sub(&nx, &t7);
    nx = fp2_sub(&nx, &t7);
    fp2 nz = fp2_add(&r->z, &t2);
    nz = fp2_sqr(&nz);
    nz = fp2_sub(&nz, &zsq);
    nz = fp2_sub(&nz, &t3);
    fp2 t10 = fp2_add(&q->y, &nz);
    fp2 t8 = fp2_sub(&t7, &nx);
    t8 = fp2_mul(&t8, &t6);
    t0 = fp2_mul(&r->y, &t5);
    t0 = fp2_dbl(&t0);
    fp2 ny = fp2_sub(&t8, &t0);
    t10 = fp2_sqr(&t10);
    t10 = fp2_sub(&t10, &ysq);
    fp2 ztsq = fp2_sqr(&nz);
    t10 = fp2_sub(&t10, &ztsq);
    t9 = fp2_dbl(&t9);
    t9 = fp2_sub(&t9, &t10);
    t10 = fp2_dbl(&nz);
    t6 = fp2_neg(&t6);
    t1 = fp2_dbl(&t6);
    r->x = nx; r->y = ny; r->z = nz;
    line_t l; l.c0 = t10; l.c1 = t1; l.c2 = t9;
    return l;
}
/* line evaluation at P and accumulation through the sparse product (src/fp12.rs:99-111) */
static fp12 ell(const fp12* f, const line_t* l, const g1a* p) {
    fp2 c0 = fp2_mul_fp(&l->c0, &p->y);
    fp2 c1 = fp2_mul_fp(&l->c1, &p->x);
    return fp12_mul_by_014(f, &l->c2, &c1, &c0);
}

/* shared-squaring Miller loop over k pairs; pairs with an infinity on either side contribute 1 (P5) */
static fp12 multi_miller_loop(const g1a* ps, const g2a* qs, size_t k) {
    g2proj* rs = (g2proj*)malloc(sizeof(g2proj) * (k ? k : 1));
    for (size_t i = 0; i < k; i++) { rs[i].x = qs[i].x; rs[i].y = qs[i].y; rs[i].z = fp2_one(); }
    fp12 f = fp12_one();
    int found_one = 0;
    for (int b = 63; b >= 0; b--) {
        int bit = (int)(((ORC_BLS_X >> 1) >> b) & 1);
        if (!found_one) { found_one = bit; continue; }
        for (size_t i = 0; i < k; i++) {
            if (ps[i].inf || qs[i].inf) continue;
            line_t l = doubling_step(&rs[i]);
            f = ell(&f, &l, &ps[i]);
        }
        if (bit)
            for (size_t i = 0; i < k; i++) {
                if (ps[i].inf || qs[i].inf) continue;
                line_t l = addition_step(&rs[i], &qs[i]);
                f = ell(&f, &l, &ps[i]);
            }
        f = fp12_sqr(&f);
    }
    for (size_t i = 0; i < k; i++) {
        if (ps[i].inf || qs[i].inf) continue;
        line_t l = doubling_step(&rs[i]);
        f = ell(&f, &l, &ps[i]);
    }
    free(rs);
    return fp12_conj(&f); /* x is negative */
}

/* f^(3 (p^12 - 1)/r): easy part f^((p^6-1)(p^2+1)), then the x-chain hard part */
static fp12 final_exponentiation(const fp12* fin) {
    fp12 f = *fin;
    fp12 t0 = f, t1, t2, t3, t4, t5, t6;
    for (int i = 0; i < 6; i++) t0 = fp12_frob(&t0);
    if (!fp12_inv(&f, &t1)) return fp12_one(); /* f == 0 cannot come out of a Miller loop */
    t2 = fp12_mul(&t0, &t1);
    t1 = t2;
    t2 = fp12_frob(&t2);
    t2 = fp12_frob(&t2);
    t2 = fp12_mul(&t2, &t1);
    t1 = fp12_cyclotomic_square(&t2);
    t1 = fp12_conj(&t1);
    t3 = cyclotomic_exp(&t2);
    t4 = fp12_cyclotomic_square(&t3);
    t5 = fp12_mul(&t1, &t3);
    t1 = cyclotomic_exp(&t5);
    t0 = cyclotomic_exp(&t1);
    t6 = cyclotomic_exp(&t0);
    t6 = fp12_mul(&t6, &t4);
    t4 = cyclotomic_exp(&t6);
    t5 = fp12_conj(&t5);
    fp12 t52 = fp12_mul(&t5, &t2);
    t4 = fp12_mul(&t4, &t52);
    t5 = fp12_conj(&t2);
    t1 = fp12_mul(&t1, &t2);
    t1 = fp12_frob(&t1); t1 = fp12_frob(&t1); t1 = fp12_frob(&t1);
    t6 = fp12_mul(&t6, &t5);
    t6 = fp12_frob(&t6);
    t3 = fp12_mul(&t3, &t0);
    t3 = fp12_frob(&t3); t3 = fp12_frob(&t3);
    t3 = fp12_mul(&t3, &t1);
    t3 = fp12_mul(&t3, &t6);
    return fp12_mul(&t3, &t4);
}

/* independent formulation: affine slopes from the reference-style G2 double/add */
static fp12 miller_loop_affine(const g1a* p, const g2a* q) {
    fp12 f = fp12_one();
    g2a t = *q;
    fp2 py; py.c0 = p->y; py.c1 = fp_zero();
    int found_one = 0;
    for (int b = 63; b >= 0; b--) {
        int bit = (int)((ORC_BLS_X >> b) & 1);
        if (!found_one) { found_one = bit; continue; }
        /* tangent at t */
        fp2 xx = fp2_sqr(&t.x), n = fp2_add(&xx, &xx);
        n = fp2_add(&n, &xx);
        fp2 d = fp2_add(&t.y, &t.y), di;
        fp2_inv(&d, &di);
        fp2 lam = fp2_mul(&n, &di);
        fp2 c0 = fp2_mul(&lam, &t.x);
        c0 = fp2_sub(&c0, &t.y);
        fp2 c1 = fp2_mul_fp(&lam, &p->x);
        c1 = fp2_neg(&c1);
        f = fp12_sqr(&f);
        f = fp12_mul_by_014(&f, &c0, &c1, &py);
        t = g2_double(&t);
        if (bit) {
            fp2 dy = fp2_sub(&q->y, &t.y), dx = fp2_sub(&q->x, &t.x), dxi;
            fp2_inv(&dx, &dxi);
            lam = fp2_mul(&dy, &dxi);
            c0 = fp2_mul(&lam, &t.x);
            c0 = fp2_sub(&c0, &t.y);
            c1 = fp2_mul_fp(&lam, &p->x);
            c1 = fp2_neg(&c1);
            f = fp12_mul_by_014(&f, &c0, &c1, &py);
            t = g2_add(&t, q);
        }
    }
    return fp12_conj(&f);
}

/* =============================================================================== exported wrappers */
#define FP_BIN(name, op) \
    void orc_fp_##name(const uint64_t a[6], const uint64_t b[6], uint64_t out[6]) { fp x = fp_load(a), y = fp_load(b), r = op(&x, &y); fp_store(out, &r); }
FP_BIN(add, fp_add)
FP_BIN(sub, fp_sub)
FP_BIN(mul, fp_mul)
void orc_fp_neg(const uint64_t a[6], uint64_t out[6]) { fp x = fp_load(a), r = fp_neg(&x); fp_store(out, &r); }
void orc_fp_square(const uint64_t a[6], uint64_t out[6]) { fp x = fp_load(a), r = fp_sqr(&x); fp_store(out, &r); }
int orc_fp_invert(const uint64_t a[6], uint64_t out[6]) { fp x = fp_load(a), r; int ok = fp_inv(&x, &r); fp_store(out, &r); return ok; }
int orc_fp_sqrt(const uint64_t a[6], uint64_t out[6]) { fp x = fp_load(a), r; int ok = fp_sqrt(&x, &r); fp_store(out, &r); return ok; }
void orc_fp_pow_vartime(const uint64_t a[6], const uint64_t e[6], uint64_t out[6]) { fp x = fp_load(a), r = fp_pow_vartime(&x, e); fp_store(out, &r); }
int orc_fp_is_canonical(const uint64_t a[6]) { fp x = fp_from_arr(a); return !fp_geq(&x, FP_P()); }
void orc_fp_to_bytes_be(const uint64_t a[6], uint8_t out[48]) {
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(a[5 - i] >> (56 - 8 * j));
}
int orc_fp_from_bytes_be(const uint8_t in[48], uint64_t out[6]) {
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | in[8 * i + j];
        out[5 - i] = v;
    }
    return orc_fp_is_canonical(out);
}

#define FP2_BIN(name, op) \
    void orc_fp2_##name(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]) { fp2 x = fp2_load(a), y = fp2_load(b), r = op(&x, &y); fp2_store(out, &r); }
#define FP2_UN(name, op) \
    void orc_fp2_##name(const uint64_t a[12], uint64_t out[12]) { fp2 x = fp2_load(a), r = op(&x); fp2_store(out, &r); }
FP2_BIN(add, fp2_add)
FP2_BIN(sub, fp2_sub)
FP2_BIN(mul, fp2_mul)
FP2_UN(neg, fp2_neg)
FP2_UN(square, fp2_sqr)
FP2_UN(conjugate, fp2_conj)
FP2_UN(mul_by_nonresidue, fp2_mul_nr)
int orc_fp2_invert(const uint64_t a[12], uint64_t out[12]) { fp2 x = fp2_load(a), r; int ok = fp2_inv(&x, &r); fp2_store(out, &r); return ok; }
int orc_fp2_sqrt(const uint64_t a[12], uint64_t out[12]) { fp2 x = fp2_load(a), r; int ok = fp2_sqrt(&x, &r); fp2_store(out, &r); return ok; }

#define FP6_BIN(name, op) \
    void orc_fp6_##name(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]) { fp6 x = fp6_load(a), y = fp6_load(b), r = op(&x, &y); fp6_store(out, &r); }
#define FP6_UN(name, op) \
    void orc_fp6_##name(const uint64_t a[36], uint64_t out[36]) { fp6 x = fp6_load(a), r = op(&x); fp6_store(out, &r); }
FP6_BIN(add, fp6_add)
FP6_BIN(sub, fp6_sub)
FP6_BIN(mul, fp6_mul)
FP6_UN(neg, fp6_neg)
FP6_UN(square, fp6_sqr)
FP6_UN(mul_by_nonresidue, fp6_mul_nr)
FP6_UN(frobenius_map, fp6_frob)
FP6_UN(frobenius_map_refcompat, fp6_frob_refcompat)
void orc_fp6_mul_by_1(const uint64_t a[36], const uint64_t c1[12], uint64_t out[36]) { fp6 x = fp6_load(a); fp2 c = fp2_load(c1); fp6 r = fp6_mul_by_1(&x, &c); fp6_store(out, &r); }
void orc_fp6_mul_by_01(const uint64_t a[36], const uint64_t c0[12], const uint64_t c1[12], uint64_t out[36]) {
    fp6 x = fp6_load(a); fp2 d0 = fp2_load(c0), d1 = fp2_load(c1); fp6 r = fp6_mul_by_01(&x, &d0, &d1); fp6_store(out, &r);
}
int orc_fp6_invert(const uint64_t a[36], uint64_t out[36]) { fp6 x = fp6_load(a), r; int ok = fp6_inv(&x, &r); fp6_store(out, &r); return ok; }

void orc_fp12_one(uint64_t out[72]) { fp12 r = fp12_one(); fp12_store(out, &r); }
void orc_fp12_add(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]) {
    fp12 x = fp12_load(a), y = fp12_load(b), r; r.c0 = fp6_add(&x.c0, &y.c0); r.c1 = fp6_add(&x.c1, &y.c1); fp12_store(out, &r);
}
void orc_fp12_sub(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]) {
    fp12 x = fp12_load(a), y = fp12_load(b), r; r.c0 = fp6_sub(&x.c0, &y.c0); r.c1 = fp6_sub(&x.c1, &y.c1); fp12_store(out, &r);
}
void orc_fp12_mul(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]) { fp12 x = fp12_load(a), y = fp12_load(b), r = fp12_mul(&x, &y); fp12_store(out, &r); }
void orc_fp12_square(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_sqr(&x); fp12_store(out, &r); }
void orc_fp12_mul_by_014(const uint64_t a[72], const uint64_t c0[12], const uint64_t c1[12], const uint64_t c4[12], uint64_t out[72]) {
    fp12 x = fp12_load(a); fp2 d0 = fp2_load(c0), d1 = fp2_load(c1), d4 = fp2_load(c4);
    fp12 r = fp12_mul_by_014(&x, &d0, &d1, &d4); fp12_store(out, &r);
}
void orc_fp12_conjugate(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_conj(&x); fp12_store(out, &r); }
int orc_fp12_invert(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r; int ok = fp12_inv(&x, &r); fp12_store(out, &r); return ok; }
void orc_fp12_frobenius_map(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_frob(&x); fp12_store(out, &r); }
void orc_fp12_frobenius_map_refcompat(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_frob_refcompat(&x); fp12_store(out, &r); }
void orc_fp12_cyclotomic_square(const uint64_t a[72], uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_cyclotomic_square(&x); fp12_store(out, &r); }
void orc_fp12_pow_u64(const uint64_t a[72], uint64_t e, uint64_t out[72]) { fp12 x = fp12_load(a), r = fp12_pow_u64(&x, e); fp12_store(out, &r); }

void orc_g1_generator(uint64_t out[12]) { g1a g; g.x = fp_from_arr(KONST(G1_X)); g.y = fp_from_arr(KONST(G1_Y)); g.inf = 0; g1_store(out, NULL, &g); }
void orc_g2_generator(uint64_t out[24]) {
    g2a g; g.x = fp2_const(KONST(G2_X0), KONST(G2_X1)); g.y = fp2_const(KONST(G2_Y0), KONST(G2_Y1)); g.inf = 0; g2_store(out, NULL, &g);
}
void orc_g1_double(const uint64_t p[12], uint8_t inf, uint64_t out[12], uint8_t* out_inf) { g1a x = g1_load(p, inf), r = g1_double(&x); g1_store(out, out_inf, &r); }
void orc_g1_add(const uint64_t p[12], uint8_t pinf, const uint64_t q[12], uint8_t qinf, uint64_t out[12], uint8_t* out_inf) {
    g1a x = g1_load(p, pinf), y = g1_load(q, qinf), r = g1_add(&x, &y); g1_store(out, out_inf, &r);
}
void orc_g1_mul(const uint64_t p[12], uint8_t inf, const uint64_t k[4], uint64_t out[12], uint8_t* out_inf) { g1a x = g1_load(p, inf), r = g1_mul(&x, k); g1_store(out, out_inf, &r); }
int orc_g1_is_on_curve(const uint64_t p[12]) { g1a x = g1_load(p, 0); return g1_on_curve(&x); }
int orc_g1_is_torsion_free(const uint64_t p[12]) { g1a x = g1_load(p, 0); return g1_torsion_free(&x); }
int orc_g1_is_valid(const uint64_t p[12], uint8_t inf) {
    if (inf) return 0;
    g1a x = g1_load(p, 0);
    if (!g1_on_curve(&x)) return 1;
    if (!g1_torsion_free(&x)) return 2;
    return 0;
}
void orc_g2_double(const uint64_t p[24], uint8_t inf, uint64_t out[24], uint8_t* out_inf) { g2a x = g2_load(p, inf), r = g2_double(&x); g2_store(out, out_inf, &r); }
void orc_g2_add(const uint64_t p[24], uint8_t pinf, const uint64_t q[24], uint8_t qinf, uint64_t out[24], uint8_t* out_inf) {
    g2a x = g2_load(p, pinf), y = g2_load(q, qinf), r = g2_add(&x, &y); g2_store(out, out_inf, &r);
}
void orc_g2_mul(const uint64_t p[24], uint8_t inf, const uint64_t k[4], uint64_t out[24], uint8_t* out_inf) { g2a x = g2_load(p, inf), r = g2_mul(&x, k); g2_store(out, out_inf, &r); }
void orc_g2_psi(const uint64_t p[24], uint64_t out[24]) { g2a x = g2_load(p, 0), r = g2_psi(&x); g2_store(out, NULL, &r); }
int orc_g2_is_on_curve(const uint64_t p[24]) { g2a x = g2_load(p, 0); return g2_on_curve(&x); }
int orc_g2_is_torsion_free(const uint64_t p[24]) { g2a x = g2_load(p, 0); return g2_torsion_free(&x); }
int orc_g2_is_valid(const uint64_t p[24], uint8_t inf) {
    if (inf) return 0;
    g2a x = g2_load(p, 0);
    if (!g2_on_curve(&x)) return 1;
    if (!g2_torsion_free(&x)) return 2;
    return 0;
}

static fp12 mml_group(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t base, size_t k) {
    g1a* ps = (g1a*)malloc(sizeof(g1a) * (k ? k : 1));
    g2a* qs = (g2a*)malloc(sizeof(g2a) * (k ? k : 1));
    for (size_t j = 0; j < k; j++) {
        ps[j] = g1_load(g1 + 12 * (base + j), inf1 ? inf1[base + j] : 0);
        qs[j] = g2_load(g2 + 24 * (base + j), inf2 ? inf2[base + j] : 0);
    }
    fp12 f = multi_miller_loop(ps, qs, k);
    free(ps);
    free(qs);
    return f;
}
void orc_multi_miller_loop_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks, size_t k, uint64_t* out_ml) {
    for (size_t i = 0; i < n_checks; i++) {
        fp12 f = mml_group(g1, g2, inf1, inf2, i * k, k);
        fp12_store(out_ml + 72 * i, &f);
    }
}
void orc_final_exponentiation_batch(const uint64_t* f, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; i++) {
        fp12 x = fp12_load(f + 72 * i), r = final_exponentiation(&x);
        fp12_store(out + 72 * i, &r);
    }
}
void orc_pairing_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n, uint64_t* out_gt) {
    for (size_t i = 0; i < n; i++) {
        fp12 f = mml_group(g1, g2, inf1, inf2, i, 1), r = final_exponentiation(&f);
        fp12_store(out_gt + 72 * i, &r);
    }
}
void orc_pairing_check_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks, size_t k, uint8_t* ok) {
    fp12 one = fp12_one();
    for (size_t i = 0; i < n_checks; i++) {
        fp12 f = mml_group(g1, g2, inf1, inf2, i * k, k), r = final_exponentiation(&f);
        ok[i] = (uint8_t)fp12_eq(&r, &one);
    }
}
void orc_miller_loop_affine(const uint64_t g1[12], const uint64_t g2[24], uint64_t out[72]) {
    g1a p = g1_load(g1, 0); g2a q = g2_load(g2, 0);
    fp12 f = miller_loop_affine(&p, &q);
    fp12_store(out, &f);
}

/* ---- pthread fan-out helpers ---- */
typedef struct { int kind; const uint64_t *a, *b; const uint8_t *i1, *i2; size_t lo, hi; uint64_t* out; } job_t;
static void* job_run(void* arg) {
    job_t* j = (job_t*)arg;
    if (j->lo >= j->hi) return NULL;
    size_t n = j->hi - j->lo;
    if (j->kind == 0)
        orc_pairing_batch(j->a + 12 * j->lo, j->b + 24 * j->lo, j->i1 ? j->i1 + j->lo : NULL, j->i2 ? j->i2 + j->lo : NULL, n, j->out + 72 * j->lo);
    else if (j->kind == 1)
        for (size_t i = j->lo; i < j->hi; i++) orc_g1_mul(j->a + 12 * i, 0, j->b + 4 * i, j->out + 12 * i, NULL);
    else
        for (size_t i = j->lo; i < j->hi; i++) orc_g2_mul(j->a + 24 * i, 0, j->b + 4 * i, j->out + 24 * i, NULL);
    return NULL;
}
static void fan_out(int kind, const uint64_t* a, const uint64_t* b, const uint8_t* i1, const uint8_t* i2, size_t n, uint64_t* out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t].kind = kind; jobs[t].a = a; jobs[t].b = b; jobs[t].i1 = i1; jobs[t].i2 = i2; jobs[t].out = out;
        jobs[t].lo = n * (size_t)t / (size_t)nthreads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
        pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
void orc_pairing_batch_mt(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n, uint64_t* out_gt, int nthreads) {
    fan_out(0, g1, g2, inf1, inf2, n, out_gt, nthreads);
}
void orc_g1_mul_batch_mt(const uint64_t* p, const uint64_t* k, size_t n, uint64_t* out, int nthreads) { fan_out(1, p, k, NULL, NULL, n, out, nthreads); }
void orc_g2_mul_batch_mt(const uint64_t* p, const uint64_t* k, size_t n, uint64_t* out, int nthreads) { fan_out(2, p, k, NULL, NULL, n, out, nthreads); }
