// zkp_field.hpp -- device-side BLS12-381 tower for the one-element-per-lane kernel family
// (gfx950 only).  Fp is 12 x 32-bit Montgomery limbs (R = 2^384), always fully reduced in [0,p).
//
// What each function stands in for in the reference (paths relative to /root/reference):
//   fp_*    src/fp.rs   (add :352-368, neg :383-405, sub :409-411, mul :416-434, invert :307-319)
//   fp2_*   src/fp2.rs  (mul :192-209, square :171-189, mul_by_nonresidue :161-168, invert :278-296)
//   fp6_*   src/fp6.rs  (mul :188-267, square :274-288, mul_by_1 :102-108, mul_by_01 :110-125)
//   fp12_*  src/fp12.rs (mul :193-210, square :173-184, mul_by_014 :99-111, invert :186-190)
// The Frobenius maps are the TRUE x -> x^p (the reference's Fp6 constants are wrong).
//
// Everything is passed by pointer and the mid-level routines are noinline on purpose: this family
// favours small code (fits the instruction cache) over register residency; the lane-cooperative
// family (zkp_coop.hip) is the throughput path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zkp_constants.h"

namespace zkp {

struct Fp { uint32_t l[12]; };
struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

#define ZKP_DECL_CONST(name, ...) __device__ __constant__ const uint32_t K_##name[12] = {__VA_ARGS__};
ZKP_CONST_LIST(ZKP_DECL_CONST)
#undef ZKP_DECL_CONST

#define ZKP_NOINLINE __attribute__((noinline))
#define ZKP_DEV __device__

ZKP_DEV inline const Fp* kfp(const uint32_t* k) { return reinterpret_cast<const Fp*>(k); }

// ------------------------------------------------------------------------------------------- Fp
ZKP_DEV inline void fp_set(Fp* r, const uint32_t* k) {
#pragma unroll
    for (int i = 0; i < 12; i++) r->l[i] = k[i];
}
ZKP_DEV inline void fp_zero(Fp* r) {
#pragma unroll
    for (int i = 0; i < 12; i++) r->l[i] = 0;
}
ZKP_DEV inline void fp_one(Fp* r) { fp_set(r, K_R); }
ZKP_DEV inline bool fp_is_zero(const Fp* a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a->l[i];
    return o == 0;
}
ZKP_DEV inline bool fp_eq(const Fp* a, const Fp* b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a->l[i] ^ b->l[i];
    return o == 0;
}

// r = a + b mod p (branch-free select between a+b and a+b-p)
ZKP_DEV ZKP_NOINLINE void fp_add(Fp* r, const Fp* a, const Fp* b) {
    uint32_t s[12], d[12];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        c += (uint64_t)a->l[i] + b->l[i];
        s[i] = (uint32_t)c;
        c >>= 32;
    }
    int64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        bw += (int64_t)s[i] - K_P[i];
        d[i] = (uint32_t)bw;
        bw >>= 32;
    }
    // a+b < 2p < 2^384: no carry out of s; take d unless the subtraction borrowed
    bool use_s = bw != 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r->l[i] = use_s ? s[i] : d[i];
}
// r = a - b mod p
ZKP_DEV ZKP_NOINLINE void fp_sub(Fp* r, const Fp* a, const Fp* b) {
    uint32_t d[12];
    int64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        bw += (int64_t)a->l[i] - b->l[i];
        d[i] = (uint32_t)bw;
        bw >>= 32;
    }
    uint32_t mask = bw != 0 ? 0xffffffffu : 0u;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        c += (uint64_t)d[i] + (K_P[i] & mask);
        r->l[i] = (uint32_t)c;
        c >>= 32;
    }
}
ZKP_DEV inline void fp_neg(Fp* r, const Fp* a) {
    Fp z;
    fp_zero(&z);
    fp_sub(r, &z, a);
}
ZKP_DEV inline void fp_dbl(Fp* r, const Fp* a) { fp_add(r, a, a); }

// Montgomery product (CIOS, 32-bit limbs, v_mad_u64_u32).  r may alias a or b.
ZKP_DEV ZKP_NOINLINE void fp_mul(Fp* r, const Fp* a, const Fp* b) {
    uint32_t A[12], B[12], t[13];
#pragma unroll
    for (int i = 0; i < 12; i++) { A[i] = a->l[i]; B[i] = b->l[i]; t[i] = 0; }
    t[12] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c += (uint64_t)A[j] * B[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[12];
        t[12] = (uint32_t)c;
        uint32_t t13 = (uint32_t)(c >> 32);
        uint32_t m = t[0] * ZKP_INV32;
        c = (uint64_t)m * K_P[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 12; j++) {
            c += (uint64_t)m * K_P[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[12];
        t[11] = (uint32_t)c;
        t[12] = t13 + (uint32_t)(c >> 32);
    }
    uint32_t d[12];
    int64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        bw += (int64_t)t[i] - K_P[i];
        d[i] = (uint32_t)bw;
        bw >>= 32;
    }
    bool use_t = (bw != 0) && (t[12] == 0);
#pragma unroll
    for (int i = 0; i < 12; i++) r->l[i] = use_t ? t[i] : d[i];
}
ZKP_DEV inline void fp_sqr(Fp* r, const Fp* a) { fp_mul(r, a, a); }

// a^(p-2), square-and-multiply over the bits of p-2 (reference src/fp.rs:264-276, :307-319).
// Returns false for a == 0 (the reference returns None).
ZKP_DEV ZKP_NOINLINE bool fp_inv(Fp* r, const Fp* a) {
    Fp base = *a, res;
    fp_one(&res);
    for (int w = 11; w >= 0; w--) {
        uint32_t e = K_P_MINUS_2[w];
        for (int i = 31; i >= 0; i--) {
            fp_sqr(&res, &res);
            if ((e >> i) & 1) fp_mul(&res, &res, &base);
        }
    }
    *r = res;
    return !fp_is_zero(a);
}

// canonical wire format (6 x u64 LE == 12 x u32 LE) <-> Montgomery
ZKP_DEV inline void fp_load(Fp* r, const uint64_t* src) {
    Fp t;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        uint64_t v = src[i];
        t.l[2 * i] = (uint32_t)v;
        t.l[2 * i + 1] = (uint32_t)(v >> 32);
    }
    fp_mul(r, &t, kfp(K_R2));
}
ZKP_DEV inline void fp_store(uint64_t* dst, const Fp* a) {
    Fp one, t;
    fp_zero(&one);
    one.l[0] = 1;
    fp_mul(&t, a, &one);
#pragma unroll
    for (int i = 0; i < 6; i++) dst[i] = (uint64_t)t.l[2 * i] | ((uint64_t)t.l[2 * i + 1] << 32);
}
ZKP_DEV inline bool fp_wire_is_canonical(const uint64_t* src) {
    int64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        uint64_t v = src[i];
        bw += (int64_t)(uint32_t)v - K_P[2 * i];
        bw >>= 32;
        bw += (int64_t)(uint32_t)(v >> 32) - K_P[2 * i + 1];
        bw >>= 32;
    }
    return bw != 0;  // borrow <=> value < p
}

// ------------------------------------------------------------------------------------------- Fp2
ZKP_DEV inline void fp2_zero(Fp2* r) { fp_zero(&r->c0); fp_zero(&r->c1); }
ZKP_DEV inline void fp2_one(Fp2* r) { fp_one(&r->c0); fp_zero(&r->c1); }
ZKP_DEV inline bool fp2_is_zero(const Fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
ZKP_DEV inline bool fp2_eq(const Fp2* a, const Fp2* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
ZKP_DEV inline void fp2_add(Fp2* r, const Fp2* a, const Fp2* b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
ZKP_DEV inline void fp2_sub(Fp2* r, const Fp2* a, const Fp2* b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
ZKP_DEV inline void fp2_neg(Fp2* r, const Fp2* a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
ZKP_DEV inline void fp2_dbl(Fp2* r, const Fp2* a) { fp2_add(r, a, a); }
ZKP_DEV inline void fp2_conj(Fp2* r, const Fp2* a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
ZKP_DEV inline void fp2_mul_nr(Fp2* r, const Fp2* a) {  // * (1 + u)
    Fp t0, t1;
    fp_sub(&t0, &a->c0, &a->c1);
    fp_add(&t1, &a->c0, &a->c1);
    r->c0 = t0;
    r->c1 = t1;
}
// Karatsuba: 3 Fp products (value identical to the reference's 4-product schoolbook)
ZKP_DEV ZKP_NOINLINE void fp2_mul(Fp2* r, const Fp2* a, const Fp2* b) {
    Fp t0, t1, sa, sb, t2;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&sa, &a->c0, &a->c1);
    fp_add(&sb, &b->c0, &b->c1);
    fp_mul(&t2, &sa, &sb);
    fp_sub(&r->c0, &t0, &t1);
    fp_sub(&t2, &t2, &t0);
    fp_sub(&r->c1, &t2, &t1);
}
ZKP_DEV ZKP_NOINLINE void fp2_sqr(Fp2* r, const Fp2* a) {
    Fp s, d, c;
    fp_add(&s, &a->c0, &a->c1);
    fp_sub(&d, &a->c0, &a->c1);
    fp_add(&c, &a->c0, &a->c0);
    fp_mul(&r->c1, &c, &a->c1);
    fp_mul(&r->c0, &s, &d);
}
ZKP_DEV inline void fp2_mul_fp(Fp2* r, const Fp2* a, const Fp* s) { fp_mul(&r->c0, &a->c0, s); fp_mul(&r->c1, &a->c1, s); }
ZKP_DEV ZKP_NOINLINE bool fp2_inv(Fp2* r, const Fp2* a) {
    Fp s0, s1, n, t, nt;
    fp_sqr(&s0, &a->c0);
    fp_sqr(&s1, &a->c1);
    fp_add(&n, &s0, &s1);
    bool ok = fp_inv(&t, &n);
    fp_neg(&nt, &t);
    fp_mul(&r->c0, &a->c0, &t);
    fp_mul(&r->c1, &a->c1, &nt);
    return ok;
}
ZKP_DEV inline void fp2_load(Fp2* r, const uint64_t* src) { fp_load(&r->c0, src); fp_load(&r->c1, src + 6); }
ZKP_DEV inline void fp2_store(uint64_t* dst, const Fp2* a) { fp_store(dst, &a->c0); fp_store(dst + 6, &a->c1); }
ZKP_DEV inline void fp2_const(Fp2* r, const uint32_t* k0, const uint32_t* k1) { fp_set(&r->c0, k0); fp_set(&r->c1, k1); }

// ------------------------------------------------------------------------------------------- Fp6
ZKP_DEV inline void fp6_zero(Fp6* r) { fp2_zero(&r->c0); fp2_zero(&r->c1); fp2_zero(&r->c2); }
ZKP_DEV inline void fp6_one(Fp6* r) { fp2_one(&r->c0); fp2_zero(&r->c1); fp2_zero(&r->c2); }
ZKP_DEV inline bool fp6_eq(const Fp6* a, const Fp6* b) { return fp2_eq(&a->c0, &b->c0) && fp2_eq(&a->c1, &b->c1) && fp2_eq(&a->c2, &b->c2); }
ZKP_DEV inline void fp6_add(Fp6* r, const Fp6* a, const Fp6* b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
ZKP_DEV inline void fp6_sub(Fp6* r, const Fp6* a, const Fp6* b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
ZKP_DEV inline void fp6_neg(Fp6* r, const Fp6* a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
ZKP_DEV inline void fp6_mul_nr(Fp6* r, const Fp6* a) {  // * v
    Fp2 t;
    fp2_mul_nr(&t, &a->c2);
    r->c2 = a->c1;
    r->c1 = a->c0;
    r->c0 = t;
}
// Karatsuba over Fp2: 6 Fp2 products
ZKP_DEV ZKP_NOINLINE void fp6_mul(Fp6* r, const Fp6* a, const Fp6* b) {
    Fp2 v0, v1, v2, s, t, u, c0, c1, c2;
    fp2_mul(&v0, &a->c0, &b->c0);
    fp2_mul(&v1, &a->c1, &b->c1);
    fp2_mul(&v2, &a->c2, &b->c2);
    fp2_add(&s, &a->c1, &a->c2);
    fp2_add(&t, &b->c1, &b->c2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v1);
    fp2_sub(&u, &u, &v2);
    fp2_mul_nr(&u, &u);
    fp2_add(&c0, &u, &v0);
    fp2_add(&s, &a->c0, &a->c1);
    fp2_add(&t, &b->c0, &b->c1);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v1);
    fp2_mul_nr(&t, &v2);
    fp2_add(&c1, &u, &t);
    fp2_add(&s, &a->c0, &a->c2);
    fp2_add(&t, &b->c0, &b->c2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v2);
    fp2_add(&c2, &u, &v1);
    r->c0 = c0;
    r->c1 = c1;
    r->c2 = c2;
}
ZKP_DEV ZKP_NOINLINE void fp6_sqr(Fp6* r, const Fp6* a) {  // CH-SQR2
    Fp2 s0, ab, s1, t, s2, bc, s3, s4, x;
    fp2_sqr(&s0, &a->c0);
    fp2_mul(&ab, &a->c0, &a->c1);
    fp2_dbl(&s1, &ab);
    fp2_sub(&t, &a->c0, &a->c1);
    fp2_add(&t, &t, &a->c2);
    fp2_sqr(&s2, &t);
    fp2_mul(&bc, &a->c1, &a->c2);
    fp2_dbl(&s3, &bc);
    fp2_sqr(&s4, &a->c2);
    fp2_mul_nr(&x, &s3);
    fp2_add(&r->c0, &x, &s0);
    fp2_mul_nr(&x, &s4);
    fp2_add(&r->c1, &x, &s1);
    fp2_add(&x, &s1, &s2);
    fp2_add(&x, &x, &s3);
    fp2_sub(&x, &x, &s0);
    fp2_sub(&r->c2, &x, &s4);
}
ZKP_DEV ZKP_NOINLINE void fp6_mul_by_1(Fp6* r, const Fp6* a, const Fp2* c1) {
    Fp2 t, u, w;
    fp2_mul(&t, &a->c2, c1);
    fp2_mul_nr(&t, &t);
    fp2_mul(&u, &a->c0, c1);
    fp2_mul(&w, &a->c1, c1);
    r->c0 = t;
    r->c1 = u;
    r->c2 = w;
}
ZKP_DEV ZKP_NOINLINE void fp6_mul_by_01(Fp6* r, const Fp6* a, const Fp2* c0, const Fp2* c1) {
    Fp2 a_a, b_b, t, s0, s1, o0, o1, o2;
    fp2_mul(&a_a, &a->c0, c0);
    fp2_mul(&b_b, &a->c1, c1);
    fp2_mul(&t, &a->c2, c1);
    fp2_mul_nr(&t, &t);
    fp2_add(&o0, &t, &a_a);
    fp2_add(&s0, c0, c1);
    fp2_add(&s1, &a->c0, &a->c1);
    fp2_mul(&t, &s0, &s1);
    fp2_sub(&t, &t, &a_a);
    fp2_sub(&o1, &t, &b_b);
    fp2_mul(&t, &a->c2, c0);
    fp2_add(&o2, &t, &b_b);
    r->c0 = o0;
    r->c1 = o1;
    r->c2 = o2;
}
ZKP_DEV ZKP_NOINLINE bool fp6_inv(Fp6* r, const Fp6* a) {
    Fp2 t, c0, c1, c2, u, v, ti;
    fp2_mul(&t, &a->c1, &a->c2);
    fp2_mul_nr(&t, &t);
    fp2_sqr(&c0, &a->c0);
    fp2_sub(&c0, &c0, &t);
    fp2_sqr(&c1, &a->c2);
    fp2_mul_nr(&c1, &c1);
    fp2_mul(&t, &a->c0, &a->c1);
    fp2_sub(&c1, &c1, &t);
    fp2_sqr(&c2, &a->c1);
    fp2_mul(&t, &a->c0, &a->c2);
    fp2_sub(&c2, &c2, &t);
    fp2_mul(&u, &a->c1, &c2);
    fp2_mul(&v, &a->c2, &c1);
    fp2_add(&t, &u, &v);
    fp2_mul_nr(&t, &t);
    fp2_mul(&u, &a->c0, &c0);
    fp2_add(&t, &t, &u);
    bool ok = fp2_inv(&ti, &t);
    fp2_mul(&r->c0, &ti, &c0);
    fp2_mul(&r->c1, &ti, &c1);
    fp2_mul(&r->c2, &ti, &c2);
    return ok;
}
ZKP_DEV ZKP_NOINLINE void fp6_frob(Fp6* r, const Fp6* a) {  // TRUE x^p
    Fp2 k, t;
    fp2_conj(&r->c0, &a->c0);
    fp2_const(&k, K_M_FROB6_C1_0, K_M_FROB6_C1_1);
    fp2_conj(&t, &a->c1);
    fp2_mul(&r->c1, &t, &k);
    fp2_const(&k, K_M_FROB6_C2_0, K_M_FROB6_C2_1);
    fp2_conj(&t, &a->c2);
    fp2_mul(&r->c2, &t, &k);
}
ZKP_DEV inline void fp6_load(Fp6* r, const uint64_t* s) { fp2_load(&r->c0, s); fp2_load(&r->c1, s + 12); fp2_load(&r->c2, s + 24); }
ZKP_DEV inline void fp6_store(uint64_t* d, const Fp6* a) { fp2_store(d, &a->c0); fp2_store(d + 12, &a->c1); fp2_store(d + 24, &a->c2); }

// ------------------------------------------------------------------------------------------- Fp12
ZKP_DEV inline void fp12_one(Fp12* r) { fp6_one(&r->c0); fp6_zero(&r->c1); }
ZKP_DEV inline bool fp12_eq(const Fp12* a, const Fp12* b) { return fp6_eq(&a->c0, &b->c0) && fp6_eq(&a->c1, &b->c1); }
ZKP_DEV inline void fp12_conj(Fp12* r, const Fp12* a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
ZKP_DEV ZKP_NOINLINE void fp12_mul(Fp12* r, const Fp12* a, const Fp12* b) {
    Fp6 aa, bb, o, s, c1, c0;
    fp6_mul(&aa, &a->c0, &b->c0);
    fp6_mul(&bb, &a->c1, &b->c1);
    fp6_add(&o, &b->c0, &b->c1);
    fp6_add(&s, &a->c1, &a->c0);
    fp6_mul(&c1, &s, &o);
    fp6_sub(&c1, &c1, &aa);
    fp6_sub(&c1, &c1, &bb);
    fp6_mul_nr(&c0, &bb);
    fp6_add(&c0, &c0, &aa);
    r->c0 = c0;
    r->c1 = c1;
}
ZKP_DEV ZKP_NOINLINE void fp12_sqr(Fp12* r, const Fp12* a) {
    Fp6 ab, c0c1, c0, c1, abn;
    fp6_mul(&ab, &a->c0, &a->c1);
    fp6_add(&c0c1, &a->c0, &a->c1);
    fp6_mul_nr(&c0, &a->c1);
    fp6_add(&c0, &c0, &a->c0);
    fp6_mul(&c0, &c0, &c0c1);
    fp6_sub(&c0, &c0, &ab);
    fp6_add(&c1, &ab, &ab);
    fp6_mul_nr(&abn, &ab);
    fp6_sub(&c0, &c0, &abn);
    r->c0 = c0;
    r->c1 = c1;
}
ZKP_DEV ZKP_NOINLINE void fp12_mul_by_014(Fp12* r, const Fp12* a, const Fp2* c0, const Fp2* c1, const Fp2* c4) {
    Fp6 aa, bb, r1, r0;
    Fp2 o;
    fp6_mul_by_01(&aa, &a->c0, c0, c1);
    fp6_mul_by_1(&bb, &a->c1, c4);
    fp2_add(&o, c1, c4);
    fp6_add(&r1, &a->c1, &a->c0);
    fp6_mul_by_01(&r1, &r1, c0, &o);
    fp6_sub(&r1, &r1, &aa);
    fp6_sub(&r1, &r1, &bb);
    fp6_mul_nr(&r0, &bb);
    fp6_add(&r0, &r0, &aa);
    r->c0 = r0;
    r->c1 = r1;
}
ZKP_DEV ZKP_NOINLINE bool fp12_inv(Fp12* r, const Fp12* a) {
    Fp6 s0, s1, n, t, nt;
    fp6_sqr(&s0, &a->c0);
    fp6_sqr(&s1, &a->c1);
    fp6_mul_nr(&s1, &s1);
    fp6_sub(&n, &s0, &s1);
    bool ok = fp6_inv(&t, &n);
    fp6_neg(&nt, &t);
    fp6_mul(&r->c0, &a->c0, &t);
    fp6_mul(&r->c1, &a->c1, &nt);
    return ok;
}
ZKP_DEV ZKP_NOINLINE void fp12_frob(Fp12* r, const Fp12* a) {  // TRUE x^p
    Fp2 g;
    Fp6 t;
    fp2_const(&g, K_M_FROB12_C1_0, K_M_FROB12_C1_1);
    fp6_frob(&r->c0, &a->c0);
    fp6_frob(&t, &a->c1);
    fp2_mul(&r->c1.c0, &t.c0, &g);
    fp2_mul(&r->c1.c1, &t.c1, &g);
    fp2_mul(&r->c1.c2, &t.c2, &g);
}
ZKP_DEV inline void fp12_load(Fp12* r, const uint64_t* s) { fp6_load(&r->c0, s); fp6_load(&r->c1, s + 36); }
ZKP_DEV inline void fp12_store(uint64_t* d, const Fp12* a) { fp6_store(d, &a->c0); fp6_store(d + 36, &a->c1); }

// Granger-Scott cyclotomic squaring
ZKP_DEV ZKP_NOINLINE void fp4_square(Fp2* o0, Fp2* o1, const Fp2* a, const Fp2* b) {
    Fp2 t0, t1, t2;
    fp2_sqr(&t0, a);
    fp2_sqr(&t1, b);
    fp2_mul_nr(&t2, &t1);
    fp2_add(o0, &t2, &t0);
    fp2_add(&t2, a, b);
    fp2_sqr(&t2, &t2);
    fp2_sub(&t2, &t2, &t0);
    fp2_sub(o1, &t2, &t1);
}
ZKP_DEV ZKP_NOINLINE void fp12_cyclotomic_square(Fp12* r, const Fp12* f) {
    Fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2;
    Fp2 t0, t1, t2, t3;
    fp4_square(&t0, &t1, &z0, &z1);
    fp2_sub(&z0, &t0, &z0); fp2_dbl(&z0, &z0); fp2_add(&z0, &z0, &t0);
    fp2_add(&z1, &t1, &z1); fp2_dbl(&z1, &z1); fp2_add(&z1, &z1, &t1);
    fp4_square(&t0, &t1, &z2, &z3);
    fp4_square(&t2, &t3, &z4, &z5);
    fp2_sub(&z4, &t0, &z4); fp2_dbl(&z4, &z4); fp2_add(&z4, &z4, &t0);
    fp2_add(&z5, &t1, &z5); fp2_dbl(&z5, &z5); fp2_add(&z5, &z5, &t1);
    fp2_mul_nr(&t0, &t3);
    fp2_add(&z2, &t0, &z2); fp2_dbl(&z2, &z2); fp2_add(&z2, &z2, &t0);
    fp2_sub(&z3, &t2, &z3); fp2_dbl(&z3, &z3); fp2_add(&z3, &z3, &t2);
    r->c0.c0 = z0; r->c0.c1 = z4; r->c0.c2 = z3;
    r->c1.c0 = z2; r->c1.c1 = z1; r->c1.c2 = z5;
}
// f^|x| then conjugate (x < 0)
ZKP_DEV ZKP_NOINLINE void cyclotomic_exp(Fp12* r, const Fp12* f) {
    Fp12 tmp;
    fp12_one(&tmp);
    bool found_one = false;
    for (int i = 63; i >= 0; i--) {
        bool bit = (ZKP_BLS_X >> i) & 1;
        if (found_one) fp12_cyclotomic_square(&tmp, &tmp); else found_one = bit;
        if (bit) fp12_mul(&tmp, &tmp, f);
    }
    fp12_conj(r, &tmp);
}

// ------------------------------------------------------------------------------------------- pairing
struct G1A { Fp x, y; };
struct G2A { Fp2 x, y; };
struct G2P { Fp2 x, y, z; };
struct Line { Fp2 c0, c1, c2; };

// ePrint 2010/354 Alg. 26
ZKP_DEV ZKP_NOINLINE void doubling_step(Line* l, G2P* r) {
    Fp2 tmp0, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6, zsq, nx, ny, nz;
    fp2_sqr(&tmp0, &r->x);
    fp2_sqr(&tmp1, &r->y);
    fp2_sqr(&tmp2, &tmp1);
    fp2_add(&tmp3, &tmp1, &r->x);
    fp2_sqr(&tmp3, &tmp3);
    fp2_sub(&tmp3, &tmp3, &tmp0);
    fp2_sub(&tmp3, &tmp3, &tmp2);
    fp2_dbl(&tmp3, &tmp3);
    fp2_add(&tmp4, &tmp0, &tmp0);
    fp2_add(&tmp4, &tmp4, &tmp0);
    fp2_add(&tmp6, &r->x, &tmp4);
    fp2_sqr(&tmp5, &tmp4);
    fp2_sqr(&zsq, &r->z);
    fp2_sub(&nx, &tmp5, &tmp3);
    fp2_sub(&nx, &nx, &tmp3);
    fp2_add(&nz, &r->z, &r->y);
    fp2_sqr(&nz, &nz);
    fp2_sub(&nz, &nz, &tmp1);
    fp2_sub(&nz, &nz, &zsq);
    fp2_sub(&ny, &tmp3, &nx);
    fp2_mul(&ny, &ny, &tmp4);
    fp2_dbl(&tmp2, &tmp2); fp2_dbl(&tmp2, &tmp2); fp2_dbl(&tmp2, &tmp2);
    fp2_sub(&ny, &ny, &tmp2);
    fp2_mul(&tmp3, &tmp4, &zsq);
    fp2_dbl(&tmp3, &tmp3);
    fp2_neg(&tmp3, &tmp3);
    fp2_sqr(&tmp6, &tmp6);
    fp2_sub(&tmp6, &tmp6, &tmp0);
    fp2_sub(&tmp6, &tmp6, &tmp5);
    fp2_dbl(&tmp1, &tmp1); fp2_dbl(&tmp1, &tmp1);
    fp2_sub(&tmp6, &tmp6, &tmp1);
    fp2_mul(&tmp0, &nz, &zsq);
    fp2_dbl(&tmp0, &tmp0);
    r->x = nx; r->y = ny; r->z = nz;
    l->c0 = tmp0; l->c1 = tmp3; l->c2 = tmp6;
}
// ePrint 2010/354 Alg. 27
ZKP_DEV ZKP_NOINLINE void addition_step(Line* l, G2P* r, const G2A* q) {
    Fp2 zsq, ysq, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, nx, ny, nz, ztsq;
    fp2_sqr(&zsq, &r->z);
    fp2_sqr(&ysq, &q->y);
    fp2_mul(&t0, &zsq, &q->x);
    fp2_add(&t1, &q->y, &r->z);
    fp2_sqr(&t1, &t1);
    fp2_sub(&t1, &t1, &ysq);
    fp2_sub(&t1, &t1, &zsq);
    fp2_mul(&t1, &t1, &zsq);
    fp2_sub(&t2, &t0, &r->x);
    fp2_sqr(&t3, &t2);
    fp2_dbl(&t4, &t3);
    fp2_dbl(&t4, &t4);
    fp2_mul(&t5, &t4, &t2);
    fp2_sub(&t6, &t1, &r->y);
    fp2_sub(&t6, &t6, &r->y);
    fp2_mul(&t9, &t6, &q->x);
    fp2_mul(&t7, &t4, &r->x);
    fp2_sqr(&nx, &t6);
    fp2_sub(&nx, &nx, &t5);
    fp2_sub(&nx, &nx, &t7);
    fp2_sub(&nx, &nx, &t7);
    fp2_add(&nz, &r->z, &t2);
    fp2_sqr(&nz, &nz);
    fp2_sub(&nz, &nz, &zsq);
    fp2_sub(&nz, &nz, &t3);
    fp2_add(&t10, &q->y, &nz);
    fp2_sub(&t8, &t7, &nx);
    fp2_mul(&t8, &t8, &t6);
    fp2_mul(&t0, &r->y, &t5);
    fp2_dbl(&t0, &t0);
    fp2_sub(&ny, &t8, &t0);
    fp2_sqr(&t10, &t10);
    fp2_sub(&t10, &t10, &ysq);
    fp2_sqr(&ztsq, &nz);
    fp2_sub(&t10, &t10, &ztsq);
    fp2_dbl(&t9, &t9);
    fp2_sub(&t9, &t9, &t10);
    fp2_dbl(&t10, &nz);
    fp2_neg(&t6, &t6);
    fp2_dbl(&t1, &t6);
    r->x = nx; r->y = ny; r->z = nz;
    l->c0 = t10; l->c1 = t1; l->c2 = t9;
}
ZKP_DEV ZKP_NOINLINE void ell(Fp12* f, const Line* l, const G1A* p) {
    Fp2 c0, c1;
    fp2_mul_fp(&c0, &l->c0, &p->y);
    fp2_mul_fp(&c1, &l->c1, &p->x);
    fp12_mul_by_014(f, f, &l->c2, &c1, &c0);
}

// f^(3 (p^12-1)/r)
ZKP_DEV ZKP_NOINLINE void final_exponentiation(Fp12* out, const Fp12* fin) {
    Fp12 t0, t1, t2, t3, t4, t5, t6;
    t0 = *fin;
    for (int i = 0; i < 6; i++) fp12_frob(&t0, &t0);
    if (!fp12_inv(&t1, fin)) { fp12_one(out); return; }
    fp12_mul(&t2, &t0, &t1);
    t1 = t2;
    fp12_frob(&t2, &t2);
    fp12_frob(&t2, &t2);
    fp12_mul(&t2, &t2, &t1);
    fp12_cyclotomic_square(&t1, &t2);
    fp12_conj(&t1, &t1);
    cyclotomic_exp(&t3, &t2);
    fp12_cyclotomic_square(&t4, &t3);
    fp12_mul(&t5, &t1, &t3);
    cyclotomic_exp(&t1, &t5);
    cyclotomic_exp(&t0, &t1);
    cyclotomic_exp(&t6, &t0);
    fp12_mul(&t6, &t6, &t4);
    cyclotomic_exp(&t4, &t6);
    fp12_conj(&t5, &t5);
    fp12_mul(&t5, &t5, &t2);
    fp12_mul(&t4, &t4, &t5);
    fp12_conj(&t5, &t2);
    fp12_mul(&t1, &t1, &t2);
    fp12_frob(&t1, &t1); fp12_frob(&t1, &t1); fp12_frob(&t1, &t1);
    fp12_mul(&t6, &t6, &t5);
    fp12_frob(&t6, &t6);
    fp12_mul(&t3, &t3, &t0);
    fp12_frob(&t3, &t3); fp12_frob(&t3, &t3);
    fp12_mul(&t3, &t3, &t1);
    fp12_mul(&t3, &t3, &t6);
    fp12_mul(out, &t3, &t4);
}

// ------------------------------------------------------------------------------------------- Jacobian G1/G2 (validity, scalar mul)
// Generic over the coordinate field via tiny adapters.
struct FpOps {
    typedef Fp E;
    ZKP_DEV static void add(E* r, const E* a, const E* b) { fp_add(r, a, b); }
    ZKP_DEV static void sub(E* r, const E* a, const E* b) { fp_sub(r, a, b); }
    ZKP_DEV static void mul(E* r, const E* a, const E* b) { fp_mul(r, a, b); }
    ZKP_DEV static void sqr(E* r, const E* a) { fp_sqr(r, a); }
    ZKP_DEV static void neg(E* r, const E* a) { fp_neg(r, a); }
    ZKP_DEV static bool inv(E* r, const E* a) { return fp_inv(r, a); }
    ZKP_DEV static bool is_zero(const E* a) { return fp_is_zero(a); }
    ZKP_DEV static bool eq(const E* a, const E* b) { return fp_eq(a, b); }
    ZKP_DEV static void one(E* r) { fp_one(r); }
    ZKP_DEV static void zero(E* r) { fp_zero(r); }
};
struct Fp2Ops {
    typedef Fp2 E;
    ZKP_DEV static void add(E* r, const E* a, const E* b) { fp2_add(r, a, b); }
    ZKP_DEV static void sub(E* r, const E* a, const E* b) { fp2_sub(r, a, b); }
    ZKP_DEV static void mul(E* r, const E* a, const E* b) { fp2_mul(r, a, b); }
    ZKP_DEV static void sqr(E* r, const E* a) { fp2_sqr(r, a); }
    ZKP_DEV static void neg(E* r, const E* a) { fp2_neg(r, a); }
    ZKP_DEV static bool inv(E* r, const E* a) { return fp2_inv(r, a); }
    ZKP_DEV static bool is_zero(const E* a) { return fp2_is_zero(a); }
    ZKP_DEV static bool eq(const E* a, const E* b) { return fp2_eq(a, b); }
    ZKP_DEV static void one(E* r) { fp2_one(r); }
    ZKP_DEV static void zero(E* r) { fp2_zero(r); }
};

template <class F>
struct Jac {
    typename F::E x, y, z;  // z == 0 <=> infinity
};

template <class F>
ZKP_DEV void jac_set_inf(Jac<F>* r) { F::one(&r->x); F::one(&r->y); F::zero(&r->z); }

// a = 0 doubling (dbl-2009-l); handles infinity and y == 0 (-> infinity) like the affine reference
template <class F>
ZKP_DEV ZKP_NOINLINE void jac_double(Jac<F>* r, const Jac<F>* p) {
    typedef typename F::E E;
    if (F::is_zero(&p->z) || F::is_zero(&p->y)) { jac_set_inf(r); return; }
    E a, b, c, d, e, f, t, nx, ny, nz;
    F::sqr(&a, &p->x);
    F::sqr(&b, &p->y);
    F::sqr(&c, &b);
    F::add(&t, &p->x, &b);
    F::sqr(&t, &t);
    F::sub(&t, &t, &a);
    F::sub(&t, &t, &c);
    F::add(&d, &t, &t);
    F::add(&e, &a, &a);
    F::add(&e, &e, &a);
    F::sqr(&f, &e);
    F::sub(&nx, &f, &d);
    F::sub(&nx, &nx, &d);
    F::mul(&nz, &p->y, &p->z);
    F::add(&nz, &nz, &nz);
    F::sub(&t, &d, &nx);
    F::mul(&ny, &e, &t);
    F::add(&c, &c, &c); F::add(&c, &c, &c); F::add(&c, &c, &c);
    F::sub(&ny, &ny, &c);
    r->x = nx; r->y = ny; r->z = nz;
}
// mixed addition r = p + (qx, qy) with every exceptional case handled
template <class F>
ZKP_DEV ZKP_NOINLINE void jac_add_affine(Jac<F>* r, const Jac<F>* p, const typename F::E* qx, const typename F::E* qy) {
    typedef typename F::E E;
    if (F::is_zero(&p->z)) { r->x = *qx; r->y = *qy; F::one(&r->z); return; }
    E z1z1, u2, s2, h, rr, t, hh, i, j, v, nx, ny, nz;
    F::sqr(&z1z1, &p->z);
    F::mul(&u2, qx, &z1z1);
    F::mul(&s2, qy, &p->z);
    F::mul(&s2, &s2, &z1z1);
    F::sub(&h, &u2, &p->x);
    F::sub(&rr, &s2, &p->y);
    if (F::is_zero(&h)) {
        if (F::is_zero(&rr)) { jac_double(r, p); return; }
        jac_set_inf(r);
        return;
    }
    F::sqr(&hh, &h);
    F::add(&i, &hh, &hh);
    F::add(&i, &i, &i);
    F::mul(&j, &h, &i);
    F::add(&rr, &rr, &rr);
    F::mul(&v, &p->x, &i);
    F::sqr(&nx, &rr);
    F::sub(&nx, &nx, &j);
    F::sub(&nx, &nx, &v);
    F::sub(&nx, &nx, &v);
    F::sub(&t, &v, &nx);
    F::mul(&ny, &rr, &t);
    F::mul(&t, &p->y, &j);
    F::add(&t, &t, &t);
    F::sub(&ny, &ny, &t);
    F::add(&nz, &p->z, &h);
    F::sqr(&nz, &nz);
    F::sub(&nz, &nz, &z1z1);
    F::sub(&nz, &nz, &hh);
    r->x = nx; r->y = ny; r->z = nz;
}
// [k]P for a 64-bit scalar chunk array (MSB first), P affine and not infinity
template <class F>
ZKP_DEV ZKP_NOINLINE void jac_mul(Jac<F>* r, const typename F::E* px, const typename F::E* py, const uint64_t* k, int nwords) {
    Jac<F> acc;
    jac_set_inf(&acc);
    for (int w = nwords - 1; w >= 0; w--) {
        uint64_t e = k[w];
        for (int i = 63; i >= 0; i--) {
            jac_double(&acc, &acc);
            if ((e >> i) & 1) jac_add_affine(&acc, &acc, px, py);
        }
    }
    *r = acc;
}
// Jacobian -> affine; returns false for infinity
template <class F>
ZKP_DEV ZKP_NOINLINE bool jac_to_affine(typename F::E* ax, typename F::E* ay, const Jac<F>* p) {
    typedef typename F::E E;
    if (F::is_zero(&p->z)) { F::zero(ax); F::one(ay); return false; }
    E zi, zi2, zi3;
    F::inv(&zi, &p->z);
    F::sqr(&zi2, &zi);
    F::mul(&zi3, &zi2, &zi);
    F::mul(ax, &p->x, &zi2);
    F::mul(ay, &p->y, &zi3);
    return true;
}
// projective comparison: Jacobian p == affine (qx, qy)?  (infinity never equals a finite point)
template <class F>
ZKP_DEV ZKP_NOINLINE bool jac_eq_affine(const Jac<F>* p, const typename F::E* qx, const typename F::E* qy) {
    typedef typename F::E E;
    if (F::is_zero(&p->z)) return false;
    E z2, z3, t;
    F::sqr(&z2, &p->z);
    F::mul(&z3, &z2, &p->z);
    F::mul(&t, qx, &z2);
    if (!F::eq(&t, &p->x)) return false;
    F::mul(&t, qy, &z3);
    return F::eq(&t, &p->y);
}

}  // namespace zkp
