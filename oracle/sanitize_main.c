/* AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU restatement (test infrastructure): every exported group of
 * functions is exercised once, results are cross-checked by algebraic identities so that the run means something.
 * Build + run: make -C oracle sanitize   (tests/test_sanitizers.py does that in the CPU suite). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bls12_381_oracle.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "sanitize_main: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

static uint64_t rng_state = 0x5EEDB15381ULL;
static uint64_t rnd(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rnd_fp(uint64_t* a) { /* < 2^380 < p */
    for (int i = 0; i < 6; i++) a[i] = rnd();
    a[5] &= 0x0fffffffffffffffULL;
}

int main(void) {
    uint64_t a[72], b[72], c[72], d[72], e[72];
    for (int i = 0; i < 12; i++) { rnd_fp(a + 6 * i); rnd_fp(b + 6 * i); rnd_fp(c + 6 * i); }
    /* Fp / Fp2 / Fp6 / Fp12: (a b) c == a (b c), a a^-1 == 1, frobenius^12 == id, conj(conj) == id */
    orc_fp_mul(a, b, d); orc_fp_mul(d, c, d); orc_fp_mul(b, c, e); orc_fp_mul(a, e, e);
    REQUIRE(memcmp(d, e, 48) == 0);
    REQUIRE(orc_fp_invert(a, d) && (orc_fp_mul(a, d, d), d[0] == 1 && d[1] == 0 && d[5] == 0));
    orc_fp_square(a, d); REQUIRE(orc_fp_sqrt(d, e)); orc_fp_square(e, e); REQUIRE(memcmp(d, e, 48) == 0);
    orc_fp2_mul(a, b, d); orc_fp2_mul(d, c, d); orc_fp2_mul(b, c, e); orc_fp2_mul(a, e, e);
    REQUIRE(memcmp(d, e, 96) == 0);
    REQUIRE(orc_fp2_invert(a, d)); orc_fp2_mul(a, d, d); REQUIRE(d[0] == 1 && d[6] == 0);
    orc_fp6_mul(a, b, d); orc_fp6_mul(d, c, d); orc_fp6_mul(b, c, e); orc_fp6_mul(a, e, e);
    REQUIRE(memcmp(d, e, 288) == 0);
    orc_fp6_square(a, d); orc_fp6_mul(a, a, e); REQUIRE(memcmp(d, e, 288) == 0);
    orc_fp12_mul(a, b, d); orc_fp12_mul(d, c, d); orc_fp12_mul(b, c, e); orc_fp12_mul(a, e, e);
    REQUIRE(memcmp(d, e, 576) == 0);
    orc_fp12_square(a, d); orc_fp12_mul(a, a, e); REQUIRE(memcmp(d, e, 576) == 0);
    REQUIRE(orc_fp12_invert(a, d)); orc_fp12_mul(a, d, d); orc_fp12_one(e); REQUIRE(memcmp(d, e, 576) == 0);
    memcpy(d, a, 576);
    for (int i = 0; i < 12; i++) orc_fp12_frobenius_map(d, d);
    REQUIRE(memcmp(d, a, 576) == 0);
    orc_fp12_mul_by_014(a, b, b + 12, b + 24, d);
    memset(e, 0, 576); memcpy(e, b, 96); memcpy(e + 12, b + 12, 96); memcpy(e + 48, b + 24, 96);
    orc_fp12_mul(a, e, e); REQUIRE(memcmp(d, e, 576) == 0);
    /* groups: generators valid, [r]G == infinity, 2G == G + G, psi / validity on a non-subgroup input */
    uint64_t g1[12], g2[24], p1[12], p2[24], q1[12], q2[24];
    uint8_t inf = 9, inf2 = 9;
    orc_g1_generator(g1); orc_g2_generator(g2);
    REQUIRE(orc_g1_is_valid(g1, 0) == 0 && orc_g2_is_valid(g2, 0) == 0);
    static const uint64_t R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    orc_g1_mul(g1, 0, R, p1, &inf); REQUIRE(inf == 1);
    orc_g2_mul(g2, 0, R, p2, &inf); REQUIRE(inf == 1);
    orc_g1_double(g1, 0, p1, &inf); orc_g1_add(g1, 0, g1, 0, q1, &inf2); REQUIRE(!inf && !inf2 && memcmp(p1, q1, 96) == 0);
    orc_g2_double(g2, 0, p2, &inf); orc_g2_add(g2, 0, g2, 0, q2, &inf2); REQUIRE(!inf && !inf2 && memcmp(p2, q2, 192) == 0);
    memcpy(q1, g1, 96); q1[6] ^= 1; REQUIRE(orc_g1_is_valid(q1, 0) == 1);
    memcpy(q2, g2, 192); q2[12] ^= 1; REQUIRE(orc_g2_is_valid(q2, 0) == 1);
    REQUIRE(orc_g1_is_valid(q1, 1) == 0);
    /* pairing: bilinearity e(2P, Q) == e(P, 2Q) == e(P, Q)^2, e(P,Q) e(-P,Q) == 1 through the check, infinities, mt == st */
    uint64_t e1[72], e2[72], e3[72];
    orc_pairing_batch(p1, g2, NULL, NULL, 1, e1);
    orc_pairing_batch(g1, p2, NULL, NULL, 1, e2);
    orc_pairing_batch(g1, g2, NULL, NULL, 1, e3);
    orc_fp12_square(e3, e3);
    REQUIRE(memcmp(e1, e2, 576) == 0 && memcmp(e1, e3, 576) == 0);
    uint64_t G1s[24], G2s[48];
    memcpy(G1s, g1, 96); memcpy(G1s + 12, g1, 48); orc_fp_neg(g1 + 6, G1s + 18);
    memcpy(G2s, g2, 192); memcpy(G2s + 24, g2, 192);
    uint8_t ok[2] = {9, 9};
    orc_pairing_check_batch(G1s, G2s, NULL, NULL, 1, 2, ok); REQUIRE(ok[0] == 1);
    uint8_t i1[2] = {1, 0}, i2[2] = {0, 0};
    orc_pairing_check_batch(G1s, G2s, i1, i2, 2, 1, ok); REQUIRE(ok[0] == 1 && ok[1] == 0);
    enum { N = 9 };
    uint64_t* bg1 = malloc(96 * N), *bg2 = malloc(192 * N), *o1 = malloc(576 * N), *o2 = malloc(576 * N), *ks = calloc(4 * N, 8);
    for (int i = 0; i < N; i++) { memcpy(bg1 + 12 * i, g1, 96); memcpy(bg2 + 24 * i, g2, 192); ks[4 * i] = 3 + i; ks[4 * i + 1] = rnd(); }
    orc_g1_mul_batch_mt(bg1, ks, N, bg1, 3);
    orc_g2_mul_batch_mt(bg2, ks, N, bg2, 4);
    orc_pairing_batch(bg1, bg2, NULL, NULL, N, o1);
    orc_pairing_batch_mt(bg1, bg2, NULL, NULL, N, o2, 5);
    REQUIRE(memcmp(o1, o2, 576 * N) == 0);
    orc_multi_miller_loop_batch(bg1, bg2, NULL, NULL, 3, 3, o1);
    orc_final_exponentiation_batch(o1, 3, o2);
    orc_miller_loop_affine(bg1, bg2, o1);
    free(bg1); free(bg2); free(o1); free(o2); free(ks);
    uint8_t bytes[48];
    orc_fp_to_bytes_be(a, bytes); REQUIRE(orc_fp_from_bytes_be(bytes, d) && memcmp(a, d, 48) == 0);
    memset(bytes, 0xff, 48); REQUIRE(!orc_fp_from_bytes_be(bytes, d));
    printf("oracle sanitize run ok\n");
    return 0;
}
