/*
 * oracle/bls12_381_oracle.h -- CPU restatement (plain C11) of the BLS12-381 tower, groups and
 * optimal-ate pairing.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (zkvm_pairings_amd/, the C-ABI library)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the CPU number reported beside the GPU.
 *
 * What it restates (reference = 0xWOLAND/zkvm-pairings @ 2024_08_07, paths relative to
 * /root/reference):
 *   Fp    src/fp.rs     canonical 6 x u64 little-endian limbs at every API edge (src/fp.rs:24);
 *                       Montgomery form is used only inside this file.
 *   Fp2   src/fp2.rs    Fp6  src/fp6.rs    Fp12  src/fp12.rs
 *   G1    src/g1.rs     G2   src/g2.rs     constants src/common.rs:68-157
 *   pairing / multi_miller_loop / final_exponentiation: src/pairings.rs is EMPTY in the reference
 *   (0 bytes).  The build defines them: optimal-ate Miller loop over |x| with the
 *   projective doubling/addition steps of ePrint 2010/354 Alg. 26/27, line placed through
 *   Fp12::mul_by_014 (src/fp12.rs:99-111), conjugation for negative x, and the upstream-shaped
 *   final exponentiation f^(3(p^12-1)/r) built on the TRUE Frobenius.
 *
 * PARITY PIN STATUS
 *   tower + groups : pinned by the reference's own known-answer tests (tests/golden/ref_kats.json,
 *                    extracted from src/fp.rs:577-588, src/g1.rs:258-341, src/g2.rs:276-443) and by
 *                    the fixed-operand identities of src/fp6.rs:561-757, src/fp12.rs:413-799.
 *   pairing        : "parity unpinned" by the reference (it has no pairing code and no pairing
 *                    vectors).  Pinned instead by an independent big-integer model
 *                    (tests/golden/bls12_381_model.py), bilinearity, e^r = 1, chain == direct
 *                    exponentiation, affine-slope Miller == projective Miller after final exp.
 *
 * Wire formats (identical to include/zkp_pairings.h):
 *   fp   = uint64_t[6]  canonical little-endian limbs, value in [0,p)
 *   fp2  = c0 | c1                      (src/fp2.rs:10-15)
 *   fp6  = c0 | c1 | c2 (each fp2)      (src/fp6.rs:13-17)
 *   fp12 = c0 | c1 (each fp6)           (src/fp12.rs:13-16)
 *   g1   = x | y (12 u64) + separate infinity byte; g2 = x.c0|x.c1|y.c0|y.c1 (24 u64) + byte
 */
#ifndef BLS12_381_ORACLE_H
#define BLS12_381_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Fp (canonical in / canonical out) ---- */
void orc_fp_add(const uint64_t a[6], const uint64_t b[6], uint64_t out[6]);
void orc_fp_sub(const uint64_t a[6], const uint64_t b[6], uint64_t out[6]);
void orc_fp_neg(const uint64_t a[6], uint64_t out[6]);
void orc_fp_mul(const uint64_t a[6], const uint64_t b[6], uint64_t out[6]);
void orc_fp_square(const uint64_t a[6], uint64_t out[6]);
int orc_fp_invert(const uint64_t a[6], uint64_t out[6]);            /* 0 if a == 0 (ref returns None) */
int orc_fp_sqrt(const uint64_t a[6], uint64_t out[6]);              /* 0 if non-residue (ref returns Err) */
void orc_fp_pow_vartime(const uint64_t a[6], const uint64_t e[6], uint64_t out[6]);
int orc_fp_is_canonical(const uint64_t a[6]);
void orc_fp_to_bytes_be(const uint64_t a[6], uint8_t out[48]);      /* src/fp.rs:195-207 */
int orc_fp_from_bytes_be(const uint8_t in[48], uint64_t out[6]);    /* CORRECT range check (ref F4 is inverted) */

/* ---- Fp2 ---- */
void orc_fp2_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]);
void orc_fp2_sub(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]);
void orc_fp2_neg(const uint64_t a[12], uint64_t out[12]);
void orc_fp2_mul(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]);
void orc_fp2_square(const uint64_t a[12], uint64_t out[12]);
void orc_fp2_conjugate(const uint64_t a[12], uint64_t out[12]);
void orc_fp2_mul_by_nonresidue(const uint64_t a[12], uint64_t out[12]);
int orc_fp2_invert(const uint64_t a[12], uint64_t out[12]);
int orc_fp2_sqrt(const uint64_t a[12], uint64_t out[12]);

/* ---- Fp6 ---- */
void orc_fp6_add(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]);
void orc_fp6_sub(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]);
void orc_fp6_neg(const uint64_t a[36], uint64_t out[36]);
void orc_fp6_mul(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]);
void orc_fp6_square(const uint64_t a[36], uint64_t out[36]);
void orc_fp6_mul_by_1(const uint64_t a[36], const uint64_t c1[12], uint64_t out[36]);
void orc_fp6_mul_by_01(const uint64_t a[36], const uint64_t c0[12], const uint64_t c1[12], uint64_t out[36]);
void orc_fp6_mul_by_nonresidue(const uint64_t a[36], uint64_t out[36]);
int orc_fp6_invert(const uint64_t a[36], uint64_t out[36]);
void orc_fp6_frobenius_map(const uint64_t a[36], uint64_t out[36]);            /* TRUE x^p */
void orc_fp6_frobenius_map_refcompat(const uint64_t a[36], uint64_t out[36]);  /* what src/fp6.rs:142-176 computes */

/* ---- Fp12 ---- */
void orc_fp12_one(uint64_t out[72]);
void orc_fp12_add(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]);
void orc_fp12_sub(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]);
void orc_fp12_mul(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]);
void orc_fp12_square(const uint64_t a[72], uint64_t out[72]);
void orc_fp12_mul_by_014(const uint64_t a[72], const uint64_t c0[12], const uint64_t c1[12], const uint64_t c4[12],
                         uint64_t out[72]);
void orc_fp12_conjugate(const uint64_t a[72], uint64_t out[72]);
int orc_fp12_invert(const uint64_t a[72], uint64_t out[72]);
void orc_fp12_frobenius_map(const uint64_t a[72], uint64_t out[72]);            /* TRUE x^p */
void orc_fp12_frobenius_map_refcompat(const uint64_t a[72], uint64_t out[72]);  /* src/fp12.rs:143-170 on top of the buggy Fp6 map */
void orc_fp12_cyclotomic_square(const uint64_t a[72], uint64_t out[72]);
void orc_fp12_pow_u64(const uint64_t a[72], uint64_t e, uint64_t out[72]);

/* ---- G1 / G2, reference-style affine arithmetic (1 inversion per op), infinity as a flag ---- */
void orc_g1_generator(uint64_t out[12]);
void orc_g2_generator(uint64_t out[24]);
void orc_g1_double(const uint64_t p[12], uint8_t inf, uint64_t out[12], uint8_t* out_inf);
void orc_g1_add(const uint64_t p[12], uint8_t pinf, const uint64_t q[12], uint8_t qinf, uint64_t out[12],
                uint8_t* out_inf);
void orc_g1_mul(const uint64_t p[12], uint8_t inf, const uint64_t k[4], uint64_t out[12], uint8_t* out_inf);
int orc_g1_is_on_curve(const uint64_t p[12]);
int orc_g1_is_torsion_free(const uint64_t p[12]);
/* 0 = ok, 1 = not on curve, 2 = not torsion free  (src/g1.rs:49-62) */
int orc_g1_is_valid(const uint64_t p[12], uint8_t inf);
void orc_g2_double(const uint64_t p[24], uint8_t inf, uint64_t out[24], uint8_t* out_inf);
void orc_g2_add(const uint64_t p[24], uint8_t pinf, const uint64_t q[24], uint8_t qinf, uint64_t out[24],
                uint8_t* out_inf);
void orc_g2_mul(const uint64_t p[24], uint8_t inf, const uint64_t k[4], uint64_t out[24], uint8_t* out_inf);
void orc_g2_psi(const uint64_t p[24], uint64_t out[24]);
int orc_g2_is_on_curve(const uint64_t p[24]);
int orc_g2_is_torsion_free(const uint64_t p[24]);
int orc_g2_is_valid(const uint64_t p[24], uint8_t inf);

/* ---- pairing (defined by the build; see header comment) ---- */
/* n_checks groups of k pairs each; out_ml[i] = multi_miller_loop(group i). inf1/inf2 may be NULL. */
void orc_multi_miller_loop_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                 size_t n_checks, size_t k, uint64_t* out_ml);
void orc_final_exponentiation_batch(const uint64_t* f, size_t n, uint64_t* out);
void orc_pairing_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n,
                       uint64_t* out_gt);
/* ok[i] = (final_exp(multi_miller_loop(group i)) == 1) */
void orc_pairing_check_batch(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                             size_t n_checks, size_t k, uint8_t* ok);
/* independent affine-slope Miller formulation (equal to the projective one only after final exp) */
void orc_miller_loop_affine(const uint64_t g1[12], const uint64_t g2[24], uint64_t out[72]);
/* multi-threaded pairing batch for the bench's cpu_baseline leg and big parity tests */
void orc_pairing_batch_mt(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n,
                          uint64_t* out_gt, int nthreads);
void orc_g1_mul_batch_mt(const uint64_t* p, const uint64_t* k, size_t n, uint64_t* out, int nthreads);
void orc_g2_mul_batch_mt(const uint64_t* p, const uint64_t* k, size_t n, uint64_t* out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
