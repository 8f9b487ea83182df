/*
 * include/zkp_pairings.h -- C ABI of libzkp_pairings.so, the MI355X (gfx950) batched BLS12-381
 * pairing engine.  This is the drop-in boundary for the pairing-check path of
 * 0xWOLAND/zkvm-pairings: a Rust/cgo/ctypes binding declares exactly these symbols
 * (see INTEGRATION.md for the Rust `extern "C"` block).
 *
 * Reference interface each entry point stands in for (paths relative to /root/reference):
 *   zkp_pairing_batch               pairing(&G1Affine,&G2Affine)->Gt        src/pairings.rs (declared at
 *   zkp_multi_miller_loop_batch     multi_miller_loop(&[(G1,G2)])            src/lib.rs:12; the file is EMPTY in
 *   zkp_final_exponentiation_batch  MillerLoopResult::final_exponentiation   this snapshot - semantics defined in
 *   zkp_pairing_check_batch         pairing(..) == Gt::identity()            SURVEY.md S6 / DESIGN.md)
 *   zkp_gt_identity                 Gt::identity() == Fp12::one()            src/fp12.rs:87-89
 *   zkp_g1_is_valid_batch           G1Affine::is_valid                       src/g1.rs:49-62  (is_on_curve :95-101,
 *                                                                            is_torsion_free :111-115)
 *   zkp_g2_is_valid_batch           G2Affine::is_valid                       src/g2.rs:57-69  (:109-120, :166-170)
 *   zkp_g1_mul_batch / g2           &G1Affine * &Fr / &G2Affine * &Fr         src/g1.rs:130-153 (bit-0 bug F5 NOT
 *                                                                            mirrored), src/g2.rs:185-208
 *   zkp_fp_op_batch                 bls12381_sys_bigint(out, op, a, b)       src/fp.rs:376,443 (op 0 = mul, 1 = add); also
 *                                   Fp::sub / neg / square / invert          src/fp.rs:307-319, 383-411, 453-455
 *   zkp_tower_op_batch              Fp2 / Fp6 / Fp12 mul, square, mul_by_014, src/fp2.rs:171-209, src/fp6.rs:188-288,
 *                                   conjugate, frobenius_map (the TRUE map),  src/fp12.rs:99-210 (:143-170 is wrong)
 *                                   invert, mul_by_nonresidue, mul_by_1 / 01  src/fp2.rs:95-102,161-168,278-296, src/fp6.rs:102-141,291-309
 *   zkp_points_check_batch          raw points -> Fp::from_bytes, is_valid, pairing check in one call  src/fp.rs:165-207, src/g1.rs:49-62, src/g2.rs:57-69
 *   zkp_pairing_*_multi             the same pairing()/check over several GPUs from ONE host thread
 *   zkp_comm_*, zkp_*_allreduce     one rank per GPU: the check + the path's one RCCL collective
 *
 * Wire formats (all little-endian, canonical representatives in [0,p), identical to the
 * reference's in-memory structs):
 *   fp    uint64_t[6]                          Fp.0                         src/fp.rs:24
 *   fp2   c0 | c1                    (12 u64)  Fp2{c0,c1}                   src/fp2.rs:10-15
 *   fp12  c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1 (72 u64)                src/fp12.rs:13-16, src/fp6.rs:13-17
 *   g1    x | y                      (12 u64)  + parallel uint8 is_infinity src/g1.rs:7-11
 *   g2    x.c0 | x.c1 | y.c0 | y.c1  (24 u64)  + parallel uint8 is_infinity src/g2.rs:8-12
 *   fr    uint64_t[4] canonical scalar                                      src/fr.rs (Fr.0)
 *
 * Ownership: the caller owns every buffer it passes (host or device); the library owns the
 * workspace inside zkp_ctx.  Errors: 0 on success, negative zkp_status otherwise; the library never
 * aborts and never unwinds across this boundary (the reference's panics F7 become status bytes).
 * Inputs must be canonical (< p): non-canonical limbs give ZKP_ERR_NONCANONICAL when validation is on
 * (zkp_set_validate), unspecified field values otherwise.
 * Threading: a zkp_ctx is bound to one GPU and is not thread-safe; use one ctx per thread / rank.  The *_dev calls
 * of one ctx may be issued on different streams: each call first makes its stream wait (an event, no host wait) for the
 * previous call's use of the context's workspace, so they never overlap on it.
 * There is NO CPU fallback: zkp_init fails with ZKP_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef ZKP_PAIRINGS_H
#define ZKP_PAIRINGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkp_ctx zkp_ctx;

typedef enum {
    ZKP_OK = 0,
    ZKP_ERR_ARG = -1,          /* null pointer / bad size */
    ZKP_ERR_NO_DEVICE = -2,    /* no usable HIP device */
    ZKP_ERR_HIP = -3,          /* a HIP runtime call failed; see zkp_last_error */
    ZKP_ERR_NONCANONICAL = -4, /* an input limb array is >= p (validation mode) */
    ZKP_ERR_OOM = -5,
    ZKP_ERR_COMM = -6          /* an RCCL call failed or the context has no communicator; see zkp_last_error */
} zkp_status;

/* zkp_fp_op_batch: the operation in bits 0..3, the limb core in bit 4.  0 and 1 are the op numbers of the zkVM precompile
 * bls12381_sys_bigint (src/fp.rs:376,443); unary operations ignore b (may be NULL). */
typedef enum {
    ZKP_FP_MUL = 0,
    ZKP_FP_ADD = 1,
    ZKP_FP_SUB = 2,
    ZKP_FP_NEG = 3,
    ZKP_FP_SQUARE = 4,
    ZKP_FP_INVERT = 5,      /* 0 gives 0 (the reference returns None, src/fp.rs:307-319) */
    ZKP_FP_CORE28 = 16      /* OR-ed in: run on the 14 x 28-bit carry-free core of the cooperative family instead of
                               the 12 x 32-bit Montgomery core */
} zkp_fp_op;

/* zkp_tower_op_batch: one tower operation per 72-u64 record.  Smaller tower elements occupy the leading coefficients
 * of a record (Fp2: 12 u64, Fp6: 36 u64); the rest of an input must be zero and the rest of the result is zero. */
typedef enum {
    ZKP_TOWER_FP2_MUL = 0,
    ZKP_TOWER_FP2_SQUARE = 1,
    ZKP_TOWER_FP6_MUL = 2,
    ZKP_TOWER_FP6_SQUARE = 3,
    ZKP_TOWER_FP6_FROBENIUS = 4,            /* x^p (NOT what src/fp6.rs:142-176 computes) */
    ZKP_TOWER_FP12_MUL = 5,
    ZKP_TOWER_FP12_SQUARE = 6,
    ZKP_TOWER_FP12_MUL_BY_014 = 7,          /* b = c0 | c1 | c4 (three Fp2) in the first 36 u64 of its record */
    ZKP_TOWER_FP12_FROBENIUS = 8,           /* x^p */
    ZKP_TOWER_FP12_CONJUGATE = 9,
    ZKP_TOWER_FP12_CYCLOTOMIC_SQUARE = 10,  /* Granger-Scott; input in the cyclotomic subgroup */
    ZKP_TOWER_FP12_CYCLOTOMIC_POW2K = 11,   /* g^(2^repeat), 1 <= repeat <= 64: compressed squarings + decompression
                                               (the building block of the final exponentiation's x-power chains) */
    ZKP_TOWER_FP12_CYCLOTOMIC_DECOMPRESS = 12, /* Karabina decompression: c0.c0 and c1.c1 of the result are recomputed from
                                               the other four Fp2 coefficients of the record (cooperative family only) */
    /* the rest of the tower functions SURVEY.md 8(a) names, as direct hooks.  The inversions return the ZERO record for a
     * non-invertible (zero) input, where the reference returns None. */
    ZKP_TOWER_FP2_INVERT = 13,              /* src/fp2.rs:278-296 */
    ZKP_TOWER_FP2_MUL_BY_NONRESIDUE = 14,   /* (c0 - c1) + (c0 + c1) u, src/fp2.rs:161-168 */
    ZKP_TOWER_FP2_MUL_FP = 15,              /* Mul<&Fp> for Fp2, src/fp2.rs:95-102: b = the Fp value in the first 6 u64 of its record */
    ZKP_TOWER_FP6_MUL_BY_1 = 16,            /* src/fp6.rs:102-108: b = c1 (Fp2) in the first 12 u64 */
    ZKP_TOWER_FP6_MUL_BY_01 = 17,           /* src/fp6.rs:110-127: b = c0 | c1 in the first 24 u64 */
    ZKP_TOWER_FP6_MUL_BY_NONRESIDUE = 18,   /* * v, src/fp6.rs:130-141 */
    ZKP_TOWER_FP6_INVERT = 19,              /* src/fp6.rs:291-309 */
    ZKP_TOWER_FP12_INVERT = 20              /* src/fp12.rs:186-190 */
} zkp_tower_op;

/* which Miller-loop/final-exp kernel family the context uses */
typedef enum {
    ZKP_KERNEL_AUTO = 0,
    ZKP_KERNEL_THREAD = 1,  /* one pairing per lane, 12x32-bit Montgomery limbs */
    ZKP_KERNEL_COOP = 2     /* lane-group-cooperative, LDS-staged tower (hot path) */
} zkp_kernel_kind;

int zkp_abi_version(void);
const char* zkp_strerror(int status);

/* device = HIP device ordinal of this process (LOCAL_RANK under torchrun). */
int zkp_init(int device, zkp_ctx** out_ctx);
void zkp_free(zkp_ctx* ctx);
const char* zkp_last_error(const zkp_ctx* ctx);
int zkp_set_validate(zkp_ctx* ctx, int on);
int zkp_set_kernel(zkp_ctx* ctx, int kind /* zkp_kernel_kind */);
int zkp_device_info(const zkp_ctx* ctx, int* cus, int* clock_khz, char* name, size_t name_len);

/* Gt::identity(): 72 u64, c0.c0.c0 = 1, rest 0. */
const uint64_t* zkp_gt_identity(void);

/* ---- host-pointer entry points (H2D copy, kernels, D2H copy, synchronous) ---------------------- */
/* out_gt[i] = pairing(g1[i], g2[i]); inf1/inf2 may be NULL (= no infinities). */
int zkp_pairing_batch(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                      size_t n, uint64_t* out_gt);
/* n_checks groups of k consecutive pairs; out_ml[c] = multi_miller_loop(group c) (72 u64 each). */
int zkp_multi_miller_loop_batch(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                                const uint8_t* inf2, size_t n_checks, size_t k, uint64_t* out_ml);
int zkp_final_exponentiation_batch(zkp_ctx* ctx, const uint64_t* f, size_t n, uint64_t* out_gt);
/* ok[c] = (final_exponentiation(multi_miller_loop(group c)) == Gt::identity()); ok may be NULL.
 * *all_ok = AND over this call's checks (the cross-GPU AND is the caller's single RCCL all-reduce). */
int zkp_pairing_check_batch(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                            const uint8_t* inf2, size_t n_checks, size_t k, uint8_t* ok, int* all_ok);
/* ---- the whole batch as ONE product check (SURVEY.md 8e variant; BLS batch verification shape) ----
 * multi_miller_loop(&[(P_0,Q_0) .. (P_n-1,Q_n-1)]) over ALL n pairs -> one Fp12 (72 u64); n == 0 gives one.
 * A multi-GPU host multiplies the per-GPU values (all-gather of 576 B per rank, zkp_fp12_product) and runs
 * ONE zkp_final_exponentiation_batch(n = 1). */
int zkp_miller_product(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                       size_t n, uint64_t* out_ml);
/* out = f_0 * f_1 * ... * f_n-1 (Fp12 Mul, reference src/fp12.rs:193-210, folded by a log-depth tree); n == 0 gives one. */
int zkp_fp12_product(zkp_ctx* ctx, const uint64_t* f, size_t n, uint64_t* out);
/* *is_one = (final_exponentiation(zkp_miller_product(..)) == Gt::identity()); out_gt (72 u64) may be NULL. */
int zkp_pairing_product_check(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                              const uint8_t* inf2, size_t n, uint64_t* out_gt, int* is_one);
/* status[i]: 0 valid (or infinity), 1 not on curve, 2 not torsion free. */
int zkp_g1_is_valid_batch(zkp_ctx* ctx, const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* status);
int zkp_g2_is_valid_batch(zkp_ctx* ctx, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* status);
/* out[i] = [k_i] p_i (affine); base_stride 0 broadcasts one base point. out_inf may be NULL. */
int zkp_g1_mul_batch(zkp_ctx* ctx, const uint64_t* base, size_t base_stride, const uint64_t* scalars, size_t n,
                     uint64_t* out, uint8_t* out_inf);
int zkp_g2_mul_batch(zkp_ctx* ctx, const uint64_t* base, size_t base_stride, const uint64_t* scalars, size_t n,
                     uint64_t* out, uint8_t* out_inf);
/* ---- uncompressed point byte codec (big-endian field elements, reference src/fp.rs:165-207 with the range
 * check done CORRECTLY - upstream's Fp::from_bytes accepts exactly the non-canonical values).
 * G1: x(48) | y(48) = 96 bytes; G2: x.c1 | x.c0 | y.c1 | y.c0 = 192 bytes (c1 first, the usual BLS12-381
 * serialisation order).  The top three bits of byte 0 are flags: 0x80 compressed (rejected), 0x40 infinity
 * (all other bytes must be zero), 0x20 must be clear.
 * status[i]: 0 ok, 1 coordinate >= p, 2 bad flags / malformed infinity.  Decoded points go out in wire
 * format (+ infinity byte) and can be fed to zkp_g*_is_valid_batch / zkp_pairing_*. */
int zkp_g1_decode_batch(zkp_ctx* ctx, const uint8_t* bytes, size_t n, uint64_t* out_g1, uint8_t* out_inf, uint8_t* status);
int zkp_g2_decode_batch(zkp_ctx* ctx, const uint8_t* bytes, size_t n, uint64_t* out_g2, uint8_t* out_inf, uint8_t* status);
int zkp_g1_encode_batch(zkp_ctx* ctx, const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* out_bytes);
int zkp_g2_encode_batch(zkp_ctx* ctx, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* out_bytes);

/* batched field op in the zkVM precompile shape (op: zkp_fp_op; 0 = mul and 1 = add as in src/fp.rs:376,443) */
int zkp_fp_op_batch(zkp_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* batched tower operation (op: zkp_tower_op) on the context's kernel family; a, b, out: n records of 72 u64; b is read
 * only by the binary operations; repeat only by ZKP_TOWER_FP12_CYCLOTOMIC_POW2K. */
int zkp_tower_op_batch(zkp_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint32_t repeat, uint64_t* out);

/* ---- device-pointer entry points (buffers already resident in HBM; asynchronous on `stream`) ---- */
/* `stream` is a hipStream_t passed as void* (NULL = default stream).  Same formats as above. */
/* hipGraph capture (round 6): a `_dev` call on a stream that is being captured enqueues kernel / memset nodes only - no allocation once
 * the context's workspaces have reached the call's size (run the call once outside the capture first), no host synchronisation, and
 * the internal pipeline streams join the capture through their fork / join events.  The hand-over of the context's workspace between
 * calls on DIFFERENT streams (an event per context) is skipped under capture: order a replayed graph against the context's other
 * calls yourself.  Measured (profiles/r06/v58_trace_n1.txt): launch gaps are 0.7 % of a single pairing's 3.8 ms - the kernels are
 * bound by one wavefront's dependent instruction chain - and a captured call replays in 3.50 ms against 3.54 ms plain (bench line,
 * batch_sweep.rows[].hipgraph): capture is supported, the library itself keeps no graph cache. */
int zkp_pairing_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1, const void* d_inf2,
                          size_t n, void* d_out_gt, void* stream);
int zkp_multi_miller_loop_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1,
                                    const void* d_inf2, size_t n_checks, size_t k, void* d_out_ml, void* stream);
int zkp_final_exponentiation_batch_dev(zkp_ctx* ctx, const void* d_f, size_t n, void* d_out_gt, void* stream);
/* d_ok: n_checks bytes (may be NULL); d_all_ok: one int32 written with the AND (may be NULL). */
int zkp_pairing_check_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1,
                                const void* d_inf2, size_t n_checks, size_t k, void* d_ok, void* d_all_ok,
                                void* stream);
/* pairing() and the Gt::identity() check in one pass: Gt out (may be NULL) + ok bytes + AND flag. */
int zkp_pairing_gt_check_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1,
                                   const void* d_inf2, size_t n_checks, size_t k, void* d_out_gt, void* d_ok,
                                   void* d_all_ok, void* stream);
/* device flavours of the one-product-check entry points; d_out_ml / d_out: 72 u64; d_out_gt (may be NULL): 72 u64;
 * d_is_one (may be NULL): one int32. */
int zkp_miller_product_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1, const void* d_inf2,
                           size_t n, void* d_out_ml, void* stream);
int zkp_fp12_product_dev(zkp_ctx* ctx, const void* d_f, size_t n, void* d_out, void* stream);
int zkp_pairing_product_check_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1,
                                  const void* d_inf2, size_t n, void* d_out_gt, void* d_is_one, void* stream);
int zkp_g1_is_valid_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_inf, size_t n, void* d_status, void* stream);
int zkp_g2_is_valid_batch_dev(zkp_ctx* ctx, const void* d_g2, const void* d_inf, size_t n, void* d_status, void* stream);
int zkp_g1_mul_batch_dev(zkp_ctx* ctx, const void* d_base, size_t base_stride, const void* d_scalars, size_t n,
                         void* d_out, void* d_out_inf, void* stream);
int zkp_g2_mul_batch_dev(zkp_ctx* ctx, const void* d_base, size_t base_stride, const void* d_scalars, size_t n,
                         void* d_out, void* d_out_inf, void* stream);

/* the uncompressed point codec on resident buffers (same formats and status bytes as zkp_g1/g2_decode_batch above) */
int zkp_g1_decode_batch_dev(zkp_ctx* ctx, const void* d_bytes, size_t n, void* d_out_g1, void* d_out_inf, void* d_status, void* stream);
int zkp_g2_decode_batch_dev(zkp_ctx* ctx, const void* d_bytes, size_t n, void* d_out_g2, void* d_out_inf, void* d_status, void* stream);
int zkp_g1_encode_batch_dev(zkp_ctx* ctx, const void* d_g1, const void* d_inf /* may be NULL */, size_t n, void* d_out_bytes, void* stream);
int zkp_g2_encode_batch_dev(zkp_ctx* ctx, const void* d_g2, const void* d_inf /* may be NULL */, size_t n, void* d_out_bytes, void* stream);

/* ---- BASELINE config 5 as ONE call: raw uncompressed points -> decode -> is_valid -> pairing check ------------------------------
 * n_checks groups of k consecutive (G1, G2) pairs given as byte strings (96 B per G1 point, 192 B per G2 point, the codec's format).
 * Every point is decoded (Fp::from_bytes with the correct range check, src/fp.rs:165-207) and validated (G1Affine::is_valid
 * src/g1.rs:49-62, G2Affine::is_valid src/g2.rs:57-69); st1[i] / st2[i] (n_checks * k bytes each, may be NULL) receive a
 * zkp_point_status.  ok[c] = (every point of check c is valid AND prod_j e(P_cj, Q_cj) == Gt::identity()); *all_ok = AND over the
 * checks (each may be NULL).  A point that is not valid takes no part in the Miller loop (no arithmetic on garbage); its check
 * fails.  Decoded points, infinity flags and intermediate status bytes stay in the context's workspace in HBM: the host flavour
 * moves only the byte strings up and the status / ok bytes down.
 * Validate first, then use (ABI version 4): only the checks whose 2k points are all valid enter the Miller loop and the final
 * exponentiation - they are listed and gathered on the device, so a check with an invalid point costs its validity tests and nothing
 * more.  The NUMBER of listed checks never leaves the device (round 6): the pairing phase is launched for the worst case and its
 * kernels read the count themselves, so the `_dev` flavour is asynchronous on `stream` like every other `_dev` call (no host
 * read-back, capturable into a hipGraph once the context's workspaces have reached their size: the first call of a size allocates). */
typedef enum {
    ZKP_POINT_OK = 0,               /* a point of the r-torsion subgroup, or a well-formed infinity */
    ZKP_POINT_NONCANONICAL = 1,     /* a coordinate >= p */
    ZKP_POINT_MALFORMED = 2,        /* compressed / sort flag set, or an infinity flag over non-zero bytes */
    ZKP_POINT_NOT_ON_CURVE = 3,     /* "Point is not on curve", src/g1.rs:54-56 */
    ZKP_POINT_NOT_IN_SUBGROUP = 4   /* "Point is not torsion free", src/g1.rs:57-59 */
} zkp_point_status;
int zkp_points_check_batch(zkp_ctx* ctx, const uint8_t* g1_bytes, const uint8_t* g2_bytes, size_t n_checks, size_t k, uint8_t* st1,
                           uint8_t* st2, uint8_t* ok, int* all_ok);
int zkp_points_check_batch_dev(zkp_ctx* ctx, const void* d_g1_bytes, const void* d_g2_bytes, size_t n_checks, size_t k, void* d_st1,
                               void* d_st2, void* d_ok, void* d_all_ok, void* stream);

/* validation mode (zkp_set_validate) on the device-pointer entry points: the range check of the inputs runs on the caller's
 * stream and ORs into a word inside the context - no host synchronisation in the *_dev call itself.  This call waits for
 * `stream`, reports whether any *_dev call since the last query saw a field element >= p (*bad = 1; the results of such a
 * call are unspecified) and clears the word. */
int zkp_take_validation_status_dev(zkp_ctx* ctx, void* stream, int* bad);

/* ---- several GPUs behind one call ------------------------------------------------------------------
 * ONE host thread, n_ctx distinct contexts (normally one per GPU of the node, each from zkp_init(device)); the checks
 * are split into contiguous blocks, one per context (a check's k pairs and its shared final exponentiation stay
 * on one GPU), every context uploads, computes and downloads its block on its own stream concurrently, and the
 * per-context AND flags are combined on the host - the data path has no collective.  Host pointers, synchronous,
 * same formats and results as zkp_pairing_check_batch / zkp_pairing_batch + the Gt::identity() flags.
 * The one-process-per-GPU form (torchrun / MPI ranks) is zkp_pairing_check_batch per rank followed by ONE
 * ncclAllReduce(&flag, &flag, 1, ncclInt32, ncclMin, comm, stream): RCCL has no bitwise AND, MIN of {0,1} is AND. */
int zkp_pairing_check_batch_multi(zkp_ctx* const* ctxs, int n_ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                                  const uint8_t* inf2, size_t n_checks, size_t k, uint8_t* ok /* may be NULL */, int* all_ok);
/* out_gt[i] = pairing(g1[i], g2[i]); ok[i] = (out_gt[i] == Gt::identity()) and *all_ok, each optional */
int zkp_pairing_batch_multi(zkp_ctx* const* ctxs, int n_ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                            const uint8_t* inf2, size_t n, uint64_t* out_gt, uint8_t* ok, int* all_ok);

/* ---- one process (rank) per GPU: the path's ONE collective behind the C ABI ------------------------------------
 * librccl is linked into libzkp_pairings.so; a Rust / C host needs nothing outside this header to finish a sharded check:
 *     rank 0: zkp_comm_unique_id(id)            -> hand the 128 bytes to every rank by the host's own means (file, socket, MPI)
 *     every rank: zkp_comm_init_rank(ctx, nranks, rank, id)        (collective: returns when all ranks have joined)
 *     every rank: zkp_pairing_check_batch_allreduce(ctx, <its contiguous block of the checks>, ok, &all_ok)
 * = zkp_pairing_check_batch on the rank's block, then ONE ncclAllReduce(count = 1, ncclInt32, ncclMin) of the AND flag on the
 * context's stream (RCCL has no bitwise AND; MIN of {0,1} is AND): *all_ok is the AND over ALL ranks' checks, ok[] stays
 * per-rank.  Nothing else crosses xGMI.  A context holds at most one communicator; zkp_free destroys it.
 * Failure on one rank: a rank whose OWN block fails (bad arguments, non-canonical limbs under zkp_set_validate, a HIP error) still
 * takes part in the collective - with flag 0 (AND-reduce) or the zero record (all-gather variant) - so no peer is left waiting and
 * every rank reads all_ok / is_one = 0; the failing rank returns its own status afterwards.  Only a call without context,
 * output flag or communicator returns before the collective (ZKP_ERR_ARG / ZKP_ERR_COMM).
 * zkp_comm_init_rank itself is a collective: every rank must call it (a rank that cannot - no context, no device - makes the others
 * wait, as with any bootstrap).  Its local failures come AFTER the bootstrap (round 6): a rank without memory for the communicator's
 * records aborts its side and returns ZKP_ERR_OOM; the peers cannot see that, so agree on the ranks' statuses over the host's own
 * channel before the first collective (bench.py does: one gloo all-reduce(MIN) of "init ok").  No reference
 * counterpart (the reference is a single-threaded host crate); the torch.distributed flavour of the same step is
 * zkvm_pairings_amd/dist.py. */
#define ZKP_COMM_ID_BYTES 128
int zkp_comm_unique_id(void* out_id /* ZKP_COMM_ID_BYTES */);
int zkp_comm_init_rank(zkp_ctx* ctx, int nranks, int rank, const void* unique_id /* ZKP_COMM_ID_BYTES */);
int zkp_comm_destroy(zkp_ctx* ctx);
/* *nranks = 0 when the context has no communicator */
int zkp_comm_info(const zkp_ctx* ctx, int* nranks, int* rank);
/* in-place AND of one int32 {0,1} flag in device memory over the ranks, asynchronous on `stream` (the building block) */
int zkp_and_allreduce_dev(zkp_ctx* ctx, void* d_flag, void* stream);
int zkp_pairing_check_batch_allreduce(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                      size_t n_checks, size_t k, uint8_t* ok /* may be NULL */, int* all_ok);
/* d_all_ok (one int32, required) receives the AND over all ranks; asynchronous on `stream` */
int zkp_pairing_check_batch_allreduce_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1, const void* d_inf2,
                                          size_t n_checks, size_t k, void* d_ok, void* d_all_ok, void* stream);
/* pairing() + the Gt::identity() check of this rank's block (zkp_pairing_gt_check_batch_dev: Gt out, may be NULL, + ok bytes) and the
 * same ONE all-reduce of the AND flag - BASELINE config 3 ("2^20 pairings sharded across 8 GPUs with RCCL AND-reduce of Gt==identity")
 * as one call per rank; it is the step `bench.py --collective abi` times.  d_all_ok is required.  (ABI version 4) */
int zkp_pairing_gt_check_batch_allreduce_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, const void* d_inf1, const void* d_inf2,
                                             size_t n_checks, size_t k, void* d_out_gt, void* d_ok, void* d_all_ok, void* stream);
/* BASELINE config 5 on a node ("subgroup check + pairing on raw points, 8 GPUs"): this rank's block of zkp_points_check_batch, then the
 * same ONE all-reduce of the AND flag; status and ok bytes stay per rank.  d_all_ok / all_ok are required. */
int zkp_points_check_batch_allreduce(zkp_ctx* ctx, const uint8_t* g1_bytes, const uint8_t* g2_bytes, size_t n_checks, size_t k, uint8_t* st1,
                                     uint8_t* st2, uint8_t* ok, int* all_ok);
int zkp_points_check_batch_allreduce_dev(zkp_ctx* ctx, const void* d_g1_bytes, const void* d_g2_bytes, size_t n_checks, size_t k,
                                         void* d_st1, void* d_st2, void* d_ok, void* d_all_ok, void* stream);
/* SURVEY.md 8e variant, ONE product check over the whole sharded batch: this rank's zkp_miller_product, ONE ncclAllGather of
 * 576 B per rank, the product of the gathered values and ONE final exponentiation on every rank (identical results):
 * *is_one = (prod over ALL ranks' pairs of e(P_i, Q_i) == Gt::identity()); out_gt (72 u64) may be NULL. */
int zkp_pairing_product_check_allgather(zkp_ctx* ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                        size_t n, uint64_t* out_gt, int* is_one);

/* ---- pinned host memory for the host-pointer entry points ("H2D/D2H per GPU straight from pinned host
 * memory on its own stream") ------------------------------------------------------------------------------------------
 * The host-pointer entry points copy with hipMemcpyAsync on their own streams.  From PAGEABLE memory such a copy stages
 * through a driver buffer and blocks the calling thread, so the next slice's upload (zkp_pairing_batch) or the next
 * context's upload (zkp_pairing_*_multi) cannot start before it is over; from page-locked memory it is a DMA that
 * overlaps the kernels and the other contexts' copies.  zkp_host_alloc returns page-locked memory usable with every
 * context / GPU of the process (hipHostMallocPortable); zkp_host_register page-locks an existing allocation in place
 * (hipHostRegister: slow, of the order of 1 ms per 4 MB - do it once for a buffer that is reused, not per call).  The
 * registered range must consist of WHOLE PAGES of its own: ptr and bytes multiples of the page size (sysconf(_SC_PAGESIZE),
 * 4096 on the GPU boxes; zkp_host_unregister(NULL) is ZKP_ERR_ARG), i.e. memory
 * from posix_memalign / aligned_alloc / mmap, not a slice of the malloc heap - the driver locks and maps pages, and two
 * registrations that share a page (two neighbouring heap arrays) left the HIP runtime with a stale entry after the
 * unregistration: a later copy from a reused heap address then went to the GPU as a DMA from an unmapped page (a GPU
 * memory fault; found in round 3).  Anything else is ZKP_ERR_ARG.  The
 * entry points themselves accept either kind of memory and give the same results.  No reference counterpart: the
 * reference is a host-only crate (its point arrays would live in a Vec; the Rust wrapper's PinnedVec in
 * integration/rust/src/lib.rs is the drop-in container). */
int zkp_host_alloc(size_t bytes, void** out_ptr);
int zkp_host_free(void* ptr);
int zkp_host_register(void* ptr, size_t bytes);
int zkp_host_unregister(void* ptr);

/* ---- measurement helper used by bench.py: times `reps` launches of the fused pairing kernel on
 * ctx's own stream with HIP events recorded on THAT stream; returns average ms per launch. ---- */
int zkp_time_pairing_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, size_t n, void* d_out_gt, int reps,
                         float* avg_ms);

/* measurement helper used by bench.py: a one-wavefront clock probe queued on `stream`; d_out (device, two u64) receives
 * the shader-clock ticks and the wall-clock ticks of about spin_us microseconds, *wall_khz the wall clock's rate.
 * Queued on a second stream beside a running pass it reports the clock the chip sustains under that load. */
int zkp_clock_probe_dev(zkp_ctx* ctx, void* stream, unsigned spin_us, void* d_out, int* wall_khz);

/* measurement: ONE pass of the fused pairing over n resident pairs with every kernel launch bracketed by HIP events on its stream
 * (a single pipeline, so that no two kernels overlap): ms[c] = summed duration and launches[c] = number of the launches of class c.
 * Cooperative family only.  bench.py prints each class against the multiply-adds it executes (roofline.kernels). */
#define ZKP_PROFILE_CLASSES 9
/* 0 k_prep_lines, 1 k_coop<30,4> (Miller program), 2 k_coop<24,34> fexp_a, 3 k_batch_inv, 4 k_ksq, 5 k_kdec_a, 6 k_kdec_b,
 * 7 k_coop<36,24> (hard-part step programs), 8 k_coop<24,34> (the other step programs of phase C) */
int zkp_profile_pairing_dev(zkp_ctx* ctx, const void* d_g1, const void* d_g2, size_t n, void* d_out_gt,
                            float* ms /* ZKP_PROFILE_CLASSES */, int* launches /* ZKP_PROFILE_CLASSES */);

/* diagnostic: time one synthetic step program of the cooperative interpreter (which: 0 T=1, 1 T=3,
 * 2 T=3+epilogue, 3 T=6, 4 T=12, 5 LIN, 6 / 7 cyclotomic squaring without / with companion slots, 8 spill + fill;
 * 9: 400 compressed squarings of k_ksq; 10: the line precomputation k_prep_lines of n pairs; 11: the one-pair Miller
 * program over the line stream that 10 left behind - n at most one chunk, 65536 by default; 15: like 10 with the upstream-shaped
 * lines of multi_miller_loop(), k_prep_lines<false>) over n checks, on ctx's own
 * stream with its own events; used by tools/ and by bench.py's per-kernel roofline split. */
int zkp_time_coop_step(zkp_ctx* ctx, int which, size_t n, float* ms);

#ifdef __cplusplus
}
#endif
#endif
