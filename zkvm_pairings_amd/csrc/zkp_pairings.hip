// zkp_pairings.hip -- libzkp_pairings.so: C ABI (include/zkp_pairings.h) + the one-element-per-lane
// kernel family for gfx950.  The lane-cooperative hot-path kernels live in zkp_coop.hip and are
// dispatched from here when the context's kernel kind selects them.
//
// There is no CPU fallback anywhere in this file: every entry point either runs HIP kernels or
// returns a negative status.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <new>
#include <string>

#include "../../include/zkp_pairings.h"
#include "zkp_field.hpp"
#include "zkp_coop.hpp"
#include "zkp_plan.hpp"

using namespace zkp;

// =============================================================================== kernels (thread family)
namespace {

constexpr int TPB = 64;   // threads per block for the scratch-heavy tower kernels
constexpr int KMAX = 4;   // pairs per shared-squaring Miller loop; larger k is split and multiplied

__device__ inline bool pair_live(const uint8_t* inf1, const uint8_t* inf2, size_t i) {
    return !((inf1 && inf1[i]) || (inf2 && inf2[i]));
}

// Miller loop over pairs [base, base+k) (k <= KMAX) with shared squarings; result NOT conjugated.
__device__ __attribute__((noinline)) void miller_group(Fp12* f, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                                                        const uint8_t* inf2, size_t base, int k) {
    G1A ps[KMAX];
    G2A qs[KMAX];
    G2P rs[KMAX];
    bool live[KMAX];
    for (int j = 0; j < k; j++) {
        live[j] = pair_live(inf1, inf2, base + j);
        fp_load(&ps[j].x, g1 + 12 * (base + j));
        fp_load(&ps[j].y, g1 + 12 * (base + j) + 6);
        fp2_load(&qs[j].x, g2 + 24 * (base + j));
        fp2_load(&qs[j].y, g2 + 24 * (base + j) + 12);
        rs[j].x = qs[j].x;
        rs[j].y = qs[j].y;
        fp2_one(&rs[j].z);
    }
    fp12_one(f);
    Line l;
    bool found_one = false;
    for (int b = 63; b >= 0; b--) {
        bool bit = ((ZKP_BLS_X >> 1) >> b) & 1;
        if (!found_one) { found_one = bit; continue; }
        for (int j = 0; j < k; j++)
            if (live[j]) { doubling_step(&l, &rs[j]); ell(f, &l, &ps[j]); }
        if (bit)
            for (int j = 0; j < k; j++)
                if (live[j]) { addition_step(&l, &rs[j], &qs[j]); ell(f, &l, &ps[j]); }
        fp12_sqr(f, f);
    }
    for (int j = 0; j < k; j++)
        if (live[j]) { doubling_step(&l, &rs[j]); ell(f, &l, &ps[j]); }
}

// multi_miller_loop of one check of k pairs (any k): product of <=KMAX-pair groups, then conjugate
__device__ __attribute__((noinline)) void miller_check(Fp12* f, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1,
                                                        const uint8_t* inf2, size_t base, size_t k) {
    fp12_one(f);
    bool first = true;
    for (size_t off = 0; off < k; off += KMAX) {
        int kk = (int)((k - off) < (size_t)KMAX ? (k - off) : (size_t)KMAX);
        if (first) { miller_group(f, g1, g2, inf1, inf2, base + off, kk); first = false; }
        else { Fp12 g; miller_group(&g, g1, g2, inf1, inf2, base + off, kk); fp12_mul(f, f, &g); }
    }
    fp12_conj(f, f);  // x < 0
}

__global__ void __launch_bounds__(TPB) k_miller(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                                 size_t n_checks, size_t k, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_checks) return;
    Fp12 f;
    miller_check(&f, g1, g2, inf1, inf2, i * k, k);
    fp12_store(out + 72 * i, &f);
}

__global__ void __launch_bounds__(TPB) k_final_exp(const uint64_t* in, size_t n, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fp12 f, r;
    fp12_load(&f, in + 72 * i);
    final_exponentiation(&r, &f);
    fp12_store(out + 72 * i, &r);
}

// fused: out[i] = final_exp(multi_miller_loop(check i)); optional ok byte + AND flag
__global__ void __launch_bounds__(TPB) k_pairing(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                                  size_t n_checks, size_t k, uint64_t* out_gt, uint8_t* ok, int* all_ok) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_checks) return;
    Fp12 f, r;
    miller_check(&f, g1, g2, inf1, inf2, i * k, k);
    final_exponentiation(&r, &f);
    if (out_gt) fp12_store(out_gt + 72 * i, &r);
    if (ok || all_ok) {
        Fp12 one;
        fp12_one(&one);
        bool is_one = fp12_eq(&r, &one);
        if (ok) ok[i] = is_one ? 1 : 0;
        if (all_ok && !is_one) atomicAnd(all_ok, 0);
    }
}

__global__ void k_set_int(int* p, int v) { *p = v; }

// one wavefront that watches the shader clock counter (s_memtime) against the constant-rate wall clock (s_memrealtime)
// for about spin_us microseconds: out[0] = shader clock ticks, out[1] = wall clock ticks.  Launched beside a running
// pass it reports the clock the chip sustains under that load (bench.py: roofline.frac_at_sustained_clock).
__global__ void k_clock_probe(unsigned long long* out, unsigned long long wall_ticks) {
    if (threadIdx.x) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(32);
        w1 = wall_clock64();
    }
    out[0] = clock64() - c0;
    out[1] = w1 - w0;
}

// one level of the Fp12 product tree, in place on wire records: buf[c] <- buf[c] * buf[c + h], c < m
__global__ void __launch_bounds__(TPB) k_fp12_mul_pairs(uint64_t* buf, size_t m, size_t h) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= m) return;
    Fp12 a, b;
    fp12_load(&a, buf + 72 * i);
    fp12_load(&b, buf + 72 * (i + h));
    fp12_mul(&a, &a, &b);
    fp12_store(buf + 72 * i, &a);
}

// *is_one = (gt == Gt::identity()) on a canonical wire record
__global__ void k_gt_is_one(const uint64_t* gt, int* is_one) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t d = gt[0] ^ 1ull;
    for (int i = 1; i < 72; i++) d |= gt[i];
    *is_one = d == 0 ? 1 : 0;
}

// reference src/g1.rs:49-62: 0 ok / 1 not on curve / 2 not torsion free
__global__ void __launch_bounds__(TPB) k_g1_valid(const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* status) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    if (inf && inf[i]) { status[i] = 0; return; }
    Fp x, y, t, u, b;
    fp_load(&x, g1 + 12 * i);
    fp_load(&y, g1 + 12 * i + 6);
    fp_sqr(&t, &y);
    fp_sqr(&u, &x);
    fp_mul(&u, &u, &x);
    fp_set(&b, K_M_B);
    fp_add(&u, &u, &b);
    if (!fp_eq(&t, &u)) { status[i] = 1; return; }
    // -[X][X]P == (beta x, y)  <=>  [X][X]P == (beta x, -y)
    uint64_t kx = ZKP_BLS_X;
    Jac<FpOps> q;
    jac_mul<FpOps>(&q, &x, &y, &kx, 1);
    Fp ax, ay;
    bool fin = jac_to_affine<FpOps>(&ax, &ay, &q);
    bool ok = false;
    if (fin) {
        jac_mul<FpOps>(&q, &ax, &ay, &kx, 1);
        Fp bx, ny, beta;
        fp_set(&beta, K_M_BETA);
        fp_mul(&bx, &x, &beta);
        fp_neg(&ny, &y);
        ok = jac_eq_affine<FpOps>(&q, &bx, &ny);
    }
    status[i] = ok ? 0 : 2;
}

// reference src/g2.rs:57-69: psi(P) == -[X]P
__global__ void __launch_bounds__(TPB) k_g2_valid(const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* status) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    if (inf && inf[i]) { status[i] = 0; return; }
    Fp2 x, y, t, u, b;
    fp2_load(&x, g2 + 24 * i);
    fp2_load(&y, g2 + 24 * i + 12);
    fp2_sqr(&t, &y);
    fp2_sqr(&u, &x);
    fp2_mul(&u, &u, &x);
    fp_set(&b.c0, K_M_B);
    fp_set(&b.c1, K_M_B);
    fp2_add(&u, &u, &b);
    if (!fp2_eq(&t, &u)) { status[i] = 1; return; }
    uint64_t kx = ZKP_BLS_X;
    Jac<Fp2Ops> q;
    jac_mul<Fp2Ops>(&q, &x, &y, &kx, 1);
    // psi(P) = (conj(x) * PSI_X, conj(y) * PSI_Y); compare with -[X]P
    Fp2 px, py, k;
    fp2_conj(&px, &x);
    fp2_const(&k, K_M_PSI_X_0, K_M_PSI_X_1);
    fp2_mul(&px, &px, &k);
    fp2_conj(&py, &y);
    fp2_const(&k, K_M_PSI_Y_0, K_M_PSI_Y_1);
    fp2_mul(&py, &py, &k);
    fp2_neg(&py, &py);
    status[i] = jac_eq_affine<Fp2Ops>(&q, &px, &py) ? 0 : 2;
}

__global__ void __launch_bounds__(TPB) k_g1_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fp x, y;
    fp_load(&x, base + stride * i);
    fp_load(&y, base + stride * i + 6);
    uint64_t k[4] = {sc[4 * i], sc[4 * i + 1], sc[4 * i + 2], sc[4 * i + 3]};
    Jac<FpOps> q;
    jac_mul<FpOps>(&q, &x, &y, k, 4);
    Fp ax, ay;
    bool fin = jac_to_affine<FpOps>(&ax, &ay, &q);
    fp_store(out + 12 * i, &ax);
    fp_store(out + 12 * i + 6, &ay);
    if (out_inf) out_inf[i] = fin ? 0 : 1;
}
__global__ void __launch_bounds__(TPB) k_g2_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fp2 x, y;
    fp2_load(&x, base + stride * i);
    fp2_load(&y, base + stride * i + 12);
    uint64_t k[4] = {sc[4 * i], sc[4 * i + 1], sc[4 * i + 2], sc[4 * i + 3]};
    Jac<Fp2Ops> q;
    jac_mul<Fp2Ops>(&q, &x, &y, k, 4);
    Fp2 ax, ay;
    bool fin = jac_to_affine<Fp2Ops>(&ax, &ay, &q);
    fp2_store(out + 24 * i, &ax);
    fp2_store(out + 24 * i + 12, &ay);
    if (out_inf) out_inf[i] = fin ? 0 : 1;
}

// batched field operation on canonical operands: 0 mul, 1 add (the zkVM precompile's two ops, src/fp.rs:376,443),
// 2 sub, 3 neg, 4 square, 5 invert (0 gives 0; the reference returns None, src/fp.rs:307-319)
__global__ void k_fp_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x, y, r;
    fp_load(&x, a + 6 * i);
    if (op <= 2) fp_load(&y, b + 6 * i);
    switch (op) {
        case 0: fp_mul(&r, &x, &y); break;
        case 1: fp_add(&r, &x, &y); break;
        case 2: fp_sub(&r, &x, &y); break;
        case 3: fp_neg(&r, &x); break;
        case 4: fp_sqr(&r, &x); break;
        default: if (!fp_inv(&r, &x)) fp_zero(&r); break;
    }
    fp_store(out + 6 * i, &r);
}

// one tower operation per 72-u64 record (zkp_tower_op_batch, thread family): smaller tower elements occupy the leading
// coefficients of a record, the rest of the result is zero
__global__ void __launch_bounds__(TPB) k_tower_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint32_t repeat, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fp12 x, y, r;
    fp12_load(&x, a + 72 * i);
    if (b) fp12_load(&y, b + 72 * i);
    fp6_zero(&r.c0);
    fp6_zero(&r.c1);
    switch (op) {
        case ZKP_TOWER_FP2_MUL: fp2_mul(&r.c0.c0, &x.c0.c0, &y.c0.c0); break;
        case ZKP_TOWER_FP2_SQUARE: fp2_sqr(&r.c0.c0, &x.c0.c0); break;
        case ZKP_TOWER_FP6_MUL: fp6_mul(&r.c0, &x.c0, &y.c0); break;
        case ZKP_TOWER_FP6_SQUARE: fp6_sqr(&r.c0, &x.c0); break;
        case ZKP_TOWER_FP6_FROBENIUS: fp6_frob(&r.c0, &x.c0); break;
        case ZKP_TOWER_FP12_MUL: fp12_mul(&r, &x, &y); break;
        case ZKP_TOWER_FP12_SQUARE: fp12_sqr(&r, &x); break;
        case ZKP_TOWER_FP12_MUL_BY_014: fp12_mul_by_014(&r, &x, &y.c0.c0, &y.c0.c1, &y.c0.c2); break;
        case ZKP_TOWER_FP12_FROBENIUS: fp12_frob(&r, &x); break;
        case ZKP_TOWER_FP12_CONJUGATE: fp12_conj(&r, &x); break;
        case ZKP_TOWER_FP12_CYCLOTOMIC_SQUARE: fp12_cyclotomic_square(&r, &x); break;
        // a non-invertible input leaves the zero record (the reference returns None)
        case ZKP_TOWER_FP2_INVERT: if (!fp2_inv(&r.c0.c0, &x.c0.c0)) fp2_zero(&r.c0.c0); break;
        case ZKP_TOWER_FP2_MUL_BY_NONRESIDUE: fp2_mul_nr(&r.c0.c0, &x.c0.c0); break;
        case ZKP_TOWER_FP2_MUL_FP: fp2_mul_fp(&r.c0.c0, &x.c0.c0, &y.c0.c0.c0); break;
        case ZKP_TOWER_FP6_MUL_BY_1: fp6_mul_by_1(&r.c0, &x.c0, &y.c0.c0); break;
        case ZKP_TOWER_FP6_MUL_BY_01: fp6_mul_by_01(&r.c0, &x.c0, &y.c0.c0, &y.c0.c1); break;
        case ZKP_TOWER_FP6_MUL_BY_NONRESIDUE: fp6_mul_nr(&r.c0, &x.c0); break;
        case ZKP_TOWER_FP6_INVERT: if (!fp6_inv(&r.c0, &x.c0)) fp6_zero(&r.c0); break;
        case ZKP_TOWER_FP12_INVERT: if (!fp12_inv(&r, &x)) { fp6_zero(&r.c0); fp6_zero(&r.c1); } break;
        default:
            r = x;
            for (uint32_t k = 0; k < repeat; k++) { Fp12 t; fp12_cyclotomic_square(&t, &r); r = t; }
            break;
    }
    fp12_store(out + 72 * i, &r);
}

// ---- uncompressed byte codec: nfp big-endian 48-byte field elements per point (2 for G1, 4 for G2).
// G2 stores c1 before c0, so element e of the byte string is wire element (e ^ 1) when nfp == 4.
// ALIGNED: the byte strings start on an 8-byte boundary (96 and 192 are multiples of 8): a big-endian word is one 64-bit load + a
// byte swap instead of eight byte loads.
template <bool ALIGNED>
__device__ __forceinline__ uint64_t be64_load(const uint8_t* p) {
    if (ALIGNED) return __builtin_bswap64(*reinterpret_cast<const uint64_t*>(p));
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) v = (v << 8) | p[b];
    return v;
}
template <bool ALIGNED>
__global__ void k_decode(const uint8_t* bytes, size_t n, int nfp, uint64_t* out, uint8_t* out_inf, uint8_t* status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = bytes + i * 48 * nfp;
    const uint8_t flags = src[0] & 0xe0;
    uint8_t st = 0, inf = 0;
    if (flags & 0xa0) st = 2;                       // compressed or sort flag: not an uncompressed point
    if (!st && (flags & 0x40)) {
        inf = 1;
        uint64_t any = be64_load<ALIGNED>(src) & 0x1fffffffffffffffULL;
        for (int w = 1; w < 6 * nfp; w++) any |= be64_load<ALIGNED>(src + 8 * w);
        if (any) st = 2;
    }
    for (int e = 0; e < nfp; e++) {
        uint64_t limbs[6];
        for (int w = 0; w < 6; w++) {
            uint64_t v = be64_load<ALIGNED>(src + 48 * e + 8 * w);
            if (e == 0 && w == 0) v &= 0x1fffffffffffffffULL;   // strip the flag bits
            limbs[5 - w] = v;
        }
        if (!st && !inf && !fp_wire_is_canonical(limbs)) st = 1;
        const int we = nfp == 4 ? (e ^ 1) : e;
        for (int w = 0; w < 6; w++) out[(i * nfp + we) * 6 + w] = (st || inf) ? 0 : limbs[w];
    }
    if (inf && !st) out[(i * nfp + nfp / 2) * 6] = 1;   // identity is (0, 1) like the reference (src/g1.rs:25-31)
    out_inf[i] = inf && !st;
    status[i] = st;
}
template <bool ALIGNED>
__global__ void k_encode(const uint64_t* pts, const uint8_t* inf, size_t n, int nfp, uint8_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* dst = out + i * 48 * nfp;
    const bool is_inf = inf && inf[i];
    for (int e = 0; e < nfp; e++) {
        const int we = nfp == 4 ? (e ^ 1) : e;
        for (int w = 0; w < 6; w++) {
            uint64_t v = is_inf ? 0 : pts[(i * nfp + we) * 6 + (5 - w)];
            if (is_inf && e == 0 && w == 0) v = 0x4000000000000000ULL;
            if (ALIGNED) {
                *reinterpret_cast<uint64_t*>(dst + 48 * e + 8 * w) = __builtin_bswap64(v);
            } else {
                for (int b = 0; b < 8; b++) dst[48 * e + 8 * w + b] = (uint8_t)(v >> (56 - 8 * b));
            }
        }
    }
}

// zkp_points_check_batch: one status byte per point from the decode status (0 ok, 1 coordinate >= p, 2 malformed) and the is_valid
// status (0, 1 not on the curve, 2 not torsion free): 0 valid (or a well-formed infinity), 1, 2 as decoded, 3 not on the curve, 4 not
// in the subgroup (zkp_point_status).  A point that is not valid is flagged as infinity for the Miller loop (it contributes the
// neutral line: no arithmetic on garbage) - its check is failed by k_checks_merge whatever the pairing product says.
__global__ void k_points_merge(const uint8_t* dec, const uint8_t* val, uint8_t* inf, size_t n, uint8_t* st_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t st = dec[i] ? dec[i] : (val[i] ? (uint8_t)(val[i] + 2) : (uint8_t)0);
    st_out[i] = st;
    if (st) inf[i] = 1;
}
__global__ void k_checks_merge(const uint8_t* st1, const uint8_t* st2, size_t n_checks, size_t k, uint8_t* ok, int* all_ok) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_checks) return;
    uint8_t bad = 0;
    for (size_t j = 0; j < k; j++) bad |= st1[c * k + j] | st2[c * k + j];
    if (bad) {
        if (ok) ok[c] = 0;
        if (all_ok) atomicAnd(all_ok, 0);
    }
}

// ---- round 5: no pairing work on checks already known to fail.  k_checks_compact lists the checks whose 2k points are all valid
// (idx[0 .. *count), any order) and zeroes the ok byte of the others; k_gather_checks copies the listed checks' points next to each
// other; k_scatter_ok puts the compact ok bytes back.  (wave-aggregated: one atomicAdd per wavefront)
__global__ void k_checks_compact(const uint8_t* st1, const uint8_t* st2, size_t n_checks, size_t k, uint32_t* idx, uint32_t* count, uint8_t* ok) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool good = false;
    if (c < n_checks) {
        uint8_t bad = 0;
        for (size_t j = 0; j < k; j++) bad |= st1[c * k + j] | st2[c * k + j];
        good = !bad;
        if (bad && ok) ok[c] = 0;
    }
    const unsigned long long m = __ballot(good);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (good) idx[base + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)c;
}
// out[j * words + w] = in[idx[j] * words + w]: `words` 64-bit words (or bytes: T = uint8_t) per check.  Round 6: the number of listed
// checks is read from DEVICE memory (*count <= n_max; the grid covers n_max checks) - the host never learns it
template <class T>
__global__ void k_gather_checks(const T* in, const uint32_t* idx, const uint32_t* count, size_t n_max, size_t words, T* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t m = *count < n_max ? *count : n_max;
    if (i >= m * words) return;
    const size_t j = i / words, w = i - j * words;
    out[i] = in[(size_t)idx[j] * words + w];
}
__global__ void k_scatter_ok(const uint8_t* okc, const uint32_t* idx, const uint32_t* count, size_t n_max, uint8_t* ok) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t m = *count < n_max ? *count : n_max;
    if (j < m) ok[idx[j]] = okc[j];
}
// the AND over all checks is 0 as soon as one check failed its validity tests, whatever the pairings of the others say
__global__ void k_flag_if_fewer(int* all_ok, const uint32_t* count, uint32_t n_checks) {
    if (*count < n_checks) *all_ok = 0;
}

// every 6-limb element of a wire buffer must be < p
__global__ void k_check_canonical(const uint64_t* a, size_t n_fp, int* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fp) return;
    if (!fp_wire_is_canonical(a + 6 * i)) atomicOr(bad, 1);
}

}  // namespace

// =============================================================================== context
struct zkp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int validate = 0;
    int kernel = ZKP_KERNEL_AUTO;
    std::string err;
    // grow-only device workspace for the host-pointer API
    void* buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int* d_flag = nullptr;      // [0] product check result, [1] AND flag of the host entry points, [2] sticky validation word of the *_dev calls
    hipEvent_t ws_busy = nullptr;   // recorded at the end of every *_dev call: the next call (on whatever stream) waits for it
    uint64_t* prod = nullptr;   // Fp12 records of the product tree (zkp_fp12_product / zkp_miller_product)
    size_t prod_cap = 0;
    // host-pointer pairing entry points on large batches: slices of host_slice pairs, two workspace slots, copies of
    // the next / previous slice on their own streams while the current slice computes
    struct HostSlot {
        void* buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // g1 g2 inf1 inf2 gt ok
        size_t cap[6] = {0, 0, 0, 0, 0, 0};
        hipEvent_t in = nullptr, done = nullptr, out = nullptr;
    } hs[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    size_t host_slice = (size_t)1 << 19;
    hipDeviceProp_t prop;
    zkp::CoopState coop;
    // zkp_points_check_batch: decoded points, infinity flags, decode / is_valid / merged status bytes, ok bytes (grow-only)
    void* pc[18] = {};
    size_t pc_cap[18] = {};
    // one-rank-per-GPU communicator (zkp_comm_init_rank); null until then
    ncclComm_t comm = nullptr;
    int comm_nranks = 0, comm_rank = 0;
    // (nranks + 3) Fp12 records owned by the communicator: [0] this rank's record, [1..R] the gathered ones, [R+1] their product, [R+2] Gt.
    // Allocated by zkp_comm_init_rank, so that no allocation stands between a rank and the all-gather its peers wait in.
    uint64_t* comm_buf = nullptr;
};

namespace {

static const uint64_t GT_IDENTITY[72] = {1};

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return e__ == hipErrorOutOfMemory ? ZKP_ERR_OOM : ZKP_ERR_HIP;                      \
        }                                                                                       \
    } while (0)

int ensure(zkp_ctx* c, int slot, size_t bytes) {
    if (bytes <= c->cap[slot]) return ZKP_OK;
    if (c->buf[slot]) { HIPCHK(c, hipFree(c->buf[slot])); c->buf[slot] = nullptr; c->cap[slot] = 0; }
    HIPCHK(c, hipMalloc(&c->buf[slot], bytes));
    zkp_dbg_alloc(slot == 0 ? "ctx.buf0" : slot == 1 ? "ctx.buf1" : slot == 4 ? "ctx.buf4" : "ctx.buf", c->buf[slot], bytes);
    c->cap[slot] = bytes;
    return ZKP_OK;
}

inline unsigned grid_for(size_t n, int tpb) { return (unsigned)((n + tpb - 1) / tpb); }

// status of a call into the cooperative family: the HIP error (launch configuration, out of memory, ...) goes to zkp_last_error
int coop_rc(zkp_ctx* c, const char* what, hipError_t e) {
    if (e == hipSuccess) return ZKP_OK;
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? ZKP_ERR_OOM : ZKP_ERR_HIP;
}

int bind(zkp_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    return ZKP_OK;
}

// ---- device-pointer implementations (shared by both API flavours)
int miller_product_dev(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n, uint64_t* out,
                       hipStream_t s);
int miller_dev(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks, size_t k,
               uint64_t* out, hipStream_t s) {
    if (n_checks == 0) return ZKP_OK;
    if (k >= 64 && n_checks <= 16) {
        // few checks with long term lists: spread each check's pairs over the GPU (eight per accumulator) and fold the
        // values with the product tree, instead of walking the check's groups one after the other
        for (size_t ck = 0; ck < n_checks; ck++) {
            int rc = miller_product_dev(c, g1 + 12 * ck * k, g2 + 24 * ck * k, i1 ? i1 + ck * k : nullptr, i2 ? i2 + ck * k : nullptr, k,
                                        out + 72 * ck, s);
            if (rc) return rc;
        }
        return ZKP_OK;
    }
    if (zkp::coop_selected(&c->coop, c->kernel) && zkp::coop_supports_k(k))
        return coop_rc(c, "coop_miller", zkp::coop_miller(&c->coop, g1, g2, i1, i2, n_checks, k, out, s));
    hipLaunchKernelGGL(k_miller, dim3(grid_for(n_checks, TPB)), dim3(TPB), 0, s, g1, g2, i1, i2, n_checks, k, out);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
int final_exp_dev(zkp_ctx* c, const uint64_t* f, size_t n, uint64_t* out, hipStream_t s) {
    if (n == 0) return ZKP_OK;
    if (zkp::coop_selected(&c->coop, c->kernel))
        return coop_rc(c, "coop_final_exp", zkp::coop_final_exp(&c->coop, f, n, out, nullptr, nullptr, s));
    hipLaunchKernelGGL(k_final_exp, dim3(grid_for(n, TPB)), dim3(TPB), 0, s, f, n, out);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
// n_dev (device pointer, cooperative family only): the number of checks that really exist (<= n_checks) - see zkp::coop_pairing
int pairing_dev(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks, size_t k,
                uint64_t* out_gt, uint8_t* ok, int* all_ok, hipStream_t s, bool reset_flag = true, const uint32_t* n_dev = nullptr) {
    if (all_ok && reset_flag) {
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, all_ok, 1);
        HIPCHK(c, hipGetLastError());
    }
    if (n_checks == 0) return ZKP_OK;
    if (zkp::coop_selected(&c->coop, c->kernel) && zkp::coop_supports_k(k))
        return coop_rc(c, "coop_pairing", zkp::coop_pairing(&c->coop, g1, g2, i1, i2, n_checks, k, out_gt, ok, all_ok, s, n_dev));
    if (n_dev) { c->err = "pairing_dev: a device-resident count needs the cooperative kernel family"; return ZKP_ERR_ARG; }
    hipLaunchKernelGGL(k_pairing, dim3(grid_for(n_checks, TPB)), dim3(TPB), 0, s, g1, g2, i1, i2, n_checks, k, out_gt, ok, all_ok);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}

int ensure_prod(zkp_ctx* c, size_t records) {
    const size_t bytes = (records + 2) * 576;   // two spare records: product and Gt of zkp_pairing_product_check
    if (bytes <= c->prod_cap) return ZKP_OK;
    if (c->prod) { HIPCHK(c, hipFree(c->prod)); c->prod = nullptr; c->prod_cap = 0; }
    HIPCHK(c, hipMalloc((void**)&c->prod, bytes));
    zkp_dbg_alloc("ctx.prod", c->prod, bytes);
    c->prod_cap = bytes;
    return ZKP_OK;
}

// buf[0] <- prod_{i<n} buf[i] (n >= 1), destroying buf[1..n): ceil(log2 n) launches, level l multiplies element c
// by element c + ceil(n_l / 2)
int fp12_product_inplace(zkp_ctx* c, uint64_t* buf, size_t n, hipStream_t s) {
    const bool coop = zkp::coop_selected(&c->coop, c->kernel);
    while (n > 1) {
        const size_t h = (n + 1) / 2, m = n - h;
        if (coop) {
            if (int rc = coop_rc(c, "coop_fp12_mul_pairs", zkp::coop_fp12_mul_pairs(&c->coop, buf, m, h, s))) return rc;
        } else {
            hipLaunchKernelGGL(k_fp12_mul_pairs, dim3(grid_for(m, TPB)), dim3(TPB), 0, s, buf, m, h);
            HIPCHK(c, hipGetLastError());
        }
        n = h;
    }
    return ZKP_OK;
}

int fp12_product_dev(zkp_ctx* c, const uint64_t* f, size_t n, uint64_t* out, hipStream_t s) {
    if (n == 0) { HIPCHK(c, hipMemcpyAsync(out, GT_IDENTITY, 576, hipMemcpyHostToDevice, s)); return ZKP_OK; }
    int rc = ensure_prod(c, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->prod, f, n * 576, hipMemcpyDeviceToDevice, s));
    if ((rc = fp12_product_inplace(c, c->prod, n, s))) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->prod, 576, hipMemcpyDeviceToDevice, s));
    return ZKP_OK;
}

// multi_miller_loop over the whole batch as ONE check: the pairs are taken eight at a time (one shared accumulator
// per eight pairs), the n/8 values are multiplied by the product tree.  Result left in c->prod[0..72) and copied to out.
int miller_product_dev(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n, uint64_t* out,
                       hipStream_t s) {
    if (n == 0) { HIPCHK(c, hipMemcpyAsync(out, GT_IDENTITY, 576, hipMemcpyHostToDevice, s)); return ZKP_OK; }
    constexpr size_t PG = 8;   // pairs per accumulator
    const size_t n4 = n / PG, r = n % PG, nv = n4 + (r ? 1 : 0);
    int rc = ensure_prod(c, nv);
    if (rc) return rc;
    if (n4 && (rc = miller_dev(c, g1, g2, i1, i2, n4, PG, c->prod, s))) return rc;
    if (r && (rc = miller_dev(c, g1 + 12 * PG * n4, g2 + 24 * PG * n4, i1 ? i1 + PG * n4 : nullptr, i2 ? i2 + PG * n4 : nullptr, 1, r,
                              c->prod + 72 * n4, s)))
        return rc;
    if ((rc = fp12_product_inplace(c, c->prod, nv, s))) return rc;
    if (out) HIPCHK(c, hipMemcpyAsync(out, c->prod, 576, hipMemcpyDeviceToDevice, s));
    return ZKP_OK;
}

// prod_i e(P_i, Q_i) == Gt::identity() with ONE final exponentiation; out_gt and is_one are device pointers, each optional
int product_check_dev(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n, uint64_t* out_gt,
                      int* is_one, hipStream_t s) {
    int rc = ensure_prod(c, n / 8 + 1);
    if (rc) return rc;
    if ((rc = miller_product_dev(c, g1, g2, i1, i2, n, n ? nullptr : c->prod, s))) return rc;
    const size_t nv = n / 8 + (n % 8 ? 1 : 0);
    uint64_t* gt = c->prod + 72 * (nv + 1);      // spare record (ensure_prod keeps two)
    if ((rc = final_exp_dev(c, c->prod, 1, gt, s))) return rc;
    if (out_gt) HIPCHK(c, hipMemcpyAsync(out_gt, gt, 576, hipMemcpyDeviceToDevice, s));
    if (is_one) {
        hipLaunchKernelGGL(k_gt_is_one, dim3(1), dim3(64), 0, s, gt, is_one);
        HIPCHK(c, hipGetLastError());
    }
    return ZKP_OK;
}

// canonical-range validation of a host-API input already copied to the device
int validate_dev(zkp_ctx* c, const uint64_t* d, size_t n_fp) {
    if (!c->validate || n_fp == 0) return ZKP_OK;
    HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_check_canonical, dim3(grid_for(n_fp, 256)), dim3(256), 0, c->stream, d, n_fp, c->d_flag);
    HIPCHK(c, hipGetLastError());
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) { c->err = "input limbs >= p"; return ZKP_ERR_NONCANONICAL; }
    return ZKP_OK;
}

struct Staged { const uint64_t *g1, *g2; const uint8_t *i1, *i2; };

int ensure_slot(zkp_ctx* c, zkp_ctx::HostSlot* h, int which, size_t bytes) {
    if (bytes <= h->cap[which]) return ZKP_OK;
    if (h->buf[which]) { HIPCHK(c, hipFree(h->buf[which])); h->buf[which] = nullptr; h->cap[which] = 0; }
    HIPCHK(c, hipMalloc(&h->buf[which], bytes));
    zkp_dbg_alloc(which == 0 ? "slot.g1" : which == 1 ? "slot.g2" : which == 4 ? "slot.gt" : "slot.other", h->buf[which], bytes);
    h->cap[which] = bytes;
    return ZKP_OK;
}

// pairing()/pairing check of n_checks x k pairs from HOST arrays in slices: while slice i computes on the context's
// stream, the host thread uploads slice i+1 (copy-in stream) and downloads the results of slice i-1 (copy-out stream).
// out_gt / ok / all_ok are host pointers, each optional.
int host_sliced_impl(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks, size_t k,
                     uint64_t* out_gt, uint8_t* ok, int* all_ok) {
    const zkp::plan::Slices sl = zkp::plan::plan_slices(c->host_slice, n_checks, k);
    const size_t sc = sl.checks_per_slice, nsl = sl.n_slices;
    int rc;
    if (!c->s_in) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(c, hipEventCreateWithFlags(&c->hs[i].in, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->hs[i].done, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->hs[i].out, hipEventDisableTiming));
        }
    }
    for (int i = 0; i < 2; i++) {
        zkp_ctx::HostSlot* h = &c->hs[i];
        if ((rc = ensure_slot(c, h, 0, sc * k * 96)) || (rc = ensure_slot(c, h, 1, sc * k * 192))) return rc;
        if (inf1 && (rc = ensure_slot(c, h, 2, sc * k))) return rc;
        if (inf2 && (rc = ensure_slot(c, h, 3, sc * k))) return rc;
        if (out_gt && (rc = ensure_slot(c, h, 4, sc * 576))) return rc;
        if ((rc = ensure_slot(c, h, 5, sc))) return rc;
    }
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->d_flag + 1, 1);
    HIPCHK(c, hipGetLastError());
    auto upload = [&](size_t i) -> int {
        zkp_ctx::HostSlot* h = &c->hs[i & 1];
        const size_t c0 = i * sc, ns = n_checks - c0 < sc ? n_checks - c0 : sc, p0 = c0 * k, np = ns * k;
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(c->s_in, h->done, 0));   // slice i-2 has finished reading these buffers
        HIPCHK(c, hipMemcpyAsync(h->buf[0], g1 + 12 * p0, np * 96, hipMemcpyHostToDevice, c->s_in));
        HIPCHK(c, hipMemcpyAsync(h->buf[1], g2 + 24 * p0, np * 192, hipMemcpyHostToDevice, c->s_in));
        if (inf1) HIPCHK(c, hipMemcpyAsync(h->buf[2], inf1 + p0, np, hipMemcpyHostToDevice, c->s_in));
        if (inf2) HIPCHK(c, hipMemcpyAsync(h->buf[3], inf2 + p0, np, hipMemcpyHostToDevice, c->s_in));
        HIPCHK(c, hipEventRecord(h->in, c->s_in));
        return ZKP_OK;
    };
    auto download = [&](size_t i) -> int {
        zkp_ctx::HostSlot* h = &c->hs[i & 1];
        const size_t c0 = i * sc, ns = n_checks - c0 < sc ? n_checks - c0 : sc;
        HIPCHK(c, hipStreamWaitEvent(c->s_out, h->done, 0));
        if (out_gt) HIPCHK(c, hipMemcpyAsync(out_gt + 72 * c0, h->buf[4], ns * 576, hipMemcpyDeviceToHost, c->s_out));
        if (ok) HIPCHK(c, hipMemcpyAsync(ok + c0, h->buf[5], ns, hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(c, hipEventRecord(h->out, c->s_out));
        return ZKP_OK;
    };
    if ((rc = upload(0))) return rc;
    for (size_t i = 0; i < nsl; i++) {
        zkp_ctx::HostSlot* h = &c->hs[i & 1];
        const size_t c0 = i * sc, ns = n_checks - c0 < sc ? n_checks - c0 : sc;
        HIPCHK(c, hipStreamWaitEvent(c->stream, h->in, 0));
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, h->out, 0));   // results of slice i-2 have left the output buffers
        if ((rc = pairing_dev(c, (const uint64_t*)h->buf[0], (const uint64_t*)h->buf[1], inf1 ? (const uint8_t*)h->buf[2] : nullptr,
                              inf2 ? (const uint8_t*)h->buf[3] : nullptr, ns, k, out_gt ? (uint64_t*)h->buf[4] : nullptr, (uint8_t*)h->buf[5],
                              c->d_flag + 1, c->stream, false)))
            return rc;
        HIPCHK(c, hipEventRecord(h->done, c->stream));
        if (i + 1 < nsl && (rc = upload(i + 1))) return rc;
        if (i >= 1 && (rc = download(i - 1))) return rc;
    }
    if ((rc = download(nsl - 1))) return rc;
    int flag = 1;
    HIPCHK(c, hipMemcpyAsync(&flag, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->s_out));
    if (all_ok) *all_ok = flag;
    return ZKP_OK;
}

// an error must not leave copies from / into the caller's arrays in flight
int host_sliced(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks, size_t k,
                uint64_t* out_gt, uint8_t* ok, int* all_ok) {
    const int rc = host_sliced_impl(c, g1, g2, inf1, inf2, n_checks, k, out_gt, ok, all_ok);
    if (rc != ZKP_OK) {
        if (c->s_in) (void)hipStreamSynchronize(c->s_in);
        (void)hipStreamSynchronize(c->stream);
        if (c->s_out) (void)hipStreamSynchronize(c->s_out);
    }
    return rc;
}

// copy a (g1,g2,inf1,inf2) pair batch to workspace slots 0..3
int stage_pairs(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t np, Staged* st) {
    int rc;
    if ((rc = ensure(c, 0, np * 96)) || (rc = ensure(c, 1, np * 192))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[0], g1, np * 96, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->buf[1], g2, np * 192, hipMemcpyHostToDevice, c->stream));
    st->g1 = (const uint64_t*)c->buf[0];
    st->g2 = (const uint64_t*)c->buf[1];
    st->i1 = st->i2 = nullptr;
    if (inf1) {
        if ((rc = ensure(c, 2, np))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->buf[2], inf1, np, hipMemcpyHostToDevice, c->stream));
        st->i1 = (const uint8_t*)c->buf[2];
    }
    if (inf2) {
        if ((rc = ensure(c, 3, np))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->buf[3], inf2, np, hipMemcpyHostToDevice, c->stream));
        st->i2 = (const uint8_t*)c->buf[3];
    }
    if ((rc = validate_dev(c, st->g1, np * 2)) || (rc = validate_dev(c, st->g2, np * 4))) return rc;
    return ZKP_OK;
}

}  // namespace

// Host-pointer entry points copy from / into the caller's arrays asynchronously: whatever way such a call returns
// (HIPCHK included), nothing may still be in flight - the context's stream is drained on the way out.
namespace {
struct HostCall {
    zkp_ctx* c;
    explicit HostCall(zkp_ctx* ctx) : c(ctx) {   // earlier *_dev calls may still be using the workspace on other streams
        if (c && c->stream && c->ws_busy) (void)hipStreamWaitEvent(c->stream, c->ws_busy, 0);
    }
    ~HostCall() {
        if (!c || !c->stream) return;
        if (c->ws_busy) (void)hipEventRecord(c->ws_busy, c->stream);
        (void)hipStreamSynchronize(c->stream);
    }
};
}  // namespace

// =============================================================================== C ABI
extern "C" {

int zkp_abi_version(void) { return 4; }

const char* zkp_strerror(int status) {
    switch (status) {
        case ZKP_OK: return "ok";
        case ZKP_ERR_ARG: return "bad argument";
        case ZKP_ERR_NO_DEVICE: return "no usable HIP device";
        case ZKP_ERR_HIP: return "HIP runtime error";
        case ZKP_ERR_NONCANONICAL: return "non-canonical field element in input";
        case ZKP_ERR_OOM: return "out of device memory";
        case ZKP_ERR_COMM: return "RCCL error / no communicator";
        default: return "unknown status";
    }
}

const uint64_t* zkp_gt_identity(void) { return GT_IDENTITY; }

int zkp_init(int device, zkp_ctx** out_ctx) {
    if (!out_ctx) return ZKP_ERR_ARG;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZKP_ERR_NO_DEVICE;
    zkp_ctx* c = new (std::nothrow) zkp_ctx();
    if (!c) return ZKP_ERR_OOM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreateWithFlags(&c->ws_busy, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&c->d_flag, 3 * sizeof(int)) != hipSuccess || hipMemset(c->d_flag, 0, 3 * sizeof(int)) != hipSuccess) {
        zkp_free(c);
        return ZKP_ERR_NO_DEVICE;
    }
    if (zkp::coop_init(&c->coop, c->prop) != hipSuccess) {
        zkp_free(c);
        return ZKP_ERR_HIP;
    }
    if (const char* hsl = getenv("ZKP_HOST_SLICE")) {
        c->host_slice = (size_t)atol(hsl);
        if (c->host_slice < 64) c->host_slice = 64;
    }
    const char* env = getenv("ZKP_KERNEL");
    if (env) {
        if (!strcmp(env, "thread")) c->kernel = ZKP_KERNEL_THREAD;
        else if (!strcmp(env, "coop")) c->kernel = ZKP_KERNEL_COOP;
    }
    *out_ctx = c;
    return ZKP_OK;
}

void zkp_free(zkp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_buf) { (void)hipFree(c->comm_buf); c->comm_buf = nullptr; }
    zkp::coop_free(&c->coop);
    for (int i = 0; i < 8; i++)
        if (c->buf[i]) (void)hipFree(c->buf[i]);
    for (int i = 0; i < 18; i++)
        if (c->pc[i]) (void)hipFree(c->pc[i]);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->prod) (void)hipFree(c->prod);
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 6; j++)
            if (c->hs[i].buf[j]) (void)hipFree(c->hs[i].buf[j]);
        if (c->hs[i].in) (void)hipEventDestroy(c->hs[i].in);
        if (c->hs[i].done) (void)hipEventDestroy(c->hs[i].done);
        if (c->hs[i].out) (void)hipEventDestroy(c->hs[i].out);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ws_busy) (void)hipEventDestroy(c->ws_busy);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* zkp_last_error(const zkp_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
int zkp_set_validate(zkp_ctx* c, int on) { if (!c) return ZKP_ERR_ARG; c->validate = on ? 1 : 0; return ZKP_OK; }
int zkp_set_kernel(zkp_ctx* c, int kind) {
    if (!c || kind < ZKP_KERNEL_AUTO || kind > ZKP_KERNEL_COOP) return ZKP_ERR_ARG;
    if (kind == ZKP_KERNEL_COOP && !c->coop.available) { c->err = "cooperative kernels not available"; return ZKP_ERR_ARG; }
    c->kernel = kind;
    return ZKP_OK;
}
int zkp_device_info(const zkp_ctx* c, int* cus, int* clock_khz, char* name, size_t name_len) {
    if (!c) return ZKP_ERR_ARG;
    if (cus) *cus = c->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = c->prop.clockRate;
    if (name && name_len) { strncpy(name, c->prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return ZKP_OK;
}

// ---------------------------------------------------------------- device-pointer API
#define S(stream) ((hipStream_t)(stream))
namespace {
// The workspace inside a ctx (line buffers, per-check state, product tree, flags) is shared by every call: a *_dev call
// first makes its stream wait for the previous call's last use of the workspace - which may have been queued on ANOTHER
// stream - and leaves an event behind for the next one.  Calls on one stream order themselves anyway.
// A stream that is being CAPTURED into a hipGraph (round 6) takes no part in that hand-over: an event recorded outside the capture may
// not be waited for inside it, and one recorded inside would poison the next ordinary call.  Whoever replays the graph orders it
// against the context's other calls himself (one stream, or his own events) - include/zkp_pairings.h, "hipGraph capture".
struct DevCall {
    zkp_ctx* c;
    hipStream_t s;
    int rc;
    bool capturing;
    DevCall(zkp_ctx* ctx, void* stream) : c(ctx), s((hipStream_t)stream), rc(ZKP_OK), capturing(false) {
        hipError_t e = hipSetDevice(c->device);
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (e == hipSuccess && s && hipStreamIsCapturing(s, &cs) == hipSuccess) capturing = cs == hipStreamCaptureStatusActive;
        if (e == hipSuccess && !capturing) e = hipStreamWaitEvent(s, c->ws_busy, 0);
        if (e != hipSuccess) { c->err = std::string("DevCall: ") + hipGetErrorString(e); rc = ZKP_ERR_HIP; }
    }
    ~DevCall() { if (rc == ZKP_OK && !capturing) (void)hipEventRecord(c->ws_busy, s); }
};
// validation mode on the device-pointer entry points: a range check of the inputs on the caller's stream that ORs into a
// sticky word (no host synchronisation); zkp_take_validation_status_dev reads and clears it
int validate_on_stream(zkp_ctx* c, const void* d, size_t n_fp, hipStream_t s) {
    if (!c->validate || !n_fp || !d) return ZKP_OK;
    hipLaunchKernelGGL(k_check_canonical, dim3(grid_for(n_fp, 256)), dim3(256), 0, s, (const uint64_t*)d, n_fp, c->d_flag + 2);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
}  // namespace
#define DEV_ENTER(ctx, stream)      \
    DevCall dc__((ctx), (stream)); \
    if (dc__.rc) return dc__.rc
// range limits shared by the entry points: every per-launch count stays in 32 bits
using zkp::plan::too_many;   // zkp_plan.hpp: n <= 2^31 - 1, k <= 65535, n * k <= 2^31 - 1

int zkp_pairing_batch_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n, void* out, void* stream) {
    if (!c || too_many(n) || (n && (!g1 || !g2 || !out))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n * 2, S(stream))) || (rc = validate_on_stream(c, g2, n * 4, S(stream)))) return rc;
    return pairing_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n, 1, (uint64_t*)out, nullptr, nullptr, S(stream));
}
int zkp_multi_miller_loop_batch_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n_checks, size_t k,
                                    void* out, void* stream) {
    if (!c || too_many(n_checks, k) || (n_checks && (!out || (k && (!g1 || !g2))))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n_checks * k * 2, S(stream))) || (rc = validate_on_stream(c, g2, n_checks * k * 4, S(stream)))) return rc;
    return miller_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n_checks, k, (uint64_t*)out, S(stream));
}
int zkp_final_exponentiation_batch_dev(zkp_ctx* c, const void* f, size_t n, void* out, void* stream) {
    if (!c || too_many(n) || (n && (!f || !out))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (int rc = validate_on_stream(c, f, n * 12, S(stream))) return rc;
    return final_exp_dev(c, (const uint64_t*)f, n, (uint64_t*)out, S(stream));
}
int zkp_fp12_product_dev(zkp_ctx* c, const void* f, size_t n, void* out, void* stream) {
    if (!c || !out || (n && !f) || too_many(n)) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (int rc = validate_on_stream(c, f, n * 12, S(stream))) return rc;
    return fp12_product_dev(c, (const uint64_t*)f, n, (uint64_t*)out, S(stream));
}
int zkp_miller_product_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n, void* out_ml, void* stream) {
    if (!c || !out_ml || (n && (!g1 || !g2)) || too_many(n)) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n * 2, S(stream))) || (rc = validate_on_stream(c, g2, n * 4, S(stream)))) return rc;
    return miller_product_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n, (uint64_t*)out_ml, S(stream));
}
int zkp_pairing_product_check_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n, void* out_gt,
                                  void* is_one, void* stream) {
    if (!c || (n && (!g1 || !g2)) || too_many(n)) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n * 2, S(stream))) || (rc = validate_on_stream(c, g2, n * 4, S(stream)))) return rc;
    return product_check_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n, (uint64_t*)out_gt, (int*)is_one,
                             S(stream));
}
int zkp_pairing_check_batch_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n_checks, size_t k,
                                void* ok, void* all_ok, void* stream) {
    if (!c || too_many(n_checks, k) || (n_checks && k && (!g1 || !g2))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n_checks * k * 2, S(stream))) || (rc = validate_on_stream(c, g2, n_checks * k * 4, S(stream)))) return rc;
    return pairing_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n_checks, k, nullptr, (uint8_t*)ok,
                       (int*)all_ok, S(stream));
}
int zkp_pairing_gt_check_batch_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n_checks, size_t k,
                                   void* out_gt, void* ok, void* all_ok, void* stream) {
    if (!c || too_many(n_checks, k) || (n_checks && k && (!g1 || !g2))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int rc;
    if ((rc = validate_on_stream(c, g1, n_checks * k * 2, S(stream))) || (rc = validate_on_stream(c, g2, n_checks * k * 4, S(stream)))) return rc;
    return pairing_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, (const uint8_t*)i1, (const uint8_t*)i2, n_checks, k, (uint64_t*)out_gt,
                       (uint8_t*)ok, (int*)all_ok, S(stream));
}
// is_valid of n G1 (which = 1) or G2 (which = 2) points on the context's kernel family, on stream s
static int valid_dev(zkp_ctx* c, int which, const void* pts, const void* inf, size_t n, void* status, hipStream_t s) {
    if (!n) return ZKP_OK;
    if (zkp::coop_selected(&c->coop, c->kernel)) {
        HIPCHK(c, which == 1 ? zkp::coop_g1_valid((const uint64_t*)pts, (const uint8_t*)inf, n, (uint8_t*)status, s)
                             : zkp::coop_g2_valid(&c->coop, (const uint64_t*)pts, (const uint8_t*)inf, n, (uint8_t*)status, s));
        return ZKP_OK;
    }
    if (which == 1)
        hipLaunchKernelGGL(k_g1_valid, dim3(grid_for(n, TPB)), dim3(TPB), 0, s, (const uint64_t*)pts, (const uint8_t*)inf, n, (uint8_t*)status);
    else
        hipLaunchKernelGGL(k_g2_valid, dim3(grid_for(n, TPB)), dim3(TPB), 0, s, (const uint64_t*)pts, (const uint8_t*)inf, n, (uint8_t*)status);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
int zkp_g1_is_valid_batch_dev(zkp_ctx* c, const void* g1, const void* inf, size_t n, void* status, void* stream) {
    if (!c || too_many(n) || (n && (!g1 || !status))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (!n) return ZKP_OK;
    if (int rc = validate_on_stream(c, g1, n * 2, S(stream))) return rc;
    return valid_dev(c, 1, g1, inf, n, status, S(stream));
}
int zkp_g2_is_valid_batch_dev(zkp_ctx* c, const void* g2, const void* inf, size_t n, void* status, void* stream) {
    if (!c || too_many(n) || (n && (!g2 || !status))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (!n) return ZKP_OK;
    if (int rc = validate_on_stream(c, g2, n * 4, S(stream))) return rc;
    return valid_dev(c, 2, g2, inf, n, status, S(stream));
}
// ---- uncompressed point codec on resident buffers (round 4): the decode / encode kernels without the PCIe round trip
static int codec_dev(zkp_ctx* c, bool decode, int nfp, const void* in, const void* inf_in, size_t n, void* out, void* out_inf, void* status,
                     hipStream_t s) {
    if (!n) return ZKP_OK;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 7u) == 0;
    if (decode) {
        if (aligned)
            hipLaunchKernelGGL(k_decode<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint8_t*)in, n, nfp, (uint64_t*)out, (uint8_t*)out_inf, (uint8_t*)status);
        else
            hipLaunchKernelGGL(k_decode<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint8_t*)in, n, nfp, (uint64_t*)out, (uint8_t*)out_inf, (uint8_t*)status);
    } else {
        if (aligned)
            hipLaunchKernelGGL(k_encode<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)in, (const uint8_t*)inf_in, n, nfp, (uint8_t*)out);
        else
            hipLaunchKernelGGL(k_encode<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)in, (const uint8_t*)inf_in, n, nfp, (uint8_t*)out);
    }
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
int zkp_g1_decode_batch_dev(zkp_ctx* c, const void* bytes, size_t n, void* out_g1, void* out_inf, void* status, void* stream) {
    if (!c || too_many(n) || (n && (!bytes || !out_g1 || !out_inf || !status))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    return codec_dev(c, true, 2, bytes, nullptr, n, out_g1, out_inf, status, S(stream));
}
int zkp_g2_decode_batch_dev(zkp_ctx* c, const void* bytes, size_t n, void* out_g2, void* out_inf, void* status, void* stream) {
    if (!c || too_many(n) || (n && (!bytes || !out_g2 || !out_inf || !status))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    return codec_dev(c, true, 4, bytes, nullptr, n, out_g2, out_inf, status, S(stream));
}
int zkp_g1_encode_batch_dev(zkp_ctx* c, const void* g1, const void* inf, size_t n, void* out_bytes, void* stream) {
    if (!c || too_many(n) || (n && (!g1 || !out_bytes))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    return codec_dev(c, false, 2, g1, inf, n, out_bytes, nullptr, nullptr, S(stream));
}
int zkp_g2_encode_batch_dev(zkp_ctx* c, const void* g2, const void* inf, size_t n, void* out_bytes, void* stream) {
    if (!c || too_many(n) || (n && (!g2 || !out_bytes))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    return codec_dev(c, false, 4, g2, inf, n, out_bytes, nullptr, nullptr, S(stream));
}

// ---- BASELINE config 5 as ONE call: raw uncompressed bytes -> decode -> is_valid -> pairing check, everything between the byte
// strings and the status / ok bytes resident in HBM (the context's workspace)
static int ensure_pc(zkp_ctx* c, int slot, size_t bytes) {
    if (bytes <= c->pc_cap[slot]) return ZKP_OK;
    if (c->pc[slot]) { HIPCHK(c, hipFree(c->pc[slot])); c->pc[slot] = nullptr; c->pc_cap[slot] = 0; }
    HIPCHK(c, hipMalloc(&c->pc[slot], bytes));
    zkp_dbg_alloc("ctx.pc", c->pc[slot], bytes);
    c->pc_cap[slot] = bytes;
    return ZKP_OK;
}
enum { PC_G1 = 0, PC_G2, PC_INF1, PC_INF2, PC_DEC, PC_VAL, PC_ST1, PC_ST2, PC_OK, PC_BYTES, PC_IDX, PC_CNT, PC_CG1, PC_CG2, PC_CINF1, PC_CINF2, PC_COK };
// Validate first, then use (reference src/g1.rs:49-62, src/g2.rs:57-69: is_valid before anything is done with a point): the checks whose
// points are all valid are listed on the device (wave-aggregated compaction), gathered next to each other, and only those enter the
// Miller loop and the final exponentiation - a check with an invalid point costs its validity tests and nothing else.  Round 6: the
// NUMBER of listed checks stays in device memory.  The pairing phase is planned for the worst case (every check valid) and its kernels
// read the count themselves (zkp::coop_pairing's n_dev: wavefronts behind the count leave at once), so the call never waits on the
// host: asynchronous on `s` like every other *_dev entry point, and capturable into a hipGraph once its workspaces exist.  (Round 5 read
// the count back - 4 bytes, one hipStreamSynchronize - to size the grids on the host.)  The thread family has no such kernels: it
// runs the fused pairing on every check and fails the invalid ones afterwards (k_checks_merge), as round 4 did.
static int points_check_dev(zkp_ctx* c, const void* b1, const void* b2, size_t n_checks, size_t k, void* st1, void* st2, void* ok, int* all_ok,
                            hipStream_t s) {
    const size_t np = n_checks * k;
    const bool compact = n_checks && k && zkp::coop_selected(&c->coop, c->kernel) && zkp::coop_supports_k(k);
    int rc;
    // every workspace first (an allocation synchronises the device): nothing below this block allocates
    if ((rc = ensure_pc(c, PC_G1, np * 96 + 8)) || (rc = ensure_pc(c, PC_G2, np * 192 + 8)) || (rc = ensure_pc(c, PC_INF1, np + 8)) ||
        (rc = ensure_pc(c, PC_INF2, np + 8)) || (rc = ensure_pc(c, PC_DEC, 2 * np + 8)) || (rc = ensure_pc(c, PC_VAL, 2 * np + 8)) ||
        (rc = ensure_pc(c, PC_ST1, np + 8)) || (rc = ensure_pc(c, PC_ST2, np + 8)) || (rc = ensure_pc(c, PC_OK, n_checks + 8)))
        return rc;
    if (compact && ((rc = ensure_pc(c, PC_IDX, n_checks * 4 + 8)) || (rc = ensure_pc(c, PC_CNT, 8)) || (rc = ensure_pc(c, PC_CG1, np * 96 + 8)) ||
                    (rc = ensure_pc(c, PC_CG2, np * 192 + 8)) || (rc = ensure_pc(c, PC_CINF1, np + 8)) || (rc = ensure_pc(c, PC_CINF2, np + 8)) ||
                    (rc = ensure_pc(c, PC_COK, n_checks + 8))))
        return rc;
    uint8_t* dec1 = (uint8_t*)c->pc[PC_DEC];
    uint8_t* dec2 = dec1 + np;
    uint8_t* val1 = (uint8_t*)c->pc[PC_VAL];
    uint8_t* val2 = val1 + np;
    uint8_t* s1 = st1 ? (uint8_t*)st1 : (uint8_t*)c->pc[PC_ST1];
    uint8_t* s2 = st2 ? (uint8_t*)st2 : (uint8_t*)c->pc[PC_ST2];
    uint8_t* okb = ok ? (uint8_t*)ok : (uint8_t*)c->pc[PC_OK];
    if (np) {
        if ((rc = codec_dev(c, true, 2, b1, nullptr, np, c->pc[PC_G1], c->pc[PC_INF1], dec1, s)) ||
            (rc = codec_dev(c, true, 4, b2, nullptr, np, c->pc[PC_G2], c->pc[PC_INF2], dec2, s)) ||
            (rc = valid_dev(c, 1, c->pc[PC_G1], c->pc[PC_INF1], np, val1, s)) || (rc = valid_dev(c, 2, c->pc[PC_G2], c->pc[PC_INF2], np, val2, s)))
            return rc;
        hipLaunchKernelGGL(k_points_merge, dim3(grid_for(np, 256)), dim3(256), 0, s, dec1, val1, (uint8_t*)c->pc[PC_INF1], np, s1);
        hipLaunchKernelGGL(k_points_merge, dim3(grid_for(np, 256)), dim3(256), 0, s, dec2, val2, (uint8_t*)c->pc[PC_INF2], np, s2);
        HIPCHK(c, hipGetLastError());
    }
    if (!compact) {
        if ((rc = pairing_dev(c, (const uint64_t*)c->pc[PC_G1], (const uint64_t*)c->pc[PC_G2], (const uint8_t*)c->pc[PC_INF1], (const uint8_t*)c->pc[PC_INF2],
                              n_checks, k, nullptr, okb, all_ok, s)))
            return rc;
        if (n_checks && k) {
            hipLaunchKernelGGL(k_checks_merge, dim3(grid_for(n_checks, 256)), dim3(256), 0, s, s1, s2, n_checks, k, okb, all_ok);
            HIPCHK(c, hipGetLastError());
        }
        return ZKP_OK;
    }
    uint32_t* const idx = (uint32_t*)c->pc[PC_IDX];
    uint32_t* const cnt = (uint32_t*)c->pc[PC_CNT];
    HIPCHK(c, hipMemsetAsync(cnt, 0, 4, s));
    hipLaunchKernelGGL(k_checks_compact, dim3(grid_for(n_checks, 256)), dim3(256), 0, s, s1, s2, n_checks, k, idx, cnt, okb);
    hipLaunchKernelGGL(k_gather_checks<uint64_t>, dim3(grid_for(np * 12, 256)), dim3(256), 0, s, (const uint64_t*)c->pc[PC_G1], idx, cnt, n_checks, k * 12, (uint64_t*)c->pc[PC_CG1]);
    hipLaunchKernelGGL(k_gather_checks<uint64_t>, dim3(grid_for(np * 24, 256)), dim3(256), 0, s, (const uint64_t*)c->pc[PC_G2], idx, cnt, n_checks, k * 24, (uint64_t*)c->pc[PC_CG2]);
    hipLaunchKernelGGL(k_gather_checks<uint8_t>, dim3(grid_for(np, 256)), dim3(256), 0, s, (const uint8_t*)c->pc[PC_INF1], idx, cnt, n_checks, k, (uint8_t*)c->pc[PC_CINF1]);
    hipLaunchKernelGGL(k_gather_checks<uint8_t>, dim3(grid_for(np, 256)), dim3(256), 0, s, (const uint8_t*)c->pc[PC_INF2], idx, cnt, n_checks, k, (uint8_t*)c->pc[PC_CINF2]);
    HIPCHK(c, hipGetLastError());
    // the listed checks (their points are all valid): fused pairing, ok bytes in list order, the AND flag over them
    if ((rc = pairing_dev(c, (const uint64_t*)c->pc[PC_CG1], (const uint64_t*)c->pc[PC_CG2], (const uint8_t*)c->pc[PC_CINF1], (const uint8_t*)c->pc[PC_CINF2],
                          n_checks, k, nullptr, (uint8_t*)c->pc[PC_COK], all_ok, s, true, cnt)))
        return rc;
    hipLaunchKernelGGL(k_scatter_ok, dim3(grid_for(n_checks, 256)), dim3(256), 0, s, (const uint8_t*)c->pc[PC_COK], idx, cnt, n_checks, okb);
    if (all_ok) hipLaunchKernelGGL(k_flag_if_fewer, dim3(1), dim3(1), 0, s, all_ok, cnt, (uint32_t)n_checks);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
int zkp_points_check_batch_dev(zkp_ctx* c, const void* g1_bytes, const void* g2_bytes, size_t n_checks, size_t k, void* st1, void* st2, void* ok,
                               void* all_ok, void* stream) {
    if (!c || too_many(n_checks, k) || (n_checks && k && (!g1_bytes || !g2_bytes))) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    return points_check_dev(c, g1_bytes, g2_bytes, n_checks, k, st1, st2, ok, (int*)all_ok, S(stream));
}
int zkp_g1_mul_batch_dev(zkp_ctx* c, const void* base, size_t stride, const void* sc, size_t n, void* out, void* out_inf, void* stream) {
    if (!c || too_many(n) || (n && (!base || !sc || !out)) || (stride != 0 && stride != 12)) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (!n) return ZKP_OK;
    if (int rc = validate_on_stream(c, base, (stride ? n : 1) * 2, S(stream))) return rc;
    if (zkp::coop_selected(&c->coop, c->kernel)) {
        HIPCHK(c, zkp::coop_g1_mul((const uint64_t*)base, stride, (const uint64_t*)sc, n, (uint64_t*)out, (uint8_t*)out_inf, S(stream)));
        return ZKP_OK;
    }
    hipLaunchKernelGGL(k_g1_mul, dim3(grid_for(n, TPB)), dim3(TPB), 0, S(stream), (const uint64_t*)base, stride, (const uint64_t*)sc, n, (uint64_t*)out, (uint8_t*)out_inf);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
int zkp_g2_mul_batch_dev(zkp_ctx* c, const void* base, size_t stride, const void* sc, size_t n, void* out, void* out_inf, void* stream) {
    if (!c || too_many(n) || (n && (!base || !sc || !out)) || (stride != 0 && stride != 24)) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    if (!n) return ZKP_OK;
    if (int rc = validate_on_stream(c, base, (stride ? n : 1) * 4, S(stream))) return rc;
    if (zkp::coop_selected(&c->coop, c->kernel)) {
        HIPCHK(c, zkp::coop_g2_mul((const uint64_t*)base, stride, (const uint64_t*)sc, n, (uint64_t*)out, (uint8_t*)out_inf, S(stream)));
        return ZKP_OK;
    }
    hipLaunchKernelGGL(k_g2_mul, dim3(grid_for(n, TPB)), dim3(TPB), 0, S(stream), (const uint64_t*)base, stride, (const uint64_t*)sc, n, (uint64_t*)out, (uint8_t*)out_inf);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}
// validation mode of the device-pointer entry points: *bad = 1 if any *_dev call since the last query saw a field
// element >= p in its inputs (the results of such a call are unspecified).  Waits for `stream`, then clears the word.
int zkp_take_validation_status_dev(zkp_ctx* c, void* stream, int* bad) {
    if (!c || !bad) return ZKP_ERR_ARG;
    DEV_ENTER(c, stream);
    int v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->d_flag + 2, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(c, hipMemsetAsync(c->d_flag + 2, 0, sizeof(int), S(stream)));
    HIPCHK(c, hipStreamSynchronize(S(stream)));
    *bad = v ? 1 : 0;
    return ZKP_OK;
}

// ---------------------------------------------------------------- host-pointer API
int zkp_pairing_batch(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n, uint64_t* out_gt) {
    if (!c || (n && (!g1 || !g2 || !out_gt))) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    if (n > c->host_slice && !c->validate) return host_sliced(c, g1, g2, inf1, inf2, n, 1, out_gt, nullptr, nullptr);
    Staged st;
    if ((rc = stage_pairs(c, g1, g2, inf1, inf2, n, &st)) || (rc = ensure(c, 4, n * 576))) return rc;
    if ((rc = pairing_dev(c, st.g1, st.g2, st.i1, st.i2, n, 1, (uint64_t*)c->buf[4], nullptr, nullptr, c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_gt, c->buf[4], n * 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_multi_miller_loop_batch(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks,
                                size_t k, uint64_t* out_ml) {
    if (!c || (n_checks && (!out_ml || (k && (!g1 || !g2))))) return ZKP_ERR_ARG;
    if (!n_checks) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    Staged st = {nullptr, nullptr, nullptr, nullptr};
    if (k && (rc = stage_pairs(c, g1, g2, inf1, inf2, n_checks * k, &st))) return rc;
    if ((rc = ensure(c, 4, n_checks * 576))) return rc;
    if ((rc = miller_dev(c, st.g1, st.g2, st.i1, st.i2, n_checks, k, (uint64_t*)c->buf[4], c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_ml, c->buf[4], n_checks * 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_final_exponentiation_batch(zkp_ctx* c, const uint64_t* f, size_t n, uint64_t* out_gt) {
    if (!c || (n && (!f || !out_gt))) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    if ((rc = ensure(c, 5, n * 576)) || (rc = ensure(c, 4, n * 576))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[5], f, n * 576, hipMemcpyHostToDevice, c->stream));
    if ((rc = validate_dev(c, (const uint64_t*)c->buf[5], n * 12))) return rc;
    if ((rc = final_exp_dev(c, (const uint64_t*)c->buf[5], n, (uint64_t*)c->buf[4], c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_gt, c->buf[4], n * 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_fp12_product(zkp_ctx* c, const uint64_t* f, size_t n, uint64_t* out) {
    if (!c || !out || (n && !f) || n > 0x7fffffffu) return ZKP_ERR_ARG;
    if (!n) { memcpy(out, GT_IDENTITY, 576); return ZKP_OK; }
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    if ((rc = ensure_prod(c, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->prod, f, n * 576, hipMemcpyHostToDevice, c->stream));
    if ((rc = validate_dev(c, c->prod, n * 12))) return rc;
    if ((rc = fp12_product_inplace(c, c->prod, n, c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->prod, 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_miller_product(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n, uint64_t* out_ml) {
    if (!c || !out_ml || (n && (!g1 || !g2)) || n > 0x7fffffffu) return ZKP_ERR_ARG;
    if (!n) { memcpy(out_ml, GT_IDENTITY, 576); return ZKP_OK; }
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    Staged st = {nullptr, nullptr, nullptr, nullptr};
    if ((rc = stage_pairs(c, g1, g2, inf1, inf2, n, &st))) return rc;
    if ((rc = miller_product_dev(c, st.g1, st.g2, st.i1, st.i2, n, nullptr, c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_ml, c->prod, 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_pairing_product_check(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n,
                              uint64_t* out_gt, int* is_one) {
    if (!c || (n && (!g1 || !g2)) || n > 0x7fffffffu) return ZKP_ERR_ARG;
    if (!n) {
        if (out_gt) memcpy(out_gt, GT_IDENTITY, 576);
        if (is_one) *is_one = 1;
        return ZKP_OK;
    }
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    Staged st = {nullptr, nullptr, nullptr, nullptr};
    if ((rc = stage_pairs(c, g1, g2, inf1, inf2, n, &st))) return rc;
    if ((rc = ensure(c, 4, 576))) return rc;
    if ((rc = product_check_dev(c, st.g1, st.g2, st.i1, st.i2, n, (uint64_t*)c->buf[4], c->d_flag, c->stream))) return rc;
    int one = 0;
    if (out_gt) HIPCHK(c, hipMemcpyAsync(out_gt, c->buf[4], 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&one, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (is_one) *is_one = one;
    return ZKP_OK;
}
int zkp_pairing_check_batch(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks,
                            size_t k, uint8_t* ok, int* all_ok) {
    if (!c || (n_checks && k && (!g1 || !g2))) return ZKP_ERR_ARG;
    if (all_ok) *all_ok = 1;
    if (!n_checks) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    // flags-only results need no large download: one shot is faster (measured) until the upload workspace gets large
    if (k && n_checks * k > 8 * c->host_slice && !c->validate) return host_sliced(c, g1, g2, inf1, inf2, n_checks, k, nullptr, ok, all_ok);
    Staged st = {nullptr, nullptr, nullptr, nullptr};
    if (k && (rc = stage_pairs(c, g1, g2, inf1, inf2, n_checks * k, &st))) return rc;
    if ((rc = ensure(c, 6, n_checks))) return rc;
    if ((rc = pairing_dev(c, st.g1, st.g2, st.i1, st.i2, n_checks, k, nullptr, (uint8_t*)c->buf[6], c->d_flag + 1, c->stream))) return rc;
    if (ok) HIPCHK(c, hipMemcpyAsync(ok, c->buf[6], n_checks, hipMemcpyDeviceToHost, c->stream));
    int flag = 1;
    HIPCHK(c, hipMemcpyAsync(&flag, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (all_ok) *all_ok = flag;
    return ZKP_OK;
}
static int valid_host(zkp_ctx* c, int which, const uint64_t* pts, const uint8_t* inf, size_t n, uint8_t* status) {
    if (!c || (n && (!pts || !status))) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    size_t sz = which == 1 ? 96 : 192;
    if ((rc = ensure(c, 0, n * sz)) || (rc = ensure(c, 6, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[0], pts, n * sz, hipMemcpyHostToDevice, c->stream));
    const uint8_t* di = nullptr;
    if (inf) {
        if ((rc = ensure(c, 2, n))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->buf[2], inf, n, hipMemcpyHostToDevice, c->stream));
        di = (const uint8_t*)c->buf[2];
    }
    if ((rc = validate_dev(c, (const uint64_t*)c->buf[0], n * sz / 48))) return rc;
    rc = which == 1 ? zkp_g1_is_valid_batch_dev(c, c->buf[0], di, n, c->buf[6], c->stream)
                    : zkp_g2_is_valid_batch_dev(c, c->buf[0], di, n, c->buf[6], c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(status, c->buf[6], n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_g1_is_valid_batch(zkp_ctx* c, const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* status) { return valid_host(c, 1, g1, inf, n, status); }
int zkp_g2_is_valid_batch(zkp_ctx* c, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* status) { return valid_host(c, 2, g2, inf, n, status); }

static int mul_host(zkp_ctx* c, int which, const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf) {
    size_t w = which == 1 ? 12 : 24;
    if (!c || (n && (!base || !sc || !out)) || (stride != 0 && stride != w)) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    size_t nb = stride ? n : 1;
    if ((rc = ensure(c, 0, nb * w * 8)) || (rc = ensure(c, 1, n * 32)) || (rc = ensure(c, 4, n * w * 8)) || (rc = ensure(c, 6, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[0], base, nb * w * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->buf[1], sc, n * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = validate_dev(c, (const uint64_t*)c->buf[0], nb * w / 6))) return rc;
    rc = which == 1 ? zkp_g1_mul_batch_dev(c, c->buf[0], stride, c->buf[1], n, c->buf[4], c->buf[6], c->stream)
                    : zkp_g2_mul_batch_dev(c, c->buf[0], stride, c->buf[1], n, c->buf[4], c->buf[6], c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->buf[4], n * w * 8, hipMemcpyDeviceToHost, c->stream));
    if (out_inf) HIPCHK(c, hipMemcpyAsync(out_inf, c->buf[6], n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_g1_mul_batch(zkp_ctx* c, const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf) {
    return mul_host(c, 1, base, stride, sc, n, out, out_inf);
}
int zkp_g2_mul_batch(zkp_ctx* c, const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf) {
    return mul_host(c, 2, base, stride, sc, n, out, out_inf);
}
static int codec_host(zkp_ctx* c, bool decode, int nfp, const void* in, const uint8_t* inf_in, size_t n, void* out, uint8_t* out_inf, uint8_t* status) {
    if (!c || (n && (!in || !out)) || (decode && n && (!out_inf || !status))) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    const size_t nb = n * 48 * nfp;
    if ((rc = ensure(c, 0, nb)) || (rc = ensure(c, 4, nb)) || (rc = ensure(c, 2, n)) || (rc = ensure(c, 6, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[0], in, nb, hipMemcpyHostToDevice, c->stream));
    if (decode) {
        hipLaunchKernelGGL(k_decode<true>, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const uint8_t*)c->buf[0], n, nfp, (uint64_t*)c->buf[4],
                           (uint8_t*)c->buf[2], (uint8_t*)c->buf[6]);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out, c->buf[4], nb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(out_inf, c->buf[2], n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(status, c->buf[6], n, hipMemcpyDeviceToHost, c->stream));
    } else {
        const uint8_t* di = nullptr;
        if (inf_in) {
            HIPCHK(c, hipMemcpyAsync(c->buf[2], inf_in, n, hipMemcpyHostToDevice, c->stream));
            di = (const uint8_t*)c->buf[2];
        }
        hipLaunchKernelGGL(k_encode<true>, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const uint64_t*)c->buf[0], di, n, nfp, (uint8_t*)c->buf[4]);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out, c->buf[4], nb, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}
int zkp_g1_decode_batch(zkp_ctx* c, const uint8_t* bytes, size_t n, uint64_t* out, uint8_t* out_inf, uint8_t* status) {
    return codec_host(c, true, 2, bytes, nullptr, n, out, out_inf, status);
}
int zkp_g2_decode_batch(zkp_ctx* c, const uint8_t* bytes, size_t n, uint64_t* out, uint8_t* out_inf, uint8_t* status) {
    return codec_host(c, true, 4, bytes, nullptr, n, out, out_inf, status);
}
int zkp_g1_encode_batch(zkp_ctx* c, const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* out) {
    return codec_host(c, false, 2, g1, inf, n, out, nullptr, nullptr);
}
int zkp_g2_encode_batch(zkp_ctx* c, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* out) {
    return codec_host(c, false, 4, g2, inf, n, out, nullptr, nullptr);
}

int zkp_points_check_batch(zkp_ctx* c, const uint8_t* g1_bytes, const uint8_t* g2_bytes, size_t n_checks, size_t k, uint8_t* st1, uint8_t* st2,
                           uint8_t* ok, int* all_ok) {
    if (!c || too_many(n_checks, k) || (n_checks && k && (!g1_bytes || !g2_bytes))) return ZKP_ERR_ARG;
    if (all_ok) *all_ok = 1;
    if (!n_checks) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    const size_t np = n_checks * k;
    if ((rc = ensure_pc(c, PC_BYTES, np * 288 + 8)) || (rc = ensure_pc(c, PC_ST1, np + 8)) || (rc = ensure_pc(c, PC_ST2, np + 8)) ||
        (rc = ensure_pc(c, PC_OK, n_checks + 8)))
        return rc;
    uint8_t* d1 = (uint8_t*)c->pc[PC_BYTES];
    uint8_t* d2 = d1 + np * 96;
    if (np) {
        HIPCHK(c, hipMemcpyAsync(d1, g1_bytes, np * 96, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d2, g2_bytes, np * 192, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = points_check_dev(c, d1, d2, n_checks, k, nullptr, nullptr, nullptr, c->d_flag + 1, c->stream))) return rc;
    if (st1 && np) HIPCHK(c, hipMemcpyAsync(st1, c->pc[PC_ST1], np, hipMemcpyDeviceToHost, c->stream));
    if (st2 && np) HIPCHK(c, hipMemcpyAsync(st2, c->pc[PC_ST2], np, hipMemcpyDeviceToHost, c->stream));
    if (ok) HIPCHK(c, hipMemcpyAsync(ok, c->pc[PC_OK], n_checks, hipMemcpyDeviceToHost, c->stream));
    int flag = 1;
    HIPCHK(c, hipMemcpyAsync(&flag, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (all_ok) *all_ok = flag;
    return ZKP_OK;
}

int zkp_fp_op_batch(zkp_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    const int base = op & ~ZKP_FP_CORE28;
    const bool unary = base == ZKP_FP_NEG || base == ZKP_FP_SQUARE || base == ZKP_FP_INVERT;
    if (!c || op < 0 || base > ZKP_FP_INVERT || n > 0x7fffffffu || (n && (!a || !out || (!unary && !b)))) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    if ((rc = ensure(c, 0, n * 48)) || (rc = ensure(c, 1, n * 48)) || (rc = ensure(c, 4, n * 48))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->buf[0], a, n * 48, hipMemcpyHostToDevice, c->stream));
    if ((rc = validate_dev(c, (const uint64_t*)c->buf[0], n))) return rc;
    if (!unary) {
        HIPCHK(c, hipMemcpyAsync(c->buf[1], b, n * 48, hipMemcpyHostToDevice, c->stream));
        if ((rc = validate_dev(c, (const uint64_t*)c->buf[1], n))) return rc;
    }
    if (op & ZKP_FP_CORE28) {   // the 14 x 28-bit carry-free core of the cooperative family
        HIPCHK(c, zkp::coop_fp28_op(base, (const uint64_t*)c->buf[0], (const uint64_t*)c->buf[1], n, (uint64_t*)c->buf[4], c->stream));
    } else {
        hipLaunchKernelGGL(k_fp_op, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, base, (const uint64_t*)c->buf[0], (const uint64_t*)c->buf[1], n, (uint64_t*)c->buf[4]);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(out, c->buf[4], n * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}

int zkp_tower_op_batch(zkp_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint32_t repeat, uint64_t* out) {
    const bool binary = op == ZKP_TOWER_FP2_MUL || op == ZKP_TOWER_FP6_MUL || op == ZKP_TOWER_FP12_MUL || op == ZKP_TOWER_FP12_MUL_BY_014 ||
                        op == ZKP_TOWER_FP2_MUL_FP || op == ZKP_TOWER_FP6_MUL_BY_1 || op == ZKP_TOWER_FP6_MUL_BY_01;
    if (!c || op < 0 || op > ZKP_TOWER_FP12_INVERT || n > 0x3fffffffu || (n && (!a || !out || (binary && !b)))) return ZKP_ERR_ARG;
    if (op == ZKP_TOWER_FP12_CYCLOTOMIC_DECOMPRESS && !zkp::coop_selected(&c->coop, c->kernel)) return ZKP_ERR_ARG;
    if (op == ZKP_TOWER_FP12_CYCLOTOMIC_POW2K && (repeat < 1 || repeat > 64)) return ZKP_ERR_ARG;
    if (!n) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    // one buffer: the a records, then the b records (the step programs read record check + n as their second operand)
    if ((rc = ensure(c, 5, 2 * n * 576)) || (rc = ensure(c, 4, n * 576))) return rc;
    uint64_t* d_a = (uint64_t*)c->buf[5];
    uint64_t* d_b = d_a + 72 * n;
    HIPCHK(c, hipMemcpyAsync(d_a, a, n * 576, hipMemcpyHostToDevice, c->stream));
    if (binary) HIPCHK(c, hipMemcpyAsync(d_b, b, n * 576, hipMemcpyHostToDevice, c->stream));
    if ((rc = validate_dev(c, d_a, (binary ? 2 : 1) * n * 12))) return rc;
    if (zkp::coop_selected(&c->coop, c->kernel)) {
        const hipError_t e = zkp::coop_tower_op(&c->coop, op, d_a, n, repeat, (uint64_t*)c->buf[4], c->stream);
        if (e != hipSuccess) { c->err = std::string("coop_tower_op: ") + hipGetErrorString(e); return ZKP_ERR_HIP; }
    } else {
        hipLaunchKernelGGL(k_tower_op, dim3(grid_for(n, TPB)), dim3(TPB), 0, c->stream, op, d_a, binary ? d_b : nullptr, n, repeat, (uint64_t*)c->buf[4]);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(out, c->buf[4], n * 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZKP_OK;
}

// ---------------------------------------------------------------- several GPUs behind one call
// one host thread drives n_ctx contexts (normally one per GPU of the node; several on one GPU work
// too), every context takes a contiguous block of the checks - a check's k pairs and its shared final exponentiation
// stay on one GPU - uploads it on its own stream, runs the same pipeline as zkp_pairing_check_batch / zkp_pairing_batch,
// and downloads its results; the per-context AND flags are combined on the host.  Nothing crosses xGMI: the data path
// has no collective.  (The one-process-per-GPU form of the same call is zkp_pairing_check_batch per rank plus ONE
// ncclAllReduce(count = 1, ncclInt32, ncclMin) of the flag - RCCL has no bitwise AND; INTEGRATION.md.)
static int multi_impl(zkp_ctx* const* ctxs, int n_ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                      size_t n_checks, size_t k, uint64_t* out_gt, uint8_t* ok, int* all_ok) {
    if (!ctxs || n_ctx <= 0 || n_ctx > 64 || too_many(n_checks, k) || (n_checks && k && (!g1 || !g2))) return ZKP_ERR_ARG;
    for (int j = 0; j < n_ctx; j++) {
        if (!ctxs[j]) return ZKP_ERR_ARG;
        for (int i = 0; i < j; i++)
            if (ctxs[i] == ctxs[j]) return ZKP_ERR_ARG;
    }
    if (all_ok) *all_ok = 1;
    if (!n_checks) return ZKP_OK;
    int flags[64];
    int rc = ZKP_OK, used = 0;
    const size_t base = n_checks / n_ctx, rem = n_checks % n_ctx;
    // workspace of every context first: an allocation synchronises its device, and must not do so between another context's
    // upload and launch.  After this pass the loop below only queues work: from page-locked arrays (zkp_host_alloc /
    // zkp_host_register) no context's start waits for another context's copy.
    for (int j = 0; j < n_ctx; j++) {
        zkp_ctx* c = ctxs[j];
        const size_t m = base + ((size_t)j < rem ? 1 : 0);
        if (!m) continue;
        if ((rc = bind(c))) return rc;
        if ((rc = ensure(c, 0, m * k * 96)) || (rc = ensure(c, 1, m * k * 192)) || (inf1 && (rc = ensure(c, 2, m * k))) ||
            (inf2 && (rc = ensure(c, 3, m * k))) || (out_gt && (rc = ensure(c, 4, m * 576))) || (rc = ensure(c, 6, m)))
            return rc;
    }
    for (int j = 0; j < n_ctx && rc == ZKP_OK; j++) {
        zkp_ctx* c = ctxs[j];
        const size_t lo = j * base + ((size_t)j < rem ? j : rem), m = base + ((size_t)j < rem ? 1 : 0), p0 = lo * k;
        flags[j] = 1;
        used = j + 1;
        if (!m) continue;
        if ((rc = bind(c))) break;
        if (c->ws_busy) (void)hipStreamWaitEvent(c->stream, c->ws_busy, 0);
        Staged st = {nullptr, nullptr, nullptr, nullptr};
        if (k && (rc = stage_pairs(c, g1 + 12 * p0, g2 + 24 * p0, inf1 ? inf1 + p0 : nullptr, inf2 ? inf2 + p0 : nullptr, m * k, &st))) break;
        if ((out_gt && (rc = ensure(c, 4, m * 576))) || (rc = ensure(c, 6, m))) break;
        if ((rc = pairing_dev(c, st.g1, st.g2, st.i1, st.i2, m, k, out_gt ? (uint64_t*)c->buf[4] : nullptr, (uint8_t*)c->buf[6], c->d_flag + 1, c->stream)))
            break;
        hipError_t e = hipSuccess;
        if (out_gt) e = hipMemcpyAsync(out_gt + 72 * lo, c->buf[4], m * 576, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && ok) e = hipMemcpyAsync(ok + lo, c->buf[6], m, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&flags[j], c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ws_busy, c->stream);
        if (e != hipSuccess) { c->err = std::string("zkp_pairing_*_multi: ") + hipGetErrorString(e); rc = ZKP_ERR_HIP; }
    }
    // whatever happened, nothing may still be copying from or into the caller's arrays when the call returns
    for (int j = 0; j < used; j++) {
        zkp_ctx* c = ctxs[j];
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            if (rc == ZKP_OK) { c->err = "zkp_pairing_*_multi: stream synchronisation failed"; rc = ZKP_ERR_HIP; }
        }
    }
    if (rc != ZKP_OK) return rc;
    int all = 1;
    for (int j = 0; j < used; j++) all &= flags[j] ? 1 : 0;
    if (all_ok) *all_ok = all;
    return ZKP_OK;
}
int zkp_pairing_check_batch_multi(zkp_ctx* const* ctxs, int n_ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                  size_t n_checks, size_t k, uint8_t* ok, int* all_ok) {
    return multi_impl(ctxs, n_ctx, g1, g2, inf1, inf2, n_checks, k, nullptr, ok, all_ok);
}
int zkp_pairing_batch_multi(zkp_ctx* const* ctxs, int n_ctx, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n,
                            uint64_t* out_gt, uint8_t* ok, int* all_ok) {
    if (n && !out_gt) return ZKP_ERR_ARG;
    return multi_impl(ctxs, n_ctx, g1, g2, inf1, inf2, n, 1, out_gt, ok, all_ok);
}

// ---------------------------------------------------------------- one rank per GPU: the RCCL collective behind the ABI
#define NCCLCHK(ctx, call)                                                                      \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + ncclGetErrorString(r__);                   \
            return ZKP_ERR_COMM;                                                                \
        }                                                                                       \
    } while (0)
static_assert(ZKP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the header's id size is RCCL's");

int zkp_comm_unique_id(void* out_id) {
    if (!out_id) return ZKP_ERR_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ZKP_ERR_COMM;
    memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ZKP_OK;
}
int zkp_comm_init_rank(zkp_ctx* c, int nranks, int rank, const void* unique_id) {
    if (!c || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return ZKP_ERR_ARG;
    if (c->comm) { c->err = "the context already has a communicator"; return ZKP_ERR_ARG; }
    int rc = bind(c);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    // EVERY rank enters the bootstrap first (round 6): ncclCommInitRank blocks until all nranks ranks have called it, so a rank that
    // returned on a local failure BEFORE it - round 5 allocated the all-gather records first - left its peers waiting for ever.  The
    // records are allocated behind it; a rank that cannot have them aborts its communicator and returns ZKP_ERR_OOM.  Its peers hold a
    // communicator with a member that is gone and cannot know: a failed zkp_comm_init_rank MUST be told to the other ranks by the
    // host's own means (bench.py: one gloo all-reduce of the status) BEFORE anyone enters a collective - include/zkp_pairings.h.
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        return ZKP_ERR_COMM;
    }
    if (hipMalloc((void**)&c->comm_buf, ((size_t)nranks + 3) * 576) != hipSuccess) {
        (void)hipGetLastError();
        (void)ncclCommAbort(c->comm);        // needs no peer (ncclCommDestroy may wait for them)
        c->comm = nullptr;
        c->comm_buf = nullptr;
        c->err = "zkp_comm_init_rank: no memory for the communicator's Fp12 records";
        return ZKP_ERR_OOM;
    }
    c->comm_nranks = nranks;
    c->comm_rank = rank;
    return ZKP_OK;
}
int zkp_comm_destroy(zkp_ctx* c) {
    if (!c) return ZKP_ERR_ARG;
    if (!c->comm) return ZKP_OK;
    int rc = bind(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ncclComm_t comm = c->comm;
    c->comm = nullptr;
    c->comm_nranks = c->comm_rank = 0;
    if (c->comm_buf) { (void)hipFree(c->comm_buf); c->comm_buf = nullptr; }
    NCCLCHK(c, ncclCommDestroy(comm));
    return ZKP_OK;
}
int zkp_comm_info(const zkp_ctx* c, int* nranks, int* rank) {
    if (!c) return ZKP_ERR_ARG;
    if (nranks) *nranks = c->comm ? c->comm_nranks : 0;
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    return ZKP_OK;
}
// AND of {0,1} flags = MIN (RCCL has no bitwise AND): ONE ncclAllReduce(count = 1, ncclInt32, ncclMin), in place
static int and_allreduce(zkp_ctx* c, int* d_flag, hipStream_t s) {
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    NCCLCHK(c, ncclAllReduce(d_flag, d_flag, 1, ncclInt32, ncclMin, c->comm, s));
    return ZKP_OK;
}
// the collective of the resident flavours after the rank's own work: a local failure turns the rank's flag into 0 and still joins
static int reduce_after_local(zkp_ctx* c, int rc_local, int* d_all_ok, hipStream_t s) {
    if (rc_local) {
        const std::string first = c->err;
        if (bind(c) != ZKP_OK || hipMemsetAsync(d_all_ok, 0, sizeof(int), s) != hipSuccess) return rc_local;      // nothing left to join with
        (void)and_allreduce(c, d_all_ok, s);
        c->err = first;
        return rc_local;
    }
    return and_allreduce(c, d_all_ok, s);
}
// ... and of the host-pointer flavours: the rank's local AND (0 when its own call failed) goes to the device, through the collective
// and back; the rank's own status is what the call returns
static int host_flag_allreduce(zkp_ctx* c, int rc_local, int local, int* all_ok) {
    if (rc_local) local = 0;
    const std::string first = c->err;
    int rc = bind(c);
    if (rc) return rc_local ? rc_local : rc;
    HostCall drain(c);
    // the flag reaches the device by a kernel argument, not by a copy that could fail before the collective: whatever happened locally,
    // this rank enters the all-reduce its peers are waiting in
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->d_flag + 1, local ? 1 : 0);
    if (hipGetLastError() != hipSuccess && hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream) != hipSuccess) {
        if (!rc_local) { c->err = "the AND flag could not be written to the device"; return ZKP_ERR_HIP; }
        return rc_local;      // nothing left to join with
    }
    if ((rc = and_allreduce(c, c->d_flag + 1, c->stream))) return rc_local ? rc_local : rc;
    int all = 0;
    HIPCHK(c, hipMemcpyAsync(&all, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *all_ok = all;
    if (rc_local) c->err = first;
    return rc_local;
}
int zkp_and_allreduce_dev(zkp_ctx* c, void* d_flag, void* stream) {
    if (!c || !d_flag) return ZKP_ERR_ARG;
    int rc = bind(c);
    if (rc) return rc;
    return and_allreduce(c, (int*)d_flag, S(stream));
}
int zkp_pairing_check_batch_allreduce_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n_checks, size_t k,
                                          void* ok, void* all_ok, void* stream) {
    if (!c || !all_ok) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    // a rank with an empty block still sets its flag to 1; a rank whose own block FAILED still takes part, with flag 0, so that no
    // peer hangs in the collective (every rank then reads 0; this rank returns its own error)
    const int rc_local = zkp_pairing_check_batch_dev(c, g1, g2, i1, i2, n_checks, k, ok, all_ok, stream);
    return reduce_after_local(c, rc_local, (int*)all_ok, S(stream));
}
int zkp_pairing_gt_check_batch_allreduce_dev(zkp_ctx* c, const void* g1, const void* g2, const void* i1, const void* i2, size_t n_checks, size_t k,
                                             void* out_gt, void* ok, void* all_ok, void* stream) {
    if (!c || !all_ok) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    const int rc_local = zkp_pairing_gt_check_batch_dev(c, g1, g2, i1, i2, n_checks, k, out_gt, ok, all_ok, stream);
    return reduce_after_local(c, rc_local, (int*)all_ok, S(stream));
}
int zkp_pairing_check_batch_allreduce(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n_checks,
                                      size_t k, uint8_t* ok, int* all_ok) {
    if (!c || !all_ok) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    // the rank's own block through the host-pointer entry point (uploads, kernels, downloads; it leaves the local AND in *all_ok),
    // then the flag goes back to the device for the collective: every rank calls this exactly once per global check, whatever
    // its own block's size or status - a rank that failed locally still takes part (with flag 0) so that no peer hangs
    int local = 1;
    const int rc_local = zkp_pairing_check_batch(c, g1, g2, inf1, inf2, n_checks, k, ok, &local);
    return host_flag_allreduce(c, rc_local, local, all_ok);
}
int zkp_points_check_batch_allreduce_dev(zkp_ctx* c, const void* b1, const void* b2, size_t n_checks, size_t k, void* st1, void* st2, void* ok,
                                         void* all_ok, void* stream) {
    if (!c || !all_ok) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    const int rc_local = zkp_points_check_batch_dev(c, b1, b2, n_checks, k, st1, st2, ok, all_ok, stream);
    return reduce_after_local(c, rc_local, (int*)all_ok, S(stream));
}
int zkp_points_check_batch_allreduce(zkp_ctx* c, const uint8_t* b1, const uint8_t* b2, size_t n_checks, size_t k, uint8_t* st1, uint8_t* st2,
                                     uint8_t* ok, int* all_ok) {
    if (!c || !all_ok) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    int local = 1;
    const int rc_local = zkp_points_check_batch(c, b1, b2, n_checks, k, st1, st2, ok, &local);
    return host_flag_allreduce(c, rc_local, local, all_ok);
}
int zkp_pairing_product_check_allgather(zkp_ctx* c, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2, size_t n,
                                        uint64_t* out_gt, int* is_one) {
    if (!c) return ZKP_ERR_ARG;
    if (!c->comm) { c->err = "no communicator: call zkp_comm_init_rank first"; return ZKP_ERR_COMM; }
    const bool bad_args = (n && (!g1 || !g2)) || n > 0x7fffffffu;
    if (bad_args) { n = 0; c->err = "zkp_pairing_product_check_allgather: null points / too many pairs"; }
    int rc = bind(c);
    if (rc) return rc;
    HostCall drain(c);
    const size_t R = (size_t)c->comm_nranks;
    // this rank's Miller product -> the communicator's record 0 (the identity for an empty block), gathered into records 1..R.  A rank
    // whose own part fails - bad arguments, staging, an allocation that runs out of memory, a launch error - still joins the collective,
    // with the ZERO record, so that the product is 0 and every rank reads is_one = 0 (no peer hangs, none passes a check this rank's
    // pairs never entered), and returns its own error afterwards.  The records the collective itself needs belong to the communicator
    // (zkp_comm_init_rank allocated them): nothing that can fail stands between this point and ncclAllGather.
    uint64_t* const mine = c->comm_buf;
    uint64_t* const gathered = c->comm_buf + 72;
    uint64_t* const total = c->comm_buf + 72 * (R + 1);
    uint64_t* const gt = c->comm_buf + 72 * (R + 2);
    Staged st = {nullptr, nullptr, nullptr, nullptr};
    int rc_local = bad_args ? ZKP_ERR_ARG : ZKP_OK;
    if (!rc_local && n) rc_local = stage_pairs(c, g1, g2, inf1, inf2, n, &st);
    if (!rc_local) rc_local = miller_product_dev(c, st.g1, st.g2, st.i1, st.i2, n, mine, c->stream);
    const std::string first = c->err;
    if (rc_local) {
        if (bind(c) != ZKP_OK || hipMemsetAsync(mine, 0, 576, c->stream) != hipSuccess) return rc_local;
    }
    NCCLCHK(c, ncclAllGather(mine, gathered, 72, ncclUint64, c->comm, c->stream));
    if ((rc = fp12_product_dev(c, gathered, R, total, c->stream))) return rc_local ? rc_local : rc;
    if ((rc = final_exp_dev(c, total, 1, gt, c->stream))) return rc_local ? rc_local : rc;
    hipLaunchKernelGGL(k_gt_is_one, dim3(1), dim3(64), 0, c->stream, (const uint64_t*)gt, c->d_flag);
    HIPCHK(c, hipGetLastError());
    int one = 0;
    if (out_gt) HIPCHK(c, hipMemcpyAsync(out_gt, gt, 576, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&one, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (is_one) *is_one = one;
    if (rc_local) c->err = first;
    return rc_local;
}

// page-locked host memory for the host-pointer entry points (header: "pinned host memory")
int zkp_host_alloc(size_t bytes, void** out_ptr) {
    if (!out_ptr || !bytes) return ZKP_ERR_ARG;
    *out_ptr = nullptr;
    const bool ok = hipHostMalloc(out_ptr, bytes, hipHostMallocPortable) == hipSuccess;
    if (ok) zkp_dbg_alloc("host_alloc", *out_ptr, bytes);
    return ok ? ZKP_OK : ZKP_ERR_HIP;
}
int zkp_host_free(void* ptr) {
    if (ptr) zkp_dbg_alloc("host_free", ptr, 0);
    return !ptr || hipHostFree(ptr) == hipSuccess ? ZKP_OK : ZKP_ERR_HIP;
}
int zkp_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return ZKP_ERR_ARG;
    // whole pages of its own only (header: pinned host memory): registrations that share a page corrupt the runtime's bookkeeping
    static const long page = sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096;
    if (((uintptr_t)ptr | bytes) & (uintptr_t)(page - 1)) return ZKP_ERR_ARG;
    zkp_dbg_alloc("host_register", ptr, bytes);
    return hipHostRegister(ptr, bytes, hipHostRegisterPortable) == hipSuccess ? ZKP_OK : ZKP_ERR_HIP;
}
int zkp_host_unregister(void* ptr) {
    if (!ptr) return ZKP_ERR_ARG;
    zkp_dbg_alloc("host_unregister", ptr, 0);
    return hipHostUnregister(ptr) == hipSuccess ? ZKP_OK : ZKP_ERR_HIP;
}

// measurement helper: a one-wavefront clock probe on `stream` (asynchronous); d_out receives two u64: shader clock ticks
// and wall clock ticks over about spin_us microseconds; *wall_khz (host) is the wall clock's rate.
int zkp_clock_probe_dev(zkp_ctx* c, void* stream, unsigned spin_us, void* d_out, int* wall_khz) {
    if (!c || !d_out || !wall_khz || spin_us == 0 || spin_us > 1000000u) return ZKP_ERR_ARG;
    int rc = bind(c);
    if (rc) return rc;
    int khz = 0;
    HIPCHK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) { c->err = "wall clock rate unknown"; return ZKP_ERR_HIP; }
    *wall_khz = khz;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, S(stream), (unsigned long long*)d_out, (unsigned long long)spin_us * (unsigned long long)khz / 1000ull);
    HIPCHK(c, hipGetLastError());
    return ZKP_OK;
}

int zkp_time_coop_step(zkp_ctx* c, int which, size_t n, float* ms) {
    if (!c || !ms) return ZKP_ERR_ARG;
    int rc = bind(c);
    if (rc) return rc;
    HostCall hc(c);   // the timing programs reuse the workspace of pipeline 0: wait for *_dev calls queued on other streams, drain on exit
    HIPCHK(c, zkp::coop_time_prog(&c->coop, which, n, c->stream, c->ev0, c->ev1, ms));
    return ZKP_OK;
}

int zkp_time_pairing_dev(zkp_ctx* c, const void* g1, const void* g2, size_t n, void* out, int reps, float* avg_ms) {
    if (!c || !g1 || !g2 || !out || reps <= 0 || !avg_ms) return ZKP_ERR_ARG;
    int rc = bind(c);
    if (rc) return rc;
    HostCall hc(c);   // same workspace discipline as every host-synchronous entry point
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++)
        if ((rc = pairing_dev(c, (const uint64_t*)g1, (const uint64_t*)g2, nullptr, nullptr, n, 1, (uint64_t*)out, nullptr, nullptr, c->stream))) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / reps;
    return ZKP_OK;
}


int zkp_profile_pairing_dev(zkp_ctx* c, const void* g1, const void* g2, size_t n, void* out, float* ms, int* launches) {
    if (!c || !g1 || !g2 || !out || !ms || !launches || !n || too_many(n)) return ZKP_ERR_ARG;
    if (!zkp::coop_selected(&c->coop, c->kernel)) { c->err = "zkp_profile_pairing_dev: the cooperative kernel family is not selected"; return ZKP_ERR_ARG; }
    int rc = bind(c);
    if (rc) return rc;
    HostCall hc(c);
    return coop_rc(c, "coop_profile_pairing", zkp::coop_profile_pairing(&c->coop, (const uint64_t*)g1, (const uint64_t*)g2, n, (uint64_t*)out, ms, launches, c->stream));
}

}  // extern "C"
