// zkp_coop.hip -- lane-cooperative kernel family (gfx950): the throughput path of the pairing engine.
//
// Pipeline for a chunk of checks (each check = k pairs sharing one Fp12 accumulator):
//   k_prep_lines   two lanes per PAIR (one per Fp2 coefficient): walks the G2 point through the 68 doubling/addition steps of
//                  the optimal-ate loop (ePrint 2010/354 Alg. 26/27), scales every line by P and
//                  streams (c2, c1*xP, c0*yP) to HBM as 28-bit-limb records (coalesced).
//   k_coop(prog)   one check per GROUP of 12 lanes (5 groups per wavefront); lane j owns Fp12
//                  coefficient j; all tower values live in LDS; a table-driven interpreter executes
//                  the step program generated (and verified on the CPU) by tools/coopgen.py:
//                    MULACC  sum of <=12 products into 64-bit columns, ONE Montgomery reduction
//                    LIN     limb-wise linear combination (no carries), weak normalisation
//                    GLOAD/GSTORE  line stream / per-check state / wire format
//                  the optional epilogue (alpha r + beta E, renormalised) and the companion store
//                  (x0 + x1 / x0 - x1 of the lane pair's Fp2 coefficient, so that later steps fetch their
//                  operand forms ready-made) ride on the MULACC step.
//   k_batch_inv    batched Fp inversions (Montgomery's trick, up to 32 values per lane, division steps): the single inversion
//                  of the final exponentiation and the six per x-power chain of the decompression.
//   k_ksq          the 63 cyclotomic squarings of an x-power chain in Karabina's compressed form: four lanes per check (one Fp2
//                  product each), 16 checks per wavefront, operands in registers, products exchanged by DPP, snapshots of the
//                  running value at the set bits of |x|.
//   k_kdec_a / _b  decompression of the snapshots (two lanes per snapshot) around one k_batch_inv call.
// Programs: miller{k}_{state|wire}, f12mul_{state|wire|pairs}, fexp_a_{state|wire}, fexp_c0..5, tw_* (zkp_coop_prog.inc).
// Host side: a super-chunk of up to 2^20 checks shares one state buffer; phase A (lines, Miller loop, fexp_a) runs per
// 2^16-check chunk on two HIP streams, then ONE k_batch_inv and the phase C plan (ZKP_FEXP_C_PLAN: six step programs
// alternating with the five x-power chains = k_ksq, k_kdec_a, k_batch_inv, k_kdec_b) cover the super-chunk (two_phase).
//
// Reference anchors: Fp12::mul_by_014 src/fp12.rs:99-111, Fp12::square :173-184, Fp12::invert
// :186-190 (+ src/fp6.rs:291-309, src/fp2.rs:278-296), conjugate :123-125; pairing semantics
// SURVEY.md S6 (src/pairings.rs is empty upstream).
#include "zkp_coop.hpp"
#include "zkp_plan.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "zkp_coop_prog.inc"
#include "zkp_fp28.hpp"
#ifndef ZKP_COOP_ASM
#define ZKP_COOP_ASM 1   // the MULACC step of k_coop as one hand-scheduled inline-asm block (tools/coopasm.py -> zkp_coop_mulacc.inc);
                         // 0 builds the C++ term loop below it, the A/B baseline
#endif
#if ZKP_COOP_ASM
#include "zkp_coop_mulacc.inc"
#endif
#ifndef ZKP_PREP_ASM
#define ZKP_PREP_ASM ZKP_COOP_ASM   // the doubling step of k_prep_lines<true> as one hand-allocated asm block (tools/prepasm.py ->
                                    // zkp_prep_dbl.inc); 0: the compiled step (dbl_step_cln), the A/B baseline
#endif
#if ZKP_PREP_ASM
#include "zkp_prep_dbl.inc"
#endif
#ifndef ZKP_VALID_ASM
#define ZKP_VALID_ASM ZKP_COOP_ASM   // the Jacobian doubling / mixed addition of the subgroup checks as asm blocks (tools/validasm.py ->
                                     // zkp_valid_steps.inc): k_g1_valid_fast / k_g2_valid_fast; 0: the compiled kernels only (round 3)
#endif
#if ZKP_VALID_ASM
#include "zkp_valid_steps.inc"
#endif

using namespace zkp28;

namespace {

#ifndef ZKP_COOP_WG_WAVES
#define ZKP_COOP_WG_WAVES 1   // wavefronts per k_coop workgroup: 1 = five checks per wavefront, lanes 60..63 idle; 3 = sixteen checks per
                              // workgroup - lanes 60..63 of the three wavefronts together run the sixteenth check, and every step that
                              // touches LDS is fenced by workgroup barriers (its operand reads before its in-place result store, its
                              // stores before the next step's reads).  Bit-exact and measured: 6.25 % fewer
                              // wavefronts, but they spend 21 % of their cycles at the barriers - the Miller program alone runs 12 %
                              // slower, the 2^20-pair pass the same (256.8 against 256.0 ms).  Kept as a build knob.
#endif
constexpr int WGW = ZKP_COOP_WG_WAVES;
static_assert(WGW == 1 || WGW == 3, "one wavefront (5 checks) or three (16 checks) per workgroup");
constexpr int GROUPS = WGW == 3 ? 16 : 5;   // checks per workgroup
constexpr int LIG = ZKP_COOP_G;        // 12 lanes per group
constexpr int ST_SIZE = ZKP_COOP_ST_SIZE;
constexpr int NLINES = ZKP_COOP_NLINES;
static_assert(zkp::plan::NLINES == NLINES && zkp::plan::ST_SIZE == ST_SIZE && zkp::plan::GROUPS == (WGW == 3 ? 16 : 5),
              "zkp_plan.hpp (the host's planning arithmetic, checked under sanitizers on the CPU) follows the generated programs");

enum { OP_END = 0, OP_MULACC = 1, OP_LIN = 2, OP_GLOAD = 3, OP_GSTORE = 4, OP_LOOP = 5, OP_ENDLOOP = 6,
       OP_PLOOP = 7, OP_PENDLOOP = 8 };   // round 5: a loop over the k pairs of the launch (run-time count); its counter offsets the line loads
enum { K_LINE = 0, K_STATE = 1, K_WIRE = 2, K_WIRE2 = 3 };   // K_WIRE2: wire record of check + chk_off

__device__ __constant__ const int32_t K_PBAL[NL] = {ZKP_COOP_P_BAL};

// Round 6: the number of checks a launch works on may live in DEVICE memory (zkp_points_check_batch_dev: the checks whose points are all
// valid are counted on the device and never read back).  A launch is planned on the host for the worst case - n checks from `base` of
// the list on - and every kernel takes min(n, *cnt - base) at its entry; wavefronts behind that leave at once.  cnt == nullptr: n as given.
struct NDev { const uint32_t* cnt; uint32_t base; };
__device__ __forceinline__ uint32_t eff_n(uint32_t n, const NDev& nd) {
    if (!nd.cnt) return n;
    const uint32_t m = *nd.cnt;                 // wave-uniform: a scalar load
    const uint32_t left = m > nd.base ? m - nd.base : 0u;
    return left < n ? left : n;
}

struct CoopArgs {
    const uint32_t* hdr;
    const uint32_t* tbl;
    const uint4* rtbl;       // resolved MULACC table: row (table offset / 12 + term) x 64 lanes of {A1 address, B1 address, A2 | B2 << 16, 0};
                             // the row behind a step's terms holds its per-lane flag words (coop_resolve_table)
    const int4* consts;      // NCONST records of 4 int4
    const int4* lines;       // [(step * k + pair) * 6 + c][check] records of 4 int4
    int4* state;             // [elem][check] records of 4 int4
    const uint64_t* wire_in;  // n_checks x 72
    uint64_t* wire_out;       // n_checks x 72 (may be null)
    uint8_t* ok;              // may be null
    int* all_ok;              // may be null
    uint32_t n_checks;
    uint32_t nc;              // record stride of the state buffer (>= n_checks: a super-chunk's state is shared by its chunks)
    uint32_t k;
    uint32_t S;               // LDS plane stride (int4) of a group region (= the program's slot count)
    uint32_t nconst;          // constants the program references (prefix of the table)
    uint32_t st_off;          // added to every K_STATE element index (where a group's Miller value lands)
    uint32_t chk_off;         // K_WIRE2 loads read the wire record of check + chk_off (product tree)
    NDev nd;                  // device-resident check count (or {nullptr, 0})
};

// ---- LDS access: quad-plane SoA, record = 4 x int4 at off, off+S, off+2S, off+3S
__device__ __forceinline__ void lds_ld(int32_t* x, const int4* lds, int off, int S) {
    int4 v0 = lds[off], v1 = lds[off + S], v2 = lds[off + 2 * S], v3 = lds[off + 3 * S];
    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w;
    x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    x[8] = v2.x; x[9] = v2.y; x[10] = v2.z; x[11] = v2.w;
    x[12] = v3.x; x[13] = v3.y;
}
__device__ __forceinline__ void lds_st(int4* lds, int off, int S, const int32_t* x) {
    lds[off] = make_int4(x[0], x[1], x[2], x[3]);
    lds[off + S] = make_int4(x[4], x[5], x[6], x[7]);
    lds[off + 2 * S] = make_int4(x[8], x[9], x[10], x[11]);
    lds[off + 3 * S] = make_int4(x[12], x[13], 0, 0);
}

// value renormalisation (tools/coopgen.py vred): q = round(top / p_top); x -= q p; weak_norm
__device__ __forceinline__ void vred(int32_t* x) {
    int32_t q = ((x[NL - 1] >> ZKP_COOP_VRED_SHIFT_IN) * ZKP_COOP_VRED_C + (1 << (ZKP_COOP_VRED_SHIFT_OUT - 1))) >> ZKP_COOP_VRED_SHIFT_OUT;
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] -= q * K_PBAL[i];
    weak_norm(x);
}

__device__ __forceinline__ int sext4(uint32_t v) { return ((int32_t)(v << 28)) >> 28; }
__device__ __forceinline__ int sext8(uint32_t v) { return ((int32_t)(v << 24)) >> 24; }

// canonical [0,p) unsigned 28-bit limbs from a reduced value in (-p, 2p) given as balanced limbs
__device__ __forceinline__ void canon28(uint32_t* f, const int32_t* x) {
    int32_t u[NL], y[NL], z[NL];
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = x[i] + carry;
        if (i < NL - 1) { u[i] = v & MASK; carry = v >> W; } else u[i] = v;
    }
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
    bool neg = u[NL - 1] < 0, ge = y[NL - 1] >= 0;
#pragma unroll
    for (int i = 0; i < NL; i++) f[i] = (uint32_t)(neg ? z[i] : (ge ? y[i] : u[i]));
}

__host__ __device__ constexpr int coop_group_stride(int S) { return S + ((4 - S % 8) + 8) % 8; }
// (+ 16 bytes behind the image with three wavefronts: the identity flags of the check that straddles them)
constexpr int coop_cfg_slots(uint32_t cfg) { return cfg == 2 ? ZKP_COOP_DEEP_NSLOT : cfg == 1 ? ZKP_COOP_WIDE_NSLOT : ZKP_COOP_NSLOT; }
constexpr int coop_cfg_consts(uint32_t cfg) { return cfg == 2 ? ZKP_COOP_DEEP_NCONST : cfg == 1 ? ZKP_COOP_WIDE_NCONST : ZKP_COOP_NCONST; }
constexpr size_t coop_lds_bytes(int S, int SC) { return (size_t)4 * (SC + GROUPS * coop_group_stride(S)) * 16 + (WGW > 1 ? 16 : 0); }
#ifndef ZKP_COOP_KARATSUBA
#define ZKP_COOP_KARATSUBA 1   // acc_mul_k: 147 multiply-adds per product block instead of 196 (zkp_fp28.hpp); measured on one box,
                               // 2^20-pair pass: 296.5 ms -> 283.3 ms
#endif
#ifndef ZKP_COOP_WAVES
#if ZKP_COOP_ASM
#define ZKP_COOP_WAVES 3   // the asm block pins 156 VGPRs (80 Karatsuba accumulators, two operand sets); with the per-lane context
                           // re-derived per step the kernel needs 166 = 3 waves per SIMD, no spills (12 x 11-12 KB of LDS per CU)
#else
#define ZKP_COOP_WAVES 2   // C++ term loop: register bound only - it allocates 168 VGPRs = 3 waves per SIMD under bound 2.  Measured
                           // alternatives: bound 3 -> 4 spilled VGPRs, 290.7 ms; bound 4 (128 VGPRs, 86 spilled) -> 925 ms; without
                           // Karatsuba 4 waves x 128 VGPRs: 296.5 ms
#endif
#endif
// S slots per group and SC constants: two instantiations - <24, 34> for programs that need the whole constants table (11,136 B of
// LDS per wavefront), <30, 4> for the Miller programs (30 slots, 4 constants; 11,776 B); twelve wavefronts per CU either way
template <int S, int SC>
__global__ void __launch_bounds__(64 * WGW, ZKP_COOP_WAVES) k_coop(CoopArgs A) {
    extern __shared__ int4 lds[];
    const uint32_t n_checks = eff_n(A.n_checks, A.nd);     // wave-uniform; also the stride of the line buffer (k_prep_lines takes the same)
    if (blockIdx.x * GROUPS >= n_checks) return;          // the whole workgroup: nothing below is reached by part of it
    const int lane = threadIdx.x;        // 0 .. 64 * WGW - 1: the lane number within the workgroup
    // per-lane values, all functions of the lane number.  With the asm MULACC block (ZKP_COOP_ASM) they are re-derived at the top of
    // every step and again behind the block from an opaque copy of `lane` (ZKP_LANE_CTX): kept in registers across the block they
    // would cost the kernel its third wavefront per SIMD (the block owns 156 of the 168 VGPRs)
    int grp, lig, gbase;
    uint32_t check;
    bool lane_ok, active;
    // S, SC: compile-time plane strides (the q * stride offsets fold into the ds_read/ds_write immediates); A.nconst
    // of the SC constants are uploaded
    // LDS image: four planes (limb quads) of PS records; a plane holds the SC constants, then SG records per group.  ONE plane
    // stride for constants and slots: the plane offsets of every ds_read / ds_write are immediates and an operand's address is
    // (its group's base or 0) + its number.  SG = S rounded up to 4 mod 8: the lane groups of a ds_read_b128 (lanes of up to
    // three check groups) then fall on different bank quads for neighbouring slots.
    constexpr int SG = coop_group_stride(S), PS = SC + GROUPS * SG;
#if ZKP_COOP_ASM
#define ZKP_LANE_OPAQUE(l) asm volatile("" : "+v"(l))
#else
#define ZKP_LANE_OPAQUE(l) (void)0
#endif
#define ZKP_LANE_CTX()                                                                                  \
    do {                                                                                                \
        int l_ = lane;                                                                                  \
        ZKP_LANE_OPAQUE(l_);                                                                            \
        if (WGW == 1) {                                                                                 \
            grp = (l_ * 43) >> 9;             /* lane / 12 for lane < 64 */                             \
            lig = l_ - grp * LIG;             /* lanes 60..63: grp 5, lig 0..3 (never store) */         \
            lane_ok = grp < GROUPS;                                                                     \
        } else {                              /* wavefront w: checks 5w .. 5w+4; its lanes 60..63 are lanes 4w .. 4w+3 of check 15 */ \
            const int wv_ = l_ >> 6, wl_ = l_ & 63, g0_ = (wl_ * 43) >> 9;                              \
            grp = g0_ < 5 ? wv_ * 5 + g0_ : 15;                                                         \
            lig = g0_ < 5 ? wl_ - g0_ * LIG : wv_ * 4 + (wl_ - 60);                                     \
            lane_ok = true;                                                                             \
        }                                                                                               \
        check = blockIdx.x * GROUPS + grp;                                                              \
        active = lane_ok && check < n_checks;                                                           \
        gbase = SC + (lane_ok ? grp : GROUPS - 1) * SG;                                                 \
    } while (0)
    // with several wavefronts per workgroup one check's lanes sit in all of them: LDS reads and writes of a step are fenced
// (LDS counter only: a global load in flight need not land before the barrier)
#define ZKP_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define ZKP_WG_FENCE() do { if (WGW > 1) ZKP_WG_BARRIER(); } while (0)

    for (int i = lane; i < (int)A.nconst * 4; i += 64 * WGW) lds[(i & 3) * PS + (i >> 2)] = A.consts[i];
    __syncthreads();
#ifndef ZKP_COOP_IDLE_LANES_OFF
#define ZKP_COOP_IDLE_LANES_OFF 1
#endif
#if ZKP_COOP_IDLE_LANES_OFF
    // round 5: lanes 60..63 of a one-wavefront workgroup own nothing (five checks of twelve lanes) - they used to run every instruction
    // on group 4's operands and store nothing.  They leave here: EXEC never holds them again (every asm block restores the EXEC it was
    // entered with; no DPP or ballot of a live lane reads them: they are a quad of their own), and their multiply-adds stop drawing power.
    if (WGW == 1 && lane >= GROUPS * LIG) return;
#endif
#if ZKP_COOP_ASM
    // the lanes that own a coefficient of a live check: wave-uniform (two SGPRs), the store mask of the asm MULACC block
    ZKP_LANE_CTX();
    const unsigned long long act_lanes = __ballot(active);
#endif

    // the step headers are read-only and wave-uniform: through the constant address space they become scalar loads
    // (s_load_dwordx4 into SGPRs, scalar cache) instead of a vector load + readfirstlane per word
    typedef const __attribute__((address_space(4))) uint32_t* chdr_t;
    chdr_t hdr = (chdr_t)(uintptr_t)A.hdr;
    const uint32_t* __restrict__ tbl = A.tbl;
    uint32_t cursor = 0;
    int pc = 0, loop_pc = 0, loop_left = 0;
    int ploop_pc = 0, ploop_left = 0;
    uint32_t pair6 = 0;      // 6 x the pair loop's counter (0 outside a pair loop): wave-uniform
    auto slot_off = [&](uint32_t s) -> int { return ((s & 64) ? 0 : gbase) + (int)(s & 63); };
    auto ld = [&](int32_t* x, uint32_t s) { lds_ld(x, lds, slot_off(s), PS); };

    for (;;) {
        const uint32_t h0 = hdr[4 * pc], h1 = hdr[4 * pc + 1], off = hdr[4 * pc + 2];
        const uint32_t op = h0 & 0xff, arg = (h0 >> 8) & 0xff;
        if (op == OP_END) break;
        ZKP_LANE_CTX();
        if (op == OP_MULACC) {
            // software pipeline: table words two terms ahead, LDS operands one term ahead, so the
            // 196 multiply-adds of term t cover the latency of everything term t+1 needs
            const uint32_t h3 = hdr[4 * pc + 3];
            const uint32_t T = arg;
#if ZKP_COOP_ASM
            // one inline-asm block (tools/coopasm.py): term loop on two operand register sets with resolved per-lane LDS
            // addresses, Karatsuba fold, row-pipelined Montgomery reduction, limb extraction; the same integers as
            // acc_mul_k / acc_fold / acc_reduce of the C++ variant below
            const uint4* rt = A.rtbl + (size_t)(off / LIG) * (64 * WGW);
            // round 4: a step without epilogue stores its result and its companion form from inside the block (resolved
            // addresses in the flag row of the table, the active lanes as a mask); with several wavefronts per workgroup the
            // stores must wait for the workgroup's fence, so they stay behind the block
            const uint32_t nost = (h1 & 1u) | (WGW > 1 ? 1u : 0u);
            int32_t r[NL];
            {
                static_assert(NL == 14, "the generated block is for 14 limbs");
                constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};
                asm volatile(ZKP_MULACC_ASM
                             : ZKP_MULACC_OUTS(r)
                             : [rt] "s"(rt), [T] "s"(T), [h1] "s"(h1), [h3] "s"(h3), [lane16] "v"(lane * 16),   // (rematerialised from the lane number: not a live value)
                               [act] "s"(act_lanes), [nost] "s"(nost),
                               [ps1] "i"(PS * 16), [ps2] "i"(PS * 32), [ps3] "i"(PS * 48), [row] "i"(1024 * WGW),
                               [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]), [p6] "s"(PL[6]),
                               [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]), [p12] "s"(PL[12]),
                               [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)
                             : ZKP_MULACC_CLOBBERS);
            }
            if (!nost) { pc++; continue; }
            ZKP_LANE_CTX();
            const uint32_t ew = tbl[off + T * LIG + lig];
#else
            Acc acc;
            acc_zero(acc);
#if ZKP_COOP_KARATSUBA
            AccMid mid;
            mid_zero(mid);
#endif
            // software pipeline: the table word of term t + 2 is requested at the top of term t and taken over at its
            // end, behind the multiply-adds (the table is padded: the read past the last term is harmless); the LDS
            // operands of term t + 1 are requested before the multiply-adds of term t
            uint32_t w = tbl[off + lig];
            uint32_t wn = tbl[off + LIG + lig];
            const uint32_t ew = tbl[off + T * LIG + lig];
            int32_t xa[NL], xb[NL];
            ld(xa, w & 127);
            ld(xb, (w >> 14) & 127);
#pragma unroll 1
            for (uint32_t t = 0; t < T; t++) {
                const uint32_t w2 = tbl[off + (t + 2) * LIG + lig];
                const bool no_a2 = (h3 >> t) & 1, no_b2 = (h3 >> (12 + t)) & 1;   // wave-uniform
                const bool no_neg = (h1 >> (4 + t)) & 1;                          // no lane negates this term
                const bool has_da = (h1 >> (16 + t)) & 1;                         // some lane doubles its A operand
                const int32_t ma = -(int32_t)((w >> 28) & 1), mb = -(int32_t)((w >> 29) & 1), mn = -(int32_t)((w >> 30) & 1);
                int32_t a[NL], b[NL];
                if (no_a2 && no_neg) {
#pragma unroll
                    for (int i = 0; i < NL; i++) a[i] = xa[i];
                } else if (no_a2) {
#pragma unroll
                    for (int i = 0; i < NL; i++) a[i] = (xa[i] ^ mn) - mn;
                } else if (no_neg) {
                    int32_t x2[NL];
                    ld(x2, (w >> 7) & 127);
#pragma unroll
                    for (int i = 0; i < NL; i++) a[i] = xa[i] + ((x2[i] ^ ma) - ma);
                } else {
                    int32_t x2[NL];
                    ld(x2, (w >> 7) & 127);
#pragma unroll
                    for (int i = 0; i < NL; i++) a[i] = ((xa[i] + ((x2[i] ^ ma) - ma)) ^ mn) - mn;
                }
                if (has_da) {
                    const uint32_t sh = w >> 31;
#pragma unroll
                    for (int i = 0; i < NL; i++) a[i] = (int32_t)((uint32_t)a[i] << sh);
                }
                if (no_b2) {
#pragma unroll
                    for (int i = 0; i < NL; i++) b[i] = xb[i];
                } else {
                    int32_t x2[NL];
                    ld(x2, (w >> 21) & 127);
#pragma unroll
                    for (int i = 0; i < NL; i++) b[i] = xb[i] + ((x2[i] ^ mb) - mb);
                }
                if (t + 1 < T) {
                    ld(xa, wn & 127);
                    ld(xb, (wn >> 14) & 127);
                }
#if ZKP_COOP_KARATSUBA
                acc_mul_k(acc, mid, a, b);
#else
                acc_mul(acc, a, b);
#endif
                w = wn;
                wn = w2;
            }
#if ZKP_COOP_KARATSUBA
            acc_fold(acc, mid);
#endif
            int32_t r[NL];
            acc_reduce(r, acc);
#endif
            if (h1 & 1) {  // step-uniform: epilogue dst = alpha r + beta E, renormalised
                int32_t e[NL];
                ld(e, (ew >> 16) & 127);
                ZKP_WG_FENCE();   // every operand of the step has been read
                const int32_t al = sext4((ew >> 8) & 15), be = sext4((ew >> 12) & 15);
                // value renormalisation folded in: q = round(value / p) from the top limb of the combination
                // (the lower limbs are normalised: their carries cannot move q), then ONE weak normalisation
                const int32_t top = al * r[NL - 1] + be * e[NL - 1];
                const int32_t q = ((top >> ZKP_COOP_VRED_SHIFT_IN) * ZKP_COOP_VRED_C + (1 << (ZKP_COOP_VRED_SHIFT_OUT - 1))) >> ZKP_COOP_VRED_SHIFT_OUT;
#pragma unroll
                for (int i = 0; i < NL; i++) r[i] = al * r[i] + be * e[i] - q * K_PBAL[i];
                weak_norm(r);
            } else {
                ZKP_WG_FENCE();
            }
            if (active && ((ew >> 7) & 1)) lds_st(lds, gbase + (int)(ew & 63), PS, r);
            if (h1 & 2) {  // step-uniform: companion store of a squaring run - the even lane of an Fp2 coefficient keeps
                           // x0 + x1, the odd lane x0 - x1, so that the next squaring reads its operand forms ready-made
                int32_t c2[NL];
                const bool odd = lig & 1;
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int32_t o = __builtin_amdgcn_update_dpp(0, r[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
                    c2[i] = odd ? o - r[i] : r[i] + o;
                }
                if (active && ((ew >> 29) & 1)) lds_st(lds, gbase + (int)((ew >> 23) & 63), PS, c2);
            }
            ZKP_WG_FENCE();
        } else if (op == OP_LIN) {
            int32_t r[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) r[i] = 0;
#pragma unroll 1
            for (uint32_t t = 0; t < arg; t++) {
                const uint32_t w = tbl[off + t * LIG + lig];
                int32_t x[NL];
                ld(x, w & 127);
                const int32_t c = sext8((w >> 8) & 0xff);
#pragma unroll
                for (int i = 0; i < NL; i++) r[i] += c * x[i];
            }
            weak_norm(r);
            if (h1 & 1) vred(r);   // only where the generator's static value bound asks for it
            const uint32_t ew = tbl[off + arg * LIG + lig];
            ZKP_WG_FENCE();
            if (active && ((ew >> 7) & 1)) lds_st(lds, gbase + (int)(ew & 63), PS, r);
            ZKP_WG_FENCE();
        } else if (op == OP_GLOAD) {
            const uint32_t w = tbl[off + lig];
            const uint32_t idx = w >> 8;
            if (active && ((w >> 7) & 1)) {
                int32_t x[NL];
                if (arg == K_WIRE || arg == K_WIRE2) {
                    const uint64_t* src = A.wire_in + ((size_t)check + (arg == K_WIRE2 ? A.chk_off : 0u)) * 72 + idx * 6;
                    uint64_t ww[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) ww[i] = src[i];
#pragma unroll
                    for (int i = 0; i < NL; i++) {
                        const int bit = W * i, word = bit >> 6, sh = bit & 63;
                        uint64_t v = ww[word] >> sh;
                        if (sh > 64 - W && word + 1 < 6) v |= ww[word + 1] << (64 - sh);
                        x[i] = (int32_t)(v & (uint64_t)MASK);
                    }
                } else {
                    size_t rec;
#ifdef ZKP_EXP_TRAFFIC4L   // timing-only experiment (wrong results): four checks read one line record
                    if (arg == K_LINE) rec = ((size_t)cursor * A.k * 6 + pair6 + idx) * n_checks + (check & ~3u);
#else
                    if (arg == K_LINE) rec = ((size_t)cursor * A.k * 6 + pair6 + idx) * n_checks + check;   // line buffer: this launch's checks only
#endif
                    else rec = (size_t)(idx + A.st_off) * A.nc + check;
                    const int4* src = (arg == K_LINE ? A.lines : (const int4*)A.state) + rec * 4;
                    int4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
                    x[8] = v2.x; x[9] = v2.y; x[10] = v2.z; x[11] = v2.w; x[12] = v3.x; x[13] = v3.y;
                }
                lds_st(lds, gbase + (int)(w & 63), PS, x);
            }
            ZKP_WG_FENCE();
            cursor += h1;
        } else if (op == OP_GSTORE) {
            const uint32_t w = tbl[off + lig];
            const uint32_t idx = w >> 8;
            const bool part = active && ((w >> 7) & 1);
            int32_t x[NL];
            lds_ld(x, lds, gbase + (int)(w & 63), PS);
            if (arg == K_STATE) {
                if (part) {
                    int4* dst = A.state + ((size_t)(idx + A.st_off) * A.nc + check) * 4;
                    dst[0] = make_int4(x[0], x[1], x[2], x[3]);
                    dst[1] = make_int4(x[4], x[5], x[6], x[7]);
                    dst[2] = make_int4(x[8], x[9], x[10], x[11]);
                    dst[3] = make_int4(x[12], x[13], 0, 0);
                }
            } else {
                uint32_t f[NL];
                canon28(f, x);
                if (part && A.wire_out) {
                    uint64_t o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < NL; i++) {
                        const int bit = W * i, word = bit >> 6, sh = bit & 63;
                        o[word] |= (uint64_t)f[i] << sh;
                        if (sh > 64 - W && word + 1 < 6) o[word + 1] |= (uint64_t)f[i] >> (64 - sh);
                    }
                    uint64_t* dst = A.wire_out + (size_t)check * 72 + idx * 6;
#pragma unroll
                    for (int i = 0; i < 6; i++) dst[i] = o[i];
                }
                if (h1 & 1) {  // Gt == identity ?  coefficient 0 must be 1, the others 0
                    uint32_t d = f[0] ^ (idx == 0 ? 1u : 0u);
#pragma unroll
                    for (int i = 1; i < NL; i++) d |= f[i];
                    const unsigned long long good = __ballot((d == 0) || !part);
                    unsigned gm;
                    if (WGW == 1) {
                        gm = (unsigned)((good >> (grp * LIG)) & 0xfffu);
                    } else {   // check 15: four lanes in each wavefront, their flags meet behind the LDS image
                        uint32_t* tailf = (uint32_t*)(lds + 4 * PS);
                        const int wv = lane >> 6, g0 = ((lane & 63) * 43) >> 9;
                        if ((lane & 63) == 60) tailf[wv] = (uint32_t)(good >> 60) & 0xfu;
                        __syncthreads();
                        gm = g0 < 5 ? (unsigned)((good >> (g0 * LIG)) & 0xfffu) : (tailf[0] | tailf[1] << 4 | tailf[2] << 8);
                    }
                    if (active && lig == 0) {
                        const bool is_one = gm == 0xfffu;
                        if (A.ok) A.ok[check] = is_one ? 1 : 0;
                        if (A.all_ok && !is_one) atomicAnd(A.all_ok, 0);
                    }
                }
            }
            ZKP_WG_FENCE();   // a GLOAD step behind this one stores without a fence of its own
        } else if (op == OP_LOOP) {
            loop_pc = pc + 1;
            loop_left = (int)h1;
        } else if (op == OP_ENDLOOP) {
            if (--loop_left > 0) { pc = loop_pc; continue; }
        } else if (op == OP_PLOOP) {
            ploop_pc = pc + 1;
            ploop_left = (int)A.k;
            pair6 = 0;
        } else if (op == OP_PENDLOOP) {
            if (--ploop_left > 0) { pair6 += 6; pc = ploop_pc; continue; }
            pair6 = 0;
            cursor += h1;
        }
        pc++;
    }
}

// =============================================================================== thread-level Fp28 helpers
__device__ __forceinline__ void f_add(Fp28& r, const Fp28& a, const Fp28& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    weak_norm(r.l);
}
__device__ __forceinline__ void f_sub(Fp28& r, const Fp28& a, const Fp28& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] - b.l[i];
    weak_norm(r.l);
}
__device__ __forceinline__ void f_set(Fp28& r, const int32_t* k) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = k[i];
}
__device__ __forceinline__ void f_zero(Fp28& r) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
}

__device__ __forceinline__ void rec_store(int4* dst, const Fp28& x) {
    dst[0] = make_int4(x.l[0], x.l[1], x.l[2], x.l[3]);
    dst[1] = make_int4(x.l[4], x.l[5], x.l[6], x.l[7]);
    dst[2] = make_int4(x.l[8], x.l[9], x.l[10], x.l[11]);
    dst[3] = make_int4(x.l[12], x.l[13], 0, 0);
}
__device__ __forceinline__ void rec_load(Fp28& x, const int4* src) {
    int4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w; x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
    x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w; x.l[12] = v3.x; x.l[13] = v3.y;
}

// =============================================================================== compressed squaring runs: four lanes per check
// The hard part of the final exponentiation is five exponentiations by |x| of values in the cyclotomic subgroup: 63
// squarings each.  In the interpreter a squaring costs 36 product blocks + 12 reductions on twelve lanes (Granger-Scott,
// every output coefficient recomputes the Fp2 squares it shares with its neighbour).  Karabina's compressed form
// (ePrint 2010/542) keeps four of the six Fp2 coefficients, z2..z5 of
//     g = (z0 + z1 s) + (z2 + z3 s) w + (z4 + z5 s) w^2,   s = w^3, s^2 = xi
// (tower positions: z0 c0.c0, z4 c0.c1, z3 c0.c2, z2 c1.c0, z1 c1.c1, z5 c1.c2), and the Granger-Scott recurrences of
// those four need only themselves:
//     z2' = 6 xi B45 + 2 z2          z3' = 3 (A45 - (1 + xi) B45) - 2 z3        A45 = (z4 + z5)(z4 + xi z5), B45 = z4 z5
//     z4' = 3 (A23 - (1 + xi) B23) - 2 z4          z5' = 6 B23 + 2 z5           A23 = (z2 + z3)(z2 + xi z3), B23 = z2 z3
// FOUR Fp2 products per squaring.  k_ksq gives each to one lane (4 product blocks + 2 reductions per lane, four lanes per
// check, SIXTEEN checks per wavefront, no idle lane), operands in registers, the products exchanged inside the lane quad
// with DPP (no LDS traffic but the parked copy of a lane's own coefficient, no step tables).  An exponentiation by |x| is
// ONE run of 57 squarings that stores a snapshot of (z2..z5) after 16, 48 and 57 squarings (the low set bits of |x|; the
// plan says which - rounds 2-3 ran all 63 with six snapshots); k_kdec_a / k_batch_inv / k_kdec_b recover z0, z1 of the
// snapshots (one shared batched inversion), and the step program squares on uncompressed through bits 60, 62, 63 and
// multiplies the six powers.  tools/coopgen.py emu_ksq / emu_kdec are the limb-exact models of these kernels.
constexpr int KS_CHECKS = 16;
#ifndef ZKP_KSQ_ASM
#define ZKP_KSQ_ASM ZKP_COOP_ASM   // a squaring of k_ksq behind its operand forms as ONE asm block (tools/coopasm.py generate_ksq): the Fp2 product
                                   // as Karatsuba terms + tail on register operands (980 multiply-adds instead of the 1,204 of two
                                   // product-scanning multiplies), DPP combinations, both carry chains, parking; 0: the compiled loop
                                   // around mont_mul_ps, the A/B baseline
#endif
#ifndef ZKP_KSQ_WAVES
#if ZKP_KSQ_ASM
#define ZKP_KSQ_WAVES 3            // the block pins 162 VGPRs (four operands, both results' homes, 78 accumulator registers)
#else
#define ZKP_KSQ_WAVES 4
#endif
#endif

__device__ __forceinline__ void park_st(int4* xch, int lane, const int32_t* re, const int32_t* im) {
    xch[0 * 64 + lane] = make_int4(re[0], re[1], re[2], re[3]);
    xch[1 * 64 + lane] = make_int4(re[4], re[5], re[6], re[7]);
    xch[2 * 64 + lane] = make_int4(re[8], re[9], re[10], re[11]);
    xch[3 * 64 + lane] = make_int4(re[12], re[13], im[0], im[1]);     // re0..re13, im0..im13 as seven quads: the asm body of k_ksq
    xch[4 * 64 + lane] = make_int4(im[2], im[3], im[4], im[5]);       // parks them straight from 28 consecutive registers
    xch[5 * 64 + lane] = make_int4(im[6], im[7], im[8], im[9]);
    xch[6 * 64 + lane] = make_int4(im[10], im[11], im[12], im[13]);
}
#if !ZKP_KSQ_ASM
__device__ __forceinline__ void park_ld(int32_t* re, int32_t* im, const int4* xch, int lane) {
    const int4 r0 = xch[0 * 64 + lane], r1 = xch[1 * 64 + lane], r2 = xch[2 * 64 + lane], r3 = xch[3 * 64 + lane];
    const int4 i0 = xch[4 * 64 + lane], i1 = xch[5 * 64 + lane], i2 = xch[6 * 64 + lane];
    re[0] = r0.x; re[1] = r0.y; re[2] = r0.z; re[3] = r0.w; re[4] = r1.x; re[5] = r1.y; re[6] = r1.z; re[7] = r1.w;
    re[8] = r2.x; re[9] = r2.y; re[10] = r2.z; re[11] = r2.w; re[12] = r3.x; re[13] = r3.y; im[0] = r3.z; im[1] = r3.w;
    im[2] = i0.x; im[3] = i0.y; im[4] = i0.z; im[5] = i0.w; im[6] = i1.x; im[7] = i1.y; im[8] = i1.z; im[9] = i1.w;
    im[10] = i2.x; im[11] = i2.y; im[12] = i2.z; im[13] = i2.w;
}
// out = 3 t + 2 sgn x - q p (sgn = -1 where neg is all ones) with q = round(value / p) taken from the top limbs, as ONE
// exact carry chain (balanced limbs, the top limb keeps the rest); |result| < 0.51 p.  (The chain stays in 64 bits: t is a
// combination of three products, |limb| <= 4 * 2^27, so 3 t + 2 x - q p reaches 20 * 2^27 - a 32-bit chain was measured 1.5 %
// faster on the pass and WRONG, round 3.)
__device__ __forceinline__ void sq_combine(int32_t* out, const int32_t* t, const int32_t* x, int32_t neg) {
    int32_t sx[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) sx[i] = (x[i] ^ neg) - neg;
    const int32_t top = 3 * t[NL - 1] + 2 * sx[NL - 1];
    const int32_t q = ((top >> ZKP_COOP_VRED_SHIFT_IN) * ZKP_COOP_VRED_C + (1 << (ZKP_COOP_VRED_SHIFT_OUT - 1))) >> ZKP_COOP_VRED_SHIFT_OUT;
    int64_t v = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        v += (int64_t)t[i] * 3;
        v += (int64_t)sx[i] * 2;
        v -= (int64_t)q * (int64_t)K_PBAL[i];
        if (i < NL - 1) {
            const int64_t u = v + (1ll << (W - 1));
            out[i] = (int32_t)((uint32_t)u & (uint32_t)MASK) - (1 << (W - 1));
            v = u >> W;
        } else {
            out[i] = (int32_t)v;
        }
    }
}
#define ZKP_QUAD(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xf, 0xf, false)
#endif

// operand forms of a lane's next product, in place: in (x, y) = (the lane's coefficient, its pair partner's).  B lanes: u v =
// mine * partner as they stand.  A lanes: X = u + v and MINUS Y = -(u + xi v), u + xi v = (u0 + v0 - v1) + (u1 + v1 + v0) u with
// v = mine on lane 0, the partner on lane 2 - the asm body wants the A product negated, and here the sign is one operand swap
#if ZKP_KSQ_ASM
__device__ __forceinline__ void ksq_forms(int32_t* xr, int32_t* xi, int32_t* yr, int32_t* yi, bool a_lane, bool v_mine) {
    if (a_lane) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int32_t tr = xr[i] + yr[i], ti = xi[i] + yi[i];
            const int32_t vr = v_mine ? xr[i] : yr[i], vi = v_mine ? xi[i] : yi[i];
            xr[i] = tr;
            xi[i] = ti;
            yr[i] = vi - tr;
            yi[i] = -ti - vr;
        }
    }
}
#endif
// nsq compressed squarings of the Fp12 value in state elements [elem_in, elem_in + 12) (only z2..z5 are read); after
// squaring number it + 1 where bit it of snap_mask is set, (z2..z5) go to the next snapshot area: 12 elements each from
// elem_snap on, laid out like an Fp12 value whose z0, z1 positions are left for k_kdec_b to fill.
__global__ void __launch_bounds__(64, ZKP_KSQ_WAVES) k_ksq(int4* state, uint32_t n_checks_in, uint32_t nc, uint32_t elem_in, uint32_t elem_snap,
                                                           uint32_t nsq, uint64_t snap_mask, NDev nd) {
    extern __shared__ int4 parked[];               // 7 x 64 quads, at LDS address 0 (the asm body addresses it by lane number)
    const uint32_t n_checks = eff_n(n_checks_in, nd);
    if (blockIdx.x * KS_CHECKS >= n_checks) return;
    const int lane = threadIdx.x;
    const int r = lane & 3;                        // 0: A23, 1: B23, 2: A45, 3: B45
    const bool b_lane = r & 1;
    const bool mine_is_v = r == 0 || r == 3;       // the lane's own new coefficient: z3' (0), z2' (1), z4' (2), z5' (3)
    const uint32_t check_raw = blockIdx.x * KS_CHECKS + (lane >> 2);
    const bool active = check_raw < n_checks;
    const uint32_t check = active ? check_raw : n_checks - 1;
    const int tu = (r & 2) ? 1 : 3, tv = (r & 2) ? 5 : 2;   // tower positions of the lane's pair (u, v) = (z4, z5) or (z2, z3)
    int4* const st = state + (size_t)check * 4;
    auto rec = [&](uint32_t e) -> int4* { return st + (size_t)e * nc * 4; };

    // X, Y: the two factors of the lane's product.  `mine` is the lane's own coefficient of its pair (u, v) - v on lanes 0
    // and 3, u on lanes 1 and 2 - and `other` the pair partner's.  B lanes: u v = mine * other.  A lanes: (u + v)(u + xi v).
    int32_t xr[NL], xi[NL], yr[NL], yi[NL];
#if !ZKP_KSQ_ASM
    auto advance = [&](const int32_t* mr, const int32_t* mi, const int32_t* o_r, const int32_t* oi) {
        park_st(parked, lane, mr, mi);            // the "2 z" term of the lane's next combination
        if (b_lane) {
#pragma unroll
            for (int i = 0; i < NL; i++) { xr[i] = mr[i]; xi[i] = mi[i]; yr[i] = o_r[i]; yi[i] = oi[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < NL; i++) {
                xr[i] = mr[i] + o_r[i];
                xi[i] = mi[i] + oi[i];
                yr[i] = xr[i] - (mine_is_v ? mi[i] : oi[i]);      // u + xi v = (u0 + v0 - v1) + (u1 + v0 + v1) u
                yi[i] = xi[i] + (mine_is_v ? mr[i] : o_r[i]);
            }
        }
    };
#endif
    {
        Fp28 u0, u1, v0, v1;
        rec_load(u0, rec(elem_in + 2 * tu));
        rec_load(u1, rec(elem_in + 2 * tu + 1));
        rec_load(v0, rec(elem_in + 2 * tv));
        rec_load(v1, rec(elem_in + 2 * tv + 1));
#if ZKP_KSQ_ASM
        const Fp28 &m0 = mine_is_v ? v0 : u0, &m1 = mine_is_v ? v1 : u1, &o0 = mine_is_v ? u0 : v0, &o1 = mine_is_v ? u1 : v1;
        park_st(parked, lane, m0.l, m1.l);
#pragma unroll
        for (int i = 0; i < NL; i++) { xr[i] = m0.l[i]; xi[i] = m1.l[i]; yr[i] = o0.l[i]; yi[i] = o1.l[i]; }
        ksq_forms(xr, xi, yr, yi, !b_lane, r == 0);
#else
        if (mine_is_v) advance(v0.l, v1.l, u0.l, u1.l); else advance(u0.l, u1.l, v0.l, v1.l);
#endif
    }
    uint32_t snap = elem_snap;
#if ZKP_KSQ_ASM
    // One inline-asm block per squaring (tools/coopasm.py generate_ksq): the Fp2 product, the other pair's products by DPP, the
    // lane-role combinations, both carry chains, the parking of the new coefficient and the pair partner's coefficient by DPP.
    // In: X = xr + xi u, Y = yr + yi u.  Out: the lane's new coefficient in (xr, xi), its partner's in (yr, yi) - the
    // snapshot store and the next product's operand forms stay here.
    static_assert(NL == 14, "the generated block is for 14 limbs");
    {
        constexpr int32_t PB[NL] = {ZKP_KSQ_BODY_P_BAL}, KB[NL] = {ZKP_COOP_P_BAL};
        constexpr int VR[3] = {ZKP_KSQ_BODY_VRED};
        static_assert(VR[0] == ZKP_COOP_VRED_C && VR[1] == ZKP_COOP_VRED_SHIFT_IN && VR[2] == ZKP_COOP_VRED_SHIFT_OUT, "regenerate zkp_coop_mulacc.inc");
        static_assert(PB[0] == KB[0] && PB[1] == KB[1] && PB[5] == KB[5] && PB[12] == KB[12] && PB[13] == KB[13], "regenerate zkp_coop_mulacc.inc");
    }
#pragma unroll 1
    for (uint32_t it = 0; it < nsq; it++) {
        {
            constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};
            asm volatile(ZKP_KSQ_BODY_ASM
                         : ZKP_KSQ_BODY_IO(xr, xi, yr, yi)
                         : [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]), [p6] "s"(PL[6]),
                           [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]), [p12] "s"(PL[12]),
                           [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)
                         : ZKP_KSQ_BODY_CLOBBERS);
        }
        // per-lane values re-derived from an opaque copy of the lane number: kept in registers across the block (which owns 162 of
        // the 168 VGPRs) they would be spilled and reloaded in every squaring
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const bool a_lane = !(l_ & 1), v_mine = (l_ & 3) == 0;     // lane 0 holds (v, u) = (mine, partner), lane 2 (u, v)
        if (it < 64 && ((snap_mask >> it) & 1)) {      // wave-uniform (a 64-bit shift by 64 or more is undefined)
            const uint32_t chk_ = blockIdx.x * KS_CHECKS + (l_ >> 2);
            int4* const st_ = state + (size_t)(chk_ < n_checks ? chk_ : n_checks - 1) * 4;
            auto rec_ = [&](uint32_t e) -> int4* { return st_ + (size_t)e * nc * 4; };
            const int tu_ = (l_ & 2) ? 1 : 3, tv_ = (l_ & 2) ? 5 : 2;
#if defined(ZKP_EXP_TRAFFIC4) && (ZKP_EXP_TRAFFIC4 & 1)     // timing-only experiment (wrong results): a quarter of the snapshot records is written
            if (chk_ < n_checks && a_lane && (chk_ & 3) == 0) {
#else
            if (chk_ < n_checks && a_lane) {
#endif
                Fp28 o;
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = xr[i];
                rec_store(rec_(snap + 2 * (v_mine ? tv_ : tu_)), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = xi[i];
                rec_store(rec_(snap + 2 * (v_mine ? tv_ : tu_) + 1), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = yr[i];
                rec_store(rec_(snap + 2 * (v_mine ? tu_ : tv_)), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = yi[i];
                rec_store(rec_(snap + 2 * (v_mine ? tu_ : tv_) + 1), o);
            }
            snap += 12;
        }
        ksq_forms(xr, xi, yr, yi, a_lane, v_mine);
    }
#else
#pragma unroll 1
    for (uint32_t it = 0; it < nsq; it++) {
        int32_t sre[NL], sim[NL];
        // X Y = (X0 Y0 - X1 Y1) + (X0 Y1 + X1 Y0) u, one reduction per coefficient
        mont_mul_ps<true>(sim, xr, yi, xi, yr);
#pragma unroll
        for (int i = 0; i < NL; i++) xi[i] = -xi[i];
        mont_mul_ps<true>(sre, xr, yr, xi, yi);
        // the other pair's products: A from its even lane, B from its odd lane.  The DPP reads stay outside the lane-role
        // branches: a DPP read from a lane that the branch has switched off returns nothing.
        int32_t tr[NL], ti[NL];
        {
            int32_t ar[NL], ai[NL], br[NL], bi[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                ar[i] = ZKP_QUAD(sre[i], 0x0A);   // quad_perm [2,2,0,0]
                ai[i] = ZKP_QUAD(sim[i], 0x0A);
                br[i] = ZKP_QUAD(sre[i], 0x5F);   // quad_perm [3,3,1,1]
                bi[i] = ZKP_QUAD(sim[i], 0x5F);
            }
            if (!b_lane) {         // 3 (A - (1 + xi) B) - 2 old,  (1 + xi) = 2 + u
#pragma unroll
                for (int i = 0; i < NL; i++) { tr[i] = ar[i] - 2 * br[i] + bi[i]; ti[i] = ai[i] - br[i] - 2 * bi[i]; }
            } else if (r == 1) {   // 3 xi (2 B) + 2 old
#pragma unroll
                for (int i = 0; i < NL; i++) { tr[i] = 2 * (br[i] - bi[i]); ti[i] = 2 * (br[i] + bi[i]); }
            } else {               // 3 (2 B) + 2 old
#pragma unroll
                for (int i = 0; i < NL; i++) { tr[i] = 2 * br[i]; ti[i] = 2 * bi[i]; }
            }
        }
        int32_t or_[NL], oi[NL];
        park_ld(or_, oi, parked, lane);
        const int32_t neg = b_lane ? 0 : -1;
        sq_combine(sre, tr, or_, neg);
        sq_combine(sim, ti, oi, neg);
        // the pair partner's new coefficient completes (u', v')
        int32_t pr[NL], pi[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) { pr[i] = ZKP_QUAD(sre[i], 0xB1); pi[i] = ZKP_QUAD(sim[i], 0xB1); }   // quad_perm [1,0,3,2]
        if (it < 64 && ((snap_mask >> it) & 1)) {      // wave-uniform (a 64-bit shift by 64 or more is undefined)
            if (active && !b_lane) {      // lane 0 holds (v, u) = (mine, partner), lane 2 (u, v)
                Fp28 o;
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = sre[i];
                rec_store(rec(snap + 2 * (mine_is_v ? tv : tu)), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = sim[i];
                rec_store(rec(snap + 2 * (mine_is_v ? tv : tu) + 1), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = pr[i];
                rec_store(rec(snap + 2 * (mine_is_v ? tu : tv)), o);
#pragma unroll
                for (int i = 0; i < NL; i++) o.l[i] = pi[i];
                rec_store(rec(snap + 2 * (mine_is_v ? tu : tv) + 1), o);
            }
            snap += 12;
        }
        advance(sre, sim, pr, pi);
    }
#endif
}

// ---- two lanes per pair: lane parity c selects the Fp2 coefficient a value's lane holds ------------------
// An Fp2 value is ONE Fp28 per lane (c = 0: real part, c = 1: imaginary part); add/sub/neg/dbl touch
// only the lane's own coefficient; products fetch the partner's coefficient with a DPP quad swap.
__device__ __forceinline__ void swap_pair(Fp28& o, const Fp28& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) o.l[i] = __builtin_amdgcn_update_dpp(0, x.l[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
}
#ifndef ZKP_PREP_KARATSUBA
#define ZKP_PREP_KARATSUBA 0   // Karatsuba product blocks (147 multiply-adds instead of 196) in the by-value routines: bit-exact, measured
                               // round 3: k_prep_lines<true> 8.03-8.16 against 8.08-8.09 ms per 2^18 pairs - the 13 extra columns cost
                               // the callers 29 spilled registers (0 without) and k_kdec_a/b their third wavefront; not the default
#endif
#ifndef ZKP_PREP_PS
#define ZKP_PREP_PS 0   // product-scanning multiply in the by-value routines: measured +0.3 % time on the 2^20 pass (at two waves
                        // per SIMD the one-column dependency chains are not covered); the accumulator form stays
#endif
// r = coefficient c of (a0 + a1 u)^2 :  c=0: (a0 + a1)(a0 - a1) ;  c=1: (2 a0) a1
// Operands and results travel BY VALUE (VGPRs): with pointers every temporary lives in scratch memory and
// the kernel becomes HBM-bound on its own stack traffic (measured: 50 GB per 2^17 pairs).
__device__ __attribute__((noinline)) Fp28 c_sqr(Fp28 mine, int c) {
    Fp28 o, r;
    swap_pair(o, mine);
    int32_t x[NL], y[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        x[i] = o.l[i] + (c ? o.l[i] : mine.l[i]);
        y[i] = mine.l[i] - (c ? 0 : o.l[i]);
    }
#if ZKP_PREP_PS
    mont_mul_ps<false>(r.l, x, y, x, y);
#elif ZKP_PREP_KARATSUBA
    Acc acc;
    AccMid mid;
    acc_zero(acc);
    mid_zero(mid);
    acc_mul_k(acc, mid, x, y);
    acc_fold(acc, mid);
    acc_reduce(r.l, acc);
#else
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, x, y);
    acc_reduce(r.l, acc);
#endif
    return r;
}
// r = coefficient c of (a0 + a1 u)(b0 + b1 u) :  c=0: a0 b0 - a1 b1 ;  c=1: a0 b1 + a1 b0  (one reduction)
// The second operand of the by-value routines travels as four int4 vectors: clang's AMDGPU ABI passes aggregates in
// registers only while they total at most 16 registers, so a second Fp28 (14 registers) went through the stack - a
// scratch store in the caller and a load the callee had to wait for, at every call.
__device__ __forceinline__ void fp28_unpack(Fp28& x, const int4& q0, const int4& q1, const int4& q2, const int4& q3) {
    x.l[0] = q0.x; x.l[1] = q0.y; x.l[2] = q0.z; x.l[3] = q0.w; x.l[4] = q1.x; x.l[5] = q1.y; x.l[6] = q1.z; x.l[7] = q1.w;
    x.l[8] = q2.x; x.l[9] = q2.y; x.l[10] = q2.z; x.l[11] = q2.w; x.l[12] = q3.x; x.l[13] = q3.y;
}
#define FP28_AS_QUADS(x) make_int4((x).l[0], (x).l[1], (x).l[2], (x).l[3]), make_int4((x).l[4], (x).l[5], (x).l[6], (x).l[7]), \
                         make_int4((x).l[8], (x).l[9], (x).l[10], (x).l[11]), make_int4((x).l[12], (x).l[13], 0, 0)
__device__ __attribute__((noinline)) Fp28 c_mul_q(Fp28 ma, int4 q0, int4 q1, int4 q2, int4 q3, int c) {
    Fp28 mb;
    fp28_unpack(mb, q0, q1, q2, q3);
    Fp28 ao, bo, r;
    swap_pair(ao, ma);
    swap_pair(bo, mb);
    int32_t x1[NL], x2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        x1[i] = c ? ao.l[i] : ma.l[i];
        x2[i] = c ? ma.l[i] : -ao.l[i];
    }
#if ZKP_PREP_PS
    mont_mul_ps<true>(r.l, x1, mb.l, x2, bo.l);
#elif ZKP_PREP_KARATSUBA
    Acc acc;
    AccMid mid;
    acc_zero(acc);
    mid_zero(mid);
    acc_mul_k(acc, mid, x1, mb.l);
    acc_mul_k(acc, mid, x2, bo.l);
    acc_fold(acc, mid);
    acc_reduce(r.l, acc);
#else
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, x1, mb.l);
    acc_mul(acc, x2, bo.l);
    acc_reduce(r.l, acc);
#endif
    return r;
}
__device__ __forceinline__ Fp28 c_mul(const Fp28& ma, const Fp28& mb, int c) { return c_mul_q(ma, FP28_AS_QUADS(mb), c); }
// r = coefficient c of s^2 - 12 e^2 with ONE reduction (the Y' = (B + F)^2 - 3 (2 E)^2 of the homogeneous doubling step): both
// squarings' operand forms as in c_sqr, the second product's first factor normalised (one pass) and scaled by -12.
// Column budget (units of 2^54 per product of limbs): s normalised, e renormalised: 2 * 2 + 12 * 2 = 28 <= 30.
[[maybe_unused]] __device__ __attribute__((noinline)) Fp28 c_sqr_sub12sqr_q(Fp28 s, int4 q0, int4 q1, int4 q2, int4 q3, int c) {
    Fp28 e, os, oe, r;
    fp28_unpack(e, q0, q1, q2, q3);
    swap_pair(os, s);
    swap_pair(oe, e);
    int32_t xs[NL], ys[NL], xe[NL], ye[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        xs[i] = os.l[i] + (c ? os.l[i] : s.l[i]);
        ys[i] = s.l[i] - (c ? 0 : os.l[i]);
        xe[i] = oe.l[i] + (c ? oe.l[i] : e.l[i]);
        ye[i] = e.l[i] - (c ? 0 : oe.l[i]);
    }
    weak_norm(xe);
#pragma unroll
    for (int i = 0; i < NL; i++) xe[i] *= -12;
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, xs, ys);
    acc_mul(acc, xe, ye);
    acc_reduce(r.l, acc);
    return r;
}
__device__ __attribute__((noinline)) Fp28 f_mul_q(Fp28 a, int4 q0, int4 q1, int4 q2, int4 q3) {
    Fp28 b, r;
    fp28_unpack(b, q0, q1, q2, q3);
#if ZKP_PREP_PS
    mont_mul_ps<false>(r.l, a.l, b.l, a.l, b.l);
#elif ZKP_PREP_KARATSUBA
    Acc acc;
    AccMid mid;
    acc_zero(acc);
    mid_zero(mid);
    acc_mul_k(acc, mid, a.l, b.l);
    acc_fold(acc, mid);
    acc_reduce(r.l, acc);
#else
    fp28_mul(r, a, b);
#endif
    return r;
}
__device__ __forceinline__ Fp28 f_mul_v(const Fp28& a, const Fp28& b) { return f_mul_q(a, FP28_AS_QUADS(b)); }
__device__ __forceinline__ Fp28 c_add(const Fp28& a, const Fp28& b) { Fp28 r; f_add(r, a, b); return r; }
__device__ __forceinline__ Fp28 c_sub(const Fp28& a, const Fp28& b) { Fp28 r; f_sub(r, a, b); return r; }
__device__ __forceinline__ Fp28 c_dbl(const Fp28& a) { Fp28 r; f_add(r, a, a); return r; }
__device__ __forceinline__ Fp28 c_neg(const Fp28& a) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = -a.l[i];
    return r;
}

struct G2C { Fp28 x, y, z; };   // this lane's coefficient of the three Jacobian coordinates

// ---- lazily normalised arithmetic of the doubling step, with its bounds carried in the TYPE: Bd<L, LO, HI> is a value whose
// limbs are at most L (2^27 + 16) in magnitude and whose value lies in [LO, HI] units of p / 64.  Additions, subtractions and
// doublings are plain limb-wise operations (no carry pass); every consumer states what it can take as a static_assert, so a
// formula that would overflow an int32 limb, a 64-bit product column or the value renormalisation does not compile:
//   * a product column holds 14 * sum(La Lb) * 2^54 < 2^63  =>  sum(La Lb) <= 30 (zkp_fp28.hpp); the operand forms of the
//     Fp2 squaring double the limb bound (x0 + x1, x0 - x1, 2 x0);
//   * a Montgomery reduction returns (-0.05 p, 1.05 p) while sum(|a| |b|) <= 100 p^2 (R = 2^392 = 2521 p);
//   * the one-pass normalisation adds 2^27 to a limb: L <= 14;  the value renormalisation subtracts q p first, q <= |v| / p + 1/2,
//     p's balanced limbs are at most 2^27: L + q + 1 <= 15.
template <int L, int LO, int HI> struct Bd { Fp28 v; };
[[maybe_unused]] constexpr int bd_k(int lo, int hi) { return -lo > hi ? -lo : hi; }
typedef Bd<1, -4, 68> BdRed;     // a reduced product
typedef Bd<1, -33, 33> BdVred;   // after the value renormalisation (|v| <= 0.51 p)
template <int L1, int A1, int B1, int L2, int A2, int B2>
__device__ __forceinline__ Bd<L1 + L2, A1 + A2, B1 + B2> bd_add(const Bd<L1, A1, B1>& a, const Bd<L2, A2, B2>& b) {
    Bd<L1 + L2, A1 + A2, B1 + B2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = a.v.l[i] + b.v.l[i];
    return r;
}
template <int L1, int A1, int B1, int L2, int A2, int B2>
__device__ __forceinline__ Bd<L1 + L2, A1 - B2, B1 - A2> bd_sub(const Bd<L1, A1, B1>& a, const Bd<L2, A2, B2>& b) {
    Bd<L1 + L2, A1 - B2, B1 - A2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = a.v.l[i] - b.v.l[i];
    return r;
}
template <int L, int A, int B>
__device__ __forceinline__ Bd<2 * L, 2 * A, 2 * B> bd_dbl(const Bd<L, A, B>& a) {
    Bd<2 * L, 2 * A, 2 * B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = a.v.l[i] + a.v.l[i];
    return r;
}
template <int L, int A, int B>
__device__ __forceinline__ Bd<L, -B, -A> bd_neg(const Bd<L, A, B>& a) {
    Bd<L, -B, -A> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = -a.v.l[i];
    return r;
}
template <int L, int A, int B>
__device__ __forceinline__ Bd<1, A, B> bd_norm(const Bd<L, A, B>& a) {
    static_assert(L <= 14, "one-pass normalisation: |limb| + 2^27 must stay below 2^31");
    Bd<1, A, B> r;
    r.v = a.v;
    weak_norm(r.v.l);
    return r;
}
template <int L, int A, int B>
__device__ __forceinline__ BdVred bd_vred(const Bd<L, A, B>& a) {
    static_assert(L + (bd_k(A, B) + 32 + 63) / 64 + 1 <= 15, "value renormalisation: |limb| + q 2^27 + 2^27 must stay below 2^31");
    BdVred r;
    r.v = a.v;
    vred(r.v.l);
    return r;
}
template <int L, int A, int B>
__device__ __forceinline__ BdRed bd_sqr(const Bd<L, A, B>& a, int c) {
    static_assert(4 * L * L <= 30, "column budget of the Fp2 squaring (its operand forms double the limbs)");
    static_assert(4 * bd_k(A, B) * bd_k(A, B) <= 100 * 64 * 64, "value budget of the reduction");
    BdRed r;
    r.v = c_sqr(a.v, c);
    return r;
}
template <int L1, int A1, int B1, int L2, int A2, int B2>
__device__ __forceinline__ BdRed bd_mul(const Bd<L1, A1, B1>& a, const Bd<L2, A2, B2>& b, int c) {
    static_assert(2 * L1 * L2 <= 30, "column budget of the Fp2 product (two products per coefficient)");
    static_assert(2 * bd_k(A1, B1) * bd_k(A2, B2) <= 100 * 64 * 64, "value budget of the reduction");
    BdRed r;
    r.v = c_mul(a.v, b.v, c);
    return r;
}
// a^2 - 12 e^2, one reduction (c_sqr_sub12sqr_q): a normalised, e renormalised
template <int A, int B>
__device__ __forceinline__ BdRed bd_sqr_sub12sqr(const Bd<1, A, B>& a, const BdVred& e, int c) {
    static_assert(4 * bd_k(A, B) * bd_k(A, B) + 12 * 4 * 33 * 33 <= 100 * 64 * 64, "value budget of the reduction");
    BdRed r;
    r.v = c_sqr_sub12sqr_q(a.v, FP28_AS_QUADS(e.v), c);
    return r;
}
// static checks for values that leave the typed code as plain Fp28
template <int L, int A, int B>
__device__ __forceinline__ const Fp28& bd_for_vred(const Bd<L, A, B>& a) {
    static_assert(L + (bd_k(A, B) + 32 + 63) / 64 + 1 <= 15, "value renormalisation: |limb| + q 2^27 + 2^27 must stay below 2^31");
    return a.v;
}
template <int L, int A, int B>
__device__ __forceinline__ const Fp28& bd_for_fmul(const Bd<L, A, B>& a) {   // multiplied by a reduced Fp value (limbs <= 2^27, |v| <= 1.05 p)
    static_assert(L <= 30 && bd_k(A, B) * 68 <= 100 * 64 * 64, "budgets of the Fp product");
    return a.v;
}

// ePrint 2010/354 Alg. 26; hands this lane's coefficient of the line (c0, c1, c2) to the three sinks and advances r.
// The operations are ordered so that few values are live at any call: a by-value call keeps the caller's values in
// the ~108 callee-saved VGPRs only, everything beyond that is spilled around EVERY call (that was 100 GB of scratch
// traffic per 2^20 pairs); the line coefficients leave through the sinks as soon as they exist.  Three one-pass
// normalisations per step are left (24 when every addition normalised its result): the types prove the rest unnecessary.
// sink_l2 renormalises its argument; sink_l0 / sink_l1 multiply theirs by a reduced Fp value.
template <class S0, class S1, class S2>
__device__ __forceinline__ void dbl_step(G2C& r, int c, S0&& sink_l0, S1&& sink_l1, S2&& sink_l2) {
    Bd<1, -33, 68> x, y, z;    // renormalised by the previous step, or a reduced input coordinate (the first step)
    x.v = r.x; y.v = r.y; z.v = r.z;
    auto zsq = bd_sqr(z, c);
    auto tmp1 = bd_sqr(y, c);
    auto nz = bd_sub(bd_sub(bd_sqr(bd_add(z, y), c), tmp1), zsq);
    sink_l0(bd_for_fmul(bd_dbl(bd_mul(nz, zsq, c))));
    auto tmp0 = bd_sqr(x, c);
    auto tmp4 = bd_norm(bd_add(bd_add(tmp0, tmp0), tmp0));
    sink_l1(bd_for_fmul(bd_neg(bd_dbl(bd_mul(tmp4, zsq, c)))));
    auto tmp5 = bd_sqr(tmp4, c);
    {
        auto tmp6 = bd_sub(bd_sub(bd_sqr(bd_add(x, tmp4), c), tmp0), tmp5);
        sink_l2(bd_for_vred(bd_sub(tmp6, bd_dbl(bd_dbl(tmp1)))));
    }
    auto tmp3s = bd_sqr(bd_add(tmp1, x), c);
    auto tmp2 = bd_sqr(tmp1, c);
    auto tmp3 = bd_norm(bd_dbl(bd_sub(bd_sub(tmp3s, tmp0), tmp2)));
    auto nx = bd_sub(bd_sub(tmp5, tmp3), tmp3);
    auto ny = bd_sub(bd_mul(bd_sub(tmp3, nx), tmp4, c), bd_norm(bd_dbl(bd_dbl(bd_dbl(tmp2)))));
    r.x = bd_vred(nx).v;
    r.y = bd_vred(ny).v;
    r.z = bd_vred(nz).v;
}
// ePrint 2010/354 Alg. 27
[[maybe_unused]] __device__ __forceinline__ void add_step(Fp28& l0, Fp28& l1, Fp28& l2, G2C& r, const Fp28& qx, const Fp28& qy, int c) {
    Fp28 zsq = c_sqr(r.z, c);
    Fp28 ysq = c_sqr(qy, c);
    Fp28 t0 = c_mul(zsq, qx, c);
    Fp28 t1 = c_sqr(c_add(qy, r.z), c);
    t1 = c_mul(c_sub(c_sub(t1, ysq), zsq), zsq, c);
    Fp28 t2 = c_sub(t0, r.x);
    Fp28 t3 = c_sqr(t2, c);
    Fp28 t4 = c_dbl(c_dbl(t3));
    Fp28 t5 = c_mul(t4, t2, c);
    Fp28 t6 = c_sub(c_sub(t1, r.y), r.y);
    Fp28 t9 = c_mul(t6, qx, c);
    Fp28 t7 = c_mul(t4, r.x, c);
    Fp28 nx = c_sqr(t6, c);
    nx = c_sub(c_sub(c_sub(nx, t5), t7), t7);
    Fp28 nz = c_sqr(c_add(r.z, t2), c);
    nz = c_sub(c_sub(nz, zsq), t3);
    Fp28 t10 = c_add(qy, nz);
    Fp28 t8 = c_mul(c_sub(t7, nx), t6, c);
    t0 = c_dbl(c_mul(r.y, t5, c));
    Fp28 ny = c_sub(t8, t0);
    t10 = c_sub(c_sqr(t10, c), ysq);
    t10 = c_sub(t10, c_sqr(nz, c));
    t9 = c_sub(c_dbl(t9), t10);
    t10 = c_dbl(nz);
    t6 = c_neg(t6);
    t1 = c_dbl(t6);
    vred(nx.l); vred(ny.l); vred(nz.l);
    r.x = nx; r.y = ny; r.z = nz;
    l0 = t10; l1 = t1; l2 = t9;
}

// ---- the same two steps in homogeneous projective coordinates, for the FUSED paths only (zkp_pairing_* / zkp_pairing_check_*):
// those expose Gt and flags, and any factor of the Miller value that lies in Fp2 dies in the final exponentiation's
// f^(p^6 - 1), so the lines may be scaled freely and the point may live in whatever coordinates are cheapest.
// zkp_multi_miller_loop_batch keeps the upstream-shaped value (Alg. 26 / 27 above).
// Costello-Lange-Naehrig doubling (ePrint 2009/615; Aranha et al. ePrint 2010/526 eq. (10)) on (X : Y : W), W = 2 Z, scaled
// by 4 so that no halving is left; b' = 4 xi (src/common.rs:69-71: B2 = (4, 4)), so 3 b' Z^2 = 3 xi W^2:
//   B = Y^2, C = W^2, H2 = (Y + W)^2 - B - C = 2 Y W, E = 3 xi C, F = 3 E,
//   X' = ((X + Y)^2 - X^2 - B) (B - F) = 2 X Y (B - F),   Y' = (B + F)^2 - 3 (2 E)^2,   W' = 4 B H2
//   line (times 2 / Z): 2 (B - E)  -  6 X^2 xP  +  H2 yP        [the (c0, c1, c4) operands of mul_by_014]
// Seven Fp2 squarings and two products: 13 products and 10 reductions per lane (the two squarings of Y' share one) against
// 16 and 13 of Alg. 26.
template <int L, int A, int B>
__device__ __forceinline__ Bd<2 * L, (A - B), (A + B) < 2 * B ? 2 * B : (A + B)> bd_xi(const Bd<L, A, B>& a, int c) {
    // (1 + u) (a0 + a1 u) = (a0 - a1) + (a0 + a1) u : this lane's coefficient, the partner's by DPP
    Bd<2 * L, (A - B), (A + B) < 2 * B ? 2 * B : (A + B)> r;
    Fp28 o;
    swap_pair(o, a.v);
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = c ? o.l[i] + a.v.l[i] : a.v.l[i] - o.l[i];
    return r;
}
template <class S0, class S1, class S2>
__device__ __forceinline__ void dbl_step_cln(G2C& r, int c, S0&& sink_l0, S1&& sink_l1, S2&& sink_l2) {
    Bd<1, -33, 68> x, y, w;    // renormalised by the previous step, or a reduced input coordinate / the constant 2 (the first step)
    x.v = r.x; y.v = r.y; w.v = r.z;
    // ordered for few live values: at most six 14-register values across any by-value call (they live in the ~108 callee-saved
    // VGPRs; what does not fit is spilled around every call)
    auto B = bd_sqr(y, c);
    auto C = bd_sqr(w, c);
    auto H2 = bd_sub(bd_sub(bd_sqr(bd_add(y, w), c), B), C);
    sink_l0(bd_for_fmul(H2));
    auto nw = bd_mul(B, bd_dbl(bd_dbl(H2)), c);      // 4 B H2 with the factor on the operand: a reduced product, no renormalisation
    auto xiC = bd_xi(C, c);
    auto E = bd_vred(bd_add(bd_add(xiC, xiC), xiC));
    sink_l2(bd_for_vred(bd_dbl(bd_sub(B, E))));
    auto X2 = bd_sqr(x, c);
    {
        auto X6 = bd_dbl(bd_add(bd_add(X2, X2), X2));
        sink_l1(bd_for_fmul(bd_neg(X6)));
    }
    auto XY2 = bd_sub(bd_sub(bd_sqr(bd_add(x, y), c), X2), B);
    auto F = bd_add(bd_add(E, E), E);
    auto nx = bd_mul(XY2, bd_sub(B, F), c);
    auto ny = bd_sqr_sub12sqr(bd_norm(bd_add(B, F)), E, c);      // (B + F)^2 - 3 (2 E)^2 under one reduction
    r.x = nx.v;                 // a reduced product is a valid input as it stands
    r.y = ny.v;
    r.z = nw.v;
}
// mixed addition T + Q on the same coordinates (Aranha et al. eq. (13), (14) with every quantity doubled: W = 2 Z):
//   th = 2 Y - y2 W, la = 2 X - x2 W, C = th^2, D = la^2, E = la D, F = W C, G = 2 X D, H = E + F - 2 G,
//   X' = la H, Y' = th (G - H) - 2 Y E, W' = 2 W E        line (times 2): la yP - th xP + (th x2 - la y2)
// Five of the 68 steps: written with normalising additions, no bound bookkeeping beyond |v| < 8 p at every product.
[[maybe_unused]] __device__ __forceinline__ void add_step_cln(Fp28& l0, Fp28& l1, Fp28& l2, G2C& r, const Fp28& qx, const Fp28& qy, int c) {
    Fp28 th = c_sub(c_dbl(r.y), c_mul(qy, r.z, c));
    Fp28 la = c_sub(c_dbl(r.x), c_mul(qx, r.z, c));
    vred(th.l); vred(la.l);
    l2 = c_sub(c_mul(th, qx, c), c_mul(la, qy, c));
    l1 = c_neg(th);
    l0 = la;
    Fp28 C = c_sqr(th, c), D = c_sqr(la, c);
    Fp28 E = c_mul(la, D, c), F = c_mul(r.z, C, c), G = c_mul(c_dbl(r.x), D, c);
    Fp28 H = c_sub(c_add(E, F), c_dbl(G));
    vred(H.l);
    Fp28 GH = c_sub(G, H);
    vred(GH.l);
    Fp28 nx = c_mul(la, H, c);
    Fp28 ny = c_sub(c_mul(th, GH, c), c_mul(c_dbl(r.y), E, c));
    Fp28 nw = c_dbl(c_mul(r.z, E, c));
    vred(ny.l); vred(nw.l);
    r.x = nx; r.y = ny; r.z = nw;
}

// two lanes per pair: write the 68-step line stream of pair `pid` (check = pid / k, j = pid % k);
// lane c writes the records of Fp2 coefficient c.  The k pairs are pairs j0 .. j0+k-1 of the check's k_in
// input pairs (k_in > k when a check is processed in groups of at most eight pairs).
#ifndef ZKP_PREP_WAVES
#define ZKP_PREP_WAVES 2   // measured: 256 VGPRs (2 waves/SIMD) 6.6 ms, 168 -> 9.3 ms, 128 -> 11.4 ms per 2^17 pairs (spill traffic)
#endif
// CLN: homogeneous projective steps with freely scaled lines (fused pairing paths); otherwise the upstream-shaped Alg. 26 / 27.
#ifdef ZKP_EXP_TRAFFIC4L
#define ZKP_EXP_LINE_KEEP(check) (((check) & 3u) == 0)     // timing-only experiment: a quarter of the line records is written
#else
#define ZKP_EXP_LINE_KEEP(check) true
#endif
template <bool CLN>
__global__ void __launch_bounds__(64, ZKP_PREP_WAVES) k_prep_lines(const uint64_t* g1, const uint64_t* g2, const uint8_t* inf1, const uint8_t* inf2,
                                                       uint32_t n_pairs_in, uint32_t k, uint32_t k_in, uint32_t j0, uint32_t nc_in, int4* lines, NDev nd) {
    // nc_in checks of k pairs each (n_pairs_in = nc_in * k); with a device-resident count the launch covers the checks that exist, and
    // their number is the stride of the line records (k_coop computes the same)
    const uint32_t nc = eff_n(nc_in, nd);
    const uint32_t n_pairs = nd.cnt ? nc * k : n_pairs_in;
    if (blockIdx.x * 32u >= n_pairs) return;
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const int c = (int)(tid & 1);
    uint32_t pid = tid >> 1;
    const bool live_lane = pid < n_pairs;
    if (!live_lane) pid = n_pairs - 1;   // keep both lanes of a pair (and the DPP swaps) well defined
    const uint32_t check = pid / k, j = pid - check * k;
    const size_t src = (size_t)check * k_in + j0 + j;
    const bool dead = (inf1 && inf1[src]) || (inf2 && inf2[src]);
    auto rec = [&](uint32_t step, uint32_t e) -> int4* { return lines + ((((size_t)step * k + j) * 6 + e) * nc + check) * 4; };
    // P = (px, py) and Q = (qx, qy) are needed once per step / five times per loop: parked in LDS (limb quad q of
    // value v at [(v * 4 + q) * 64 + lane], conflict-free) instead of occupying 56 VGPRs across every call
    extern __shared__ int4 park[];           // 4 values x 4 quads x 64 lanes, at LDS address 0 (the asm step reads xP, yP by lane number)
    const int lane = threadIdx.x;
    enum { PX = 0, PY = 1, QX = 2, QY = 3 };
    auto park_st = [&](int v, const Fp28& x) {
        park[(v * 4 + 0) * 64 + lane] = make_int4(x.l[0], x.l[1], x.l[2], x.l[3]);
        park[(v * 4 + 1) * 64 + lane] = make_int4(x.l[4], x.l[5], x.l[6], x.l[7]);
        park[(v * 4 + 2) * 64 + lane] = make_int4(x.l[8], x.l[9], x.l[10], x.l[11]);
        park[(v * 4 + 3) * 64 + lane] = make_int4(x.l[12], x.l[13], 0, 0);
    };
    [[maybe_unused]] auto park_ld = [&](int v) -> Fp28 {
        asm volatile("" ::: "memory");   // keep the load at its use (no hoisting out of the step loop)
        const int4 v0 = park[(v * 4 + 0) * 64 + lane], v1 = park[(v * 4 + 1) * 64 + lane], v2 = park[(v * 4 + 2) * 64 + lane],
                   v3 = park[(v * 4 + 3) * 64 + lane];
        Fp28 x;
        x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w; x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
        x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w; x.l[12] = v3.x; x.l[13] = v3.y;
        return x;
    };
    G2C r;
    {
        Fp28 t;
        fp28_from_wire(t, g1 + 12 * src);
        park_st(PX, t);
        fp28_from_wire(t, g1 + 12 * src + 6);
        park_st(PY, t);
        fp28_from_wire(r.x, g2 + 24 * src + 6 * c);
        park_st(QX, r.x);
        fp28_from_wire(r.y, g2 + 24 * src + 12 + 6 * c);
        park_st(QY, r.y);
    }
    if (c == 0) f_set(r.z, K28_ONE); else f_zero(r.z);
    if (CLN) r.z = c_dbl(r.z);    // W = 2 Z
    uint32_t step = 0;
    // stream order (c2, c1 * xP, c0 * yP) = the (c0, c1, c4) operands of mul_by_014; a pair with an infinity streams
    // the neutral line (1, 0, 0)
#if !ZKP_PREP_ASM
    auto put = [&](uint32_t e, const Fp28& v) {
        if (!live_lane) return;
        if (dead) {      // a branch, not selects: as selects the limbs of ONE stay in (spilled) registers for the whole kernel
            Fp28 o;
            if (e == 0 && c == 0) f_set(o, K28_ONE); else f_zero(o);
            asm volatile("" ::: "memory");
            rec_store(rec(step, e + c), o);
        } else {
            rec_store(rec(step, e + c), v);
        }
    };
    auto sink_l0 = [&](const Fp28& l0) { put(4, f_mul_v(l0, park_ld(PY))); };
    auto sink_l1 = [&](const Fp28& l1) { put(2, f_mul_v(l1, park_ld(PX))); };
    auto sink_l2 = [&](Fp28 l2) { vred(l2.l); put(0, l2); };
#else
    // the asm steps store the lines of live pairs without an infinity only; a pair with an infinity streams the neutral line
    // (1, 0, 0) at every step - written here, ahead of the loop, so that none of this is alive across the blocks
    if (live_lane && dead) {
        Fp28 o;
        for (uint32_t st = 0; st < (uint32_t)NLINES; st++) {
            for (uint32_t e = 0; e < 6; e += 2) {
                if (e == 0 && c == 0) f_set(o, K28_ONE); else f_zero(o);
                rec_store(rec(st, e + c), o);
            }
        }
    }
#endif
    // bits of |x| below its leading one: a doubling step each, an addition step after it where the bit is set
    // (the lowest bit is clear: the loop ends with the final doubling) -> 63 + 5 = 68 line records
    const uint64_t xs = 0xd201000000010000ULL;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
#if ZKP_PREP_ASM
        // the step's three line records leave from inside the block (lanes of live pairs without an infinity); a pair with an
        // infinity gets the neutral line from put() behind it
#define ZKP_PREP_STEP_ASM(BLOCK)                                                                                                          \
        do {                                                                                                                              \
            constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};                                                                                  \
            const unsigned long long smask = __ballot(live_lane && !dead && ZKP_EXP_LINE_KEEP(check));                                    \
            const uint64_t sb_ = (uint64_t)(uintptr_t)lines + (uint64_t)step * k * 6 * nc * 64;   /* wave-uniform: made scalar by hand */ \
            const char* const sbase = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb_ >> 32)) << 32) |  \
                                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb_));                  \
            const uint32_t voff = (uint32_t)(((size_t)j * 6 + c) * nc + check) * 64u;                                                     \
            asm volatile(BLOCK                                                                                                            \
                         : ZKP_PREP_DBL_IO(r.x.l, r.y.l, r.z.l)                                                                           \
                         : [estride] "s"(2u * nc * 64u), [voff] "v"(voff), [smask] "s"(smask), [base] "s"(sbase),                         \
                           [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]),         \
                           [p6] "s"(PL[6]), [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]),     \
                           [p12] "s"(PL[12]), [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)                                                   \
                         : ZKP_PREP_DBL_CLOBBERS);                                                                                        \
        } while (0)
        static_assert(NL == 14, "the generated blocks are for 14 limbs");
        // <false>: the Jacobian doubling with the upstream-shaped lines (round 4: generate_jac, 18 Karatsuba blocks + 12 reductions)
        if (CLN) ZKP_PREP_STEP_ASM(ZKP_PREP_DBL_ASM); else ZKP_PREP_STEP_ASM(ZKP_PREP_JAC_DBL_ASM);
#else
        if (CLN) dbl_step_cln(r, c, sink_l0, sink_l1, sink_l2); else dbl_step(r, c, sink_l0, sink_l1, sink_l2);
#endif
        step++;
        if ((xs >> b) & 1) {
#if ZKP_PREP_ASM
            if (CLN) ZKP_PREP_STEP_ASM(ZKP_PREP_ADD_ASM); else ZKP_PREP_STEP_ASM(ZKP_PREP_JAC_ADD_ASM);
#else
            {
                Fp28 l0, l1, l2;
                if (CLN) add_step_cln(l0, l1, l2, r, park_ld(QX), park_ld(QY), c); else add_step(l0, l1, l2, r, park_ld(QX), park_ld(QY), c);
                sink_l2(l2);
                sink_l1(l1);
                sink_l0(l0);
            }
#endif
            step++;
        }
    }
}

// =============================================================================== validity checks on the 28-bit core
// G1Affine::is_valid / G2Affine::is_valid (reference src/g1.rs:49-62, src/g2.rs:57-69) with Jacobian arithmetic and
// no inversion.  Zero tests canonicalise (one reduction); every exceptional case of the group law is handled so
// that the status agrees with the affine reference semantics on EVERY input (small-order points included).
__device__ __forceinline__ bool f_is_zero(const Fp28& a) {
    Acc acc;
    acc_zero(acc);
#pragma unroll
    for (int i = 0; i < NL; i++) acc.c[i] = a.l[i];
    int32_t x[NL];
    acc_reduce(x, acc);
    uint32_t f[NL];
    canon28(f, x);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= f[i];
    return o == 0;
}
__device__ __forceinline__ Fp28 f_const(const int32_t* k) { Fp28 r; f_set(r, k); return r; }
__device__ __forceinline__ Fp28 f_vred(Fp28 a) { vred(a.l); return a; }

// ---- generic Jacobian arithmetic over an "element" E with by-value ops supplied by the policy F
//      (F1: Fp, one lane per point; F2: Fp2 spread over a lane pair)
// the affine input point of a validity check, parked in LDS (limb quad q of value v at [(v * 4 + q) * 64 + lane])
__device__ __forceinline__ void valid_park(int4* park, int lane, int v, const Fp28& x) {
    park[(v * 4 + 0) * 64 + lane] = make_int4(x.l[0], x.l[1], x.l[2], x.l[3]);
    park[(v * 4 + 1) * 64 + lane] = make_int4(x.l[4], x.l[5], x.l[6], x.l[7]);
    park[(v * 4 + 2) * 64 + lane] = make_int4(x.l[8], x.l[9], x.l[10], x.l[11]);
    park[(v * 4 + 3) * 64 + lane] = make_int4(x.l[12], x.l[13], 0, 0);
}
__device__ __forceinline__ Fp28 valid_unpark(const int4* park, int lane, int v) {
    asm volatile("" ::: "memory");   // keep the load at its use
    const int4 v0 = park[(v * 4 + 0) * 64 + lane], v1 = park[(v * 4 + 1) * 64 + lane], v2 = park[(v * 4 + 2) * 64 + lane],
               v3 = park[(v * 4 + 3) * 64 + lane];
    Fp28 x;
    x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w; x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
    x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w; x.l[12] = v3.x; x.l[13] = v3.y;
    return x;
}
struct F1 {
    int c;
    __device__ Fp28 sqr(const Fp28& a) const { return f_mul_v(a, a); }
    __device__ Fp28 mul(const Fp28& a, const Fp28& b) const { return f_mul_v(a, b); }
    __device__ bool is_zero(const Fp28& a) const { return f_is_zero(a); }
    __device__ Fp28 one() const { return f_const(K28_ONE); }
};
struct F2 {
    int c;   // which Fp2 coefficient this lane holds
    __device__ Fp28 sqr(const Fp28& a) const { return c_sqr(a, c); }
    __device__ Fp28 mul(const Fp28& a, const Fp28& b) const { return c_mul(a, b, c); }
    __device__ bool is_zero(const Fp28& a) const {
        const int z = f_is_zero(a) ? 1 : 0;
        const int zo = __builtin_amdgcn_update_dpp(0, z, 0xB1, 0xf, 0xf, false);
        return z && zo;
    }
    __device__ Fp28 one() const { Fp28 r; if (c == 0) f_set(r, K28_ONE); else f_zero(r); return r; }
};
struct JacP { Fp28 x, y, z; };   // z == 0 <=> infinity

// a = 0 doubling (dbl-2009-l).  Infinity and y = 0 need no special case: Z3 = 2 Y Z vanishes.
// Ordered for few live values (at most five 14-register values across any call: the by-value routines keep the caller's
// values in the ~108 callee-saved VGPRs, anything beyond that is spilled around every call).
template <class F>
__device__ __forceinline__ void jac_dbl(const F& f, JacP& p) {
    Fp28 B = f.sqr(p.y);
    Fp28 Z3 = c_dbl(f.mul(p.y, p.z));
    Fp28 A = f.sqr(p.x);
    Fp28 t = f.sqr(c_add(p.x, B));
    Fp28 C = f.sqr(B);
    Fp28 D = c_dbl(c_sub(c_sub(t, A), C));
    Fp28 E = c_add(c_add(A, A), A);
    Fp28 X3 = c_sub(c_sub(f.sqr(E), D), D);
    Fp28 Y3 = c_sub(f.mul(E, c_sub(D, X3)), c_dbl(c_dbl(c_dbl(C))));
    p.x = f_vred(X3); p.y = f_vred(Y3); p.z = f_vred(Z3);
}
// mixed addition p += (qx, qy) (madd-2007-bl) with every exceptional case; the affine point comes through a loader (it is
// parked in LDS: held in registers it would be ten live values across every doubling of the scalar multiplication)
template <class F, class Q>
__device__ __forceinline__ void jac_madd(const F& f, JacP& p, Q&& ldq) {
    if (f.is_zero(p.z)) { p.x = ldq(0); p.y = ldq(1); p.z = f.one(); return; }
    Fp28 Z1Z1 = f.sqr(p.z);
    Fp28 H = c_sub(f.mul(ldq(0), Z1Z1), p.x);
    Fp28 rr = c_sub(f.mul(f.mul(ldq(1), p.z), Z1Z1), p.y);
    if (f.is_zero(H)) {
        if (f.is_zero(rr)) { p.x = ldq(0); p.y = ldq(1); p.z = f.one(); jac_dbl(f, p); return; }
        f_zero(p.z);   // P + (-P)
        return;
    }
    Fp28 HH = f.sqr(H);
    Fp28 Z3 = c_sub(c_sub(f.sqr(c_add(p.z, H)), Z1Z1), HH);
    rr = c_dbl(rr);
    Fp28 I = c_dbl(c_dbl(HH));
    Fp28 V = f.mul(p.x, I);
    Fp28 J = f.mul(H, I);
    Fp28 X3 = c_sub(c_sub(c_sub(f.sqr(rr), J), V), V);
    Fp28 Y3 = c_sub(f.mul(rr, c_sub(V, X3)), c_dbl(f.mul(p.y, J)));
    p.x = f_vred(X3); p.y = f_vred(Y3); p.z = f_vred(Z3);
}
// p = [k] Q, Q behind the loader, k given as nwords 64-bit words, MSB first
template <class F, class Q>
__device__ __forceinline__ void jac_mul(const F& f, JacP& p, Q&& ldq, const uint64_t* k, int nwords) {
    p.x = f.one(); p.y = f.one(); f_zero(p.z);
#pragma unroll 1
    for (int w = nwords - 1; w >= 0; w--) {
        const uint64_t e = k[w];
#pragma unroll 1
        for (int b = 63; b >= 0; b--) {
            jac_dbl(f, p);
            if ((e >> b) & 1) jac_madd(f, p, ldq);
        }
    }
}
// Jacobian p == affine (qx, qy) ?  (infinity never equals a finite point)
template <class F>
__device__ __forceinline__ bool jac_eq_affine(const F& f, const JacP& p, const Fp28& qx, const Fp28& qy) {
    if (f.is_zero(p.z)) return false;
    Fp28 z2 = f.sqr(p.z);
    Fp28 z3 = f.mul(z2, p.z);
    const bool ex = f.is_zero(c_sub(f.mul(qx, z2), p.x));
    const bool ey = f.is_zero(c_sub(f.mul(qy, z3), p.y));
    return ex && ey;
}

// one lane per point: 0 valid / 1 not on curve / 2 not torsion free  (-[X^2]P == (beta x, y), src/g1.rs:111-115)
// redo_only: the asm kernel ran first and left VALID_REDO where its chain met an exceptional case of the group law (infinity,
// P + P, P - P, order two: points outside the prime-order subgroup only) - this kernel, which handles every case, redoes those
constexpr uint8_t VALID_REDO = 0xff;
__global__ void __launch_bounds__(64, 2) k_g1_valid28(const uint64_t* g1, const uint8_t* inf, uint32_t n, uint8_t* status, int redo_only) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (redo_only && status[i] != VALID_REDO) return;
    if (inf && inf[i]) { status[i] = 0; return; }
    F1 f{0};
    Fp28 x, y;
    fp28_from_wire(x, g1 + 12 * (size_t)i);
    fp28_from_wire(y, g1 + 12 * (size_t)i + 6);
    Fp28 lhs = f.sqr(y);
    Fp28 rhs = c_add(f.mul(f.sqr(x), x), f_const(K28_B));
    if (!f.is_zero(c_sub(lhs, rhs))) { status[i] = 1; return; }
    const uint64_t x2[2] = {0x0000000100000000ULL, 0xac45a4010001a402ULL};   // X^2, X = 0xd201000000010000
    __shared__ int4 park[2 * 4 * 64];
    const int lane = threadIdx.x;
    valid_park(park, lane, 0, x);
    valid_park(park, lane, 1, y);
    auto ldq = [&](int v) -> Fp28 { return valid_unpark(park, lane, v); };
    JacP p;
    jac_mul(f, p, ldq, x2, 2);
    Fp28 bx = f.mul(ldq(0), f_const(K28_BETA));
    status[i] = jac_eq_affine(f, p, bx, c_neg(ldq(1))) ? 0 : 2;
}

// two lanes per point: psi(P) == -[X]P  (src/g2.rs:166-170)
__global__ void __launch_bounds__(64, 2) k_g2_valid28(const uint64_t* g2, const uint8_t* inf, uint32_t n, uint8_t* status, int redo_only) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const int c = (int)(tid & 1);
    uint32_t i = tid >> 1;
    const bool live = i < n;
    if (!live) i = n - 1;
    if (redo_only && !__any(live && status[i] == VALID_REDO)) return;      // wave-uniform: both lanes of a pair stay together
    const bool keep = redo_only && status[i] != VALID_REDO;
    const bool is_inf = inf && inf[i];
    __shared__ int4 park[2 * 4 * 64];
    const int lane = threadIdx.x;
    F2 f{c};
    Fp28 x, y;
    fp28_from_wire(x, g2 + 24 * (size_t)i + 6 * c);
    fp28_from_wire(y, g2 + 24 * (size_t)i + 12 + 6 * c);
    uint8_t st;
    Fp28 lhs = f.sqr(y);
    Fp28 rhs = c_add(f.mul(f.sqr(x), x), f_const(K28_B));     // b' = 4 (1 + u): both coefficients are 4
    if (!f.is_zero(c_sub(lhs, rhs))) {
        st = 1;
    } else {
        const uint64_t xs[1] = {0xd201000000010000ULL};
        valid_park(park, lane, 0, x);
        valid_park(park, lane, 1, y);
        auto ldq = [&](int v) -> Fp28 { return valid_unpark(park, lane, v); };
        JacP p;
        jac_mul(f, p, ldq, xs, 1);
        x = ldq(0); y = ldq(1);
        // psi(P) = (conj(x) PSI_X, conj(y) PSI_Y); compare [X]P with (psi_x, -psi_y)
        Fp28 cx = c ? c_neg(x) : x, cy = c ? c_neg(y) : y;
        Fp28 kx = f_const(c ? K28_PSI_X_1 : K28_PSI_X_0), ky = f_const(c ? K28_PSI_Y_1 : K28_PSI_Y_0);
        Fp28 px = f.mul(cx, kx), py = f.mul(cy, ky);
        st = jac_eq_affine(f, p, px, c_neg(py)) ? 0 : 2;
    }
    if (live && c == 0 && !keep) status[i] = is_inf ? 0 : st;
}

#if ZKP_VALID_ASM
// ---- the same two checks with the chain's steps as asm blocks (tools/validasm.py).  No exceptional case of the group law is
// handled in the steps: each of them sends Z to 0, Z = 0 is absorbing, and a chain that ends with Z = 0 mod p is handed to
// the generic kernel above (status VALID_REDO; launched behind this one with redo_only).  A point of the prime-order subgroup
// never takes that route: the chains' scalars (x^2, |x|) are below r.
// one lane per point, three waves per SIMD: -[x^2] P == (beta x, y)  (src/g1.rs:111-115)
__global__ void __launch_bounds__(64, 3) k_g1_valid_fast(const uint64_t* g1, const uint8_t* inf, uint32_t n, uint8_t* status) {
    extern __shared__ int4 park[];          // slots 0, 1: the affine point, 2: the addition's parked value; at LDS address 0
    const int lane = threadIdx.x;
    uint32_t i = blockIdx.x * 64 + lane;
    const bool live = i < n;
    if (!live) i = n - 1;
    const bool is_inf = inf && inf[i];
    bool on;
    int32_t X[NL], Y[NL], Z[NL];
    {
        Fp28 x, y;
        fp28_from_wire(x, g1 + 12 * (size_t)i);
        fp28_from_wire(y, g1 + 12 * (size_t)i + 6);
        Fp28 lhs, t, rhs;
        fp28_mul(lhs, y, y);
        fp28_mul(t, x, x);
        fp28_mul(rhs, t, x);
        on = f_is_zero(c_sub(lhs, c_add(rhs, f_const(K28_B))));
        valid_park(park, lane, 0, x);
        valid_park(park, lane, 1, y);
#pragma unroll
        for (int k = 0; k < NL; k++) { X[k] = x.l[k]; Y[k] = y.l[k]; Z[k] = K28_ONE[k]; }
    }
    if (!__any(on && !is_inf)) {            // wave-uniform: nothing to multiply
        if (live) status[i] = is_inf ? 0 : 1;
        return;
    }
    // the blocks own v6..v167 of the 168 registers of three waves per SIMD: what the epilogue needs of the lane is ONE packed
    // register here and is re-derived from an opaque copy of the lane number behind the loop
    int flags = (on ? 1 : 0) | (is_inf ? 2 : 0);
    static_assert(NL == 14, "the generated blocks are for 14 limbs");
#define ZKP_G1_STEP(BLOCK)                                                                                                               \
    do {                                                                                                                                 \
        constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};                                                                                     \
        asm volatile(BLOCK                                                                                                               \
                     : ZKP_G1_STEP_IO(X, Y, Z)                                                                                           \
                     : [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]),            \
                       [p6] "s"(PL[6]), [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]),        \
                       [p12] "s"(PL[12]), [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)                                                      \
                     : ZKP_G1_STEP_CLOBBERS);                                                                                            \
    } while (0)
    // X^2 = 0xac45a4010001a402_0000000100000000 (X = 0xd201000000010000): the accumulator starts at P (the leading one), then
    // one doubling per remaining bit and an addition where the bit is set
    const uint64_t hi = 0xac45a4010001a402ULL, lo = 0x0000000100000000ULL;
#pragma unroll 1
    for (int b = 126; b >= 0; b--) {
        ZKP_G1_STEP(ZKP_G1_DBL_ASM);
        if (((b >= 64 ? hi >> (b - 64) : lo >> b) & 1) != 0) ZKP_G1_STEP(ZKP_G1_MADD_ASM);
    }
    int l_ = threadIdx.x;
    asm volatile("" : "+v"(l_), "+v"(flags));
    Fp28 px, py, pz;
#pragma unroll
    for (int k = 0; k < NL; k++) { px.l[k] = X[k]; py.l[k] = Y[k]; pz.l[k] = Z[k]; }
    uint8_t st;
    if (f_is_zero(pz)) {
        st = VALID_REDO;
    } else {
        Fp28 z2, z3, bx, t;
        fp28_mul(z2, pz, pz);
        fp28_mul(z3, z2, pz);
        fp28_mul(bx, valid_unpark(park, l_, 0), f_const(K28_BETA));
        fp28_mul(t, bx, z2);
        const bool ex = f_is_zero(c_sub(t, px));
        fp28_mul(t, c_neg(valid_unpark(park, l_, 1)), z3);
        const bool ey = f_is_zero(c_sub(t, py));
        st = ex && ey ? 0 : 2;
    }
    const uint32_t i_ = blockIdx.x * 64 + (uint32_t)l_;
    if (i_ < n) status[i_] = (flags & 2) ? 0 : ((flags & 1) ? st : 1);
}

// two lanes per point: psi(P) == -[X] P  (src/g2.rs:166-170)
__global__ void __launch_bounds__(64, 2) k_g2_valid_fast(const uint64_t* g2, const uint8_t* inf, uint32_t n, uint8_t* status) {
    extern __shared__ int4 park[];          // values 0, 1: this lane's coefficient of the affine point; at LDS address 0
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x * 64 + lane;
    const int c = (int)(tid & 1);
    uint32_t i = tid >> 1;
    const bool live = i < n;
    if (!live) i = n - 1;
    const bool is_inf = inf && inf[i];
    F2 f{c};
    bool on;
    G2C r;
    {
        Fp28 x, y;
        fp28_from_wire(x, g2 + 24 * (size_t)i + 6 * c);
        fp28_from_wire(y, g2 + 24 * (size_t)i + 12 + 6 * c);
        Fp28 lhs = f.sqr(y);
        Fp28 rhs = c_add(f.mul(f.sqr(x), x), f_const(K28_B));     // b' = 4 (1 + u): both coefficients are 4
        on = f.is_zero(c_sub(lhs, rhs));
        valid_park(park, lane, 0, x);
        valid_park(park, lane, 1, y);
        r.x = x;
        r.y = y;
        r.z = f.one();
    }
    if (!__any(on && !is_inf)) {
        if (live && c == 0) status[i] = is_inf ? 0 : 1;
        return;
    }
#define ZKP_G2_STEP(BLOCK)                                                                                                               \
    do {                                                                                                                                 \
        constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};                                                                                     \
        asm volatile(BLOCK                                                                                                               \
                     : ZKP_G2_STEP_IO(r.x.l, r.y.l, r.z.l)                                                                               \
                     : [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]),            \
                       [p6] "s"(PL[6]), [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]),        \
                       [p12] "s"(PL[12]), [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)                                                      \
                     : ZKP_G2_STEP_CLOBBERS);                                                                                            \
    } while (0)
    const uint64_t xs = 0xd201000000010000ULL;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        ZKP_G2_STEP(ZKP_G2_DBL_ASM);
        if ((xs >> b) & 1) ZKP_G2_STEP(ZKP_G2_MADD_ASM);
    }
    uint8_t st;
    JacP p{r.x, r.y, r.z};
    if (f.is_zero(p.z)) {
        st = VALID_REDO;
    } else {
        Fp28 x = valid_unpark(park, lane, 0), y = valid_unpark(park, lane, 1);
        // psi(P) = (conj(x) PSI_X, conj(y) PSI_Y); compare [X] P with (psi_x, -psi_y)
        Fp28 cx = c ? c_neg(x) : x, cy = c ? c_neg(y) : y;
        Fp28 kx = f_const(c ? K28_PSI_X_1 : K28_PSI_X_0), ky = f_const(c ? K28_PSI_Y_1 : K28_PSI_Y_0);
        Fp28 qx = f.mul(cx, kx), qy = f.mul(cy, ky);
        st = jac_eq_affine(f, p, qx, c_neg(qy)) ? 0 : 2;
    }
    if (live && c == 0) status[i] = is_inf ? 0 : (on ? st : 1);
}

// the same check at THREE waves per SIMD (tools/validasm.py g2_dbl3 / g2_madd3): five register blocks + the Karatsuba product set =
// 168 VGPRs.  Between steps only Y is in registers; X and Z live in LDS slots 0 and 1 (slot 2 parks a value inside the addition), the
// affine point's limbs - needed by the five additions - in `qs`: quad q of value v (0: x, 1: y) of global lane t at
// qs[(v * 4 + q) * lanes + t].
__global__ void __launch_bounds__(64, 3) k_g2_valid_fast3(const uint64_t* g2, const uint8_t* inf, uint32_t n, uint8_t* status, int4* qs) {
    extern __shared__ int4 park[];
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x * 64 + lane;
    const uint32_t lanes = gridDim.x * 64;
    const int c = (int)(tid & 1);
    uint32_t i = tid >> 1;
    if (i >= n) i = n - 1;
    const bool is_inf = inf && inf[i];
    bool on;
    int32_t Y[NL];
    {
        F2 f{c};
        Fp28 x, y;
        fp28_from_wire(x, g2 + 24 * (size_t)i + 6 * c);
        fp28_from_wire(y, g2 + 24 * (size_t)i + 12 + 6 * c);
        Fp28 lhs = f.sqr(y);
        Fp28 rhs = c_add(f.mul(f.sqr(x), x), f_const(K28_B));
        on = f.is_zero(c_sub(lhs, rhs));
        valid_park(park, lane, 0, x);
        valid_park(park, lane, 1, f.one());
        const Fp28* v[2] = {&x, &y};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            qs[(size_t)(k * 4 + 0) * lanes + tid] = make_int4(v[k]->l[0], v[k]->l[1], v[k]->l[2], v[k]->l[3]);
            qs[(size_t)(k * 4 + 1) * lanes + tid] = make_int4(v[k]->l[4], v[k]->l[5], v[k]->l[6], v[k]->l[7]);
            qs[(size_t)(k * 4 + 2) * lanes + tid] = make_int4(v[k]->l[8], v[k]->l[9], v[k]->l[10], v[k]->l[11]);
            qs[(size_t)(k * 4 + 3) * lanes + tid] = make_int4(v[k]->l[12], v[k]->l[13], 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NL; k++) Y[k] = y.l[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the lane reads its own scratch records back inside the additions
    if (!__any(on && !is_inf)) {
        if (tid < 2 * n && c == 0) status[i] = is_inf ? 0 : 1;
        return;
    }
    int flags = (on ? 1 : 0) | (is_inf ? 2 : 0);
    const uint32_t qoff = tid * 16u, qstride = lanes * 16u;
    const uint32_t qlo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qs);
    const uint32_t qhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qs >> 32));
#define ZKP_G2W3_STEP(BLOCK)                                                                                                             \
    do {                                                                                                                                 \
        constexpr uint32_t PL[NL] = {ZKP28_P_LIMBS};                                                                                     \
        asm volatile(BLOCK                                                                                                               \
                     : ZKP_G2W3_STEP_IO(Y)                                                                                               \
                     : [qoff] "v"(qoff), [qstride] "s"(qstride), [qlo] "s"(qlo), [qhi] "s"(qhi),                                         \
                       [p0] "s"(PL[0]), [p1] "s"(PL[1]), [p2] "s"(PL[2]), [p3] "s"(PL[3]), [p4] "s"(PL[4]), [p5] "s"(PL[5]),            \
                       [p6] "s"(PL[6]), [p7] "s"(PL[7]), [p8] "s"(PL[8]), [p9] "s"(PL[9]), [p10] "s"(PL[10]), [p11] "s"(PL[11]),        \
                       [p12] "s"(PL[12]), [p13] "s"(PL[13]), [pinv] "s"(ZKP28_PINV)                                                      \
                     : ZKP_G2W3_STEP_CLOBBERS);                                                                                          \
    } while (0)
    const uint64_t xs = 0xd201000000010000ULL;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        ZKP_G2W3_STEP(ZKP_G2W3_DBL_ASM);
        if ((xs >> b) & 1) ZKP_G2W3_STEP(ZKP_G2W3_MADD_ASM);
    }
    int l_ = threadIdx.x;
    asm volatile("" : "+v"(l_), "+v"(flags));
    const uint32_t t_ = blockIdx.x * 64 + (uint32_t)l_;
    const int c_ = (int)(t_ & 1);
    uint32_t i_ = t_ >> 1;
    const bool live_ = i_ < n;
    if (!live_) i_ = n - 1;
    F2 f{c_};
    JacP p;
    p.x = valid_unpark(park, l_, 0);
    p.z = valid_unpark(park, l_, 1);
#pragma unroll
    for (int k = 0; k < NL; k++) p.y.l[k] = Y[k];
    uint8_t st;
    if (f.is_zero(p.z)) {
        st = VALID_REDO;
    } else {
        Fp28 x, y;
        fp28_from_wire(x, g2 + 24 * (size_t)i_ + 6 * c_);
        fp28_from_wire(y, g2 + 24 * (size_t)i_ + 12 + 6 * c_);
        Fp28 cx = c_ ? c_neg(x) : x, cy = c_ ? c_neg(y) : y;
        Fp28 kx = f_const(c_ ? K28_PSI_X_1 : K28_PSI_X_0), ky = f_const(c_ ? K28_PSI_Y_1 : K28_PSI_Y_0);
        Fp28 qx = f.mul(cx, kx), qy = f.mul(cy, ky);
        st = jac_eq_affine(f, p, qx, c_neg(qy)) ? 0 : 2;
    }
    if (live_ && c_ == 0) status[i_] = (flags & 2) ? 0 : ((flags & 1) ? st : 1);
}
#endif

// a^(p-2) (Fermat; reference src/fp.rs:307-319); a == 0 gives 0.  Kept as the cross-check of f_inv (ZKP_INV_FERMAT).
__device__ __forceinline__ Fp28 f_inv_fermat(const Fp28& a) {
    Fp28 res = f_const(K28_ONE);
#pragma unroll 1
    for (int w = NL - 1; w >= 0; w--) {
        const uint32_t e = (uint32_t)K28_P[w] - (w == 0 ? 2u : 0u);
#pragma unroll 1
        for (int b = W - 1; b >= 0; b--) {
            fp28_mul(res, res, res);
            if ((e >> b) & 1) fp28_mul(res, res, a);
        }
    }
    return res;
}

// a^-1 by Bernstein-Yang division steps ("safegcd", delta = 1 variant; 0 gives 0): the same value as the reference's
// a^(p-2) (src/fp.rs:307-319) with ~26 k instead of ~240 k instructions on the single-lane critical path of the final
// exponentiation.  f = p, g = a as 13 signed limbs of 30 bits; 37 batches of 30 division steps (1110 >= the proven
// bound (49 * 381 + 57) / 17 = 1101 for 381-bit inputs); each batch runs on the low limbs, yields a 2x2 transition matrix
// (entries < 2^30 in magnitude) and is applied to (f, g) exactly and to (d, e) modulo p, d and e staying in (-2p, p).
// Straight-line code: no lane diverges.  Layout of the arithmetic follows libsecp256k1's modinv32 (public domain
// algorithm description in its safegcd_implementation.md); tools/safegcd_model.py is the limb-exact model it was
// checked against on the CPU.
namespace sg {
constexpr int N = 13;
constexpr int32_t M30 = 0x3fffffff;
__device__ __constant__ const int32_t PL[N] = {ZKP30_P_LIMBS};

// Round 6: the transition matrix is tracked in PACKED 16-bit halves - (u, v) in one register, (q, r) in another - over three runs of ten steps
// (entries <= 2^10), and the three 2 x 2 matrices are multiplied together afterwards (24-bit multiplies; the products wrap mod 2^32 and
// the true entries are <= 2^30).  A conditional negation / masked addition / doubling of BOTH entries of a row is one v_pk_* instruction:
// 19 instead of 24 instructions per step.  The same matrix as the plain 30-step loop (ZKP_SG_PLAIN_DIVSTEPS=1 builds that one: the A/B
// baseline), hence the same inverse: tests/test_gpu_parity.py::test_divstep_inversion_equals_fermat.
#ifndef ZKP_SG_PLAIN_DIVSTEPS
#define ZKP_SG_PLAIN_DIVSTEPS 0
#endif
typedef short zkp_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void divsteps30(int32_t& eta, uint32_t f, uint32_t g, int32_t& u_, int32_t& v_, int32_t& q_, int32_t& r_) {
#if ZKP_SG_PLAIN_DIVSTEPS
    uint32_t u = 1, v = 0, q = 0, r = 1, e = (uint32_t)eta;
#pragma unroll 6      // five trips instead of thirty - a taken branch costs a lone wavefront about three of the step's 24 instructions
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)((int32_t)e >> 31);          // eta < 0  <=>  delta > 0
        const uint32_t c2 = 0u - (g & 1u);
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                              // delta > 0 and g odd: swap roles
        e = (e ^ c1) - (c1 + 1u);
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    eta = (int32_t)e; u_ = (int32_t)u; v_ = (int32_t)v; q_ = (int32_t)q; r_ = (int32_t)r;
#else
    uint32_t e = (uint32_t)eta;
    int32_t U = 1, V = 0, Q = 0, R = 1;                        // the product of the runs so far
#pragma unroll
    for (int run = 0; run < 3; run++) {
        zkp_s2 P = {1, 0}, T = {0, 1};                         // rows (u, v) and (q, r) of this run's matrix
#pragma unroll 5
        for (int i = 0; i < 10; i++) {
            uint32_t c1 = (uint32_t)((int32_t)e >> 31);        // eta < 0  <=>  delta > 0
            const uint32_t c2 = 0u - (g & 1u);
            const uint32_t x = (f ^ c1) - c1;
            const zkp_s2 m1 = __builtin_bit_cast(zkp_s2, c1), m2 = __builtin_bit_cast(zkp_s2, c2);
            const zkp_s2 Y = (P ^ m1) - m1;                    // (-u, -v) where delta > 0
            g += x & c2;
            T += Y & m2;
            c1 &= c2;                                          // delta > 0 and g odd: swap roles
            e = (e ^ c1) - (c1 + 1u);
            f += g & c1;
            P += T & __builtin_bit_cast(zkp_s2, c1);
            g >>= 1;
            P += P;                                            // (u, v) <<= 1
        }
        const int32_t u2 = P.x, v2 = P.y, q2 = T.x, r2 = T.y;
        if (run == 0) {
            U = u2; V = v2; Q = q2; R = r2;
        } else {                                               // M <- M_run * M: entries of M_run <= 2^10, of M <= 2^20: 24-bit operands
            const int32_t nU = __mul24(u2, U) + __mul24(v2, Q), nV = __mul24(u2, V) + __mul24(v2, R);
            const int32_t nQ = __mul24(q2, U) + __mul24(r2, Q), nR = __mul24(q2, V) + __mul24(r2, R);
            U = nU; V = nV; Q = nQ; R = nR;
        }
    }
    eta = (int32_t)e; u_ = U; v_ = V; q_ = Q; r_ = R;
#endif
}
// Round 6: the limbs and matrix entries are pinned to 32-bit registers (an empty asm the optimiser cannot see through).  Without it
// LLVM carries a limb as the masked 64-bit carry word it came from and expands every (int64) u * limb into the 64 x 64-bit pattern
// (v_mad_u64_u32 x 2 + v_mul_lo_u32 x 2 + moves: 562 instructions for the two updates of a batch); with 32-bit operands each product
// is ONE v_mad_i64_i32.  k_batch_inv is one lane's dependent chain - the six launches of a pass are what a small or medium batch pays in
// full (profiles/r06: 0.11-0.18 ms each, whatever the batch) - so instructions are time here.
#define ZKP_SG_PIN32(x) asm volatile("" : "+v"(x))
// (f, g) <- (u f + v g, q f + r g) / 2^30, exactly
__device__ __forceinline__ void update_fg(int32_t* f, int32_t* g, int32_t u, int32_t v, int32_t q, int32_t r) {
    ZKP_SG_PIN32(u); ZKP_SG_PIN32(v); ZKP_SG_PIN32(q); ZKP_SG_PIN32(r);
    int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
    int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
    cf >>= 30; cg >>= 30;                                      // the low 30 bits are zero by construction
#pragma unroll
    for (int i = 1; i < N; i++) {
        cf += (int64_t)u * f[i] + (int64_t)v * g[i];
        cg += (int64_t)q * f[i] + (int64_t)r * g[i];
        f[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g[i - 1] = (int32_t)cg & M30; cg >>= 30;
        ZKP_SG_PIN32(f[i - 1]); ZKP_SG_PIN32(g[i - 1]);
    }
    f[N - 1] = (int32_t)cf; g[N - 1] = (int32_t)cg;
    ZKP_SG_PIN32(f[N - 1]); ZKP_SG_PIN32(g[N - 1]);
}
// (d, e) <- (u d + v e, q d + r e) / 2^30 mod p; inputs and outputs in (-2p, p)
__device__ __forceinline__ void update_de(int32_t* d, int32_t* e, int32_t u, int32_t v, int32_t q, int32_t r) {
    ZKP_SG_PIN32(u); ZKP_SG_PIN32(v); ZKP_SG_PIN32(q); ZKP_SG_PIN32(r);
    const int32_t sd = d[N - 1] >> 31, se = e[N - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);   // add p to a negative d / e first
    int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
    int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
    md -= (int32_t)((ZKP30_PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);   // multiple of p that clears the low 30 bits
    me -= (int32_t)((ZKP30_PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    ZKP_SG_PIN32(md); ZKP_SG_PIN32(me);
    cd += (int64_t)PL[0] * md; ce += (int64_t)PL[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < N; i++) {
        cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)PL[i] * md;
        ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)PL[i] * me;
        d[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e[i - 1] = (int32_t)ce & M30; ce >>= 30;
        ZKP_SG_PIN32(d[i - 1]); ZKP_SG_PIN32(e[i - 1]);
    }
    d[N - 1] = (int32_t)cd; e[N - 1] = (int32_t)ce;
    ZKP_SG_PIN32(d[N - 1]); ZKP_SG_PIN32(e[N - 1]);
}
// x in (-2p, p), negated when sign < 0, brought to [0, p)
__device__ __forceinline__ void normalize(int32_t* x, int32_t sign) {
    int32_t add = x[N - 1] >> 31;
    const int32_t neg = sign >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        int32_t t = x[i] + (PL[i] & add);
        t = (t ^ neg) - neg;
        t += c;
        if (i < N - 1) { x[i] = t & M30; c = t >> 30; } else x[i] = t;
    }
    add = x[N - 1] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int32_t t = x[i] + (PL[i] & add) + c;
        if (i < N - 1) { x[i] = t & M30; c = t >> 30; } else x[i] = t;
    }
}
}  // namespace sg

__device__ __noinline__ Fp28 f_inv(Fp28 a) {
    // canonical value of the stored (Montgomery) representative, repacked from 14 x 28 to 13 x 30 bits.  a * R / R
    // first: the same value, but certainly inside canon28's input range whatever linear combination a came from
    uint32_t c28[NL];
    {
        Fp28 t;
        fp28_mul(t, a, f_const(K28_ONE));
        canon28(c28, t.l);
    }
    int32_t f[sg::N], g[sg::N], d[sg::N], e[sg::N];
#pragma unroll
    for (int i = 0; i < sg::N; i++) {
        const int bit = 30 * i, wd = bit / W, sh = bit % W;
        uint64_t v = (uint64_t)c28[wd] >> sh;
        if (wd + 1 < NL) v |= (uint64_t)c28[wd + 1] << (W - sh);
        if (wd + 2 < NL) v |= (uint64_t)c28[wd + 2] << (2 * W - sh);
        g[i] = (int32_t)(v & (uint64_t)sg::M30);
        f[i] = sg::PL[i];
        d[i] = 0;
        e[i] = i == 0 ? 1 : 0;
    }
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 37; it++) {
        int32_t u, v, q, r;
        sg::divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], u, v, q, r);
        sg::update_de(d, e, u, v, q, r);
        sg::update_fg(f, g, u, v, q, r);
    }
    sg::normalize(d, f[sg::N - 1]);            // f = +-1 (or +-p with d = 0 when a == 0)
    // 13 x 30 -> 14 x 28 bits; the value is (a R)^-1, and (a R)^-1 * R^3 / R = a^-1 R: the Montgomery form of a^-1
    Fp28 w;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = W * i, wd = bit / 30, sh = bit % 30;
        uint64_t v = (uint64_t)(uint32_t)d[wd] >> sh;
        if (wd + 1 < sg::N) v |= (uint64_t)(uint32_t)d[wd + 1] << (30 - sh);
        w.l[i] = (int32_t)(v & (uint64_t)MASK);
    }
    Fp28 res;
    fp28_mul(res, w, f_const(K28_R3));
    return res;
}

// [k] P for 256-bit scalars, affine in / affine out (&G1Affine * &Fr, reference src/g1.rs:130-153 without its dropped
// bit 0; src/g2.rs:185-208).  One lane per G1 point; two lanes per G2 point.
__global__ void __launch_bounds__(64, 2) k_g1_mul28(const uint64_t* base, size_t stride, const uint64_t* sc, uint32_t n, uint64_t* out, uint8_t* out_inf) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    F1 f{0};
    Fp28 x, y;
    fp28_from_wire(x, base + stride * i);
    fp28_from_wire(y, base + stride * i + 6);
    const uint64_t k[4] = {sc[4 * (size_t)i], sc[4 * (size_t)i + 1], sc[4 * (size_t)i + 2], sc[4 * (size_t)i + 3]};
    __shared__ int4 park[2 * 4 * 64];
    const int lane = threadIdx.x;
    valid_park(park, lane, 0, x);
    valid_park(park, lane, 1, y);
    auto ldq = [&](int v) -> Fp28 { return valid_unpark(park, lane, v); };
    JacP p;
    jac_mul(f, p, ldq, k, 4);
    const bool inf = f.is_zero(p.z);
    Fp28 zi = f_inv(p.z);
    Fp28 zi2 = f.sqr(zi);
    Fp28 ax = f.mul(p.x, zi2), ay = f.mul(p.y, f.mul(zi2, zi));
    if (inf) { f_zero(ax); ay = f_const(K28_ONE); }      // identity is (0, 1, infinity), reference src/g1.rs:25-31
    fp28_to_wire(out + 12 * (size_t)i, ax);
    fp28_to_wire(out + 12 * (size_t)i + 6, ay);
    if (out_inf) out_inf[i] = inf ? 1 : 0;
}
__global__ void __launch_bounds__(64, 2) k_g2_mul28(const uint64_t* base, size_t stride, const uint64_t* sc, uint32_t n, uint64_t* out, uint8_t* out_inf) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const int c = (int)(tid & 1);
    uint32_t i = tid >> 1;
    const bool live = i < n;
    if (!live) i = n - 1;
    F2 f{c};
    Fp28 x, y;
    fp28_from_wire(x, base + stride * i + 6 * c);
    fp28_from_wire(y, base + stride * i + 12 + 6 * c);
    const uint64_t k[4] = {sc[4 * (size_t)i], sc[4 * (size_t)i + 1], sc[4 * (size_t)i + 2], sc[4 * (size_t)i + 3]};
    __shared__ int4 park[2 * 4 * 64];
    const int lane = threadIdx.x;
    valid_park(park, lane, 0, x);
    valid_park(park, lane, 1, y);
    auto ldq = [&](int v) -> Fp28 { return valid_unpark(park, lane, v); };
    JacP p;
    jac_mul(f, p, ldq, k, 4);
    const bool inf = f.is_zero(p.z);
    // 1 / (z0 + z1 u) = (z0 - z1 u) / (z0^2 + z1^2)   (reference src/fp2.rs:278-296); both lanes invert the norm
    Fp28 o;
    swap_pair(o, p.z);
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, p.z.l, p.z.l);
    acc_mul(acc, o.l, o.l);
    Fp28 nrm;
    acc_reduce(nrm.l, acc);
    Fp28 ninv = f_inv(nrm);
    Fp28 zi = f_mul_v(c ? c_neg(p.z) : p.z, ninv);
    Fp28 zi2 = f.sqr(zi);
    Fp28 ax = f.mul(p.x, zi2), ay = f.mul(p.y, f.mul(zi2, zi));
    if (inf) { f_zero(ax); ay = f.one(); }
    if (live) {
        fp28_to_wire(out + 24 * (size_t)i + 6 * c, ax);
        fp28_to_wire(out + 24 * (size_t)i + 12 + 6 * c, ay);
        if (out_inf && c == 0) out_inf[i] = inf ? 1 : 0;
    }
}

// out = in^-1 for `count` planes of per-check Fp elements: plane j of the input is state element elem_n + j, of the output
// elem_ninv + j (n_checks values each, record stride nc).  The final exponentiation's single inversion is
// (ST_N, ST_NINV, 1), the decompression of the snapshots of an x-power chain (ST_KN, ST_KNINV, 3).  One lane inverts
// B of the count * n_checks values (i, i + L, i + 2L, ...; L lanes) with Montgomery's simultaneous inversion: exclusive
// prefix products parked in the output records, ONE inversion (a^(p-2) in the reference, src/fp.rs:307-319) of the total,
// then two multiplications per value on the way back.
// A zero element (a non-invertible final_exponentiation input, the identity's compressed form) is replaced by one in the
// chain and gets 0, as Fermat gives.
__global__ void __launch_bounds__(64) k_batch_inv(int4* state, uint32_t n_checks_in, uint32_t nc, uint32_t Bf, uint32_t elem_n, uint32_t elem_ninv,
                                                  uint32_t count, NDev nd) {
    const uint32_t n_checks = eff_n(n_checks_in, nd);
    const uint32_t B = Bf & 0x7fffffffu;
    const bool fermat = Bf >> 31;               // cross-check path (ZKP_COOP_INV_FERMAT=1)
    const uint32_t total = n_checks * count;
    const uint32_t L = (total + B - 1) / B;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= L) return;
    auto recno = [&](uint32_t idx) -> size_t { const uint32_t j = idx / n_checks; return ((size_t)j * nc + (idx - j * n_checks)) * 4; };
    int4* N = state + (size_t)elem_n * nc * 4;
    int4* I = state + (size_t)elem_ninv * nc * 4;
    Fp28 acc = f_const(K28_ONE);
    uint32_t cnt = 0;
    // both loops ask for the next element's records before they work on the current one: a lane's chain is strictly
    // sequential and only a few wavefronts share a SIMD here, so an exposed load latency per step is not covered by anyone
    Fp28 e_next;
    rec_load(e_next, N + recno(i));
#pragma unroll 1
    for (uint32_t idx = i; idx < total && cnt < B; idx += L, cnt++) {
        Fp28 e = e_next;
        if (idx + L < total && cnt + 1 < B) rec_load(e_next, N + recno(idx + L));
        if (B > 1) {
            rec_store(I + recno(idx), acc);
            if (f_is_zero(e)) e = f_const(K28_ONE);
            fp28_mul(acc, acc, e);
        } else {
            acc = e;
        }
    }
    Fp28 inv = fermat ? f_inv_fermat(acc) : f_inv(acc);
    if (B == 1) { rec_store(I + recno(i), inv); return; }
    Fp28 pre_next;
    if (cnt) {
        const size_t at = recno(i + (cnt - 1) * L);
        rec_load(e_next, N + at);
        rec_load(pre_next, I + at);
    }
#pragma unroll 1
    for (uint32_t j = cnt; j-- > 0;) {
        const size_t at = recno(i + j * L);
        Fp28 e = e_next, pre = pre_next, r;
        if (j) {
            const size_t nx = recno(i + (j - 1) * L);
            rec_load(e_next, N + nx);
            rec_load(pre_next, I + nx);
        }
        if (f_is_zero(e)) {
            f_zero(r);
        } else {
            fp28_mul(r, inv, pre);
            fp28_mul(inv, inv, e);
        }
        rec_store(I + at, r);
    }
}

// ---- decompression of snapshots (z2..z5 known, z0 and z1 to recover; Karabina, ePrint 2010/542 Theorem 3.1 in this
// tower's coordinates - tools/coopgen.py checks the identities against the big-integer model):
//     z2 != 0:  z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)          z2 == 0:  z1 = 2 z4 z5 / z3   (0 / 0 := 0: the identity)
//     z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1
// Two lanes per (snapshot, check) - lane parity = Fp2 coefficient, as in k_prep_lines.  k_kdec_a leaves the numerator N in the
// snapshot's z1 records and n = |D|^2 (x 4 where D = z2) in plane elem_n + snapshot; k_batch_inv inverts the planes; k_kdec_b
// finishes: 1 / D = conj(D) / |D|^2.  The denominator D is z2 or z3 itself: since round 5 it is not stored a second time - which of
// the two it is travels in the padding of N's record (third dword of the last quad), and k_kdec_b reads z2 and z3 anyway.
__global__ void __launch_bounds__(64, 2) k_kdec_a(int4* state, uint32_t n_checks_in, uint32_t nc, uint32_t elem_snap, uint32_t count, uint32_t elem_n,
                                                  NDev nd) {
    const uint32_t n_checks = eff_n(n_checks_in, nd);
    if (blockIdx.x * 32u >= n_checks * count) return;
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const int c = (int)(tid & 1);
    uint32_t e = tid >> 1;
    bool live = e < n_checks * count;
    if (!live) e = n_checks * count - 1;
    const uint32_t sn = e / n_checks, check = e - sn * n_checks;
#if defined(ZKP_EXP_TRAFFIC4) && (ZKP_EXP_TRAFFIC4 & 2)
    int4* const st = state + (size_t)(check & ~3u) * 4;
    if (check & 3) live = false;
#else
    int4* const st = state + (size_t)check * 4;
#endif
    const uint32_t base = elem_snap + 12 * sn;
    auto rec = [&](uint32_t el) -> int4* { return st + (size_t)el * nc * 4; };
    F2 f{c};
    Fp28 z2, z3, z4, z5;
    rec_load(z2, rec(base + 6 + c));
    rec_load(z3, rec(base + 4 + c));
    rec_load(z4, rec(base + 2 + c));
    rec_load(z5, rec(base + 10 + c));
    const bool z2_zero = f.is_zero(z2);
    Fp28 s5 = f.sqr(z5), o;
    swap_pair(o, s5);
    Fp28 na = c ? c_add(o, s5) : c_sub(s5, o);                 // xi z5^2
    Fp28 s4 = f.sqr(z4);
    na = c_sub(c_add(na, c_add(c_dbl(s4), s4)), c_dbl(z3));
    Fp28 nb = na;
    if (__any(z2_zero)) nb = c_dbl(f.mul(z4, z5));             // wave-uniform: almost never taken (z2 = 0: the identity)
    Fp28 N = z2_zero ? nb : na, D = z2_zero ? z3 : z2;
    swap_pair(o, D);
    Acc acc;
    acc_zero(acc);
    acc_mul(acc, D.l, D.l);
    acc_mul(acc, o.l, o.l);
    Fp28 n;
    acc_reduce(n.l, acc);
    if (!z2_zero) n = c_dbl(c_dbl(n));
    if (live) {
        int4* dn = rec(base + 8 + c);
        dn[0] = make_int4(N.l[0], N.l[1], N.l[2], N.l[3]);
        dn[1] = make_int4(N.l[4], N.l[5], N.l[6], N.l[7]);
        dn[2] = make_int4(N.l[8], N.l[9], N.l[10], N.l[11]);
        dn[3] = make_int4(N.l[12], N.l[13], z2_zero ? 1 : 0, 0);       // the tag: D = z3 (1) or z2 (0)
        if (c == 0) rec_store(rec(elem_n + sn), n);
    }
}
#ifndef ZKP_KDEC_MERGED
#define ZKP_KDEC_MERGED 1
#endif
__global__ void __launch_bounds__(64, 3) k_kdec_b(int4* state, uint32_t n_checks_in, uint32_t nc, uint32_t elem_snap, uint32_t count, uint32_t elem_ninv,
                                                  NDev nd) {
    const uint32_t n_checks = eff_n(n_checks_in, nd);
    if (blockIdx.x * 32u >= n_checks * count) return;
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const int c = (int)(tid & 1);
    uint32_t e = tid >> 1;
    bool live = e < n_checks * count;
    if (!live) e = n_checks * count - 1;
    const uint32_t sn = e / n_checks, check = e - sn * n_checks;
#if defined(ZKP_EXP_TRAFFIC4) && (ZKP_EXP_TRAFFIC4 & 2)
    int4* const st = state + (size_t)(check & ~3u) * 4;
    if (check & 3) live = false;
#else
    int4* const st = state + (size_t)check * 4;
#endif
    const uint32_t base = elem_snap + 12 * sn;
    auto rec = [&](uint32_t el) -> int4* { return st + (size_t)el * nc * 4; };
    F2 f{c};
    Fp28 N, D, ninv, z2, z3;
    int d_is_z3;
    {
        const int4* sn_ = rec(base + 8 + c);
        const int4 v0 = sn_[0], v1 = sn_[1], v2 = sn_[2], v3 = sn_[3];
        N.l[0] = v0.x; N.l[1] = v0.y; N.l[2] = v0.z; N.l[3] = v0.w; N.l[4] = v1.x; N.l[5] = v1.y; N.l[6] = v1.z; N.l[7] = v1.w;
        N.l[8] = v2.x; N.l[9] = v2.y; N.l[10] = v2.z; N.l[11] = v2.w; N.l[12] = v3.x; N.l[13] = v3.y;
        d_is_z3 = v3.z;                                                 // k_kdec_a's tag
    }
    rec_load(z2, rec(base + 6 + c));
    rec_load(ninv, rec(elem_ninv + sn));
    D = z2;
    if (__any(d_is_z3 != 0)) {       // wave-uniform and almost never taken (z2 = 0: the identity, pairs with an infinity): D = z3
        Fp28 z3e;
        rec_load(z3e, rec(base + 4 + c));
#pragma unroll
        for (int i = 0; i < NL; i++) D.l[i] = d_is_z3 ? z3e.l[i] : z2.l[i];
    }
    Fp28 dinv = f_mul_v(c ? c_neg(D) : D, ninv);               // conj(D) / |D|^2
    Fp28 z1 = f.mul(N, dinv);
#if ZKP_KDEC_MERGED
    // t = 2 z1^2 + z2 z5 - 3 z3 z4 under ONE reduction (five products; column budget 2 * 2 * 2 + 2 + 3 * 2 = 16 <= 30; a reduced
    // value, no renormalisation): coefficient c of a b is x1 b + x2 b' with (x1, x2) = (a, -a') or (a', a), ' = the partner's
    Fp28 t;
    {
        Acc acc;
        acc_zero(acc);
        auto mulacc = [&](const Fp28& a, const Fp28& b, int k) {
            Fp28 ao, bo;
            swap_pair(ao, a);
            swap_pair(bo, b);
            int32_t x1[NL], x2[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                x1[i] = k * (c ? ao.l[i] : a.l[i]);
                x2[i] = k * (c ? a.l[i] : -ao.l[i]);
            }
            acc_mul(acc, x1, b.l);
            acc_mul(acc, x2, bo.l);
        };
        {
            Fp28 o1;
            swap_pair(o1, z1);
            int32_t x[NL], y[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                x[i] = 2 * (o1.l[i] + (c ? o1.l[i] : z1.l[i]));
                y[i] = z1.l[i] - (c ? 0 : o1.l[i]);
            }
            acc_mul(acc, x, y);
        }
        {
            Fp28 z5;
            rec_load(z5, rec(base + 10 + c));
            mulacc(z2, z5, 1);
        }
        {
            Fp28 z4;
            rec_load(z3, rec(base + 4 + c));
            rec_load(z4, rec(base + 2 + c));
            mulacc(z3, z4, -3);
        }
        acc_reduce(t.l, acc);
    }
    Fp28 o;
    swap_pair(o, t);
#else
    Fp28 z4, z5;
    rec_load(z5, rec(base + 10 + c));
    Fp28 t = c_add(c_dbl(f.sqr(z1)), f.mul(z2, z5));
    rec_load(z3, rec(base + 4 + c));
    rec_load(z4, rec(base + 2 + c));
    Fp28 m34 = f.mul(z3, z4);
    t = f_vred(c_sub(t, c_add(c_dbl(m34), m34)));
    Fp28 o;
    swap_pair(o, t);
#endif
    Fp28 z0 = c ? c_add(o, t) : c_add(c_sub(t, o), f_const(K28_ONE));   // xi t + 1
    z0 = f_vred(z0);
    if (live) {
        rec_store(rec(base + c), z0);
        rec_store(rec(base + 8 + c), z1);
    }
}

// batched field operation on wire operands through the 28-bit core (zkp_fp_op_batch with ZKP_FP_CORE28): 0 mul, 1 add,
// 2 sub, 3 neg, 4 square, 5 invert (0 gives 0).  Reference: Fp::mul / add / sub / neg / square / invert, src/fp.rs:352-455.
__global__ void k_fp28_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp28 x, y, r;
    fp28_from_wire(x, a + 6 * i);
    if (op <= 2) fp28_from_wire(y, b + 6 * i);
    switch (op) {
        case 0: fp28_mul(r, x, y); break;
        case 1: f_add(r, x, y); break;
        case 2: f_sub(r, x, y); break;
        case 3: r = c_neg(x); break;
        case 4: fp28_mul(r, x, x); break;
        default: r = f_inv(x); break;
    }
    fp28_to_wire(out + 6 * i, r);
}

// the primer of prime(): a grid of one-wavefront workgroups that leave at once
__global__ void __launch_bounds__(64) k_primer() {}

// measurement only (ZKP_TIME_FILL=256): pseudo-random balanced 28-bit limbs as the timing hook's synthetic state - what real operands look like
__global__ void k_fill_rand(int32_t* p, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (int32_t)((uint32_t)x >> 4) - (1 << 27);
}

}  // namespace

// =============================================================================== host side
namespace zkp {

struct CoopProgDev { uint32_t* hdr; uint32_t* tbl; uint4* rtbl; uint32_t nslot; uint32_t nconst; uint32_t wide; };
constexpr int MAX_PIPES = plan::MAX_PIPES;
struct CoopDev;
struct CoopPipe {            // one in-flight chunk: its own workspace and (for pipes > 0) its own stream
    int4* lines;  size_t lines_bytes;
    int4* state;  size_t state_bytes;
    hipStream_t stream;
    hipEvent_t done;
    CoopDev* owner;
    // transient, set on the copy a chunk's launches are given: the check count in device memory and this chunk's offset into the list
    // it counts (coop_pairing's n_dev argument); {nullptr, 0} otherwise
    NDev nd;
};
struct CoopDev {
    CoopProgDev progs[ZKP_PROG_COUNT];
    int4* consts;
    CoopPipe pipe[MAX_PIPES];
    plan::Knobs kn;          // chunking / splitting knobs, clamped (zkp_plan.hpp: the arithmetic the CPU test walks under sanitizers)
    bool inv_fermat;         // a^(p-2) instead of the division-step inversion (cross-check)
    int cus;                 // compute units of the device
    int prime_mask;          // ZKP_COOP_PRIME_MASK (experiments): which kernel classes are primed
    bool prime_now;          // set per super-chunk by two_phase: batches of at most one chunk only (a larger one keeps the GPU full: nothing to place)
    int prime;               // ZKP_COOP_PRIME: small grids in the dispatcher's bad bands are preceded by an empty grid of the same size (prime())
    int4* big_state;         // per-check state of a whole super-chunk (7.9 KB per check)
    size_t big_state_bytes;
    int4* vscratch;          // k_g2_valid_fast3: Montgomery limbs of the affine points (2 values x 4 quads per lane)
    size_t vscratch_bytes;
    hipEvent_t ready;
    // coop_profile_pairing: every launch of one pass bracketed by events on its stream (ONE pipeline, so that no two kernels overlap)
    struct ProfEv { int cls; hipEvent_t a, b; };
    std::vector<ProfEv>* prof;     // null: not profiling
    bool prof_phase_c;
    const uint32_t* n_dev;         // for the duration of one coop_pairing call: the device-resident check count (or null)
};

enum { PRIME_COOP = 1, PRIME_PREP = 2, PRIME_KSQ = 4, PRIME_INV = 8, PRIME_KDEC = 16 };      // kernel classes of prime()
// kernel classes of coop_profile_pairing (include/zkp_pairings.h ZKP_PROFILE_*)
enum { PROF_PREP = 0, PROF_MILLER, PROF_FEXP_A, PROF_INV, PROF_KSQ, PROF_KDEC_A, PROF_KDEC_B, PROF_C_DEEP, PROF_C_PLAIN, PROF_CLASSES };
struct ProfScope {
    CoopDev* d; hipStream_t s; hipEvent_t b = nullptr;
    ProfScope(CoopDev* d_, hipStream_t s_, int cls) : d(d_), s(s_) {      // d may be null: a launch outside any context's profile
        if (!d || !d->prof) return;
        hipEvent_t a = nullptr;
        if (hipEventCreate(&a) != hipSuccess) return;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); b = nullptr; return; }
        (void)hipEventRecord(a, s);
        d->prof->push_back({cls, a, b});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, s); }
};

// The MULACC table of a program with every LDS address resolved per LANE (the asm block of k_coop reads it: no address
// arithmetic in the term loop).  Row r = table offset / 12 + term, 64 lanes per row, one uint4 per lane:
//   x = byte address of the A1 record (plane 0), y = B1, z = A2 | B2 << 16, w = 0;
// the row behind a step's last term carries the step's per-lane flag words: x = negate bits (bit t = term t) | doubled-A
// bits << 12, y = subtract-A2 bits | subtract-B2 bits << 12.  Lane -> (group, lane in group) and the slot -> address map are
// k_coop's (gbase, slot_off); lanes 60..63 compute on group 4's addresses and never store.
static void coop_resolve_table(const ZkpProgDesc& p, std::vector<uint4>& rt) {
    const int S = coop_cfg_slots(p.wide), SC = coop_cfg_consts(p.wide);
    const int SG = coop_group_stride(S);
    const size_t rows = p.n_tbl / LIG + 2;
    constexpr int RL = 64 * WGW;   // lanes per row
    rt.assign(rows * RL, make_uint4(0, 0, 0, 0));
    auto addr = [&](uint32_t slot, int grp) -> uint32_t { return 16u * (((slot & 64) ? 0 : SC + grp * SG) + (slot & 63)); };
    for (uint32_t pc = 0; pc * 4 + 3 < p.n_hdr; pc++) {
        const uint32_t h0 = p.hdr[4 * pc], off = p.hdr[4 * pc + 2];
        if ((h0 & 0xff) != OP_MULACC) continue;
        const uint32_t T = (h0 >> 8) & 0xff;
        for (int lane = 0; lane < RL; lane++) {
            int g0 = lane / LIG, lig = lane - g0 * LIG, grp = g0 < GROUPS ? g0 : GROUPS - 1;
            if (WGW > 1) {
                const int wv = lane >> 6, wl = lane & 63;
                g0 = wl / LIG;
                grp = g0 < 5 ? wv * 5 + g0 : 15;
                lig = g0 < 5 ? wl - g0 * LIG : wv * 4 + (wl - 60);
            }
            uint32_t f1 = 0, f2 = 0;
            for (uint32_t t = 0; t < T; t++) {
                const uint32_t w = p.tbl[off + t * LIG + lig];
                rt[(off / LIG + t) * RL + lane] =
                    make_uint4(addr(w & 127, grp), addr((w >> 14) & 127, grp), addr((w >> 7) & 127, grp) | (addr((w >> 21) & 127, grp) << 16), 0);
                f1 |= ((w >> 30) & 1) << t | ((w >> 31) & 1) << (12 + t);
                f2 |= ((w >> 28) & 1) << t | ((w >> 29) & 1) << (12 + t);
            }
            // z, w: LDS byte address of the step's result slot / companion slot for this lane, -1 where it stores nothing
            // (a padding lane of the step, lanes 60..63 of a wavefront)
            const uint32_t ew = p.tbl[off + T * LIG + lig];
            const bool owner = WGW > 1 || g0 < GROUPS;
            const uint32_t dst = (owner && ((ew >> 7) & 1)) ? addr(ew & 63, grp) : 0xffffffffu;
            const uint32_t sd = (owner && ((ew >> 29) & 1)) ? addr((ew >> 23) & 63, grp) : 0xffffffffu;
            rt[(off / LIG + T) * RL + lane] = make_uint4(f1, f2, dst, sd);
        }
    }
}

hipError_t coop_init(CoopState* st, const hipDeviceProp_t& prop) {
    st->cus = prop.multiProcessorCount;
    CoopDev* d = new CoopDev();
    memset(d, 0, sizeof(*d));
    st->d_prog = d;          // coop_free releases whatever exists if the initialisation stops half way
    hipError_t e;
    for (int i = 0; i < ZKP_PROG_COUNT; i++) {
        const ZkpProgDesc& p = ZKP_PROGS[i];
        if ((e = hipMalloc((void**)&d->progs[i].hdr, p.n_hdr * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&d->progs[i].tbl, (p.n_tbl + 2 * LIG + 4) * 4)) != hipSuccess) return e;
        if ((e = hipMemset(d->progs[i].tbl, 0, (p.n_tbl + 2 * LIG + 4) * 4)) != hipSuccess) return e;   // the kernel reads table words two terms ahead
        if ((e = hipMemcpy(d->progs[i].hdr, p.hdr, p.n_hdr * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(d->progs[i].tbl, p.tbl, p.n_tbl * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        d->progs[i].nslot = p.nslot;
        d->progs[i].nconst = p.nconst;
        d->progs[i].wide = p.wide;
        std::vector<uint4> rt;
        coop_resolve_table(p, rt);
        if ((e = hipMalloc((void**)&d->progs[i].rtbl, rt.size() * sizeof(uint4))) != hipSuccess) return e;
        zkp_dbg_alloc("prog.hdr", d->progs[i].hdr, p.n_hdr * 4);
        zkp_dbg_alloc("prog.tbl", d->progs[i].tbl, (p.n_tbl + 2 * LIG + 4) * 4);
        zkp_dbg_alloc("prog.rtbl", d->progs[i].rtbl, rt.size() * sizeof(uint4));
        if ((e = hipMemcpy(d->progs[i].rtbl, rt.data(), rt.size() * sizeof(uint4), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    if ((e = hipMalloc((void**)&d->consts, sizeof(ZKP_COOP_CONSTS))) != hipSuccess) return e;
    if ((e = hipMemcpy(d->consts, ZKP_COOP_CONSTS, sizeof(ZKP_COOP_CONSTS), hipMemcpyHostToDevice)) != hipSuccess) return e;
    // chunks of a batch are processed by n_pipes independent pipelines on separate HIP streams so that the
    // tail of one kernel (and the latency-bound inversion) overlaps the next chunk's work
    // every knob is read ONCE per context, here, and clamped by plan::clamp (zkp_plan.hpp): measured defaults - 2 pipes x 2^16 checks;
    // phase C in one launch sequence above one chunk (2^17 / 2^18 / 2^19 checks: 37.1 / 72.9 / 142.4 ms against 37.6 / 73.6 / 142.9 per
    // chunk), in two parts on the pipelines from 2^18 checks on; inversion: half a wavefront per SIMD runs its chain fastest
    auto env_l = [](const char* name, long dflt) -> long { const char* v = getenv(name); return v ? atol(v) : dflt; };
    plan::Knobs kn;
    kn.n_pipes = (int)env_l("ZKP_COOP_STREAMS", kn.n_pipes);
    kn.chunk = (size_t)env_l("ZKP_COOP_CHUNK", (long)kn.chunk);
    kn.super = (size_t)env_l("ZKP_COOP_SUPER", (long)kn.super);
    kn.c_single = env_l("ZKP_COOP_C_SINGLE", 1) != 0;
    kn.inv_batch = (uint32_t)env_l("ZKP_COOP_INV_BATCH", (long)kn.inv_batch);
    kn.inv_lanes = (size_t)env_l("ZKP_COOP_INV_LANES", (long)kn.inv_lanes);
    // ZKP_COOP_NO_STREAM=1: groups of eight pairs joined by f12mul, the flow of rounds 1-4 (A/B baseline, cross-check);
    // ZKP_COOP_MAX_STREAM: pairs per streamed group, 9 .. 64 (sweeps)
    const long ms_env = env_l("ZKP_COOP_MAX_STREAM", 0);
    kn.max_stream = env_l("ZKP_COOP_NO_STREAM", 0) != 0 ? plan::MAX_GROUP
                    : (ms_env > (long)plan::MAX_GROUP && ms_env <= (long)plan::MAX_STREAM_LIMIT ? (size_t)ms_env : plan::MAX_STREAM);
    kn.c_split = (int)env_l("ZKP_COOP_C_SPLIT", 0);
    kn.c_split_min = (size_t)env_l("ZKP_COOP_C_SPLIT_MIN", (long)kn.c_split_min);
    kn.split_min = (size_t)env_l("ZKP_COOP_SPLIT_MIN", (long)kn.split_min);
    kn = plan::clamp(kn);
    kn.c_single_min = (size_t)env_l("ZKP_COOP_C_SINGLE_MIN", (long)kn.chunk);
    d->kn = kn;
    d->inv_fermat = env_l("ZKP_COOP_INV_FERMAT", 0) != 0;
    d->cus = prop.multiProcessorCount;
    d->prime = (int)env_l("ZKP_COOP_PRIME", 1);
    d->prime_mask = (int)env_l("ZKP_COOP_PRIME_MASK", PRIME_PREP | PRIME_KSQ | PRIME_INV);
    for (int i = 0; i < MAX_PIPES; i++) d->pipe[i].owner = d;
    for (int i = 0; i < d->kn.n_pipes; i++) {
        if ((e = hipStreamCreateWithFlags(&d->pipe[i].stream, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&d->pipe[i].done, hipEventDisableTiming)) != hipSuccess) return e;
    }
    if ((e = hipEventCreateWithFlags(&d->ready, hipEventDisableTiming)) != hipSuccess) return e;
    st->available = true;
    return hipSuccess;
}

void coop_free(CoopState* st) {
    CoopDev* d = (CoopDev*)st->d_prog;
    if (!d) return;
    for (int i = 0; i < ZKP_PROG_COUNT; i++) {
        if (d->progs[i].hdr) (void)hipFree(d->progs[i].hdr);
        if (d->progs[i].tbl) (void)hipFree(d->progs[i].tbl);
        if (d->progs[i].rtbl) (void)hipFree(d->progs[i].rtbl);
    }
    if (d->consts) (void)hipFree(d->consts);
    if (d->big_state) (void)hipFree(d->big_state);
    if (d->vscratch) (void)hipFree(d->vscratch);
    for (int i = 0; i < MAX_PIPES; i++) {
        if (d->pipe[i].lines) (void)hipFree(d->pipe[i].lines);
        if (d->pipe[i].state) (void)hipFree(d->pipe[i].state);
        if (d->pipe[i].stream) (void)hipStreamDestroy(d->pipe[i].stream);
        if (d->pipe[i].done) (void)hipEventDestroy(d->pipe[i].done);
    }
    if (d->ready) (void)hipEventDestroy(d->ready);
    delete d;
    st->d_prog = nullptr;
    st->available = false;
}

// AUTO (0) and COOP (2) route through this family; THREAD (1) forces the one-pairing-per-lane kernels
bool coop_selected(const CoopState* st, int kind) { return st->available && kind != 1; }

static hipError_t ensure_buf(int4** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return hipSuccess;
    if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) return e; *p = nullptr; *cap = 0; }
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e == hipSuccess) { *cap = bytes; zkp_dbg_alloc("coop.buf", *p, bytes); }
    return e;
}

// Round 6: where the wavefronts of a SMALL grid land depends on what the dispatcher placed last.  On an idle GPU, or behind a kernel of
// another shape, 1,024 one-wavefront workgroups (one per SIMD if spread evenly) come to lie two to a SIMD on half of the SIMDs and take as
// long as 1,536 (k_ksq: 375 us where 768 wavefronts take 220); 2,048 come to lie three to a SIMD (550 us where 1,536 take 385).  Behind a
// grid of the SAME size they spread evenly - even when that grid is an empty kernel (tools/occupancy_probe.py, profiles/r06/placement_probe.txt).
// So a launch whose grid falls into a bad band is preceded by a PRIMER: an empty kernel on the same grid (3-4 us).  The bands, measured per
// kernel class (profiles/r06/knob_sweeps.txt r6o, r6p): k_ksq more than 3 and at most 4, or more than 6 and at most 8 workgroups per compute
// unit (16,384 / 32,768 checks: 6.5 -> 5.8 / 10.4 -> 9.6 ms per pass); k_prep_lines and k_batch_inv more than 2 and at most 3 (20,480 /
// 24,576 pairs: 7.5 -> 6.8 / 8.3 -> 7.5 ms); the interpreter's and the decompression's launches gain nothing and are not primed.  Only
// batches of at most one chunk are primed (a larger one keeps the GPU full).  LDS padding to cap the workgroups per
// compute unit does nothing here (the imbalance is between the SIMDs of a compute unit).  ZKP_COOP_PRIME: 0 off, 1 the bands (default),
// 2 every grid of 1 .. 12 workgroups per compute unit; ZKP_COOP_PRIME_MASK: the kernel classes (A/B).
static hipError_t prime(const CoopDev* d, hipStream_t s, size_t blocks, int cls) {
    if (!d || !d->prime || d->cus <= 0 || !blocks || !(d->prime_mask & cls) || !d->prime_now) return hipSuccess;
    const size_t c = (size_t)d->cus;
    const bool band = cls == PRIME_KSQ ? (blocks > 3 * c && blocks <= 4 * c) || (blocks > 6 * c && blocks <= 8 * c)      // three wavefronts per SIMD
                                       : (blocks > 2 * c && blocks <= 3 * c);
    if (!(band || (d->prime >= 2 && blocks > c && blocks <= 12 * c))) return hipSuccess;
    hipLaunchKernelGGL(k_primer, dim3((unsigned)blocks), dim3(64), 0, s);
    return hipGetLastError();
}

static hipError_t run_prog(CoopDev* d, CoopPipe* pp, int prog, uint32_t n_checks, uint32_t nc, uint32_t k, const uint64_t* wire_in,
                           uint64_t* wire_out, uint8_t* ok, int* all_ok, uint32_t st_off = 0, uint32_t chk_off = 0) {
    hipStream_t s = pp->stream;
    CoopArgs a;
    a.hdr = d->progs[prog].hdr;
    a.tbl = d->progs[prog].tbl;
    a.rtbl = d->progs[prog].rtbl;
    a.consts = d->consts;
    a.lines = pp->lines;
    a.state = pp->state;
    a.wire_in = wire_in;
    a.wire_out = wire_out;
    a.ok = ok;
    a.all_ok = all_ok;
    a.n_checks = n_checks;
    a.nc = nc;
    a.k = k;
    const uint32_t cfg = d->progs[prog].wide;      // LDS configuration: 0 plain, 1 wide, 2 deep (tools/coopgen.py lds_config)
    if (cfg > 2 || d->progs[prog].nslot > (uint32_t)coop_cfg_slots(cfg) || d->progs[prog].nconst > (uint32_t)coop_cfg_consts(cfg))
        return hipErrorInvalidValue;
    a.S = coop_cfg_slots(cfg);
    a.nconst = d->progs[prog].nconst;
    a.st_off = st_off;
    a.chk_off = chk_off;
    a.nd = pp->nd;

    static_assert(12 / WGW * coop_lds_bytes(ZKP_COOP_NSLOT, ZKP_COOP_NCONST) <= 160 * 1024 &&
                      12 / WGW * coop_lds_bytes(ZKP_COOP_WIDE_NSLOT, ZKP_COOP_WIDE_NCONST) <= 160 * 1024 &&
                      12 / WGW * coop_lds_bytes(ZKP_COOP_DEEP_NSLOT, ZKP_COOP_DEEP_NCONST) <= 160 * 1024,
                  "twelve wavefronts (three per SIMD, the register bound) must fit the 160 KB of LDS of a CU");
    size_t lds_bytes = coop_lds_bytes(coop_cfg_slots(cfg), coop_cfg_consts(cfg));
    // occupancy experiments only (the VALUE is cached, not the pointer getenv returned: a later setenv may move the environment)
    static const long lds_pad = getenv("ZKP_COOP_LDS_PAD") ? atol(getenv("ZKP_COOP_LDS_PAD")) : 0;
    if (lds_pad > 0) lds_bytes += (size_t)lds_pad;
    unsigned blocks = (n_checks + GROUPS - 1) / GROUPS;
    { hipError_t ep = prime(d, s, blocks, PRIME_COOP); if (ep != hipSuccess) return ep; }
    ProfScope prof(d, s, cfg == 1 ? PROF_MILLER : cfg == 2 ? PROF_C_DEEP : d->prof_phase_c ? PROF_C_PLAIN : PROF_FEXP_A);
    if (cfg == 2)
        hipLaunchKernelGGL((k_coop<ZKP_COOP_DEEP_NSLOT, ZKP_COOP_DEEP_NCONST>), dim3(blocks), dim3(64 * WGW), lds_bytes, s, a);
    else if (cfg == 1)
        hipLaunchKernelGGL((k_coop<ZKP_COOP_WIDE_NSLOT, ZKP_COOP_WIDE_NCONST>), dim3(blocks), dim3(64 * WGW), lds_bytes, s, a);
    else
        hipLaunchKernelGGL((k_coop<ZKP_COOP_NSLOT, ZKP_COOP_NCONST>), dim3(blocks), dim3(64 * WGW), lds_bytes, s, a);
    return hipGetLastError();
}

static int miller_prog(size_t k, bool wire) {
    switch (k) {
        case 1: return wire ? ZKP_PROG_MILLER1_WIRE : ZKP_PROG_MILLER1_STATE;
        case 2: return wire ? ZKP_PROG_MILLER2_WIRE : ZKP_PROG_MILLER2_STATE;
        case 3: return wire ? ZKP_PROG_MILLER3_WIRE : ZKP_PROG_MILLER3_STATE;
        case 4: return wire ? ZKP_PROG_MILLER4_WIRE : ZKP_PROG_MILLER4_STATE;
        case 5: return wire ? ZKP_PROG_MILLER5_WIRE : ZKP_PROG_MILLER5_STATE;
        case 6: return wire ? ZKP_PROG_MILLER6_WIRE : ZKP_PROG_MILLER6_STATE;
        case 7: return wire ? ZKP_PROG_MILLER7_WIRE : ZKP_PROG_MILLER7_STATE;
        case 8: return wire ? ZKP_PROG_MILLER8_WIRE : ZKP_PROG_MILLER8_STATE;
        default: return -1;
    }
}

constexpr size_t MAX_GROUP = plan::MAX_GROUP;   // pairs per UNROLLED Miller program (miller1..8)
// round 5: more pairs run through the run-time-k program (millern: a pair loop inside every iteration, ONE accumulator, the 63 squarings
// shared by all of them) in groups of at most kn.max_stream pairs; a check with more pairs is joined from its groups' Miller values by
// f12mul as before.  16 is the measured optimum (multi_miller_loop() of 2^20 / 2^18 pairs in checks of k, same box, ms; groups of 8 |
// 16 | 64): k = 9: 89.4 | 85.9 | 84.5, 12: 90.6 | 87.9 | 86.7, 16: 85.6 | 87.8 | 85.9, 32: 93.1 | 86.1 | 93.9, 48: 105.0 | 93.6 | 113.1,
// 64: 107.6 | 102.7 | 116.3, 96: 114.6 | 110.4 | 141.0 (2^18 pairs: 9: 26.5 | 23.0 | 23.3, 16: 27.0 | 25.8 | 25.8, 64: 36.0 | 33.9 | 41.3) -
// longer groups shrink the chunks the line buffer allows (26 KB per pair) until a launch no longer fills the GPU
bool coop_supports_k(size_t k) { return k >= 1 && k <= 0xffffu; }

// line stream of pairs j0 .. j0+g-1 of each of the n checks starting at base_check (k_in pairs per check)
// fused: the Miller value only feeds a final exponentiation inside the same call (homogeneous projective steps, freely scaled lines)
static hipError_t prep(CoopPipe* pp, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t base_check, uint32_t n,
                       uint32_t k_in, uint32_t j0, uint32_t g, bool fused) {
    hipStream_t s = pp->stream;
    size_t p0 = base_check * k_in;
    uint32_t n_pairs = n * g;
    const size_t prep_lds = 4 * 4 * 64 * sizeof(int4);
    { hipError_t ep = prime(pp->owner, s, (2 * (size_t)n_pairs + 63) / 64, PRIME_PREP); if (ep != hipSuccess) return ep; }
    ProfScope prof(pp->owner, s, PROF_PREP);
    static const bool no_cln = getenv("ZKP_PREP_NO_CLN") && atoi(getenv("ZKP_PREP_NO_CLN"));   // A/B knob (value cached, not the pointer)
    if (fused && !no_cln)
        hipLaunchKernelGGL(k_prep_lines<true>, dim3((2 * n_pairs + 63) / 64), dim3(64), prep_lds, s, g1 + 12 * p0, g2 + 24 * p0, i1 ? i1 + p0 : nullptr,
                           i2 ? i2 + p0 : nullptr, n_pairs, g, k_in, j0, n, pp->lines, pp->nd);
    else
        hipLaunchKernelGGL(k_prep_lines<false>, dim3((2 * n_pairs + 63) / 64), dim3(64), prep_lds, s, g1 + 12 * p0, g2 + 24 * p0, i1 ? i1 + p0 : nullptr,
                           i2 ? i2 + p0 : nullptr, n_pairs, g, k_in, j0, n, pp->lines, pp->nd);
    return hipGetLastError();
}

// multi_miller_loop of n checks of k pairs each on one pipeline.  k <= 8: one program.  k > 8: groups of at most
// eight pairs run their own Miller loop (the line buffer is reused), the group values are joined by f12mul
// (prod_i f_i is the multi-Miller value; only the shared squarings are lost).  Result: state ST_F, or the
// canonical wire record in `wire_out` when that is not null.
static hipError_t miller_on_pipe(CoopDev* d, CoopPipe* pp, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2,
                                 size_t base, uint32_t n, uint32_t nc, size_t k, uint64_t* wire_out, bool fused = false) {
    hipError_t e;
    if (k <= MAX_GROUP) {
        if ((e = prep(pp, g1, g2, i1, i2, base, n, (uint32_t)k, 0, (uint32_t)k, fused)) != hipSuccess) return e;
        return run_prog(d, pp, miller_prog(k, wire_out != nullptr), n, nc, (uint32_t)k, nullptr, wire_out, nullptr, nullptr);
    }
    const size_t MS = d->kn.max_stream;
    if (k <= MS) {               // one accumulator for all k pairs
        if ((e = prep(pp, g1, g2, i1, i2, base, n, (uint32_t)k, 0, (uint32_t)k, fused)) != hipSuccess) return e;
        return run_prog(d, pp, wire_out ? ZKP_PROG_MILLERN_WIRE : ZKP_PROG_MILLERN_STATE, n, nc, (uint32_t)k, nullptr, wire_out, nullptr, nullptr);
    }
    for (size_t j0 = 0; j0 < k; j0 += MS) {
        const size_t g = k - j0 < MS ? k - j0 : MS;
        if ((e = prep(pp, g1, g2, i1, i2, base, n, (uint32_t)k, (uint32_t)j0, (uint32_t)g, fused)) != hipSuccess) return e;
        if ((e = run_prog(d, pp, g <= MAX_GROUP ? miller_prog(g, false) : ZKP_PROG_MILLERN_STATE, n, nc, (uint32_t)g, nullptr, nullptr, nullptr, nullptr,
                          j0 ? ZKP_COOP_ST_G : 0)) != hipSuccess)
            return e;
        if (j0) {
            const bool last = j0 + g == k;
            if ((e = run_prog(d, pp, (last && wire_out) ? ZKP_PROG_F12MUL_WIRE : ZKP_PROG_F12MUL_STATE, n, nc, 1, nullptr, last ? wire_out : nullptr,
                              nullptr, nullptr)) != hipSuccess)
                return e;
        }
    }
    return hipSuccess;
}

// Run `body(pipe, base, n)` for every chunk of the batch, round-robin over the pipelines.  The caller's
// stream `s` is forked into the pipeline streams (they wait for everything already queued on `s`) and
// joined again at the end, so from the caller's point of view the call is ordered on `s`.
template <class Body>
static hipError_t for_chunks(CoopDev* d, size_t n_total, size_t k, bool need_lines, bool need_state, hipStream_t s, Body body, size_t chunk_override = 0) {
    if (!n_total) return hipSuccess;
    hipError_t e;
    const plan::Chunks pc = plan::plan_chunks(d->kn, n_total, k, need_lines, chunk_override, d->prof != nullptr);
    const size_t chunk = pc.chunk;
    const int pipes = pc.pipes;
    // workspace first: hipMalloc/hipFree synchronise the device, so never (re)allocate between launches
    for (int i = 0; i < pipes; i++) {
        if (need_lines && (e = ensure_buf(&d->pipe[i].lines, &d->pipe[i].lines_bytes, plan::lines_bytes(plan::group_size(d->kn, k), pc.cmax))) != hipSuccess) return e;
        if (need_state && (e = ensure_buf(&d->pipe[i].state, &d->pipe[i].state_bytes, plan::state_bytes(pc.cmax))) != hipSuccess) return e;
    }
    if ((e = hipEventRecord(d->ready, s)) != hipSuccess) return e;
    for (int i = 0; i < pipes; i++)
        if ((e = hipStreamWaitEvent(d->pipe[i].stream, d->ready, 0)) != hipSuccess) return e;
    int c = 0;
    for (size_t base = 0; base < n_total; base += chunk, c++) {
        uint32_t n = (uint32_t)((n_total - base) < chunk ? (n_total - base) : chunk);
        if ((e = body(&d->pipe[c % pipes], base, n)) != hipSuccess) return e;
    }
    for (int i = 0; i < pipes; i++) {
        if ((e = hipEventRecord(d->pipe[i].done, d->pipe[i].stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(s, d->pipe[i].done, 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t coop_miller(CoopState* st, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks, size_t k,
                       uint64_t* out, hipStream_t s) {
    CoopDev* d = (CoopDev*)st->d_prog;
    if (!coop_supports_k(k)) return hipErrorNotSupported;
    d->prime_now = n_checks <= d->kn.chunk;
    return for_chunks(d, n_checks, k, true, true, s, [&](CoopPipe* pp, size_t base, uint32_t n) -> hipError_t {
        return miller_on_pipe(d, pp, g1, g2, i1, i2, base, n, n, k, out + 72 * base);
    });
}

// nsq compressed squarings of the Fp12 value in state elements [elem_in, elem_in + 12) of every check, snapshots of
// (z2..z5) after the squarings whose bit is set in snap_mask into 12-element areas from elem_snap on
static hipError_t run_ksq(hipStream_t s, int4* state, uint32_t n_checks, uint32_t nc, uint32_t elem_in, uint32_t elem_snap, uint32_t nsq,
                          uint64_t snap_mask, CoopDev* d = nullptr, NDev nd = NDev{nullptr, 0}) {
    if (!n_checks || !nsq) return hipSuccess;
    if (snap_mask && nsq > 64) return hipErrorInvalidValue;   // snapshot bits exist for the first 64 squarings only
    { hipError_t ep = prime(d, s, (n_checks + KS_CHECKS - 1) / KS_CHECKS, PRIME_KSQ); if (ep != hipSuccess) return ep; }
    ProfScope prof(d, s, PROF_KSQ);
    hipLaunchKernelGGL(k_ksq, dim3((n_checks + KS_CHECKS - 1) / KS_CHECKS), dim3(64), 7 * 64 * sizeof(int4), s, state, n_checks, nc, elem_in, elem_snap, nsq, snap_mask, nd);
    return hipGetLastError();
}

// batched inversion of `count` planes of n per-check values (k_batch_inv): few lanes with long batches, because the kernel
// is bound by the latency of one lane's chain (610 + 3 B multiplications)
static hipError_t run_inv(CoopDev* d, hipStream_t s, int4* state, uint32_t n, uint32_t nc, uint32_t elem_n, uint32_t elem_ninv, uint32_t count,
                          NDev nd = NDev{nullptr, 0}) {
    const size_t total = (size_t)n * count;
    if (!total) return hipSuccess;
    const plan::Inv pi = plan::plan_inv(d->kn, n, count);
    const uint32_t B = pi.batch;
    const size_t lanes = pi.lanes;
    { hipError_t ep = prime(d, s, (lanes + 63) / 64, PRIME_INV); if (ep != hipSuccess) return ep; }
    ProfScope prof(d, s, PROF_INV);
    hipLaunchKernelGGL(k_batch_inv, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, state, n, nc, B | (d->inv_fermat ? 0x80000000u : 0u), elem_n,
                       elem_ninv, count, nd);
    return hipGetLastError();
}

// phase C of the final exponentiation: the generated plan (tools/coopgen.py prog_fexp_c) alternates step programs of the
// interpreter with the compressed squaring runs of the five x-power chains and the decompression of their snapshots.
// Everything runs on pp->stream over the n checks whose state starts at pp->state (record stride nc).
static hipError_t run_fexp_c(CoopDev* d, CoopPipe* pp, uint32_t n, uint32_t nc, uint64_t* wire_out, uint8_t* ok, int* all_ok) {
    hipError_t e;
    d->prof_phase_c = true;
    for (int i = 0; i < ZKP_FEXP_C_PLAN_LEN; i++) {
        const ZkpPlanStep& ps = ZKP_FEXP_C_PLAN[i];
        switch (ps.kind) {
            case ZKP_PLAN_PROG:
                e = run_prog(d, pp, (int)ps.a, n, nc, 1, nullptr, wire_out, ok, all_ok);
                break;
            case ZKP_PLAN_KSQ:
                e = run_ksq(pp->stream, pp->state, n, nc, ps.a, ps.b, ps.c, ps.mask, d, pp->nd);
                break;
            case ZKP_PLAN_KDEC_A: {
                { hipError_t ep = prime(d, pp->stream, (2 * (size_t)n * ps.b + 63) / 64, PRIME_KDEC); if (ep != hipSuccess) return ep; }
            ProfScope prof(d, pp->stream, PROF_KDEC_A);
                hipLaunchKernelGGL(k_kdec_a, dim3((unsigned)((2 * (size_t)n * ps.b + 63) / 64)), dim3(64), 0, pp->stream, pp->state, n, nc, ps.a, ps.b, ps.c, pp->nd);
                e = hipGetLastError();
                break;
            }
            case ZKP_PLAN_INV:
                e = run_inv(d, pp->stream, pp->state, n, nc, ps.a, ps.b, ps.c, pp->nd);
                break;
            case ZKP_PLAN_KDEC_B: {
                { hipError_t ep = prime(d, pp->stream, (2 * (size_t)n * ps.b + 63) / 64, PRIME_KDEC); if (ep != hipSuccess) return ep; }
            ProfScope prof(d, pp->stream, PROF_KDEC_B);
                hipLaunchKernelGGL(k_kdec_b, dim3((unsigned)((2 * (size_t)n * ps.b + 63) / 64)), dim3(64), 0, pp->stream, pp->state, n, nc, ps.a, ps.b, ps.c, pp->nd);
                e = hipGetLastError();
                break;
            }
            default:
                e = hipErrorInvalidValue;
        }
        if (e != hipSuccess) break;
    }
    d->prof_phase_c = false;
    return e;
}

// Final exponentiation in two phases over a super-chunk of checks that share ONE state buffer: phase A per chunk on
// the pipelines (whatever produces ST_F, then fexp_a down to the single Fp inversion), ONE batched inversion over all
// checks of the super-chunk (k_batch_inv is latency bound: per 2^16-check chunk it would leave the GPU idle for ~1.5 ms),
// phase C (fexp_c) per chunk.
template <class PhaseA>
static hipError_t two_phase(CoopDev* d, size_t n_total, size_t k, bool need_lines, hipStream_t s, uint64_t* out, uint8_t* ok, int* all_ok,
                            PhaseA phase_a) {
    hipError_t e;
    for (size_t sb = 0; sb < n_total; sb += d->kn.super) {
        const size_t ns = n_total - sb < d->kn.super ? n_total - sb : d->kn.super;
        d->prime_now = ns <= d->kn.chunk && n_total <= d->kn.chunk;
        if ((e = ensure_buf(&d->big_state, &d->big_state_bytes, plan::state_bytes(ns))) != hipSuccess) return e;
        e = for_chunks(d, ns, k, need_lines, false, s, [&](CoopPipe* pp, size_t base, uint32_t n) -> hipError_t {
            CoopPipe v = *pp;
            v.state = d->big_state + 4 * base;
            v.nd = NDev{d->n_dev, (uint32_t)(sb + base)};
            return phase_a(&v, sb + base, n, (uint32_t)ns);
        });
        if (e != hipSuccess) return e;
        if ((e = run_inv(d, s, d->big_state, (uint32_t)ns, (uint32_t)ns, ZKP_COOP_ST_N, ZKP_COOP_ST_NINV, 1, NDev{d->n_dev, (uint32_t)sb})) != hipSuccess) return e;
        // phase C needs no line buffer.  The plan (zkp_plan.hpp plan_phase_c): two parts on the two pipelines from 2^18 checks on (round 5,
        // same-box A/B one sequence -> two parts: 2^20 234.1 / 236.0 -> 233.4 / 233.4 ms; 2^19 117.7 / 117.5 -> 115.1 / 114.8; 2^18 60.2 /
        // 60.1 -> 59.1 / 58.6; 2^17 a wash), ONE launch sequence on the caller's stream below.  Round 6 measured the small and medium
        // batches (profiles/r06/knob_sweeps.txt, the v60 traces beside it): what a 2^16 / 2^17-check batch loses against the 2^20 rate is
        // the six k_batch_inv launches - 0.11-0.18 ms each whatever the batch, one lane's dependent chain - and the round quantisation of
        // k_ksq (4,096 / 8,192 wavefronts on 3,072 slots).  Two parts at these sizes lose (2^16: 16.4 -> 17.1 ms) - each half-size launch
        // has its own partial round - and two parts kept OUT of phase by events (the follower's squaring run c starts when the leader's
        // has ended, so that one part's inversion runs beside the other's squaring run) hide the inversion but not for free: beside a
        // squaring run it takes 0.44 instead of 0.13 ms and the run 7 % longer; 16.6 / 31.5 ms against 16.4 / 31.5.  More inversion
        // lanes, an occupancy cap and chunks of 2^13 .. 2^15 all lose as well.  ZKP_COOP_C_SPLIT=n forces n parts (A/B, tests).
        const plan::PhaseC pcc = plan::plan_phase_c(d->kn, ns, d->prof != nullptr);
        if (pcc.mode == plan::C_PARTS) {
            const size_t part = pcc.part;
            e = for_chunks(d, ns, 1, false, false, s, [&](CoopPipe* pp, size_t base, uint32_t n) -> hipError_t {
                CoopPipe v = *pp;
                v.state = d->big_state + 4 * base;
                v.nd = NDev{d->n_dev, (uint32_t)(sb + base)};
                return run_fexp_c(d, &v, n, (uint32_t)ns, out ? out + 72 * (sb + base) : nullptr, ok ? ok + sb + base : nullptr, all_ok);
            }, part);
        } else if (pcc.mode == plan::C_SINGLE) {
            CoopPipe v = d->pipe[0];
            v.state = d->big_state;
            v.stream = s;
            v.nd = NDev{d->n_dev, (uint32_t)sb};
            e = run_fexp_c(d, &v, (uint32_t)ns, (uint32_t)ns, out ? out + 72 * sb : nullptr, ok ? ok + sb : nullptr, all_ok);
        } else {
            e = for_chunks(d, ns, 1, false, false, s, [&](CoopPipe* pp, size_t base, uint32_t n) -> hipError_t {
                CoopPipe v = *pp;
                v.state = d->big_state + 4 * base;
                v.nd = NDev{d->n_dev, (uint32_t)(sb + base)};
                return run_fexp_c(d, &v, n, (uint32_t)ns, out ? out + 72 * (sb + base) : nullptr, ok ? ok + sb + base : nullptr, all_ok);
            });
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t coop_final_exp(CoopState* st, const uint64_t* f, size_t n_total, uint64_t* out, uint8_t* ok, int* all_ok, hipStream_t s) {
    CoopDev* d = (CoopDev*)st->d_prog;
    return two_phase(d, n_total, 1, false, s, out, ok, all_ok, [&](CoopPipe* pp, size_t base, uint32_t n, uint32_t nc) -> hipError_t {
        return run_prog(d, pp, ZKP_PROG_FEXP_A_WIRE, n, nc, 1, f + 72 * base, nullptr, nullptr, nullptr);
    });
}

// one level of the Fp12 product tree, in place: buf[c] <- buf[c] * buf[c + h] for c < m  (m <= h)
hipError_t coop_fp12_mul_pairs(CoopState* st, uint64_t* buf, size_t m, size_t h, hipStream_t s) {
    CoopDev* d = (CoopDev*)st->d_prog;
    if (m > 0xffffffffu || h > 0xffffffffu) return hipErrorInvalidValue;
    CoopPipe on_s = d->pipe[0];   // no workspace is touched: only the stream matters
    on_s.stream = s;
    return run_prog(d, &on_s, ZKP_PROG_F12MUL_PAIRS, (uint32_t)m, (uint32_t)m, 1, buf, buf, nullptr, nullptr, 0, (uint32_t)h);
}

hipError_t coop_pairing(CoopState* st, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks, size_t k,
                        uint64_t* out_gt, uint8_t* ok, int* all_ok, hipStream_t s, const uint32_t* n_dev) {
    CoopDev* d = (CoopDev*)st->d_prog;
    if (!coop_supports_k(k)) return hipErrorNotSupported;
    // n_dev (device memory, may be null): only the first min(n_checks, *n_dev) checks exist - the launches are planned for n_checks and
    // size themselves on the device (NDev); nothing is read back
    d->n_dev = n_dev;
    const hipError_t e2 = two_phase(d, n_checks, k, true, s, out_gt, ok, all_ok, [&](CoopPipe* pp, size_t base, uint32_t n, uint32_t nc) -> hipError_t {
        hipError_t e = miller_on_pipe(d, pp, g1, g2, i1, i2, base, n, nc, k, nullptr, true);
        if (e != hipSuccess) return e;
        return run_prog(d, pp, ZKP_PROG_FEXP_A_STATE, n, nc, 1, nullptr, nullptr, nullptr, nullptr);
    });
    d->n_dev = nullptr;
    return e2;
}

// one pass of the fused pairing with every launch bracketed by events (one pipeline: no two kernels overlap); ms[c] = the
// summed duration of the launches of class c, launches[c] their number.  Synchronises `s`.
hipError_t coop_profile_pairing(CoopState* st, const uint64_t* g1, const uint64_t* g2, size_t n, uint64_t* out_gt, float* ms, int* launches, hipStream_t s) {
    CoopDev* d = (CoopDev*)st->d_prog;
    std::vector<CoopDev::ProfEv> evs;
    struct ProfGuard {       // d->prof points at this frame's vector: it is taken back on EVERY way out (push_back may throw)
        CoopDev* d;
        ~ProfGuard() { d->prof = nullptr; d->prof_phase_c = false; }
    } guard{d};
    d->prof = &evs;
    hipError_t e;
    try {
        e = coop_pairing(st, g1, g2, nullptr, nullptr, n, 1, out_gt, nullptr, nullptr, s, nullptr);
    } catch (...) {
        e = hipErrorOutOfMemory;
    }
    d->prof = nullptr;
    d->prof_phase_c = false;
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
    for (int c = 0; c < PROF_CLASSES; c++) { ms[c] = 0.f; launches[c] = 0; }
    for (auto& ev : evs) {
        float t = 0.f;
        if (e == hipSuccess && (e = hipEventElapsedTime(&t, ev.a, ev.b)) == hipSuccess) { ms[ev.cls] += t; launches[ev.cls]++; }
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    return e;
}

// zkp_tower_op_batch on the cooperative family: `ab` holds the n a-records followed by the n b-records (wire format);
// op as in zkp_tower_op (include/zkp_pairings.h).  The cyclotomic power g^(2^repeat) (and the decompression alone) takes the route of the final
// exponentiation's x-power chains: wire -> state, one compressed squaring run with a single snapshot, decompression, -> wire.
hipError_t coop_tower_op(CoopState* st, int op, const uint64_t* ab, size_t n, uint32_t repeat, uint64_t* out, hipStream_t s) {
    CoopDev* d = (CoopDev*)st->d_prog;
    static const int progs[11] = {ZKP_PROG_TW_FP2_MUL, ZKP_PROG_TW_FP2_SQR, ZKP_PROG_TW_FP6_MUL, ZKP_PROG_TW_FP6_SQR, ZKP_PROG_TW_FP12_FROB,
                                  ZKP_PROG_TW_FP12_MUL, ZKP_PROG_TW_FP12_SQR, ZKP_PROG_TW_FP12_014, ZKP_PROG_TW_FP12_FROB, ZKP_PROG_TW_FP12_CONJ,
                                  ZKP_PROG_TW_CYC_SQR};
    if (op < 0 || op > 20 || n > 0x3fffffffu) return hipErrorInvalidValue;
    d->prime_now = n <= d->kn.chunk;
    hipError_t e;
    for (size_t base = 0; base < n; base += d->kn.chunk) {
        const uint32_t m = (uint32_t)(n - base < d->kn.chunk ? n - base : d->kn.chunk);
        CoopPipe v = d->pipe[0];
        v.stream = s;
        if (op >= 13) {
            // round 4: the remaining tower functions of SURVEY 8(a).  One program for the sparse products and the nonresidue maps;
            // an inversion is program A (wire -> state, the norm chain down to ONE Fp value), the batched inversion kernel (0 gives
            // 0: a non-invertible input ends as the zero record), program B (state -> wire) - for Fp12 exactly the route
            // final_exponentiation() takes (fexp_a, k_batch_inv, the opening of fexp_c)
            int one = -1, pa = -1, pb = -1;
            switch (op) {
                case 13: pa = ZKP_PROG_TW_FP2_INV_A; pb = ZKP_PROG_TW_FP2_INV_B; break;
                case 14: one = ZKP_PROG_TW_FP2_NR; break;
                case 15: one = ZKP_PROG_TW_FP2_MULFP; break;
                case 16: one = ZKP_PROG_TW_FP6_BY1; break;
                case 17: one = ZKP_PROG_TW_FP6_BY01; break;
                case 18: one = ZKP_PROG_TW_FP6_NR; break;
                case 19: pa = ZKP_PROG_TW_FP6_INV_A; pb = ZKP_PROG_TW_FP6_INV_B; break;
                default: pa = ZKP_PROG_FEXP_A_WIRE; pb = ZKP_PROG_TW_FP12_INV_B; break;
            }
            if (one >= 0) {
                if ((e = run_prog(d, &v, one, m, m, 1, ab + 72 * base, out + 72 * base, nullptr, nullptr, 0, (uint32_t)n)) != hipSuccess) return e;
                continue;
            }
            if ((e = ensure_buf(&d->pipe[0].state, &d->pipe[0].state_bytes, plan::state_bytes(d->kn.chunk))) != hipSuccess) return e;
            v.state = d->pipe[0].state;
            if ((e = run_prog(d, &v, pa, m, m, 1, ab + 72 * base, nullptr, nullptr, nullptr)) != hipSuccess) return e;
            if ((e = run_inv(d, s, v.state, m, m, ZKP_COOP_ST_N, ZKP_COOP_ST_NINV, 1)) != hipSuccess) return e;
            if ((e = run_prog(d, &v, pb, m, m, 1, nullptr, out + 72 * base, nullptr, nullptr)) != hipSuccess) return e;
            continue;
        }
        if (op <= 10) {
            if ((e = run_prog(d, &v, progs[op], m, m, 1, ab + 72 * base, out + 72 * base, nullptr, nullptr, 0, (uint32_t)n)) != hipSuccess) return e;
            // ZKP_TOWER_FP6_FROBENIUS runs the Fp12 program; the Fp6 result is its c0 half, the rest of the record is zero whatever
            // the input's tail held (as the thread family's k_tower_op writes it)
            if (op == 4 && (e = hipMemset2DAsync(out + 72 * base + 36, 576, 0, 288, m, s)) != hipSuccess) return e;
            continue;
        }
        if (op == 11 && (repeat < 1 || repeat > 64)) return hipErrorInvalidValue;
        if ((e = ensure_buf(&d->pipe[0].state, &d->pipe[0].state_bytes, plan::state_bytes(d->kn.chunk))) != hipSuccess) return e;
        v.state = d->pipe[0].state;
        if (op == 12) {      // decompression alone: the record's z2..z5 are the snapshot
            if ((e = run_prog(d, &v, ZKP_PROG_TW_TO_SNAP, m, m, 1, ab + 72 * base, nullptr, nullptr, nullptr)) != hipSuccess) return e;
        } else {
            if ((e = run_prog(d, &v, ZKP_PROG_TW_TO_STATE, m, m, 1, ab + 72 * base, nullptr, nullptr, nullptr)) != hipSuccess) return e;
            if ((e = run_ksq(s, v.state, m, m, 0, ZKP_COOP_ST_SNAP, repeat, 1ull << (repeat - 1), d)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_kdec_a, dim3((2 * m + 63) / 64), dim3(64), 0, s, v.state, m, m, (uint32_t)ZKP_COOP_ST_SNAP, 1u, (uint32_t)ZKP_COOP_ST_KN, NDev{nullptr, 0});
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = run_inv(d, s, v.state, m, m, ZKP_COOP_ST_KN, ZKP_COOP_ST_KNINV, 1)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_kdec_b, dim3((2 * m + 63) / 64), dim3(64), 0, s, v.state, m, m, (uint32_t)ZKP_COOP_ST_SNAP, 1u, (uint32_t)ZKP_COOP_ST_KNINV, NDev{nullptr, 0});
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = run_prog(d, &v, ZKP_PROG_TW_FROM_SNAP, m, m, 1, nullptr, out + 72 * base, nullptr, nullptr)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// timing hook (diagnostic): one of the synthetic programs (tools/coopgen.py prog_timing; which 0..8), 400 squarings of the
// compressed squaring-run kernel (which 9), the line precomputation of n pairs (which 10) or the one-pair Miller program
// over the line stream it left behind (which 11; call 10 first) - on the caller's stream, timed with the events handed in.
// The kernels' running time does not depend on the data (zeroed buffers serve as inputs).
hipError_t coop_time_prog(CoopState* st, int which, size_t n, hipStream_t s, hipEvent_t e0, hipEvent_t e1, float* ms) {
    CoopDev* d = (CoopDev*)st->d_prog;
    static const int ids[9] = {ZKP_PROG_TIME_T1, ZKP_PROG_TIME_T3, ZKP_PROG_TIME_T3E, ZKP_PROG_TIME_T6, ZKP_PROG_TIME_T12, ZKP_PROG_TIME_LIN,
                               ZKP_PROG_TIME_CYC, ZKP_PROG_TIME_CYCSD, ZKP_PROG_TIME_FILL};
    static const int ids2[3] = {ZKP_PROG_TIME_T6S, ZKP_PROG_TIME_T12S, ZKP_PROG_TIME_T12B};   // which 12..14: one-slot / B-two-slot operand forms
    if (which < 0 || which > 19 || !n || n > 0x7fffffffu || ((which == 10 || which == 11 || which == 15) && n > d->kn.chunk)) return hipErrorInvalidValue;
    hipError_t e = ensure_buf(&d->pipe[0].state, &d->pipe[0].state_bytes, (size_t)ST_SIZE * n * 64);
    if (e != hipSuccess) return e;
    if ((which == 10 || which == 11 || which == 15) && (e = ensure_buf(&d->pipe[0].lines, &d->pipe[0].lines_bytes, (size_t)NLINES * 6 * n * 64)) != hipSuccess) return e;
    CoopPipe v = d->pipe[0];
    v.stream = s;
    bool timed = false;      // which 16 / 17: 57 squarings (one run of the pass) behind ANOTHER kernel (a step program) / behind an idle GPU
    auto once = [&]() -> hipError_t {
        if (which == 9) return run_ksq(s, v.state, (uint32_t)n, (uint32_t)n, 0, 12, 400, 0);   // timing run: no snapshots (a mask needs nsq <= 64)
        if (which >= 16 && which <= 19) {      // 18: behind a ONE-squaring launch of the same grid; 19: behind a step program and that
            if (!timed) {
                timed = true;
                if (which == 17) return hipStreamSynchronize(s);
                static const bool empty_primer = getenv("ZKP_PRIMER_EMPTY") && atoi(getenv("ZKP_PRIMER_EMPTY")) != 0;
                if (which == 18 && empty_primer) {
                    hipLaunchKernelGGL(k_primer, dim3((unsigned)((n + KS_CHECKS - 1) / KS_CHECKS)), dim3(64), 0, s);
                    return hipGetLastError();
                }
                if (which == 18) return run_ksq(s, v.state, (uint32_t)n, (uint32_t)n, 0, 12, 1, 0);
                hipError_t e1 = run_prog(d, &v, ZKP_PROG_TIME_T1, (uint32_t)n, (uint32_t)n, 1, nullptr, nullptr, nullptr, nullptr);
                if (e1 != hipSuccess || which == 16) return e1;
                return run_ksq(s, v.state, (uint32_t)n, (uint32_t)n, 0, 12, 1, 0);
            }
            return run_ksq(s, v.state, (uint32_t)n, (uint32_t)n, 0, 12, 57, 0);
        }
        if (which >= 12 && which <= 14) return run_prog(d, &v, ids2[which - 12], (uint32_t)n, (uint32_t)n, 1, nullptr, nullptr, nullptr, nullptr);
        if (which == 10 || which == 15) {                          // 15: the upstream-shaped lines (k_prep_lines<false>)
            const uint64_t* zero = (const uint64_t*)v.state;      // 36 u64 of zeros per pair: the state buffer is far larger
            return prep(&v, zero, zero + 12 * n, nullptr, nullptr, 0, (uint32_t)n, 1, 0, 1, which == 10);
        }
        if (which == 11) return run_prog(d, &v, ZKP_PROG_MILLER1_STATE, (uint32_t)n, (uint32_t)n, 1, nullptr, nullptr, nullptr, nullptr);
        return run_prog(d, &v, ids[which], (uint32_t)n, (uint32_t)n, 1, nullptr, nullptr, nullptr, nullptr);
    };
    // ZKP_TIME_FILL=<byte> (measurement only): the synthetic inputs are this byte repeated instead of zeros - the clock a kernel gets depends
    // on the operand data (DESIGN_HISTORY section 4, round 5), so zeros flatter a multiply-add-dense kernel
    static const int fill = getenv("ZKP_TIME_FILL") ? atoi(getenv("ZKP_TIME_FILL")) : 0;
    if (which != 11 && fill < 256 && (e = hipMemsetAsync(v.state, fill & 0xff, (size_t)ST_SIZE * n * 64, s)) != hipSuccess) return e;
    if (which != 11 && fill >= 256) {
        const size_t words = (size_t)ST_SIZE * n * 16;
        hipLaunchKernelGGL(k_fill_rand, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (int32_t*)v.state, words);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if ((e = once()) != hipSuccess) return e;
    if ((e = hipEventRecord(e0, s)) != hipSuccess) return e;
    if ((e = once()) != hipSuccess) return e;
    if ((e = hipEventRecord(e1, s)) != hipSuccess) return e;
    if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
    return hipEventElapsedTime(ms, e0, e1);
}

// ZKP_VALID_GENERIC=1 (environment, read once): the compiled kernels alone, the round-3 path - the A/B baseline and the cross-check
[[maybe_unused]] static bool valid_generic_only() {
    static const bool v = getenv("ZKP_VALID_GENERIC") && atoi(getenv("ZKP_VALID_GENERIC")) != 0;
    return v;
}
hipError_t coop_g1_valid(const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* status, hipStream_t s) {
    if (!n) return hipSuccess;
#if ZKP_VALID_ASM
    if (!valid_generic_only()) {
        hipLaunchKernelGGL(k_g1_valid_fast, dim3((unsigned)((n + 63) / 64)), dim3(64), 3 * 4 * 64 * sizeof(int4), s, g1, inf, (uint32_t)n, status);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_g1_valid28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, inf, (uint32_t)n, status, 1);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(k_g1_valid28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, inf, (uint32_t)n, status, 0);
    return hipGetLastError();
}
hipError_t coop_g2_valid(CoopState* st, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* status, hipStream_t s) {
    if (!n) return hipSuccess;
#if ZKP_VALID_ASM
    if (!valid_generic_only()) {
        // ZKP_G2_VALID_WAVES=2 (environment, read once): the two-wave kernel (no scratch buffer), the A/B baseline of the three-wave one
        static const bool two_waves = getenv("ZKP_G2_VALID_WAVES") && atoi(getenv("ZKP_G2_VALID_WAVES")) == 2;
        const unsigned blocks = (unsigned)((2 * n + 63) / 64);
        hipError_t e;
        if (two_waves) {
            hipLaunchKernelGGL(k_g2_valid_fast, dim3(blocks), dim3(64), 2 * 4 * 64 * sizeof(int4), s, g2, inf, (uint32_t)n, status);
        } else {
            // at most 2^22 points per launch: the kernel addresses its scratch with 32-bit lane offsets and plane strides (2 lanes x 16 B
            // x 8 planes per point wrap at 2^27 points), and the scratch stays at 1 GiB whatever the batch (launches on one stream
            // run one after the other, so they may share it)
            const size_t CH = plan::VALID_CHUNK;
            CoopDev* d = (CoopDev*)st->d_prog;
            const size_t big = n < CH ? n : CH;
            if ((e = ensure_buf(&d->vscratch, &d->vscratch_bytes, plan::vscratch_bytes(big))) != hipSuccess) return e;
            for (size_t lo = 0; lo < n; lo += CH) {
                const size_t m = n - lo < CH ? n - lo : CH;
                hipLaunchKernelGGL(k_g2_valid_fast3, dim3((unsigned)((2 * m + 63) / 64)), dim3(64), 3 * 4 * 64 * sizeof(int4), s, g2 + 24 * lo,
                                   inf ? inf + lo : nullptr, (uint32_t)m, status + lo, d->vscratch);
            }
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_g2_valid28, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, g2, inf, (uint32_t)n, status, 1);
        return hipGetLastError();
    }
#endif
    (void)st;
    hipLaunchKernelGGL(k_g2_valid28, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, g2, inf, (uint32_t)n, status, 0);
    return hipGetLastError();
}

hipError_t coop_g1_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_g1_mul28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, base, stride, sc, (uint32_t)n, out, out_inf);
    return hipGetLastError();
}
hipError_t coop_g2_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_g2_mul28, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, base, stride, sc, (uint32_t)n, out, out_inf);
    return hipGetLastError();
}

hipError_t coop_fp28_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, hipStream_t s) {
    if (!n) return hipSuccess;
    if (op < 0 || op > 5) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fp28_op, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, op, a, b, n, out);
    return hipGetLastError();
}

}  // namespace zkp
