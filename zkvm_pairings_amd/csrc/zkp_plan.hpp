// zkp_plan.hpp -- the host side's PURE arithmetic: how a batch is cut into super-chunks, chunks, phase-C parts, inversion batches and
// host slices, what every workspace holds and what every launch is given.  No HIP type, no allocation, no I/O: the file is included by
// zkp_coop.hip / zkp_pairings.hip (the product) AND compiled with gcc -fsanitize=address,undefined into tests/plan_check.cpp, which
// walks it over the sizes the C ABI admits (too_many) and asserts that every product fits the type the kernel receives it in
// (the host side under sanitizers).  Reference anchor: none - the reference has no batching
// (src/lib.rs:1-14 is its whole surface); this is the drop-in's own plumbing.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zkp {
namespace plan {

// generated constants the plans depend on (zkp_coop.hip static_asserts them against zkp_coop_prog.inc)
constexpr uint32_t NLINES = 68;        // line records per pair: 63 doubling + 5 addition steps of the Miller loop over |x|
constexpr uint32_t ST_SIZE = 208;      // 64-byte state records per check
constexpr uint32_t REC_BYTES = 64;     // one record: 14 limbs of 28 bits in four int4
constexpr uint32_t GROUPS = 5;         // checks per k_coop wavefront (twelve lanes each)
constexpr uint32_t KS_CHECKS = 16;     // checks per k_ksq wavefront (four lanes each)
constexpr size_t MAX_GROUP = 8;        // pairs per UNROLLED Miller program (miller1..8)
constexpr size_t MAX_STREAM = 16;      // pairs per group of the run-time-k Miller program (measured optimum, zkp_coop.hip)
constexpr size_t MAX_STREAM_LIMIT = 64;
constexpr size_t MIN_CHUNK = 320, MAX_CHUNK = (size_t)1 << 20, MAX_SUPER = (size_t)1 << 22;
constexpr int MAX_PIPES = 4;
constexpr size_t VALID_CHUNK = (size_t)1 << 22;   // points per k_g2_valid_fast3 launch (32-bit scratch offsets)

// range limits shared by the C ABI's entry points: every per-launch count stays in 32 bits
constexpr bool too_many(size_t n, size_t k = 1) { return n > 0x7fffffffu || k > 0xffffu || (k && n > 0x7fffffffu / k); }

struct Knobs {
    int n_pipes = 2;                          // ZKP_COOP_STREAMS
    size_t chunk = (size_t)1 << 16;           // ZKP_COOP_CHUNK: checks per pipeline pass (bounds the line buffer: 26 KB per pair)
    size_t super = (size_t)1 << 20;           // ZKP_COOP_SUPER: checks per two-phase final exponentiation
    bool c_single = true;                     // ZKP_COOP_C_SINGLE
    size_t c_single_min = (size_t)1 << 16;    // ZKP_COOP_C_SINGLE_MIN (default: chunk)
    uint32_t inv_batch = 32;                  // ZKP_COOP_INV_BATCH
    size_t inv_lanes = (size_t)1 << 15;       // ZKP_COOP_INV_LANES
    size_t max_stream = MAX_STREAM;           // ZKP_COOP_MAX_STREAM / ZKP_COOP_NO_STREAM
    int c_split = 0;                          // ZKP_COOP_C_SPLIT: phase C in this many parts (0: the default rule)
    size_t c_split_min = (size_t)1 << 18;     // ZKP_COOP_C_SPLIT_MIN: ... the default rule: two parts from this many checks on
    size_t split_min = (size_t)1 << 14;       // ZKP_COOP_SPLIT_MIN: a batch of at most one chunk still goes over all pipelines when every part
                                              // keeps at least this many checks (round 6)
};

// the clamps coop_init applies to what the environment asked for
inline Knobs clamp(Knobs k) {
    if (k.n_pipes < 1) k.n_pipes = 1;
    if (k.n_pipes > MAX_PIPES) k.n_pipes = MAX_PIPES;
    if (k.chunk < MIN_CHUNK) k.chunk = MIN_CHUNK;
    if (k.chunk > MAX_CHUNK) k.chunk = MAX_CHUNK;      // 8 pairs x 2^20 checks x 2 lanes: every per-launch count stays in 32 bits
    if (k.super < k.chunk) k.super = k.chunk;
    if (k.super > MAX_SUPER) k.super = MAX_SUPER;      // 56 GB of state; keeps every per-launch count in 32 bits
    if (k.inv_batch < 1) k.inv_batch = 1;
    if (k.inv_batch > 4096) k.inv_batch = 4096;
    if (k.inv_lanes < 1) k.inv_lanes = 1;
    if (k.max_stream < MAX_GROUP) k.max_stream = MAX_GROUP;
    if (k.max_stream > MAX_STREAM_LIMIT) k.max_stream = MAX_STREAM_LIMIT;
    if (k.c_split < 0) k.c_split = 0;
    if (k.c_split > 64) k.c_split = 64;
    if (k.split_min < 1) k.split_min = 1;
    return k;
}

// pairs per Miller launch of a check with k pairs: the whole check up to max_stream pairs, groups of max_stream beyond
inline size_t group_size(const Knobs& kn, size_t k) { return k <= MAX_GROUP ? k : (k < kn.max_stream ? k : kn.max_stream); }

// ---- workspace sizes (bytes)
inline size_t lines_bytes(size_t pairs_per_group, size_t checks) { return (size_t)NLINES * pairs_per_group * 6 * checks * REC_BYTES; }
inline size_t state_bytes(size_t checks) { return (size_t)ST_SIZE * checks * REC_BYTES; }
inline size_t vscratch_bytes(size_t points) { return ((2 * points + 63) / 64) * 64 * 8 * 16; }

// ---- grids (one-wavefront workgroups)
inline size_t blocks_coop(size_t checks) { return (checks + GROUPS - 1) / GROUPS; }
inline size_t blocks_ksq(size_t checks) { return (checks + KS_CHECKS - 1) / KS_CHECKS; }
inline size_t blocks_two_lanes(size_t units) { return (2 * units + 63) / 64; }      // k_prep_lines (pairs), k_kdec_a / _b (snapshots), G2 points
inline size_t blocks_one_lane(size_t units) { return (units + 63) / 64; }

// ---- the chunks of one batch on the pipelines (for_chunks)
struct Chunks {
    size_t chunk;        // checks per chunk (the last one may be shorter)
    size_t n_chunks;
    int pipes;           // pipelines in use: chunk c runs on pipeline c % pipes
    size_t cmax;         // checks the largest chunk holds: what the pipelines' workspaces are sized for
};
inline Chunks plan_chunks(const Knobs& kn, size_t n_total, size_t k, bool need_lines, size_t chunk_override, bool profiling) {
    Chunks c{};
    size_t chunk = chunk_override ? chunk_override : kn.chunk;
    if (need_lines && k > 4) {   // keep the line buffer at the size four pairs per check need
        chunk = chunk * 4 / group_size(kn, k);
        if (chunk < MIN_CHUNK) chunk = MIN_CHUNK;
    }
    int pipes = profiling ? 1 : kn.n_pipes;
    size_t n_chunks = n_total ? (n_total + chunk - 1) / chunk : 0;
    if (pipes > 1 && n_chunks && !chunk_override) {
        // round 6: the pipelines end together - the number of chunks becomes a multiple of the pipelines and the chunks equal, when the
        // (shorter) chunks still hold split_min checks each; in particular a batch of at most one chunk is cut in `pipes` parts (it
        // used to run on ONE pipeline: no kernel's tail overlapped anything)
        const size_t up = (n_chunks + pipes - 1) / pipes * pipes;
        size_t even = (n_total + up - 1) / up;
        even = (even + 15) / 16 * 16;
        if (even >= kn.split_min && even <= chunk) {
            chunk = even;
            n_chunks = (n_total + chunk - 1) / chunk;
        }
    }
    if (n_chunks <= 1) pipes = 1;
    c.chunk = chunk;
    c.n_chunks = n_chunks;
    c.pipes = pipes;
    c.cmax = n_total < chunk ? n_total : chunk;
    return c;
}

// ---- phase C of a super-chunk of ns checks (two_phase)
enum PhaseCMode { C_PARTS = 0, C_SINGLE = 1, C_CHUNKS = 2 };
struct PhaseC { PhaseCMode mode; size_t part; };     // C_PARTS: parts of `part` checks alternate over the pipelines
inline PhaseC plan_phase_c(const Knobs& kn, size_t ns, bool profiling) {
    const int parts = profiling ? 0 : (kn.c_split > 0 ? kn.c_split : (ns >= kn.c_split_min && kn.n_pipes >= 2 ? 2 : 0));
    if (parts > 1 && ns >= (size_t)16 * parts) return {C_PARTS, ((ns + parts - 1) / parts + 15) / 16 * 16};
    if (kn.c_single || ns <= kn.c_single_min) return {C_SINGLE, ns};     // one launch sequence on the caller's stream
    return {C_CHUNKS, 0};                                                // ZKP_COOP_C_SINGLE=0: per chunk on the pipelines (rounds 1-2)
}

// ---- batched inversion of `count` planes of n values: few lanes with long batches (the kernel is bound by one lane's chain)
struct Inv { uint32_t batch; size_t lanes; };
inline Inv plan_inv(const Knobs& kn, size_t n, size_t count) {
    const size_t total = n * count;
    size_t b = total / kn.inv_lanes;
    b = b < 1 ? 1 : (b > kn.inv_batch ? kn.inv_batch : b);
    return {(uint32_t)b, total ? (total + b - 1) / b : 0};
}

// ---- 32-bit quantities inside the kernels (what tests/plan_check.cpp asserts for every launch the plans above produce)
// k_prep_lines: lane number 2 * pairs, the 32-bit lane offset of a line record and the step stride it adds as a 32-bit scalar
inline bool prep_fits(size_t n, size_t g) {
    const uint64_t pairs = (uint64_t)n * g;
    const uint64_t voff_max = ((uint64_t)(g - 1) * 6 + 5) * n + (n - 1);                // record index of the last coefficient of the last pair
    return n && g && 2 * pairs + 63 <= 0xffffffffull && voff_max * REC_BYTES + REC_BYTES <= 0xffffffffull &&
           2ull * n * REC_BYTES <= 0xffffffffull;
}
// every kernel takes its check count, its state stride and (k_coop) the pairs per check as uint32_t
inline bool u32(size_t v) { return v <= 0xffffffffull; }

// ---- host-pointer entry points: slices of host_slice pairs through two workspace slots (host_sliced_impl)
struct Slices { size_t checks_per_slice, n_slices; };
inline Slices plan_slices(size_t host_slice, size_t n_checks, size_t k) {
    const size_t sc = (k && host_slice / k) ? host_slice / k : 1;
    return {sc, (n_checks + sc - 1) / sc};
}

}  // namespace plan
}  // namespace zkp
