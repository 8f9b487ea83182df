// plan_check.cpp -- the host planners of the product (zkvm_pairings_amd/csrc/zkp_plan.hpp: chunk / group / slice plans, workspace sizes,
// launch counts) walked over the sizes the C ABI admits, built with gcc -fsanitize=address,undefined by tests/test_sanitizers.py.
// Every product is formed in checked 64-bit arithmetic and compared with the type the kernel receives it in; the loops are the ones
// zkp_coop.hip runs (for_chunks, two_phase, miller_on_pipe, run_fexp_c), driven by the same plan functions.  CPU only: nothing here
// touches HIP.  Exit code 0 and "plan_check ok" on success, the first violated bound otherwise.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../zkvm_pairings_amd/csrc/zkp_plan.hpp"

using namespace zkp::plan;

static unsigned long long g_launches = 0, g_cases = 0;

[[noreturn]] static void fail(const char* what, unsigned long long a, unsigned long long b, unsigned long long c) {
    std::fprintf(stderr, "plan_check: %s (%llu, %llu, %llu)\n", what, a, b, c);
    std::exit(1);
}
#define REQUIRE(cond, a, b, c) do { if (!(cond)) fail(#cond, (unsigned long long)(a), (unsigned long long)(b), (unsigned long long)(c)); } while (0)

static uint64_t mul(uint64_t a, uint64_t b) {
    uint64_t r;
    if (__builtin_mul_overflow(a, b, &r)) fail("64-bit product wraps", a, b, 0);
    return r;
}
static uint64_t add(uint64_t a, uint64_t b) {
    uint64_t r;
    if (__builtin_add_overflow(a, b, &r)) fail("64-bit sum wraps", a, b, 0);
    return r;
}
constexpr uint64_t U32 = 0xffffffffull, I32 = 0x7fffffffull;
constexpr uint64_t HBM = 288ull << 30;      // nothing a plan asks for may exceed the device

// one k_coop launch over n checks of a state with stride nc, k pairs per check
static void launch_coop(uint64_t n, uint64_t nc, uint64_t k) {
    g_launches++;
    REQUIRE(n >= 1 && n <= U32 && nc <= U32 && n <= nc && k <= U32, n, nc, k);
    REQUIRE(blocks_coop(n) <= I32 && mul(blocks_coop(n), GROUPS) >= n, n, nc, k);
    // record indices are size_t in the kernel; the line index of the last record: ((67 * k + k - 1) * 6 + 5) * n + n - 1
    (void)mul(add(mul(add(mul(NLINES - 1, k), k - 1), 6), 5), n);
    (void)mul(mul(ST_SIZE, nc), REC_BYTES);
}
// one k_prep_lines launch: g pairs of each of n checks into a line buffer of stride n
static void launch_prep(uint64_t n, uint64_t g, uint64_t lines_cap) {
    g_launches++;
    REQUIRE(prep_fits(n, g), n, g, 0);
    REQUIRE(mul(n, g) <= U32 && blocks_two_lanes(mul(n, g)) <= I32, n, g, 0);
    REQUIRE(lines_bytes(g, n) <= lines_cap, n, g, lines_cap);      // what the launch writes fits what for_chunks allocated
    REQUIRE(lines_bytes(g, n) == mul(mul(mul(mul(NLINES, g), 6), n), REC_BYTES), n, g, 0);
}
static void launch_inv(const Knobs& kn, uint64_t n, uint64_t nc, uint64_t count) {
    g_launches++;
    const Inv pi = plan_inv(kn, n, count);
    const uint64_t total = mul(n, count);
    REQUIRE(total <= U32 && n <= nc && nc <= U32, n, nc, count);          // the kernel's element index is a uint32_t
    REQUIRE(pi.batch >= 1 && pi.batch <= kn.inv_batch, pi.batch, kn.inv_batch, total);
    REQUIRE(mul(pi.lanes, pi.batch) >= total && (pi.lanes == 0 || mul(pi.lanes - 1, pi.batch) < total), pi.lanes, pi.batch, total);
    REQUIRE(blocks_one_lane(pi.lanes) <= I32, pi.lanes, 0, 0);
}
static void launch_ksq(uint64_t n, uint64_t nc) {
    g_launches++;
    REQUIRE(n >= 1 && n <= nc && nc <= U32 && blocks_ksq(n) <= I32, n, nc, 0);
}
static void launch_kdec(uint64_t n, uint64_t nc, uint64_t count) {
    g_launches++;
    REQUIRE(n <= nc && nc <= U32 && mul(2, mul(n, count)) + 63 <= U32, n, nc, count);   // lane number = 2 * (snapshot index)
}

// run_fexp_c over n checks of a state with stride nc: six step programs, five x-power chains of (k_ksq, k_kdec_a, k_batch_inv, k_kdec_b)
static void phase_c(const Knobs& kn, uint64_t n, uint64_t nc) {
    for (int chain = 0; chain < 5; chain++) {
        launch_coop(n, nc, 1);
        launch_ksq(n, nc);
        launch_kdec(n, nc, 6);          // at most six snapshots per chain (three in the shipped plan)
        launch_inv(kn, n, nc, 6);
        launch_kdec(n, nc, 6);
    }
    launch_coop(n, nc, 1);
}

// the chunks of one for_chunks call: cover [0, n_total) exactly once, each within the workspace
template <class Body>
static void walk_chunks(const Knobs& kn, uint64_t n_total, uint64_t k, bool need_lines, uint64_t override_, Body body) {
    const Chunks pc = plan_chunks(kn, n_total, k, need_lines, override_, false);
    REQUIRE(pc.chunk >= 1 && pc.chunk <= (override_ ? MAX_SUPER : MAX_CHUNK) && pc.pipes >= 1 && pc.pipes <= kn.n_pipes, pc.chunk, pc.pipes, n_total);   // a phase-C part has no line buffer
    REQUIRE(pc.cmax <= pc.chunk && pc.cmax <= n_total, pc.cmax, pc.chunk, n_total);
    REQUIRE(pc.n_chunks == (n_total + pc.chunk - 1) / pc.chunk, pc.n_chunks, pc.chunk, n_total);
    const uint64_t lines_cap = need_lines ? lines_bytes(group_size(kn, k), pc.cmax) : 0;
    if (need_lines) REQUIRE(lines_cap <= HBM / 2, lines_cap, k, pc.cmax);
    uint64_t covered = 0, c = 0;
    for (uint64_t base = 0; base < n_total; base += pc.chunk, c++) {
        const uint64_t n = n_total - base < pc.chunk ? n_total - base : pc.chunk;
        REQUIRE(base == covered && n >= 1 && n <= pc.cmax, base, covered, n);
        body(base, n, lines_cap);
        covered += n;
    }
    REQUIRE(covered == n_total && c == pc.n_chunks, covered, n_total, c);
}

// miller_on_pipe: the Miller launches of one chunk (n checks of k pairs, state stride nc)
static void miller_chunk(const Knobs& kn, uint64_t n, uint64_t nc, uint64_t k, uint64_t lines_cap) {
    if (k <= kn.max_stream) {
        launch_prep(n, k, lines_cap);
        launch_coop(n, nc, k);
        return;
    }
    uint64_t seen = 0;
    for (uint64_t j0 = 0; j0 < k; j0 += kn.max_stream) {
        const uint64_t g = k - j0 < kn.max_stream ? k - j0 : kn.max_stream;
        launch_prep(n, g, lines_cap);
        launch_coop(n, nc, g);
        if (j0) launch_coop(n, nc, 1);
        seen += g;
    }
    REQUIRE(seen == k, seen, k, 0);
}

// coop_pairing / coop_final_exp (two_phase) over n_total checks of k pairs
static void pairing(const Knobs& kn, uint64_t n_total, uint64_t k) {
    g_cases++;
    REQUIRE(!too_many(n_total, k), n_total, k, 0);
    uint64_t done = 0;
    for (uint64_t sb = 0; sb < n_total; sb += kn.super) {
        const uint64_t ns = n_total - sb < kn.super ? n_total - sb : kn.super;
        REQUIRE(ns <= MAX_SUPER && state_bytes(ns) <= HBM / 4 && state_bytes(ns) == mul(mul(ST_SIZE, ns), REC_BYTES), ns, 0, 0);
        uint64_t pairs_off = mul(add(sb, ns), k);       // wire offsets: 12 / 24 u64 per pair, 72 per check - size_t in the product
        (void)mul(pairs_off, 192);
        (void)mul(add(sb, ns), 576);
        walk_chunks(kn, ns, k, true, 0, [&](uint64_t, uint64_t n, uint64_t cap) {
            miller_chunk(kn, n, ns, k, cap);
            launch_coop(n, ns, 1);      // fexp_a
        });
        launch_inv(kn, ns, ns, 1);
        const PhaseC pcc = plan_phase_c(kn, ns, false);
        if (pcc.mode == C_PARTS) {
            REQUIRE(pcc.part >= 1 && pcc.part % 16 == 0, pcc.part, ns, 0);
            walk_chunks(kn, ns, 1, false, pcc.part, [&](uint64_t, uint64_t n, uint64_t) { phase_c(kn, n, ns); });
        } else if (pcc.mode == C_SINGLE) {
            phase_c(kn, ns, ns);
        } else {
            walk_chunks(kn, ns, 1, false, 0, [&](uint64_t, uint64_t n, uint64_t) { phase_c(kn, n, ns); });
        }
        done += ns;
    }
    REQUIRE(done == n_total, done, n_total, 0);
}

// coop_miller (multi_miller_loop alone): chunks with their own state
static void miller_only(const Knobs& kn, uint64_t n_total, uint64_t k) {
    g_cases++;
    walk_chunks(kn, n_total, k, true, 0, [&](uint64_t, uint64_t n, uint64_t cap) { miller_chunk(kn, n, n, k, cap); });
}

// host-pointer entry points: slices through two workspace slots; the points check's workspaces; the G2 subgroup check's launches
static void host_side(uint64_t n_checks, uint64_t k) {
    for (uint64_t hs : {64ull, 1ull << 19, 1ull << 24}) {
        const Slices sl = plan_slices(hs, n_checks, k);
        REQUIRE(sl.checks_per_slice >= 1 && mul(sl.n_slices, sl.checks_per_slice) >= n_checks, sl.checks_per_slice, sl.n_slices, n_checks);
        REQUIRE(sl.n_slices == 0 || mul(sl.n_slices - 1, sl.checks_per_slice) < n_checks, sl.checks_per_slice, sl.n_slices, n_checks);
        REQUIRE(mul(sl.checks_per_slice, k) <= (hs > k ? hs : k), sl.checks_per_slice, k, hs);
        (void)mul(mul(sl.checks_per_slice, k), 192);
        (void)mul(sl.checks_per_slice, 576);
    }
    const uint64_t np = mul(n_checks, k);
    REQUIRE(np <= I32, n_checks, k, 0);
    (void)add(mul(np, 288), 8);                                   // PC_BYTES
    REQUIRE((add(mul(np, 24), 255)) / 256 <= I32, np, 0, 0);      // k_gather_checks over the u64 words of the G2 points
    for (uint64_t lo = 0; lo < np; lo += VALID_CHUNK) {           // k_g2_valid_fast3: 32-bit scratch offsets, 2 lanes x 16 B x 8 planes per point
        const uint64_t m = np - lo < VALID_CHUNK ? np - lo : VALID_CHUNK;
        REQUIRE(mul(mul(2, m) + 63, 16 * 8) <= U32 && vscratch_bytes(m) <= (1ull << 30) + 8192, m, 0, 0);
    }
}

int main() {
    const uint64_t ns_list[] = {1, 4, 5, 16, 319, 320, 321, 4095, 4096, (1u << 14) - 1, 1u << 14, (1u << 15) + 1, (1u << 16) - 1, 1u << 16, (1u << 16) + 1,
                                (1u << 17), (1u << 17) + 16, (1u << 18) - 1, 1u << 18, (1u << 20), (1u << 20) + 1, (1u << 22) - 1, 1u << 22, (1u << 22) + 1,
                                1u << 27, (1u << 27) + 5, 0x7fffffffu};
    const uint64_t ks[] = {1, 2, 3, 4, 5, 8, 9, 16, 17, 63, 64, 65, 96, 1000, 65535};
    std::vector<Knobs> knobs;
    for (long chunk : {1L, 321L, 4096L, 1L << 15, 1L << 16, 1L << 20, 1L << 30}) {
        for (int pipes : {1, 2, 3, 9}) {
            for (long ms : {8L, 16L, 64L}) {
                Knobs k;
                k.chunk = (size_t)chunk;
                k.n_pipes = pipes;
                k.max_stream = (size_t)ms;
                k = clamp(k);
                k.c_single_min = k.chunk;
                knobs.push_back(k);
                Knobs k2 = k;
                k2.c_split = 3; k2.split_min = 1; k2.super = (size_t)1 << 22; k2.inv_batch = 1; k2.inv_lanes = 1;
                k2 = clamp(k2);
                knobs.push_back(k2);
                Knobs k3 = k;
                k3.c_single = false; k3.split_min = (size_t)1 << 40; k3.super = 1; k3.inv_batch = 100000; k3.inv_lanes = (size_t)1 << 40;
                k3 = clamp(k3);
                knobs.push_back(k3);
            }
        }
    }
    for (const Knobs& kn : knobs) {
        REQUIRE(kn.chunk >= MIN_CHUNK && kn.chunk <= MAX_CHUNK && kn.super >= kn.chunk && kn.super <= MAX_SUPER && kn.n_pipes >= 1 && kn.n_pipes <= MAX_PIPES,
                kn.chunk, kn.super, kn.n_pipes);
        for (uint64_t n : ns_list) {
            for (uint64_t k : ks) {
                if (too_many(n, k)) continue;
                // the walk is linear in n / chunk: the small-chunk knob sets take the small and medium batches, the default-sized ones everything
                if (n / kn.chunk > 1200 || n / kn.super > 1200) continue;
                pairing(kn, n, k);
                miller_only(kn, n, k);
            }
        }
    }
    for (uint64_t n : ns_list)
        for (uint64_t k : ks)
            if (!too_many(n, k)) host_side(n, k);
    // the limits themselves
    REQUIRE(too_many(0x80000000ull) && !too_many(0x7fffffffull) && too_many(1, 65536) && too_many(32768, 65536) && !too_many(32767, 65535) && too_many(32769, 65535), 0, 0, 0);
    // round 6: what a batch of at most one chunk does by default - two pipelines from 2 x split_min checks on, one below
    {
        const Knobs kn = clamp(Knobs());
        const Chunks a = plan_chunks(kn, (size_t)1 << 16, 1, true, 0, false), b = plan_chunks(kn, 2 * kn.split_min - 32, 1, true, 0, false),
                     c = plan_chunks(kn, (size_t)1 << 20, 1, true, 0, false), d = plan_chunks(kn, 3 << 15, 3, true, 0, false);
        REQUIRE(a.pipes == 2 && a.n_chunks == 2 && a.chunk == (size_t)1 << 15, a.pipes, a.n_chunks, a.chunk);
        REQUIRE(b.pipes == 1 && b.n_chunks == 1, b.pipes, b.n_chunks, b.chunk);
        REQUIRE(c.pipes == 2 && c.n_chunks == 16 && c.chunk == (size_t)1 << 16, c.pipes, c.n_chunks, c.chunk);
        REQUIRE(d.pipes == 2 && d.n_chunks == 2 && d.chunk == 3 << 14, d.pipes, d.n_chunks, d.chunk);
    }
    std::printf("plan_check ok: %llu cases, %llu launches checked\n", g_cases, g_launches);
    return 0;
}
