/* Pure-C consumer of the multi-GPU entry points of include/zkp_pairings.h: ONE host thread, several
 * contexts, contiguous blocks of checks per context, AND of the per-context flags on the host.
 *   gcc -O2 -I include integration/c/zkp_multi.c -L zkvm_pairings_amd -lzkp_pairings -Wl,-rpath,$PWD/zkvm_pairings_amd -o zkp_multi
 *   ./zkp_multi [n_ctx [n_pairs]]        context j is created on device j % (number of GPUs): on a one-GPU box all of
 *                                         them share device 0, on an 8-GPU node `./zkp_multi 8` uses every GPU.
 * Checks: zkp_pairing_batch_multi == zkp_pairing_batch (every Gt), zkp_pairing_check_batch_multi == the single-context
 * call on 2-pair checks e(a P, Q) e(-a P, Q) == 1 with every 7th check spoiled, and the AND flags; the same batch from
 * page-locked memory (zkp_host_alloc / zkp_host_register) with both timings printed. */
#define _POSIX_C_SOURCE 199309L
#define _DEFAULT_SOURCE 1 /* usleep, aligned_alloc under a strict -std= */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zkp_pairings.h"

static const uint64_t G1[12] = {
    0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL,
    0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t G2[24] = {
    0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL,
    0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL,
    0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL,
    0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
static const uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};

static void fp_neg(uint64_t* out, const uint64_t* y) { /* p - y, y != 0 */
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
        const uint64_t d = P[i] - y[i], e = d - borrow;
        borrow = (uint64_t)(P[i] < y[i]) | (uint64_t)(d < borrow);
        out[i] = e;
    }
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv) {
    int n_ctx = argc > 1 ? atoi(argv[1]) : 3;
    size_t n = argc > 2 ? (size_t)atol(argv[2]) : 1000;
    if (n_ctx < 1 || n_ctx > 16 || n < 1) return 2;
    zkp_ctx* ctx[16] = {0};
    int ndev = 0;
    for (int j = 0; j < n_ctx; j++) {          /* device j while zkp_init accepts it, then wrap around */
        int rc = zkp_init(ndev == j ? j : j % ndev, &ctx[j]);
        if (rc == ZKP_OK && ndev == j) ndev = j + 1;
        else if (rc == ZKP_ERR_NO_DEVICE && ndev == j && j > 0) rc = zkp_init(j % ndev, &ctx[j]);
        CHECK(rc == ZKP_OK, "zkp_init for context %d: %s", j, zkp_strerror(rc));
    }
    printf("%d contexts on %d device(s), %zu pairs\n", n_ctx, ndev, n);

    /* pairs (a_i G1, b_i G2) with small distinct scalars, made on the GPU */
    uint64_t* sc = calloc(4 * n, 8), *sb = calloc(4 * n, 8);
    /* whole pages of their own, so that they can be page-locked in place below (zkp_host_register takes nothing else) */
    const size_t b1 = (96 * n + 4095) / 4096 * 4096, b2 = (192 * n + 4095) / 4096 * 4096;
    uint64_t* g1 = aligned_alloc(4096, b1), *g2 = aligned_alloc(4096, b2);
    CHECK(g1 && g2, "aligned_alloc failed");
    for (size_t i = 0; i < n; i++) { sc[4 * i] = 3 + 2 * i; sc[4 * i + 1] = 0x9e3779b97f4a7c15ULL * (i + 1); sb[4 * i] = 5 + 7 * i; }
    CHECK(zkp_g1_mul_batch(ctx[0], G1, 0, sc, n, g1, NULL) == ZKP_OK, "g1 mul: %s", zkp_last_error(ctx[0]));
    CHECK(zkp_g2_mul_batch(ctx[0], G2, 0, sb, n, g2, NULL) == ZKP_OK, "g2 mul: %s", zkp_last_error(ctx[0]));

    /* pairing(): multi == single, every Gt */
    uint64_t* gt1 = malloc(576 * n), *gtm = malloc(576 * n);
    uint8_t* okm = malloc(n);
    int all = -1;
    CHECK(zkp_pairing_batch(ctx[0], g1, g2, NULL, NULL, n, gt1) == ZKP_OK, "single: %s", zkp_last_error(ctx[0]));
    CHECK(zkp_pairing_batch_multi(ctx, n_ctx, g1, g2, NULL, NULL, n, gtm, okm, &all) == ZKP_OK, "multi pairing failed");
    CHECK(memcmp(gt1, gtm, 576 * n) == 0, "multi-context Gt differs from the single-context Gt");
    CHECK(all == 0, "random pairings must not all be the identity");
    for (size_t i = 0; i < n; i++) CHECK(okm[i] == 0, "pairing %zu reported as identity", i);

    /* the same call from page-locked memory (zkp_host_alloc) and from the page-aligned arrays page-locked in place
     * (zkp_host_register): same Gt; the copies are then DMAs that overlap the kernels and the other contexts' copies */
    {
        void *p1 = NULL, *p2 = NULL, *pg = NULL;
        CHECK(zkp_host_alloc(96 * n, &p1) == ZKP_OK && zkp_host_alloc(192 * n, &p2) == ZKP_OK && zkp_host_alloc(576 * n, &pg) == ZKP_OK,
              "zkp_host_alloc failed");
        memcpy(p1, g1, 96 * n);
        memcpy(p2, g2, 192 * n);
        struct timespec t0, t1, t2;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        CHECK(zkp_pairing_batch_multi(ctx, n_ctx, g1, g2, NULL, NULL, n, gtm, okm, &all) == ZKP_OK, "multi pairing (pageable) failed");
        clock_gettime(CLOCK_MONOTONIC, &t1);
        CHECK(zkp_pairing_batch_multi(ctx, n_ctx, (const uint64_t*)p1, (const uint64_t*)p2, NULL, NULL, n, (uint64_t*)pg, okm, &all) == ZKP_OK,
              "multi pairing (page-locked) failed");
        clock_gettime(CLOCK_MONOTONIC, &t2);
        CHECK(memcmp(gt1, pg, 576 * n) == 0, "Gt from page-locked arrays differs");
        printf("zkp_pairing_batch_multi, %zu pairs, Gt out: %.2f ms from pageable memory, %.2f ms from zkp_host_alloc memory\n", n,
               (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) * 1e-6);
        CHECK(zkp_host_register(g1, b1) == ZKP_OK && zkp_host_register(g2, b2) == ZKP_OK, "zkp_host_register failed");
        CHECK(zkp_host_register((char*)gt1 + 8, 4096) == ZKP_ERR_ARG, "a range that is not whole pages must be refused");
        memset(gtm, 0, 576 * n);
        CHECK(zkp_pairing_batch_multi(ctx, n_ctx, g1, g2, NULL, NULL, n, gtm, NULL, NULL) == ZKP_OK, "multi pairing (registered) failed");
        CHECK(memcmp(gt1, gtm, 576 * n) == 0, "Gt from registered arrays differs");
        CHECK(zkp_host_unregister(g1) == ZKP_OK && zkp_host_unregister(g2) == ZKP_OK, "zkp_host_unregister failed");
        CHECK(zkp_host_free(p1) == ZKP_OK && zkp_host_free(p2) == ZKP_OK && zkp_host_free(pg) == ZKP_OK, "zkp_host_free failed");
        CHECK(zkp_host_alloc(0, &p1) == ZKP_ERR_ARG && zkp_host_free(NULL) == ZKP_OK, "argument checks of the host memory calls");
    }

    /* 2-pair checks e(P_i, Q_i) e(-P_i, Q_i) == 1, every 7th spoiled by pairing with Q_(i+1) instead */
    uint64_t* c1 = malloc(2 * 96 * n), *c2 = malloc(2 * 192 * n);
    uint8_t* ok1 = malloc(n), *ok2 = malloc(n);
    for (size_t i = 0; i < n; i++) {
        memcpy(c1 + 24 * i, g1 + 12 * i, 96);
        memcpy(c1 + 24 * i + 12, g1 + 12 * i, 48);
        fp_neg(c1 + 24 * i + 18, g1 + 12 * i + 6);
        memcpy(c2 + 48 * i, g2 + 24 * i, 192);
        memcpy(c2 + 48 * i + 24, g2 + 24 * ((i % 7 == 3) ? (i + 1) % n : i), 192);
    }
    int all1 = -1, all2 = -1;
    CHECK(zkp_pairing_check_batch(ctx[0], c1, c2, NULL, NULL, n, 2, ok1, &all1) == ZKP_OK, "single check: %s", zkp_last_error(ctx[0]));
    CHECK(zkp_pairing_check_batch_multi(ctx, n_ctx, c1, c2, NULL, NULL, n, 2, ok2, &all2) == ZKP_OK, "multi check failed");
    CHECK(memcmp(ok1, ok2, n) == 0 && all1 == all2, "multi-context flags differ from the single-context flags");
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const int expect = (i % 7 == 3 && n > 1) ? 0 : 1;
        CHECK(ok2[i] == expect, "check %zu: flag %d, expected %d", i, ok2[i], expect);
        bad += !expect;
    }
    CHECK(all2 == (bad ? 0 : 1), "AND flag %d with %zu failing checks", all2, bad);
    /* all checks good -> AND flag 1 (first min(n, 3) checks only contain good ones) */
    CHECK(zkp_pairing_check_batch_multi(ctx, n_ctx, c1, c2, NULL, NULL, n < 3 ? n : 3, 2, NULL, &all2) == ZKP_OK && all2 == 1, "AND flag of good checks");
    /* fewer checks than contexts, and none */
    CHECK(zkp_pairing_check_batch_multi(ctx, n_ctx, c1, c2, NULL, NULL, 1, 2, ok2, &all2) == ZKP_OK && ok2[0] == 1 && all2 == 1, "one check");
    CHECK(zkp_pairing_check_batch_multi(ctx, n_ctx, c1, c2, NULL, NULL, 0, 2, NULL, &all2) == ZKP_OK && all2 == 1, "no checks");
    CHECK(zkp_pairing_check_batch_multi(ctx, 2, c1, c2, NULL, NULL, 1, 2, NULL, &all2) == ZKP_OK || n_ctx < 2, "two contexts");
    zkp_ctx* twice[2] = {ctx[0], ctx[0]};
    CHECK(zkp_pairing_check_batch_multi(twice, 2, c1, c2, NULL, NULL, 1, 2, NULL, &all2) == ZKP_ERR_ARG, "a context listed twice must be rejected");
    for (int j = 0; j < n_ctx; j++) zkp_free(ctx[j]);
    printf("C ABI multi-context ok: %zu pairings and %zu checks (%zu spoiled) over %d contexts equal the single-context results\n", n, n, bad, n_ctx);
    return 0;
}
