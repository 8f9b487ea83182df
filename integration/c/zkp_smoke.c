/* Pure-C consumer of include/zkp_pairings.h: no Python, no torch.  Build:
 *   gcc -O2 -I include integration/c/zkp_smoke.c -L zkvm_pairings_amd -lzkp_pairings -Wl,-rpath,$PWD/zkvm_pairings_amd -o zkp_smoke
 * Computes e(G1gen, G2gen) (generators: reference src/common.rs:92-144), checks it is not Gt::identity(), checks
 * e(P,Q) * e(-P,Q) == identity through one shared final exponentiation, and validates the generators. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

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

int main(void) {
    zkp_ctx* ctx = NULL;
    int rc = zkp_init(0, &ctx);
    if (rc != ZKP_OK) { fprintf(stderr, "zkp_init: %s\n", zkp_strerror(rc)); return 2; }
    uint64_t gt[72];
    rc = zkp_pairing_batch(ctx, G1, G2, NULL, NULL, 1, gt);
    if (rc != ZKP_OK) { fprintf(stderr, "pairing: %s (%s)\n", zkp_strerror(rc), zkp_last_error(ctx)); return 3; }
    printf("e(G1,G2).c0.c0.c0 = 0x%016llx...%016llx\n", (unsigned long long)gt[5], (unsigned long long)gt[0]);
    if (memcmp(gt, zkp_gt_identity(), sizeof gt) == 0) { fprintf(stderr, "degenerate pairing\n"); return 4; }
    /* -P = (x, p - y) */
    uint64_t g1s[24], g2s[48];
    memcpy(g1s, G1, sizeof G1);
    memcpy(g1s + 12, G1, sizeof G1);
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
        const uint64_t d = P[i] - G1[6 + i], e = d - borrow;
        borrow = (uint64_t)(P[i] < G1[6 + i]) | (uint64_t)(d < borrow);
        g1s[12 + 6 + i] = e;
    }
    memcpy(g2s, G2, sizeof G2);
    memcpy(g2s + 24, G2, sizeof G2);
    uint8_t ok = 0, st1 = 9, st2 = 9;
    int all_ok = 0;
    rc = zkp_pairing_check_batch(ctx, g1s, g2s, NULL, NULL, 1, 2, &ok, &all_ok);
    if (rc != ZKP_OK || !ok || !all_ok) { fprintf(stderr, "product check failed rc=%d ok=%d\n", rc, ok); return 5; }
    if (zkp_g1_is_valid_batch(ctx, G1, NULL, 1, &st1) != ZKP_OK || zkp_g2_is_valid_batch(ctx, G2, NULL, 1, &st2) != ZKP_OK || st1 || st2) {
        fprintf(stderr, "generators not valid? %d %d\n", st1, st2);
        return 6;
    }
    zkp_free(ctx);
    printf("C ABI smoke ok\n");
    return 0;
}
