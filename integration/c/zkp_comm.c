/* Plain-C rank of a sharded pairing check: everything SURVEY.md 8(e) needs comes from include/zkp_pairings.h - the check of the
 * rank's block AND the path's one RCCL collective (all-reduce(MIN) of the flag), no torch, no MPI.  Build:
 *   gcc -O2 -I include integration/c/zkp_comm.c -L zkvm_pairings_amd -lzkp_pairings -Wl,-rpath,$PWD/zkvm_pairings_amd -Wl,-rpath,/opt/rocm/lib -o zkp_comm
 * Run one process per GPU:   zkp_comm <nranks> <rank> <id file> [device]
 * Rank 0 writes the 128-byte communicator id to <id file> (created under a temporary name, then renamed); the other ranks wait for
 * it.  The global batch is 4 * nranks two-pair checks e(P,Q) e(-P,Q) == 1 (P = [s] G1gen by zkp_g1_mul_batch), rank r owns checks
 * 4r .. 4r+3.  Pass 1: all checks hold -> every rank must see all_ok = 1.  Pass 2: the LAST rank breaks one of its checks -> every
 * rank must see all_ok = 0 while the other ranks' own ok bytes stay 1.  Pass 3: the product check over the whole batch through
 * one all-gather of 576 B per rank (zkp_pairing_product_check_allgather) is the identity on every rank.
 * The GPU box has one GPU: the test suite runs it with nranks = 1 (a one-rank communicator); on a node, nranks = number of GPUs. */
#define _DEFAULT_SOURCE 1 /* usleep, aligned_alloc under a strict -std= */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
#define CHECKS 4

static void neg_y(uint64_t* dst, const uint64_t* y) {
    uint64_t borrow = 0;      /* p - y, limb by limb (y < p) */
    for (int i = 0; i < 6; i++) {
        const uint64_t d = P[i] - y[i], e = d - borrow;
        borrow = (uint64_t)(P[i] < y[i]) | (uint64_t)(d < borrow);
        dst[i] = e;
    }
}

#define DIE(code, ...) do { fprintf(stderr, __VA_ARGS__); return code; } while (0)

int main(int argc, char** argv) {
    if (argc < 4) DIE(1, "usage: %s <nranks> <rank> <id file> [device]\n", argv[0]);
    const int nranks = atoi(argv[1]), rank = atoi(argv[2]);
    const char* idfile = argv[3];
    const int device = argc > 4 ? atoi(argv[4]) : rank;
    unsigned char id[ZKP_COMM_ID_BYTES];
    if (rank == 0) {
        if (zkp_comm_unique_id(id) != ZKP_OK) DIE(2, "zkp_comm_unique_id failed\n");
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, idfile) != 0) DIE(2, "cannot write %s\n", idfile);
    } else {
        FILE* f = NULL;
        for (int i = 0; i < 600 && !(f = fopen(idfile, "rb")); i++) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) DIE(2, "rank %d: no id in %s\n", rank, idfile);
        fclose(f);
    }
    zkp_ctx* ctx = NULL;
    int rc = zkp_init(device, &ctx);
    if (rc != ZKP_OK) DIE(3, "zkp_init(%d): %s\n", device, zkp_strerror(rc));
    int nr = -1, rk = -1;
    if (zkp_comm_info(ctx, &nr, &rk) != ZKP_OK || nr != 0) DIE(3, "a fresh context claims a communicator\n");
    int all = 7;
    if (zkp_pairing_check_batch_allreduce(ctx, G1, G2, NULL, NULL, 1, 1, NULL, &all) != ZKP_ERR_COMM) DIE(3, "allreduce without a communicator must fail\n");
    if ((rc = zkp_comm_init_rank(ctx, nranks, rank, id)) != ZKP_OK) DIE(4, "zkp_comm_init_rank: %s (%s)\n", zkp_strerror(rc), zkp_last_error(ctx));
    if (zkp_comm_info(ctx, &nr, &rk) != ZKP_OK || nr != nranks || rk != rank) DIE(4, "zkp_comm_info\n");

    /* this rank's block: checks (P_j, Q), (-P_j, Q) with P_j = [rank * CHECKS + j + 2] G1gen */
    uint64_t sc[CHECKS][4], pj[CHECKS][12], g1[CHECKS * 2][12], g2[CHECKS * 2][24];
    memset(sc, 0, sizeof sc);
    for (int j = 0; j < CHECKS; j++) sc[j][0] = (uint64_t)(rank * CHECKS + j + 2);
    if ((rc = zkp_g1_mul_batch(ctx, G1, 0, &sc[0][0], CHECKS, &pj[0][0], NULL)) != ZKP_OK) DIE(5, "g1_mul: %s\n", zkp_last_error(ctx));
    for (int j = 0; j < CHECKS; j++) {
        memcpy(g1[2 * j], pj[j], 96);
        memcpy(g1[2 * j + 1], pj[j], 48);
        neg_y(&g1[2 * j + 1][6], &pj[j][6]);
        memcpy(g2[2 * j], G2, 192);
        memcpy(g2[2 * j + 1], G2, 192);
    }
    uint8_t ok[CHECKS];
    all = 0;
    if ((rc = zkp_pairing_check_batch_allreduce(ctx, &g1[0][0], &g2[0][0], NULL, NULL, CHECKS, 2, ok, &all)) != ZKP_OK)
        DIE(6, "pass 1: %s (%s)\n", zkp_strerror(rc), zkp_last_error(ctx));
    for (int j = 0; j < CHECKS; j++)
        if (!ok[j]) DIE(6, "pass 1: check %d of rank %d fails\n", j, rank);
    if (all != 1) DIE(6, "pass 1: rank %d sees all_ok = %d\n", rank, all);
    /* pass 2: the last rank breaks its check 1 (second pair: P instead of -P) */
    if (rank == nranks - 1) memcpy(g1[3], pj[1], 96);
    all = 1;
    if ((rc = zkp_pairing_check_batch_allreduce(ctx, &g1[0][0], &g2[0][0], NULL, NULL, CHECKS, 2, ok, &all)) != ZKP_OK)
        DIE(7, "pass 2: %s (%s)\n", zkp_strerror(rc), zkp_last_error(ctx));
    for (int j = 0; j < CHECKS; j++)
        if (ok[j] != !(rank == nranks - 1 && j == 1)) DIE(7, "pass 2: ok[%d] = %d on rank %d\n", j, ok[j], rank);
    if (all != 0) DIE(7, "pass 2: rank %d sees all_ok = %d, a peer's failing check did not arrive\n", rank, all);
    /* pass 3: the whole (repaired) batch as ONE product check: all-gather of the ranks' Miller products */
    if (rank == nranks - 1) { memcpy(g1[3], pj[1], 48); neg_y(&g1[3][6], &pj[1][6]); }
    uint64_t gt[72];
    int one = 0;
    if ((rc = zkp_pairing_product_check_allgather(ctx, &g1[0][0], &g2[0][0], NULL, NULL, CHECKS * 2, gt, &one)) != ZKP_OK)
        DIE(8, "pass 3: %s (%s)\n", zkp_strerror(rc), zkp_last_error(ctx));
    if (!one || memcmp(gt, zkp_gt_identity(), sizeof gt) != 0) DIE(8, "pass 3: the product over all ranks is not the identity on rank %d\n", rank);
    if ((rc = zkp_comm_destroy(ctx)) != ZKP_OK) DIE(9, "zkp_comm_destroy: %s\n", zkp_last_error(ctx));
    zkp_free(ctx);
    printf("rank %d of %d: RCCL AND-reduce behind the C ABI ok\n", rank, nranks);
    return 0;
}
