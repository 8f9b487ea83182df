// zkp_coop.hpp -- interface between the C-ABI dispatcher (zkp_pairings.hip) and the
// lane-cooperative kernel family (zkp_coop.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace zkp {

struct CoopState {
    bool available = false;
    int cus = 0;
    void* d_prog = nullptr;   // microcode tables resident in HBM
    void* d_ws = nullptr;     // per-launch workspace (Miller outputs between the two phases)
    size_t ws_bytes = 0;
};

hipError_t coop_init(CoopState* st, const hipDeviceProp_t& prop);
void coop_free(CoopState* st);
// true when `kind` (zkp_kernel_kind) routes the pairing path through the cooperative kernels
bool coop_selected(const CoopState* st, int kind);
// pairs per check the cooperative path takes (checks with more than eight pairs run in groups of eight)
bool coop_supports_k(size_t k);
// batched Fp operation on wire operands through the 28-bit core: op 0 mul, 1 add, 2 sub, 3 neg, 4 square, 5 invert
hipError_t coop_fp28_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, hipStream_t s);
// G1/G2 is_valid on the 28-bit core (status 0 / 1 / 2)
hipError_t coop_g1_valid(const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* status, hipStream_t s);
// st: the context's cooperative state (the three-wave G2 kernel keeps the affine points' limbs in a scratch buffer of its own)
hipError_t coop_g2_valid(CoopState* st, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* status, hipStream_t s);
hipError_t coop_g1_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf, hipStream_t s);
hipError_t coop_g2_mul(const uint64_t* base, size_t stride, const uint64_t* sc, size_t n, uint64_t* out, uint8_t* out_inf, hipStream_t s);
// one tower operation per record (zkp_tower_op_batch); ab = n a-records followed by n b-records
hipError_t coop_tower_op(CoopState* st, int op, const uint64_t* ab, size_t n, uint32_t repeat, uint64_t* out, hipStream_t s);
hipError_t coop_time_prog(CoopState* st, int which, size_t n, hipStream_t s, hipEvent_t e0, hipEvent_t e1, float* ms);
hipError_t coop_miller(CoopState* st, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks,
                       size_t k, uint64_t* out, hipStream_t s);
// out (wire Gt), ok (per element Gt == identity) and all_ok (AND-ed into *all_ok) are each optional
hipError_t coop_final_exp(CoopState* st, const uint64_t* f, size_t n, uint64_t* out, uint8_t* ok, int* all_ok, hipStream_t s);
// one level of the Fp12 product tree, in place on wire records: buf[c] <- buf[c] * buf[c + h] for c < m
hipError_t coop_fp12_mul_pairs(CoopState* st, uint64_t* buf, size_t m, size_t h, hipStream_t s);
// n_dev (device pointer, may be null): the number of checks that exist, <= n_checks, read by the kernels themselves (never by the host)
hipError_t coop_pairing(CoopState* st, const uint64_t* g1, const uint64_t* g2, const uint8_t* i1, const uint8_t* i2, size_t n_checks,
                        size_t k, uint64_t* out_gt, uint8_t* ok, int* all_ok, hipStream_t s, const uint32_t* n_dev = nullptr);

// measurement: one pass of the fused pairing, ms / launches per kernel class (ZKP_PROFILE_CLASSES of them, include/zkp_pairings.h)
hipError_t coop_profile_pairing(CoopState* st, const uint64_t* g1, const uint64_t* g2, size_t n, uint64_t* out_gt, float* ms, int* launches, hipStream_t s);

}  // namespace zkp

// debugging aid: ZKP_DEBUG_ALLOC=<file> appends one line per device / pinned allocation and release (address, bytes, what for), so
// that the address of a GPU memory fault can be matched to a buffer
#include <cstdio>
#include <cstdlib>
#include <string>
static inline void zkp_dbg_alloc(const char* what, const void* p, size_t bytes) {
    static const std::string path = getenv("ZKP_DEBUG_ALLOC") ? getenv("ZKP_DEBUG_ALLOC") : "";   // a copy: setenv may move the environment
    if (path.empty()) return;
    if (FILE* f = fopen(path.c_str(), "a")) {
        fprintf(f, "%s %p %zu\n", what, p, bytes);
        fclose(f);
    }
}
