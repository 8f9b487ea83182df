// zkp_coop.hip -- lane-cooperative kernel family (placeholder until the microcoded kernels land:
// reports "not available", so the dispatcher keeps using the thread family).
#include "zkp_coop.hpp"

namespace zkp {
hipError_t coop_init(CoopState* st, const hipDeviceProp_t& prop) { st->available = false; st->cus = prop.multiProcessorCount; return hipSuccess; }
void coop_free(CoopState*) {}
bool coop_selected(const CoopState* st, int kind) { return st->available && kind != 1; }
hipError_t coop_miller(CoopState*, const uint64_t*, const uint64_t*, const uint8_t*, const uint8_t*, size_t, size_t, uint64_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t coop_final_exp(CoopState*, const uint64_t*, size_t, uint64_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t coop_pairing(CoopState*, const uint64_t*, const uint64_t*, const uint8_t*, const uint8_t*, size_t, size_t, uint64_t*, uint8_t*, int*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace zkp
