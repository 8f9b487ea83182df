// Throughput probe for the 28-bit carry-free core as hipcc compiles it: each lane repeatedly does
// T LDS-fed products into one accumulator set, one Montgomery reduction, one LDS store
// (the MULACC step of the cooperative kernels).  Reports cycles per step per wave and the implied
// issue interval per v_mad_i64_i32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../zkvm_pairings_amd/csrc/zkp_fp28.hpp"
using namespace zkp28;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NSLOT = 40;            // Fp slots per lane-group region
constexpr int ITER = 64;

// LDS layout: quad-plane SoA: int4 at [(q * NSLOT + slot)] for q = 0..3 (limbs 4q..4q+3; 14,15 unused)
__device__ __forceinline__ void lds_load(int32_t* x, const int4* base, int slot) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int4 v = base[q * NSLOT + slot];
        x[4 * q] = v.x; x[4 * q + 1] = v.y;
        if (q < 3) { x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
    }
}
__device__ __forceinline__ void lds_store(int4* base, int slot, const int32_t* x) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int4 v;
        v.x = x[4 * q]; v.y = x[4 * q + 1];
        v.z = q < 3 ? x[4 * q + 2] : 0; v.w = q < 3 ? x[4 * q + 3] : 0;
        base[q * NSLOT + slot] = v;
    }
}

template <int T>
__global__ void __launch_bounds__(512) k(unsigned long long* out, const unsigned* tbl, int waves_per_block) {
    extern __shared__ int4 lds[];
    int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    int grp = lane / 12;                         // 5 groups of 12 lanes, lanes 60..63 idle-ish
    int4* base = lds + (wave * 6 + grp) * (4 * NSLOT + 3);
    // init region
    for (int s = lane % 12; s < NSLOT; s += 12) {
        int32_t x[16];
        for (int i = 0; i < 16; i++) x[i] = (int32_t)((s * 2654435761u + i * 40503u + lane) & 0x7ffffff) - (1 << 26);
        lds_store(base, s, x);
    }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        Acc acc;
        acc_zero(acc);
        unsigned dst = 0;
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            unsigned w = tbl[(it * T + t) * 64 + lane];
            int32_t a[16], b[16];
            lds_load(a, base, w & 63);
            lds_load(b, base, (w >> 8) & 63);
            dst = (w >> 16) & 63;
            acc_mul(acc, a, b);
        }
        int32_t r[16];
        acc_reduce(r, acc);
        lds_store(base, dst, r);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        int4 v = base[0];
        out[2 * (blockIdx.x * waves_per_block + wave)] = t1 - t0;
        out[2 * (blockIdx.x * waves_per_block + wave) + 1] = v.x;
    }
}

template <int T>
void run(int cus, unsigned long long* dout, unsigned* dtbl) {
    for (int wpb : {4, 8}) {
        size_t lds_bytes = (size_t)wpb * 6 * (4 * NSLOT + 3) * 16;
        CHECK(hipFuncSetAttribute((const void*)k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<T><<<cus, wpb * 64, lds_bytes>>>(dout, dtbl, wpb);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<T><<<cus, wpb * 64, lds_bytes>>>(dout, dtbl, wpb);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int nw = cus * wpb;
        std::vector<unsigned long long> h(2 * nw);
        CHECK(hipMemcpy(h.data(), dout, 16 * nw, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> c(nw);
        for (int i = 0; i < nw; i++) c[i] = h[2 * i];
        std::sort(c.begin(), c.end());
        double per_step = (double)c[nw / 2] / ITER;
        double mads = T * 196.0 + 196.0;
        printf("T=%2d waves/SIMD=%d  cycles/step/wave=%9.1f  => per SIMD %8.1f cyc/step ; %.2f cyc per mad (incl. everything) ; wall %.3f ms lds %zu B\n",
               T, wpb / 4, per_step, per_step / (wpb / 4), per_step / (wpb / 4) / mads, ms, lds_bytes);
    }
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    unsigned long long* dout; CHECK(hipMalloc(&dout, 16 * cus * 8));
    std::vector<unsigned> tbl(ITER * 12 * 64);
    for (size_t i = 0; i < tbl.size(); i++) { unsigned r = (unsigned)(i * 2654435761u); tbl[i] = (r % 37) | ((r / 61 % 37) << 8) | ((r / 3721 % 37) << 16); }
    unsigned* dtbl; CHECK(hipMalloc(&dtbl, tbl.size() * 4)); CHECK(hipMemcpy(dtbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
    run<1>(cus, dout, dtbl);
    run<3>(cus, dout, dtbl);
    run<6>(cus, dout, dtbl);
    run<12>(cus, dout, dtbl);
    return 0;
}
