// Instruction-rate microbenchmark for gfx950 integer/fp64 VALU ops that matter for
// 384-bit modular arithmetic. Each wave runs ITER iterations of 16 independent
// instructions of one kind; s_memtime brackets the loop. Reports cycles per
// wave-instruction per SIMD at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 4096;

template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, unsigned seed) {
    unsigned a[16], b[16];
    unsigned long long c[16];
    double d[16];
    unsigned long long e[16];
    for (int i = 0; i < 16; i++) e[i] = threadIdx.x * 77u + i;
    for (int i = 0; i < 16; i++) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed * (i + 7) ^ threadIdx.x; c[i] = (unsigned long long)a[i] * b[i]; d[i] = (double)a[i]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 5) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (KIND == 6) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (KIND == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) & 15]));
            if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 10) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i+1)&15]), "v"(b[i]) : "vcc");
            if (KIND == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 12) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 13) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "s10", "s11");
            if (KIND == 14) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 15) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 16) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 17) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 18) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            // round 5: does an INDEPENDENT cheap instruction hide in the shadow of a multiply-add of the same wavefront?
            if (KIND == 19) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_add_u32 %1, %1, %3" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]) : "s10", "s11");
            if (KIND == 20) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_add_u32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]) : "s10", "s11");
            if (KIND == 21) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_lshl_add_u64 %1, %1, 0, %4" : "+v"(c[i]), "+v"(e[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]), "v"(e[(i + 1) & 15]) : "s10", "s11");
            if (KIND == 22) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]) : "s10", "s11");
            // are the 64-bit integer multiply-add and the FP64 / FP32 FMA separate pipes?  (wall time of the pair against the sum of the two alone)
            if (KIND == 24) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_fma_f64 %1, %1, %4, %1" : "+v"(c[i]), "+v"(d[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]), "v"(d[(i + 1) & 15]) : "s10", "s11");
            if (KIND == 25) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_fma_f32 %1, %1, %3, %1" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]) : "s10", "s11");
            if (KIND == 26) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_pk_fma_f32 %1, %1, %4, %1" : "+v"(c[i]), "+v"(e[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]), "v"(e[(i + 1) & 15]) : "s10", "s11");
            if (KIND == 27) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(e[i]) : "v"(e[(i + 1) & 15]));
            if (KIND == 23) asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n\tv_add_u32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 15]), "v"(b[i]) : "s10", "s11");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
    for (int i = 0; i < 16; i++) acc += a[i] + b[i] + c[i] + e[i] + (unsigned long long)d[i];
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = acc;
    }
}

typedef void (*kern_t)(unsigned long long*, unsigned);
struct Entry { const char* name; kern_t fn; };

int main() {
    Entry es[] = {
        {"v_mad_u64_u32 (vcc)", k<0>}, {"v_mul_lo_u32", k<1>}, {"v_mul_hi_u32", k<2>}, {"v_mad_u32_u24", k<3>},
        {"v_mul_hi_u32_u24", k<4>}, {"v_add_co_u32", k<5>}, {"v_addc_co_u32", k<6>}, {"v_fma_f64", k<7>},
        {"v_lshl_add_u64", k<8>}, {"v_add_u32", k<9>}, {"mad_u64_u32+addc pair", k<10>}, {"v_mul_u32_u24", k<11>},
        {"v_mad_i32_i24", k<12>}, {"v_mad_u64_u32 (sgpr carry)", k<13>}, {"v_dot4_u32_u8", k<14>},
        {"v_pk_mul_lo_u16", k<15>}, {"v_pk_mad_u16", k<16>}, {"v_fma_f32", k<17>}, {"v_mad_u32_u16", k<18>},
        {"mad + 1 indep add_u32", k<19>}, {"mad + add_u32 + xor", k<20>}, {"mad + lshl_add_u64", k<21>}, {"mad + mul_lo_u32", k<22>}, {"mad + 4 simple", k<23>},
        {"mad + fma_f64", k<24>}, {"mad + fma_f32", k<25>}, {"mad + pk_fma_f32", k<26>}, {"v_pk_fma_f32", k<27>},
    };
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    unsigned long long* dout; CHECK(hipMalloc(&dout, sizeof(unsigned long long) * 2 * 16 * 2 * cus * 4));
    for (auto& e : es) {
        for (int wps : {1, 2, 3, 4}) {           // waves per SIMD
            int threads = 64 * 4 * wps;         // one block per CU fills all 4 SIMDs
            int blocks = cus;
            hipEvent_t ev0, ev1; CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1));
            e.fn<<<blocks, threads>>>(dout, 12345u);   // warmup
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(ev0));
            e.fn<<<blocks, threads>>>(dout, 12345u);
            CHECK(hipEventRecord(ev1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            int nw = blocks * threads / 64;
            std::vector<unsigned long long> h(2 * nw);
            CHECK(hipMemcpy(h.data(), dout, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> cyc(nw);
            for (int i = 0; i < nw; i++) cyc[i] = h[2 * i];
            std::sort(cyc.begin(), cyc.end());
            double med = (double)cyc[nw / 2];
            double ninstr = (double)ITER * 16;
            // s_memtime ticks at 100MHz const clock on some parts; report both tick-based and wall-based
            double per_wave = med / ninstr;
            double wall_cyc_per_instr_per_simd = (ms * 1e-3) * 2.4e9 / (ninstr * wps);
            printf("%-28s wps=%d  memtime/instr/wave=%.3f  => per-SIMD issue interval=%.3f ticks ; wall: %.3f ms => %.3f cyc@2.4GHz per instr per SIMD\n",
                   e.name, wps, per_wave, per_wave / wps, ms, wall_cyc_per_instr_per_simd);
        }
    }
    return 0;
}
