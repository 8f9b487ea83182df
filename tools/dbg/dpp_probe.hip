// what do DPP forms of VOP2 integer instructions return on this GPU?   hipcc --offload-arch=gfx950 dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int x = 100 + lane, y = 1000 * lane, zero = 0, a, b, c, d, e;
    asm volatile("s_nop 4\n\t"
                 "v_subrev_u32_dpp %0, %5, %7 quad_perm:[3,3,1,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_sub_u32_dpp %1, %5, %6 quad_perm:[2,2,0,0] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32_dpp %2, %5, %6 quad_perm:[2,2,0,0] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %4, %5 quad_perm:[3,3,1,1] row_mask:0xf bank_mask:0xf\n\t"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(x), "v"(y), "v"(zero));
    out[lane * 5 + 0] = a; out[lane * 5 + 1] = b; out[lane * 5 + 2] = c; out[lane * 5 + 3] = d; out[lane * 5 + 4] = e;
}
int main() {
    int* d; int h[64 * 5];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int l = 0; l < 8; l++) printf("lane %d: subrev_dpp[3,3,1,1](x,0)=%d  sub_dpp[2,2,0,0](x,y)=%d  add_dpp=%d  mov_dpp[1,0,3,2]=%d mov_dpp[3,3,1,1]=%d   (x = 100 + lane, y = 1000 lane)\n", l, h[l*5], h[l*5+1], h[l*5+2], h[l*5+3], h[l*5+4]);
    return 0;
}
