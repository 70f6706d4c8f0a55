// ubench_simd.hip -- which SIMD does wavefront w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8] ...)
// k3_sep_ws assigns its roles by wavefront index; whether a producer shares its SIMD with a consumer depends on this placement.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_simd ubench_simd.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if (lane == 0) out[blockIdx.x * 8 + wave] = id;
}
int main() {
    unsigned *d, h[64 * 8];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; b++) {
        printf("workgroup %2d: SIMD of wavefronts 0..7 =", b);
        for (int w = 0; w < 8; w++) printf(" %u", (h[b * 8 + w] >> 4) & 3u);
        printf("   (cu %u, xcc/se bits %x)\n", (h[b * 8] >> 8) & 15u, h[b * 8] >> 12);
    }
    return 0;
}
