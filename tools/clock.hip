// clock.hip -- shader clock under a 1000 x 1-wave load: s_memtime (core cycles) vs s_memrealtime (100 MHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *out, int iters, double seed) {
    double a = seed + threadIdx.x, b = seed;
    float f = (float)seed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f32 %2, %2, %2\n v_add_f64 %0, %0, %1\n v_add_f32 %2, %2, %2\n"
                     "v_add_f64 %0, %0, %1\n v_add_f32 %2, %2, %2\n v_add_f64 %0, %0, %1\n v_add_f32 %2, %2, %2" : "+v"(a), "+v"(b), "+v"(f));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a + f == 1.2345) out[0] = 0;
}
int main() {
    unsigned long long *d; static unsigned long long h[8192];
    hipMalloc(&d, sizeof h);
    int cfgs[4][2] = {{1, 64}, {1000, 64}, {2000, 64}, {1000, 128}};
    for (int c = 0; c < 4; c++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(cfgs[c][0]), dim3(cfgs[c][1]), 0, 0, d, 400000, 1.5);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, sizeof(unsigned long long) * 2 * cfgs[c][0], hipMemcpyDeviceToHost);
            double cyc = (double)h[0], rt = (double)h[1];
            printf("blocks %4d x %3d thr: %.2f ms wall, memtime %.0f ticks, realtime %.0f ticks (100MHz => %.2f ms), clock = %.3f GHz, %.2f cycles/instr\n",
                   cfgs[c][0], cfgs[c][1], ms, cyc, rt, rt / 1e5, cyc / rt * 0.1, cyc / (400000.0 * 8));
        }
    }
    return 0;
}
