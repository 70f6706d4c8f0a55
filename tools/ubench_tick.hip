// ubench_tick.hip -- what does s_memtime count on gfx950?  (1) its rate against s_memrealtime (100 MHz) in an idle and in an MFMA-saturated
// kernel; (2) ticks taken by a known number of CYCLES: s_sleep 127 (64 x 127 cycles each) and a chain of dependent v_add_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_tick ubench_tick.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(unsigned long long *out, int mode, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4]; for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    f16x8 a, b; for (int q = 0; q < 8; q++) { a[q] = (_Float16)(lane * 0.01f + q); b[q] = (_Float16)(q - lane * 0.02f); }
    float f = (float)lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) { for (int i = 0; i < iters; i++) asm volatile("s_sleep 127"); }
    if (mode == 1) { for (int i = 0; i < iters; i++) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(f)); }
    if (mode == 2) { for (int i = 0; i < iters; i++) { for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0); } }
    if (mode == 3) { for (int i = 0; i < iters; i++) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15"); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = f; for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) s += acc[i][q];
    if (s == 1234.5f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
int main() {
    unsigned long long *d, h[2]; (void)hipMalloc(&d, 64);
    struct { const char *name; int mode, iters, blocks, threads; double cycles_per_iter; } cfg[] = {
        {"s_sleep 127 (8128 cycles each), 1 wave on the chip", 0, 200, 1, 64, 127 * 64.0},
        {"s_nop 15 x 4 (64 cycles), 1 wave", 3, 20000, 1, 64, 64.0},
        {"dependent v_add_f32 x 4, 1 wave", 1, 20000, 1, 64, 0},
        {"dependent v_add_f32 x 4, every SIMD busy (1024 waves)", 1, 20000, 256, 256, 0},
        {"MFMA 32x32x16 x 4, 1 wave per SIMD on every CU", 2, 20000, 256, 256, 0},
        {"MFMA 32x32x16 x 4, 2 waves per SIMD on every CU", 2, 20000, 256, 512, 0},
        {"MFMA 32x32x16 x 4, 4 waves per SIMD on every CU", 2, 20000, 256, 1024, 0},
        {"MFMA 32x32x16 x 4, 1 wave on the chip", 2, 20000, 1, 64, 0},
    };
    for (auto &c : cfg) {
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), 0, 0, d, c.mode, c.iters);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), 0, 0, d, c.mode, c.iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double ticks = (double)h[0], us = (double)h[1] / 100.0;
        printf("%-58s s_memtime %.0f MHz | %.2f ticks per iteration", c.name, ticks / us, ticks / c.iters);
        if (c.cycles_per_iter > 0) printf(" = %.3f ticks per cycle -> core clock %.0f MHz", ticks / c.iters / c.cycles_per_iter, c.cycles_per_iter * c.iters / us);
        if (c.mode == 2) printf(" | wall %.3f ms = %.2f PFLOP/s f16 dense", ms, (double)c.blocks * (c.threads / 64) * c.iters * 4.0 * 32768.0 / (ms * 1e-3) / 1e15);
        printf("\n");
    }
    return 0;
}
