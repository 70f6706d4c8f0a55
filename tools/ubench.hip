// ubench.hip -- lone-wave / 2-wave issue cost of the instruction kinds in the K2 fill loop (gfx950).
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2000
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int MODE>
__global__ void k(unsigned long long *out, double seed) {
    double a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5;
    float f = (float)seed + threadIdx.x, g = f * 2, h = f * 3, e = f * 5;
    int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; i++) {
        if (MODE == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g));) }
        if (MODE == 1) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (MODE == 2) { REP4(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (MODE == 3) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
        if (MODE == 4) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f)); asm volatile("" :: "v"(a));) }
        if (MODE == 5) { REP16(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a)); asm volatile("" :: "v"(f));) }
        if (MODE == 6) { REP4(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f), "+v"(g), "+v"(h), "+v"(e) : "v"((float)seed));) }
        if (MODE == 7) { REP16(asm volatile("s_nop 0\n v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(g));) }
        if (MODE == 8) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(g) : "vcc");) }
        if (MODE == 9) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (MODE == 10) {  // alternating f64 / f32 independent
            REP4(asm volatile("v_add_f64 %0, %0, %2\n v_add_f32 %1, %1, %3\n v_add_f64 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a), "+v"(f) : "v"(b), "v"(g));) }
        if (MODE == 11) { // readlane -> scalar compare -> branch chain
            REP16({ int s = __builtin_amdgcn_readlane(__float_as_int(f), 5); if (s == 12345 + i) f += 1.0f; asm volatile("" : "+v"(f)); }) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a + b + c + d + f + g + h + e == 12345.678) out[100] = 1;
}
template <int MODE>
void run(const char *name, int threads, int blocks, int per) {
    unsigned long long *d, h[256];
    hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.5);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.5);
    hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    printf("%-44s threads/block %4d: %.2f cycles/instr (memtime ticks)\n", name, threads, (double)h[0] / (N * (double)per));
    hipFree(d);
}
int main() {
    int cfg[3] = {64, 256, 512};
    for (int c = 0; c < 3; c++) {
        int t = cfg[c];
        run<0>("v_add_f32 dependent", t, 1, 16);
        run<6>("v_add_f32 4 independent chains", t, 1, 16);
        run<1>("v_add_f64 dependent", t, 1, 16);
        run<2>("v_add_f64 4 independent chains", t, 1, 16);
        run<9>("v_mul_f64 dependent", t, 1, 16);
        run<3>("v_fma_f64 dependent", t, 1, 16);
        run<4>("v_cvt_f64_f32 independent", t, 1, 16);
        run<5>("v_cvt_f32_f64 independent", t, 1, 16);
        run<7>("s_nop+v_mov_dpp wave_shl", t, 1, 16);
        run<8>("v_cndmask dependent", t, 1, 16);
        run<10>("f64/f32 alternating", t, 1, 16);
        run<11>("readlane+scmp+branch", t, 1, 16);
        printf("\n");
    }
    return 0;
}
