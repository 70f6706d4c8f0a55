// ubench_mfmasize.hip -- does the vector issue rate of a wavefront beside a DENSE MFMA stream of its SIMD's sibling depend on the MFMA's SIZE?
// tools/ubench_duty.hip: beside back-to-back v_mfma_f32_32x32x16_f16 (8 passes, 32 ticks each) the sibling issues one v_pk_fma_f32 per ~31 ticks,
// i.e. one per MFMA.  Here the same with v_mfma_f32_16x16x32_f16 (4 passes, 16 ticks, half the flops each: the same flops per tick).
// Also: does s_setprio on the vector wavefront change its share?
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfmasize ubench_mfmasize.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int PRIO>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks, int iters, int viters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        f16x8 a, b;
        for (int q = 0; q < 8; q++) { a[q] = (_Float16)(lane * 0.01f + q); b[q] = (_Float16)(q - lane * 0.02f); }
        float sum = 0.f;
        if (KIND == 0) {
            f32x16 acc[4];
            for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) sum += acc[i][q];
        } else {
            f32x4 acc[8];
            for (int i = 0; i < 8; i++) for (int q = 0; q < 4; q++) acc[i][q] = 0.f;
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            for (int i = 0; i < 8; i++) for (int q = 0; q < 4; q++) sum += acc[i][q];
        }
        if (sum == 1234.5f) out[0] = sum;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        f32x2 o[8], x = {lane * 0.5f, 1.f}, w = {1.0001f, 0.9999f};
        for (int i = 0; i < 8; i++) o[i] = f32x2{(float)i, 0.f};
        for (int it = 0; it < viters; it++)
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x), "v"(w));
        float sum = 0.f;
        for (int i = 0; i < 8; i++) sum += o[i].x + o[i].y;
        if (sum == 1234.5f) out[1] = sum;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}
template <int KIND, int PRIO> void run(const char *name, int per_iter, float *out, unsigned long long *tk) {
    const int vi = 1000, mi = 24 * vi * 40 / 256 + 64;        // the matrix wavefronts outlast the vector wavefronts' 24 x vi instructions
    unsigned long long h[16];
    for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(tk, 0, 128);
        hipLaunchKernelGGL((k<KIND, PRIO>), dim3(256), dim3(512), 0, 0, out, tk, mi, vi);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, tk, 128, hipMemcpyDeviceToHost);
    printf("%-28s %6.1f ticks per MFMA | sibling wavefront: %6.2f ticks per v_pk_fma_f32\n", name, h[0] / (double)mi / per_iter, h[4] / (24.0 * vi));
}
int main() {
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 128);
    run<0, 0>("v_mfma_f32_32x32x16_f16", 8, out, tk);
    run<1, 0>("v_mfma_f32_16x16x32_f16", 16, out, tk);
    run<0, 1>("32x32x16, sibling s_setprio 1", 8, out, tk);
    run<0, 3>("32x32x16, sibling s_setprio 3", 8, out, tk);
    run<1, 3>("16x16x32, sibling s_setprio 3", 16, out, tk);
    return 0;
}
