// ubench_duty.hip -- how does the vector issue rate of a wavefront depend on the DUTY CYCLE of the matrix pipe of its SIMD?  One workgroup of 8
// wavefronts per CU (two per SIMD, as k3_sep_ws): waves 0-3 issue bursts of 8 x v_mfma_f32_32x32x16_f16 (256 pipe cycles) separated by
// s_sleep(n) (n x 64 cycles idle), waves 4-7 issue v_pk_fma_f32 on 8 independent registers back to back.  If a vector instruction is only
// slowed WHILE MFMAs execute, ticks per vector instruction = d x 16.2 + (1 - d) x 5.5 with d = the duty cycle; tools/ubench_coissue.hip is d = 1.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_duty ubench_duty.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int SLEEP>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks, int iters, int viters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        f32x16 acc[4];
        for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
        f16x8 a, b;
        for (int q = 0; q < 8; q++) { a[q] = (_Float16)(lane * 0.01f + q); b[q] = (_Float16)(q - lane * 0.02f); }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        }
        float sum = 0.f;
        for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) sum += acc[i][q];
        if (sum == 1234.5f) out[0] = sum;
    } else {
        f32x2 o[8], x = {lane * 0.5f, 1.f}, w = {1.0001f, 0.9999f};
        for (int i = 0; i < 8; i++) o[i] = f32x2{(float)i, 0.f};
        // the vector wavefront runs until the matrix wavefronts are done: a fixed number of instructions sized for the longest setting
        for (int it = 0; it < viters; it++) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x), "v"(w));
        }
        float sum = 0.f;
        for (int i = 0; i < 8; i++) sum += o[i].x + o[i].y;
        if (sum == 1234.5f) out[1] = sum;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}
template <int SLEEP> void run(float *out, unsigned long long *tk) {
    // the vector wavefronts issue 24 x vi instructions (at most ~20 ticks each) INSIDE the matrix wavefronts' run, which is sized to last longer
    const int vi = 1000;
    const int mi = (int)(24.0 * vi * 25.0 / (256.0 + 64.0 * SLEEP)) + 64;
    unsigned long long h[16];
    for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(tk, 0, 128);
        hipLaunchKernelGGL((k<SLEEP>), dim3(256), dim3(512), 0, 0, out, tk, mi, vi);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, tk, 128, hipMemcpyDeviceToHost);
    const double period = h[0] / (double)mi, duty = 256.0 / period;
    printf("s_sleep %2d: burst period %7.1f ticks (matrix pipe duty %.2f) | %6.2f ticks per v_pk_fma_f32 of the vector wavefront (d x 16.2 + (1 - d) x 5.5 = %5.2f)\n", SLEEP, period, duty,
           h[4] / (24.0 * vi), duty * 16.2 + (1 - duty) * 5.5);
}
int main() {
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 128);
    run<0>(out, tk); run<1>(out, tk); run<2>(out, tk); run<4>(out, tk); run<8>(out, tk); run<16>(out, tk); run<32>(out, tk);
    return 0;
}
