// ubench_coissue.hip -- do the matrix pipe and the vector pipe of a gfx950 SIMD run side by side when the instructions come from
// DIFFERENT wavefronts?  One workgroup of 8 wavefronts per CU (two per SIMD, as k3_sep_ws): waves 0-3 issue v_mfma_f32_32x32x16_f16
// on 8 independent accumulators, waves 4-7 issue one kind of vector instruction on 8 independent registers (asm volatile: exactly
// the instruction named).  Shader-clock ticks per instruction for each role alone and for both together.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_coissue ubench_coissue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND> __device__ __forceinline__ void vec24(f32x2 (&o)[8], float (&s)[8], f32x2 x, f32x2 w) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x), "v"(w));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x.x), "v"(w.x));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(o[i]) : "v"(w));
            if (KIND == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(s[i]));
            if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(x.x), "v"(w.x));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(o[i]) : "v"(w));
        }
}
template <int KIND, int NV>
__global__ __launch_bounds__(256 + 256 * NV) void k(float *out, unsigned long long *ticks, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    if ((mode == 0 && !mf) || (mode == 1 && mf)) return;
    __syncthreads();   // (only the waves that stay take part: the others have exited)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        f32x16 acc[4];                                   // 64 registers: fits beside three vector wavefronts per SIMD without spilling
        for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
        f16x8 a, b;
        for (int q = 0; q < 8; q++) { a[q] = (_Float16)(lane * 0.01f + q); b[q] = (_Float16)(q - lane * 0.02f); }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float sum = 0.f;
        for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) sum += acc[i][q];
        if (sum == 1234.5f) out[0] = sum;
    } else {
        f32x2 o[8], x = {lane * 0.5f, 1.f}, w = {1.0001f, 0.9999f};
        float s[8];
        for (int i = 0; i < 8; i++) { o[i] = f32x2{(float)i, 0.f}; s[i] = (float)i; }
        for (int it = 0; it < iters; it++) vec24<KIND>(o, s, x, w);
        float sum = 0.f;
        for (int i = 0; i < 8; i++) sum += o[i].x + o[i].y + s[i];
        if (sum == 1234.5f) out[1] = sum;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}
template <int KIND> void run(const char *name, float *out, unsigned long long *tk) {
    const int iters = 2000;
    unsigned long long h[16];
    for (int nv = 1; nv <= 3; nv++) {                      // vector wavefronts per SIMD beside ONE matrix wavefront per SIMD
        double r[3][2];
        for (int mode = 0; mode < 3; mode++) {
            (void)hipMemset(tk, 0, 128);
            if (nv == 1) hipLaunchKernelGGL((k<KIND, 1>), dim3(256), dim3(512), 0, 0, out, tk, iters, mode);
            if (nv == 2) hipLaunchKernelGGL((k<KIND, 2>), dim3(256), dim3(768), 0, 0, out, tk, iters, mode);
            if (nv == 3) hipLaunchKernelGGL((k<KIND, 3>), dim3(256), dim3(1024), 0, 0, out, tk, iters, mode);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, tk, 128, hipMemcpyDeviceToHost);
            r[mode][0] = h[0] / (8.0 * iters); r[mode][1] = h[4] / (24.0 * iters);
        }
        printf("%-14s %d vector wave(s)/SIMD: alone %6.2f ticks/instr/wave (aggregate %5.2f) | beside MFMA %6.2f (aggregate %5.2f) | MFMA alone %.1f, beside %.1f ticks\n", name, nv,
               r[1][1], r[1][1] / nv, r[2][1], r[2][1] / nv, r[0][0], r[2][0]);
    }
}
int main() {
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 128);
    run<0>("v_pk_fma_f32", out, tk); run<1>("v_fma_f32", out, tk); run<3>("v_cvt_f16_f32", out, tk);
    return 0;
}
