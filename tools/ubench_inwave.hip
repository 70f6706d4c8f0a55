// ubench_inwave.hip -- vector instructions in the shadow of the SAME wavefront's MFMAs.  tools/ubench_coissue.hip showed that a vector
// wavefront BESIDE a matrix wavefront on a SIMD is throttled to one v_pk_fma_f32 per 16 cycles (5.5 alone).  Here ONE wavefront per SIMD
// issues  { v_mfma_f32_32x32x16_f16 ; K independent vector instructions }  repeatedly: cycles per group for K = 0 .. 10, i.e. how many
// vector instructions ride for free in an MFMA's 32-cycle slot when they come from the wavefront that issued it.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_inwave ubench_inwave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND> __device__ __forceinline__ void vop(f32x2 &o, float &s, f32x2 x, f32x2 w) {
    if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o) : "v"(x), "v"(w));
    if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s) : "v"(x.x), "v"(w.x));
    if (KIND == 2) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(s));
    if (KIND == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(o) : "v"((int)(threadIdx.x * 8)) : "memory");
}
template <int KIND, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, unsigned long long *ticks, int iters) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    f16x8 a, b;
    for (int q = 0; q < 8; q++) { a[q] = (_Float16)(lane * 0.01f + q); b[q] = (_Float16)(q - lane * 0.02f); }
    f32x2 o[10], x = {lane * 0.5f, 1.f}, w = {1.0001f, 0.9999f};
    float s[10];
    for (int i = 0; i < 10; i++) { o[i] = f32x2{(float)i, 0.f}; s[i] = (float)i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < K; v++) vop<KIND>(o[v], s[v], x, w);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = lds[lane];
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) sum += acc[i][q];
    for (int i = 0; i < 10; i++) sum += o[i].x + o[i].y + s[i];
    if (sum == 1234.5f) out[0] = sum;
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) ticks[0] = t1 - t0;
}
template <int KIND, int K, int WAVES> double run1(float *out, unsigned long long *tk) {
    const int iters = 1000;
    unsigned long long h = 0;
    hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, tk, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    return h / (8.0 * iters);
}
template <int KIND, int WAVES> void run(const char *name, float *out, unsigned long long *tk) {
    printf("%-14s %d wave(s)/SIMD, cycles per {MFMA + K x %s}: K=0 %.1f | 1 %.1f | 2 %.1f | 3 %.1f | 4 %.1f | 5 %.1f | 6 %.1f | 8 %.1f | 10 %.1f\n", name, WAVES / 4, name,
           run1<KIND, 0, WAVES>(out, tk), run1<KIND, 1, WAVES>(out, tk), run1<KIND, 2, WAVES>(out, tk), run1<KIND, 3, WAVES>(out, tk), run1<KIND, 4, WAVES>(out, tk),
           run1<KIND, 5, WAVES>(out, tk), run1<KIND, 6, WAVES>(out, tk), run1<KIND, 8, WAVES>(out, tk), run1<KIND, 10, WAVES>(out, tk));
}
int main() {
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 64);
    run<0, 4>("v_pk_fma_f32", out, tk); run<1, 4>("v_fma_f32", out, tk); run<2, 4>("v_cvt_f16_f32", out, tk); run<3, 4>("ds_read_b64", out, tk);
    run<0, 8>("v_pk_fma_f32", out, tk); run<1, 8>("v_fma_f32", out, tk);
    return 0;
}
