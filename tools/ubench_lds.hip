// ubench_lds.hip -- lone-wave cost of the scan's serial pattern pieces (cycles per element, s_memtime)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int MODE>
__global__ void k(unsigned long long *out, double seed) {
    __shared__ double2 buf[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) buf[i] = make_double2(seed + i, seed * i);
    __syncthreads();
    double s = seed, q = seed * 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < N / 256; rep++) {
        if (MODE < 10 || lane == 0) {
            for (int g = 0; g < 256; g += 16) {
                double2 v[16];
#pragma unroll
                for (int j = 0; j < 16; j++) { if (MODE % 10 == 2) v[j] = make_double2(seed + j, seed - j); else v[j] = buf[g + j]; }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (MODE % 10 != 1) { s = s + v[j].x; q = q + v[j].y; v[j] = make_double2(s, q); }
                    else asm volatile("" :: "v"(v[j].x), "v"(v[j].y));
                }
                if (MODE % 10 == 0 || MODE % 10 == 3) {
#pragma unroll
                    for (int j = 0; j < 16; j++) buf[g + j] = v[j];
                }
                if (MODE % 10 == 4) { asm volatile("" :: "v"(s), "v"(q)); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    if (s + q == 1.2345) out[1] = 1;
}
template <int MODE> void run(const char *name) {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, 1.5);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, 1.5);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-60s %.1f cycles/element\n", name, (double)h[0] / N);
    hipFree(d);
}
int main() {
    run<0>("all lanes: read + 2 adds + write");
    run<1>("all lanes: read only");
    run<2>("all lanes: 2 adds only (register inputs)");
    run<4>("all lanes: read + 2 adds (no write)");
    run<10>("lane 0 : read + 2 adds + write");
    run<11>("lane 0 : read only");
    run<12>("lane 0 : 2 adds only");
    run<14>("lane 0 : read + 2 adds (no write)");
    return 0;
}
