// ubench_sstore.hip -- does gfx950 execute scalar stores (s_store_dwordx4 + s_dcache_wb), and at what rate?
// Every wavefront writes ROWS rows of 32 bytes through the scalar unit (two s_store_dwordx4 per row) the way a k2_fill6
// that kept its trace rows in SGPRs would; the host checks every byte.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_sstore ubench_sstore.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_sstore(unsigned *out, int rows) {
    const unsigned w = blockIdx.x;
    unsigned long long p = (unsigned long long)(out + (size_t)w * rows * 8);
    for (int r = 0; r < rows; r++) {
        const unsigned b = w * 1000003u + (unsigned)r * 8u;
        u4 a = {b, b + 1, b + 2, b + 3}, c = {b + 4, b + 5, b + 6, b + 7};
        asm volatile("s_store_dwordx4 %0, %2, 0x0\n\ts_store_dwordx4 %1, %2, 0x10" :: "s"(a), "s"(c), "s"(p) : "memory");
        p += 32;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
__global__ __launch_bounds__(64) void k_vstore(unsigned *out, int rows) {
    const unsigned w = blockIdx.x, lane = threadIdx.x;
    unsigned *p = out + (size_t)w * rows * 8;
    for (int r = 0; r + 8 <= rows; r += 8) {
        const unsigned rr = r + (lane >> 3);
        p[(size_t)r * 8 + lane] = w * 1000003u + rr * 8u + (lane & 7);
    }
}
int main() {
    const int waves = 4096, rows = 4096;
    unsigned *d; hipMalloc(&d, (size_t)waves * rows * 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(d, 0, (size_t)waves * rows * 32);
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k_sstore, dim3(waves), dim3(64), 0, 0, d, rows);
            else hipLaunchKernelGGL(k_vstore, dim3(waves), dim3(64), 0, 0, d, rows);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> h((size_t)waves * rows * 8);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t w = 0; w < (size_t)waves; w++) for (size_t r = 0; r < (size_t)rows; r++) for (int j = 0; j < 8; j++)
            bad += h[(w * rows + r) * 8 + j] != (unsigned)(w * 1000003u + r * 8u + j);
        printf("%s: %s, %.3f ms, %.1f GB/s, mismatches %zu of %zu dwords\n", mode == 0 ? "scalar stores (s_store_dwordx4 x2 per 32-B row)" : "vector stores (256 B per 8 rows)",
               hipGetErrorString(e), best, (double)waves * rows * 32 / best / 1e6, bad, h.size());
    }
    return 0;
}
