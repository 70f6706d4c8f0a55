// ubench_glds.hip -- semantics check of the LDS-DMA load used by k3_conv_dma: __builtin_amdgcn_global_load_lds(gptr, lds, 16, 0, 0) writes
// lane i's 16 bytes to lds_base + 16 i (wave-uniform base); the swizzle of the weight tile is therefore applied to the per-lane SOURCE address.
// A [128 rows][32 halfs] tile is brought in by 8 instructions (16 rows each) with chunk' = chunk ^ ((row >> 2) & 3) and read back as MFMA
// fragments (lane: row = lane & 31 (+ 32 j), chunk = 2 k16 + (lane >> 5)); every value is checked.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t *W, unsigned *bad) {
    __shared__ __attribute__((aligned(16))) uint16_t Bs[128 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 8 pieces of 16 rows; wave w takes pieces w, w + 4
    for (int pc = wave; pc < 8; pc += 4) {
        const int row = pc * 16 + (lane >> 2), pch = lane & 3, ch = pch ^ ((row >> 2) & 3);
        const uint16_t *src = W + row * 32 + ch * 8;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(Bs + pc * 16 * 32), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned wrong = 0;
    for (int j = 0; j < 4; j++)
        for (int k16 = 0; k16 < 2; k16++) {
            const int row = j * 32 + (lane & 31), ch = 2 * k16 + (lane >> 5);
            const u32x4 v = *reinterpret_cast<const u32x4 *>(&Bs[row * 32 + ((ch ^ ((row >> 2) & 3)) * 8)]);
            const u32x4 w = *reinterpret_cast<const u32x4 *>(W + row * 32 + ch * 8);
            for (int q = 0; q < 4; q++) wrong += v[q] != w[q];
        }
    if (wrong) atomicAdd(bad, wrong);
}
int main() {
    uint16_t h[128 * 32]; for (int i = 0; i < 128 * 32; i++) h[i] = (uint16_t)(i * 7 + 3);
    uint16_t *d; unsigned *bad, hb = 0;
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, 0, d, bad);
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA swizzled tile: %u wrong words (%s)\n", hb, hipGetErrorString(hipGetLastError()));
    return 0;
}
