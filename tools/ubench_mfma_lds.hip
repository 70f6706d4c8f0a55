// ubench_mfma_lds.hip -- what an MFMA costs a SIMD when its A fragment comes from LDS and its B fragment sits in registers (the inner loop of k3_block64's shortcut
// wavefronts and of tools/k3_conv_ws_experiment.h, which both measured ~46 cycles per MFMA per SIMD with two wavefronts on it): wavefronts per SIMD x
// accumulation chains per wavefront x how far ahead of their use the fragment reads are issued.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_mfma_lds tools/ubench_mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WAVES, int CHAINS, int AHEAD, int RANDOM = 0>   // RANDOM: operands with random mantissas / signs / exponents (fp16 values in [-4, 4]) instead of 1.0 everywhere.  AHEAD: groups (1 group = 2 reads + 3 MFMAs) between a read and its use
__global__ __launch_bounds__(64 * WAVES) void k(float *out, unsigned long long *ticks, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto rnd16 = [](unsigned x) -> unsigned { x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; const unsigned m = x & 0x3ff, e = 13 + ((x >> 10) & 3), sg = (x >> 12) & 1; return (sg << 15) | (e << 10) | m; };
    for (int i = tid; i < 32768; i += blockDim.x) lds[i] = RANDOM ? (rnd16(2 * i + 1) << 16 | rnd16(2 * i + 2)) : 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x16 acc[2];
    for (int j = 0; j < 2; j++) for (int q = 0; q < 16; q++) acc[j][q] = 0.f;
    u32x4 bw[8];
    for (int i = 0; i < 8; i++) bw[i] = RANDOM ? u32x4{rnd16(lane * 64 + i * 8 + 1) << 16 | rnd16(lane * 64 + i * 8 + 2), rnd16(lane * 64 + i * 8 + 3) << 16 | rnd16(lane * 64 + i * 8 + 4), rnd16(lane * 64 + i * 8 + 5) << 16 | rnd16(lane * 64 + i * 8 + 6), rnd16(lane * 64 + i * 8 + 7) << 16 | rnd16(lane * 64 + i * 8 + 8)}
                                              : u32x4{0x3c003c00u, 0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u};
    const unsigned *base = lds + wave * 2048 + (lane & 31) * 20 + (lane >> 5) * 4;       // rows of 80 B, 16 B per lane half
    constexpr int G = 12;                                  // groups per iteration
    u32x4 fh[AHEAD + 1], fl[AHEAD + 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < AHEAD; g++) { fh[g] = *reinterpret_cast<const u32x4 *>(base + g * 8); fl[g] = *reinterpret_cast<const u32x4 *>(base + 1024 + g * 8); }
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g + AHEAD < G) { fh[(g + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const u32x4 *>(base + ((g + AHEAD) % 16) * 8); fl[(g + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const u32x4 *>(base + 1024 + ((g + AHEAD) % 16) * 8); }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 h = fh[g % (AHEAD + 1)], l = fl[g % (AHEAD + 1)];
            f32x16 &a = acc[CHAINS == 2 ? (g & 1) : 0];
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, l), __builtin_bit_cast(f16x8, bw[g % 8]), a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, h), __builtin_bit_cast(f16x8, bw[(g + 3) % 8]), a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, h), __builtin_bit_cast(f16x8, bw[g % 8]), a, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int j = 0; j < 2; j++) for (int q = 0; q < 16; q++) sum += acc[j][q];
    if (sum == 1234.5f) out[0] = sum;
    if (lane == 0 && blockIdx.x == 100) ticks[wave] = t1 - t0;
}
template <int WAVES, int CHAINS, int AHEAD, int RANDOM = 0> void run(float *out, unsigned long long *tk) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<WAVES, CHAINS, AHEAD, RANDOM>), dim3(256), dim3(64 * WAVES), 0, 0, out, tk, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; (void)hipMemcpy(h, tk, 64, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < WAVES; w++) mx = h[w] > mx ? h[w] : mx;
    // MFMAs per SIMD = waves on it x 36 per iteration
    const double nm = iters * 36.0 * (WAVES / 4);            // MFMAs per SIMD
    printf("%d wavefront(s) per SIMD, %d chain(s), reads %d group(s) ahead, %s operands: %.1f ticks per MFMA on the SIMD; %.2f ns per MFMA by the host's events = %.2f ticks per ns; %.0f TFLOP/s\n", WAVES / 4, CHAINS, AHEAD,
           RANDOM ? "RANDOM" : "all-ones", (double)mx / nm, ms * 1e6 / nm, (double)mx / (ms * 1e6), nm * 1024.0 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
}
int main() {
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 64);
    run<4, 1, 1>(out, tk); run<4, 1, 2>(out, tk); run<4, 1, 4>(out, tk); run<4, 2, 2>(out, tk);
    run<8, 1, 1>(out, tk); run<8, 1, 2>(out, tk); run<8, 1, 4>(out, tk); run<8, 2, 2>(out, tk); run<8, 2, 4>(out, tk);
    run<4, 1, 1, 1>(out, tk); run<4, 1, 2, 1>(out, tk); run<8, 1, 1, 1>(out, tk); run<8, 1, 2, 1>(out, tk); run<8, 2, 2, 1>(out, tk);
    run<8, 1, 1, 0>(out, tk); run<8, 1, 1, 1>(out, tk);
    return 0;
}
