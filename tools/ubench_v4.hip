// ubench_v4.hip -- the premise of a one-wavefront-per-SIMD k3_block64, tested on the shape of its inner step before anything is built:
// a "phase" = one layer's multiply for a 32-row chunk (8 ds_read_b128 of A fragments, 24 dependent-pair MFMAs) + another layer's post-processing
// (~270 vector instructions: plain + packed, 32 ds_write_b32).  Three arrangements, one barrier per step everywhere:
//   A  4 wavefronts per CU (one per SIMD), TWO phases per step each, vector instructions INTERLEAVED into the wavefront's own MFMA slots (6 per MFMA, rest behind)
//   B  the same, NOT interleaved (24 MFMAs, then the vector block): what a wavefront does today
//   C  8 wavefronts per CU (two per SIMD), ONE phase per step each, not interleaved: today's k3_block64
// Same work per CU and step in all three.  Prints shader-clock ticks per step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_v4 tools/ubench_v4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(d, x, y) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y))
#define VMAX(d, x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(x))
#define VCVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VPK(d, x, y) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y))

template <int MODE>   // 0 = A, 1 = B, 2 = C
__global__ __launch_bounds__(MODE == 2 ? 512 : 256) void k(float *out, unsigned long long *ticks, int steps) {
    __shared__ __attribute__((aligned(16))) unsigned lds[40000];                 // 160 000 B: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40000; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    f32x16 acc[2][2];
    for (int p = 0; p < 2; p++) for (int j = 0; j < 2; j++) for (int q = 0; q < 16; q++) acc[p][j][q] = 0.f;
    u32x4 bw[16];
    for (int i = 0; i < 16; i++) bw[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    float v[24]; f32x2 pk[8]; unsigned h[8];
    for (int i = 0; i < 24; i++) v[i] = 1.0f + i * 0.001f + lane * 1e-4f;
    for (int i = 0; i < 8; i++) { pk[i] = f32x2{1.0f + i, 2.0f}; h[i] = 0; }
    const float c1 = 0.999f, c2 = 1e-3f; const f32x2 cp = {0.999f, 1.001f};
    const unsigned *ab = lds + wave * 4096 + (lane & 31) * 10 * 4 + (lane >> 5) * 4;      // this wavefront's "planes"
    unsigned *wb = lds + 32768 / 4 * 0 + 20000 + wave * 2048 + lane;
    auto vector_block = [&](int from, int to) {                                   // instructions [from, to) of the ~270 of a post-processing
        for (int i = from; i < to; i++) {
            const int r = i % 24;
            if (i % 9 == 4 && i / 9 < 30) VPK(pk[(i / 9) % 8], cp, cp);           // 30 packed ones
            else if (i % 5 == 0) VMAX(v[r], c1);
            else if (i % 5 == 1) VCVT(h[r % 8], v[r], v[(r + 1) % 24]);
            else VFMA(v[r], c1, c2);
        }
    };
    auto phase = [&](int p, bool interleave) {
        u32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = *reinterpret_cast<const u32x4 *>(ab + i * 320 + p * 2560);
        if (interleave) {
#pragma unroll
            for (int m = 0; m < 24; m++) {
                MFMA(acc[p][m & 1], a[m % 8], bw[m % 16]);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int g = m * 6 + i, r = g % 24;
                    if (g % 5 == 0) VMAX(v[r], c1); else if (g % 5 == 1) VCVT(h[r % 8], v[r], v[(r + 1) % 24]); else VFMA(v[r], c1, c2);
                }
            }
#pragma unroll
            for (int i = 0; i < 126; i++) {
                const int r = i % 24;
                if (i % 4 == 1 && i / 4 < 30) VPK(pk[(i / 4) % 8], cp, cp); else if (i % 5 == 0) VMAX(v[r], c1); else VFMA(v[r], c1, c2);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 24; m++) MFMA(acc[p][m & 1], a[m % 8], bw[m % 16]);
#pragma unroll
            for (int blk = 0; blk < 9; blk++) {            // (one loop of 270 is left rolled by the compiler: register indexing through s_set_gpr_idx)
#pragma unroll
                for (int j = 0; j < 30; j++) {
                    const int i = blk * 30 + j, r = i % 24;
                    if (i % 9 == 4 && i / 9 < 30) VPK(pk[(i / 9) % 8], cp, cp); else if (i % 5 == 0) VMAX(v[r], c1); else if (i % 5 == 1) VCVT(h[r % 8], v[r], v[(r + 1) % 24]); else VFMA(v[r], c1, c2);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 32; i++) wb[i * 64 + p * 8192 / 4 * 0] = h[i % 8] + i;
    };
    unsigned long long t0 = 0;
    for (int s = 0; s < steps; s++) {
        if (s == 8) t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 2) phase(0, false);
        else { phase(0, MODE == 0); phase(1, MODE == 0); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int p = 0; p < 2; p++) for (int j = 0; j < 2; j++) for (int q = 0; q < 16; q++) sum += acc[p][j][q];
    for (int i = 0; i < 24; i++) sum += v[i];
    for (int i = 0; i < 8; i++) sum += pk[i][0] + pk[i][1] + (float)h[i];
    if (sum == 1234.5f) out[0] = sum;
    if (tid == 0 && blockIdx.x == 100) ticks[0] = (t1 - t0) / (unsigned long long)(steps - 8);
}
int main() {
    float *out; unsigned long long *tk, h = 0;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 64);
    const char *names[3] = {"A: 4 wavefronts, two phases each, vector work in the wavefront's own MFMA slots", "B: 4 wavefronts, two phases each, not interleaved", "C: 8 wavefronts, one phase each (today's shape)"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, tk, 160);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, tk, 160);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, tk, 160);
            (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
            printf("%-90s %6llu ticks per step   (%.0f us for 160 steps)\n", names[mode], h, ms * 1e3);
        }
    return 0;
}
