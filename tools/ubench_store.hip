// ubench_store.hip -- how should a 32 x 32 fp32 MFMA accumulator tile leave for HBM?  Rows of 256 channels (1 KB pitch), one wavefront per tile.
//   A: the layout conv_epilogue has today -- a lane owns ONE column and 16 rows: 16 x buffer/global store b32, each instruction two 128-byte row segments
//   B: operands swapped in the MFMA (weights as A, activations as B) -- a lane owns ONE row and 16 channels (weight rows permuted so that they are
//      consecutive): 4 x store b128, each instruction 32 rows x 2 pieces of 16 bytes
//   C: as B but without the permutation: a lane's 4-channel groups lie 32 bytes apart (channels 8 g + 4 h + k)
// Chip-wide GB/s for each (every tile written once, 1.2 M rows x 256 channels = 1.2 GB per pass).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_store ubench_store.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *Y, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup = 128 rows x 128 columns (4 wavefronts: 64 rows x 64 columns each = 2 x 2 tiles), like k3_conv_split
    const int m0 = (blockIdx.x >> 1) * 128 + (wave >> 1) * 64, n0 = (blockIdx.x & 1) * 128 + (wave & 1) * 64;
    if (m0 >= rows) return;
    float v[16];
    for (int q = 0; q < 16; q++) v[q] = (float)(lane + q);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), col = n0 + j * 32 + (lane & 31);
                    Y[(size_t)row * 256 + col] = v[q];
                }
            } else {
                const int row = m0 + i * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = n0 + j * 32 + (MODE == 1 ? 16 * h + 4 * g : 8 * g + 4 * h);
                    *reinterpret_cast<f32x4 *>(&Y[(size_t)row * 256 + col]) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                }
            }
        }
}
template <int MODE> void run(const char *name, float *Y, int rows) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)(rows / 128 * 2);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, Y, rows);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, Y, rows);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-58s %.3f ms  %.0f GB/s\n", name, best, (double)rows * 1024.0 / best / 1e6);
}
int main() {
    const int rows = 1200128;
    float *Y; (void)hipMalloc(&Y, (size_t)rows * 1024);
    run<0>("A  16 x store b32, lane = column (today)", Y, rows);
    run<1>("B  4 x store b128, lane = row, 16 consecutive channels", Y, rows);
    run<2>("C  4 x store b128, lane = row, groups 32 bytes apart", Y, rows);
    run<0>("A  again", Y, rows);
    return 0;
}
