// ubench_slice_read.hip -- does HBM care in which ORDER a workgroup reads its [144 rows][C] fp32 activation tile?
// k3_sep_ws reads the tile one 32-channel block at a time: 128-byte pieces 4 * C bytes apart, 8 passes over the same rows with
// compute in between.  Mode 0 reads the same bytes in that order, mode 1 in address order (whole rows); DELAY inserts idle time
// between the passes (the real kernel computes there).  Same bytes, same number of loads, same tile -> the difference is the
// memory system's.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_slice_read ubench_slice_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int C, int MODE>
__global__ __launch_bounds__(256) void k_read(const float4 *x, float *out, int rows, int delay) {
    constexpr int Q = C / 4, NB = C / 32;                // float4 per row, 32-channel blocks
    const long r0 = (long)blockIdx.x * 128 - 8;
    float4 acc = {0, 0, 0, 0};
    for (int cb = 0; cb < NB; cb++) {
        for (int f = threadIdx.x; f < 144 * 8; f += 256) {
            long row; int q;
            if (MODE == 0) { row = r0 + f / 8; q = cb * 8 + f % 8; }                          // a 128-byte piece of every row
            else { const int g = cb * 144 * 8 + f; row = r0 + g / Q; q = g % Q; }             // the same amount, in address order
            if (row >= 0 && row < rows) { const float4 v = x[row * Q + q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        for (int d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(16);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}
template <int C> void run(const float4 *x, float *out, int rows) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int delay : {0, 8, 32})
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9f;
            for (int it = 0; it < 4; it++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL((k_read<C, 0>), dim3(rows / 128), dim3(256), 0, 0, x, out, rows, delay);
                else hipLaunchKernelGGL((k_read<C, 1>), dim3(rows / 128), dim3(256), 0, 0, x, out, rows, delay);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("C %3d delay %2d  %s  %.3f ms  %.0f GB/s of tile bytes (144/128 halo included)\n", C, delay, mode == 0 ? "32-channel slices" : "address order    ",
                   best, (double)rows / 128 * 144 * C * 4 / best * 1e-6);
        }
}
int main() {
    const int rows = 1200128;
    float4 *x; float *out; hipMalloc(&x, (size_t)rows * 256 * 4); hipMalloc(&out, 64);
    hipMemset(x, 0, (size_t)rows * 256 * 4);
    run<256>(x, out, rows); run<128>(x, out, rows); run<64>(x, out, rows);
    return 0;
}
