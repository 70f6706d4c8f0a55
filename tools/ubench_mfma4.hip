// ubench_mfma4.hip -- can the depthwise (per-channel) filter of a SeparableConv1D run on the matrix cores?
// v_mfma_f32_4x4x4_16B_f16 multiplies 16 independent 4x4 blocks: block = channel, A = a 4x4 Toeplitz slice of the channel's taps,
// B = 4 row groups x 4 consecutive rows of the channel's input, D = 4 rows x 4 row groups of its output.
//   (1) operand layout check against a CPU product        (2) issue cost alone / in one wave with 32x32x16 / beside other waves' 32x32x16
//   (3) a 17-tap depthwise filter as 5 Toeplitz products x 3 fp16 piece products against the fp32 fmaf chain
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma4 ubench_mfma4.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k_layout(const _Float16 *A, const _Float16 *B, float *D) {     // A[b][i][k], B[b][k][j] -> D[b][i][j]
    const int lane = threadIdx.x, b = lane >> 2, q = lane & 3;
    f16x4 a, bb;
    for (int k = 0; k < 4; k++) { a[k] = A[(b * 4 + q) * 4 + k]; bb[k] = B[(b * 4 + k) * 4 + q]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(b * 4 + i) * 4 + q] = c[i];
}

// mode 0: 4x4x4 alone (waves 0-3) | 1: 32x32x16 alone | 2: waves 0-3 32x32x16, waves 4-7 4x4x4 | 3: one wave alternating 8 + 8
__global__ __launch_bounds__(512) void k_tput(float *out, unsigned long long *ticks, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool first = wave < 4;
    if ((mode == 0 || mode == 1 || mode == 3) && !first) return;
    __syncthreads();
    f32x16 big[4]; f32x4 sm[8];
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) big[i][q] = 0.f;
    for (int i = 0; i < 8; i++) sm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8; f16x4 a4, b4;
    for (int q = 0; q < 8; q++) { a8[q] = (_Float16)(lane * 0.01f + q); b8[q] = (_Float16)(q - lane * 0.02f); }
    for (int q = 0; q < 4; q++) { a4[q] = a8[q]; b4[q] = b8[q]; }
    const bool do_big = mode == 1 || (mode == 2 && first) || mode == 3;
    const bool do_small = mode == 0 || (mode == 2 && !first) || mode == 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_big) {
#pragma unroll
            for (int i = 0; i < 4; i++) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, big[i], 0, 0, 0);
        }
        if (do_small) {
#pragma unroll
            for (int i = 0; i < 8; i++) sm[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, sm[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) s += big[i][q];
    for (int i = 0; i < 8; i++) for (int q = 0; q < 4; q++) s += sm[i][q];
    if (s == 1234.5f) out[0] = s;
    if (lane == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

// (3) depthwise: X [rows + 16][16 channels] fp32, W [17][16]; one wavefront computes 16 output rows x 16 channels per chain.
// lane = (channel c = lane >> 2, row group j = lane & 3).  out[4 j + i] = sum_t w[t] x[4 j + i + t]; with i + t = 4 m + k:
// A_m[i][k] = w[4 m + k - i] (0 outside 0..16), B_m[k][j] = x[4 (j + m) + k], m = 0..4.  x = xh + xl, w = wh + wl (fp16 pieces):
// three products per m, smallest first.
__global__ void k_dw(const float *X, const float *W, float *Y, int rows) {
    const int lane = threadIdx.x, c = lane >> 2, q = lane & 3;
    f16x4 ah[5], al[5];
    for (int m = 0; m < 5; m++)
        for (int k = 0; k < 4; k++) {
            const int t = 4 * m + k - q;                    // A: lane (c, i = q), element k
            const float w = (t >= 0 && t < 17) ? W[t * 16 + c] : 0.0f;
            const _Float16 h = (_Float16)w;
            ah[m][k] = h; al[m][k] = (_Float16)(w - (float)h);
        }
    for (int r0 = 0; r0 < rows; r0 += 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < 5; m++) {
            f16x4 xh, xl;
            for (int k = 0; k < 4; k++) {                   // B: lane (c, j = q), element k
                const float x = X[(size_t)(r0 + 4 * (q + m) + k) * 16 + c];
                const _Float16 h = (_Float16)x;
                xh[k] = h; xl[k] = (_Float16)(x - (float)h);
            }
            acc = __builtin_amdgcn_mfma_f32_4x4x4f16(al[m], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ah[m], xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ah[m], xh, acc, 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) Y[(size_t)(r0 + 4 * q + i) * 16 + c] = acc[i];      // D: lane (c, j = q), element i
    }
}

int main() {
    // ---- (1) layout ----
    _Float16 hA[256], hB[256]; float hD[256];
    srand(7);
    for (int i = 0; i < 256; i++) { hA[i] = (_Float16)(float)(rand() % 9 - 4); hB[i] = (_Float16)(float)(rand() % 9 - 4); }
    _Float16 *dA, *dB; float *dD;
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 16; b++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float s = 0.f;
        for (int k = 0; k < 4; k++) s += (float)hA[(b * 4 + i) * 4 + k] * (float)hB[(b * 4 + k) * 4 + j];
        if (s != hD[(b * 4 + i) * 4 + j]) bad++;
    }
    printf("layout: A lane (b, i) elem k, B lane (b, j) elem k, D lane (b, j) elem i: %d of 256 wrong\n", bad);
    // ---- (2) issue cost ----
    float *out; unsigned long long *tk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&tk, 64);
    const int iters = 2000;
    unsigned long long h[8];
    const char *names[4] = {"4x4x4 alone", "32x32x16 alone", "32x32x16 (w0-3) beside 4x4x4 (w4-7)", "one wave: 4 x 32x32x16 + 8 x 4x4x4 per iteration"};
    for (int mode = 0; mode < 4; mode++) {
        (void)hipMemset(tk, 0, 64);
        hipLaunchKernelGGL(k_tput, dim3(256), dim3(512), 0, 0, out, tk, iters, mode);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, tk, 64, hipMemcpyDeviceToHost);
        if (mode == 0) printf("%-50s %.2f ticks per 4x4x4\n", names[mode], h[0] / (8.0 * iters));
        if (mode == 1) printf("%-50s %.2f ticks per 32x32x16\n", names[mode], h[0] / (4.0 * iters));
        if (mode == 2) printf("%-50s %.2f ticks per 32x32x16, %.2f per 4x4x4\n", names[mode], h[0] / (4.0 * iters), h[4] / (8.0 * iters));
        if (mode == 3) printf("%-50s %.2f ticks per iteration (4 x big alone + 8 x small alone would add up)\n", names[mode], (double)h[0] / iters);
    }
    // ---- (3) depthwise ----
    const int rows = 64;
    float *hX = (float *)malloc((rows + 16) * 16 * 4), *hW = (float *)malloc(17 * 16 * 4), *hY = (float *)malloc(rows * 16 * 4);
    for (int i = 0; i < (rows + 16) * 16; i++) hX[i] = (float)rand() / RAND_MAX * 4.f - 1.f;
    for (int i = 0; i < 17 * 16; i++) hW[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.6f;
    float *dX, *dW, *dY;
    (void)hipMalloc(&dX, (rows + 16) * 64); (void)hipMalloc(&dW, 17 * 64); (void)hipMalloc(&dY, rows * 64);
    (void)hipMemcpy(dX, hX, (rows + 16) * 64, hipMemcpyHostToDevice); (void)hipMemcpy(dW, hW, 17 * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dw, dim3(1), dim3(64), 0, 0, dX, dW, dY, rows);
    (void)hipMemcpy(hY, dY, rows * 64, hipMemcpyDeviceToHost);
    double emax = 0., ref_max = 0.;
    for (int r = 0; r < rows; r++) for (int c = 0; c < 16; c++) {
        float s = 0.f; double d = 0.;
        for (int t = 0; t < 17; t++) { s = fmaf(hX[(r + t) * 16 + c], hW[t * 16 + c], s); d += (double)hX[(r + t) * 16 + c] * hW[t * 16 + c]; }
        emax = fmax(emax, fabs((double)hY[r * 16 + c] - d)); ref_max = fmax(ref_max, fabs((double)s - d));
    }
    printf("depthwise 17 taps on 4x4x4 MFMA (3 fp16 piece products): max |err| vs exact %.3e; the fp32 fmaf chain: %.3e\n", emax, ref_max);
    return 0;
}
