// k3_cnn.hip -- K3: the BrdU/EdU residual CNN (runCNN, detect.cpp:577-675, TensorFlow graph in the reference) on gfx950.
//
// A data-driven executor: the model is a list of ops (dn_cnn_op, include/dnascent_hip.h) over a few activation
// buffers + one fp32 weight blob (dnascent_amd/cnn_model.py explains why the topology is data).  Activations of the
// whole batch live in HBM as [row][channel] fp32, one row per aligned position (r.refCoordToAP entry), reads laid end to end
// with CNN_PAD all-zero rows between them so that "same"-padded convolutions need no per-tap boundary test: a tap that
// leaves a read lands on a zero row; every epilogue re-zeroes the padding rows (valid[] mask).
//
//   k3_encode         one thread per position: two stacked GRUs (16 units, Keras reset_after, gate order z r h) over the 20
//                     raw samples (zero samples masked, reads.h:161), + one-hot base digits of the core / residual k-mer
//                     index -> 64 channels.  Weights are staged in LDS and read by broadcast.
//   k3_conv<BN>       Conv1D [k, cin, cout] as an implicit GEMM on the fp32 MATRIX cores: M = rows (128 per workgroup),
//                     N = cout (BN = 64 or 128 per workgroup), K = k * cin walked in 32-deep steps (one tap per step since cin
//                     is a multiple of 32); v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD), A/B staged through LDS
//                     k-major so fragment reads are conflict-free; epilogue = folded BatchNorm + bias (scale, shift),
//                     ReLU, padding-row mask.  fp32 because 1e-4 absolute on the probabilities rules out bf16.
//   k3_dwconv         depthwise part of SeparableConv1D: per-channel k-tap FIR, float4 per thread (HBM-bound).
//   k3_add_relu       residual join.   k3_dense_softmax   TimeDistributed Dense(3) + softmax -> class probabilities.
#include "dn_dev.h"
#include "dnascent_hip.h"

#define CNN_PAD 8            // zero rows between reads (>= (17 - 1) / 2, the widest kernel)
#define CNN_BM 128

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CnnRows {             // per batch
    const unsigned *row_off; // [n_reads] first activation row of each read
    const uint8_t *valid;    // [rows] 1 for a real position, 0 for padding
    unsigned rows;           // padded to a multiple of CNN_BM
    unsigned r0, r1;         // reads [r0, r1) are resident in this pass (row_off is only defined for them)
};

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(64) void k3_encode(BatchDev B, const float *core, const float *resid, const float *sig, CnnRows R,
                                                uint8_t *valid_out, float *out, const float *wts, dn_cnn_op op) {
    __shared__ float W[2 * (48 + 768 + 96) + 768];       // g1: K[1][48] R[16][48] b[2][48]; g2: K[16][48] R[16][48] b[2][48]
    float *K1 = W, *R1 = W + 48, *b1 = W + 48 + 768, *K2 = W + 912, *R2 = W + 912 + 768, *b2 = W + 912 + 1536;
    for (int i = threadIdx.x; i < 48; i += 64) K1[i] = wts[op.aux[0] + i];
    for (int i = threadIdx.x; i < 768; i += 64) { R1[i] = wts[op.aux[1] + i]; K2[i] = wts[op.aux[3] + i]; R2[i] = wts[op.aux[4] + i]; }
    for (int i = threadIdx.x; i < 96; i += 64) { b1[i] = wts[op.aux[2] + i]; b2[i] = wts[op.aux[5] + i]; }
    __syncthreads();
    const int r = R.r0 + blockIdx.y;
    const unsigned p = blockIdx.x * 64 + threadIdx.x;
    if (B.res[r].status != 0 || p >= B.res[r].n_positions) return;
    const uint64_t src = B.ref_off[r] + p;
    const unsigned row = R.row_off[r] + p;
    float h1[16], h2[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { h1[u] = 0.f; h2[u] = 0.f; }
    for (int t = 0; t < DN_RAWDEPTH_DEV; t++) {
        const float x = sig[src * DN_RAWDEPTH_DEV + t];
        if (x == 0.0f) continue;                          // masked time step: both layers keep their state
        float n1[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float hz = b1[48 + u], hr = b1[48 + 16 + u], hh = b1[48 + 32 + u];
#pragma unroll
            for (int j = 0; j < 16; j++) { hz += h1[j] * R1[j * 48 + u]; hr += h1[j] * R1[j * 48 + 16 + u]; hh += h1[j] * R1[j * 48 + 32 + u]; }
            const float z = sigmoidf_(x * K1[u] + b1[u] + hz);
            const float rr = sigmoidf_(x * K1[16 + u] + b1[16 + u] + hr);
            const float c = tanhf(x * K1[32 + u] + b1[32 + u] + rr * hh);
            n1[u] = z * h1[u] + (1.0f - z) * c;
        }
        float n2[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float xz = b2[u], xr = b2[16 + u], xh = b2[32 + u];
            float hz = b2[48 + u], hr = b2[48 + 16 + u], hh = b2[48 + 32 + u];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                xz += n1[j] * K2[j * 48 + u]; xr += n1[j] * K2[j * 48 + 16 + u]; xh += n1[j] * K2[j * 48 + 32 + u];
                hz += h2[j] * R2[j * 48 + u]; hr += h2[j] * R2[j * 48 + 16 + u]; hh += h2[j] * R2[j * 48 + 32 + u];
            }
            const float z = sigmoidf_(xz + hz);
            const float rr = sigmoidf_(xr + hr);
            const float c = tanhf(xh + rr * hh);
            n2[u] = z * h2[u] + (1.0f - z) * c;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) { h1[u] = n1[u]; h2[u] = n2[u]; }
    }
    float *o = out + (size_t)row * 64;
#pragma unroll
    for (int u = 0; u < 16; u++) o[u] = h2[u];
    const unsigned ci = (unsigned)core[src] - 1u, ri = (unsigned)resid[src] - 1u;     // reads.h:112-138 indices are 1-based
#pragma unroll
    for (int j = 0; j < 5; j++) { const unsigned d = (ci >> (2 * (4 - j))) & 3u;
#pragma unroll
        for (int q = 0; q < 4; q++) o[16 + j * 4 + q] = (d == (unsigned)q) ? 1.0f : 0.0f; }
#pragma unroll
    for (int j = 0; j < 4; j++) { const unsigned d = (ri >> (2 * (3 - j))) & 3u;
#pragma unroll
        for (int q = 0; q < 4; q++) o[36 + j * 4 + q] = (d == (unsigned)q) ? 1.0f : 0.0f; }
#pragma unroll
    for (int q = 52; q < 64; q++) o[q] = 0.0f;
    valid_out[row] = 1;
}

// ---------------------------------------------------------------------------------------------------------
// implicit-GEMM Conv1D on the fp32 matrix cores
// ---------------------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256) void k3_conv(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wt,
                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                               const uint8_t *__restrict__ valid, int rows, int k, int cin, int cout, int relu) {
    __shared__ float As[32][CNN_BM + 4];                  // [k][m]
    __shared__ float Bs[32][BN + 4];                      // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;              // 2 x 2 wavefronts, each 64 rows x BN/2 columns
    const int m0 = blockIdx.x * CNN_BM, n0 = blockIdx.y * BN;
    constexpr int NJ = BN / 64;                           // 32-wide column tiles per wavefront
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    const int half = (k - 1) / 2;
    const int la_m = tid >> 3, la_k = (tid & 7) * 4;      // A loader: 32 rows x 32 k per pass
    for (int tap = 0; tap < k; tap++) {
        for (int c0 = 0; c0 < cin; c0 += 32) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int m = p * 32 + la_m;
                const int src = m0 + m + tap - half;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src >= 0 && src < rows) v = *reinterpret_cast<const float4 *>(X + (size_t)src * cin + c0 + la_k);
                As[la_k + 0][m] = v.x; As[la_k + 1][m] = v.y; As[la_k + 2][m] = v.z; As[la_k + 3][m] = v.w;
            }
            {
                constexpr int TPR = BN / 4;               // threads per B row
                constexpr int RPP = 256 / TPR;            // rows per pass
#pragma unroll
                for (int p = 0; p < 32 / RPP; p++) {
                    const int kk = p * RPP + tid / TPR, nq = (tid % TPR) * 4;
                    const float4 v = *reinterpret_cast<const float4 *>(Wt + ((size_t)(tap * cin + c0 + kk)) * cout + n0 + nq);
                    *reinterpret_cast<float4 *>(&Bs[kk][nq]) = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 2) {
                const int kr = k0 + (lane >> 5);
                float a[2], b[NJ];
#pragma unroll
                for (int i = 0; i < 2; i++) a[i] = As[kr][wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NJ; j++) b[j] = Bs[kr][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                float y = acc[i][j][q] * sc + sh;
                if (relu) y = fmaxf(y, 0.0f);
                if (!valid[row]) y = 0.0f;
                Y[(size_t)row * cout + col] = y;
            }
        }
}

__global__ __launch_bounds__(256) void k3_dwconv(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wt,
                                                 const uint8_t *__restrict__ valid, int rows, int k, int c) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = c / 4;
    const size_t row = idx / c4;
    const int ch = (int)(idx % c4) * 4;
    if (row >= (size_t)rows) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid[row]) {
        const int half = (k - 1) / 2;
        for (int tap = 0; tap < k; tap++) {
            const long src = (long)row + tap - half;
            if (src < 0 || src >= rows) continue;
            const float4 x = *reinterpret_cast<const float4 *>(X + (size_t)src * c + ch);
            const float4 w = *reinterpret_cast<const float4 *>(Wt + (size_t)tap * c + ch);
            acc.x += x.x * w.x; acc.y += x.y * w.y; acc.z += x.z * w.z; acc.w += x.w * w.w;
        }
    }
    *reinterpret_cast<float4 *>(Y + row * c + ch) = acc;
}

__global__ __launch_bounds__(256) void k3_add_relu(const float *__restrict__ A, const float *__restrict__ Bv, float *__restrict__ Y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4 *>(A)[i], b = reinterpret_cast<const float4 *>(Bv)[i];
    reinterpret_cast<float4 *>(Y)[i] = make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}

// Dense(cin -> 3) + softmax per position; writes the probabilities next to the other per-position outputs (at ref_off)
__global__ __launch_bounds__(64) void k3_dense_softmax(BatchDev B, const float *__restrict__ X, const float *__restrict__ Wt,
                                                       const float *__restrict__ bias, CnnRows R, int cin, float *probs) {
    const int r = R.r0 + blockIdx.y;
    const unsigned p = blockIdx.x * 64 + threadIdx.x;
    if (B.res[r].status != 0 || p >= B.res[r].n_positions) return;
    const unsigned row = R.row_off[r] + p;
    const float *x = X + (size_t)row * cin;
    float z0 = bias[0], z1 = bias[1], z2 = bias[2];
    for (int c = 0; c < cin; c++) { const float v = x[c]; z0 += v * Wt[c * 3 + 0]; z1 += v * Wt[c * 3 + 1]; z2 += v * Wt[c * 3 + 2]; }
    const float m = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
    const float s = e0 + e1 + e2;
    float *o = probs + (B.ref_off[r] + p) * 3;
    o[0] = e0 / s; o[1] = e1 / s; o[2] = e2 / s;
}

// ---------------------------------------------------------------------------------------------------------
// host-side walker over the op list
// ---------------------------------------------------------------------------------------------------------
struct CnnRun {
    const dn_cnn_op *ops; int n_ops;
    const float *wts;                 // device
    float *buf[8]; int n_buf;         // device activation buffers, rows x 256 floats each
    CnnRows rows; uint8_t *valid;
    const float *core, *resid, *sig; float *probs;
    unsigned max_pos;
};

int k3_run(const BatchDev &B, const CnnRun &c, hipStream_t st) {
    const unsigned rows = c.rows.rows;
    for (int i = 0; i < c.n_ops; i++) {
        const dn_cnn_op &o = c.ops[i];
        switch (o.op) {
            case DN_CNN_ENCODE_GRU:
                hipMemsetAsync(c.buf[o.dst], 0, (size_t)rows * 64 * sizeof(float), st);
                hipLaunchKernelGGL(k3_encode, dim3((c.max_pos + 63) / 64, c.rows.r1 - c.rows.r0), dim3(64), 0, st, B, c.core, c.resid, c.sig, c.rows,
                                   c.valid, c.buf[o.dst], c.wts, o);
                break;
            case DN_CNN_CONV:
                if (o.cin % 32 || o.cout % 64) return -1;
                if (o.cout % 128 == 0)
                    hipLaunchKernelGGL(k3_conv<128>, dim3(rows / CNN_BM, o.cout / 128), dim3(256), 0, st, c.buf[o.src], c.buf[o.dst],
                                       c.wts + o.w, c.wts + o.scale, c.wts + o.shift, c.valid, (int)rows, o.k, o.cin, o.cout, o.relu);
                else
                    hipLaunchKernelGGL(k3_conv<64>, dim3(rows / CNN_BM, o.cout / 64), dim3(256), 0, st, c.buf[o.src], c.buf[o.dst],
                                       c.wts + o.w, c.wts + o.scale, c.wts + o.shift, c.valid, (int)rows, o.k, o.cin, o.cout, o.relu);
                break;
            case DN_CNN_DWCONV: {
                const size_t n = (size_t)rows * (o.cin / 4);
                hipLaunchKernelGGL(k3_dwconv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c.buf[o.src], c.buf[o.dst], c.wts + o.w,
                                   c.valid, (int)rows, o.k, o.cin);
                break;
            }
            case DN_CNN_ADD_RELU: {
                const size_t n4 = (size_t)rows * o.cin / 4;
                hipLaunchKernelGGL(k3_add_relu, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, c.buf[o.a], c.buf[o.b], c.buf[o.dst], n4);
                break;
            }
            case DN_CNN_DENSE_SOFTMAX:
                if (o.cout != 3) return -1;
                hipLaunchKernelGGL(k3_dense_softmax, dim3((c.max_pos + 63) / 64, c.rows.r1 - c.rows.r0), dim3(64), 0, st, B, c.buf[o.src], c.wts + o.w,
                                   c.wts + o.shift, c.rows, o.cin, c.probs);
                break;
            default: return -1;
        }
    }
    return 0;
}
