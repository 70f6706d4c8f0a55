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
    unsigned r0, r1;         // sequences [r0, r1) are resident in this pass (row_off is only defined for them)
    const unsigned *n_pos;   // [n_seq] positions of each sequence (0 = nothing to do: failed read)
    const uint64_t *io_off;  // [n_seq] first position of each sequence in the input tensors / the probability output
};

// Row layout of one pass, computed ON THE DEVICE from the live position counts: sequence r owns n_pos[r] rows + CNN_PAD zero rows, the
// sequences of the pass lie end to end behind CNN_PAD leading rows, and the live row count (rounded up to 256) goes to *live.  The host
// only knows the BOUND of every count (it sizes the buffers and the grids from it); workgroups beyond the live rows return at once.
// One wavefront; a pass holds at most a few thousand sequences.
__global__ __launch_bounds__(64) void k3_layout(unsigned *row_off, const unsigned *n_pos, unsigned r0, unsigned r1, int *live) {
    const int lane = threadIdx.x;
    unsigned rows = CNN_PAD;
    for (unsigned b = r0; b < r1; b += 64) {
        const unsigned r = b + lane;
        const unsigned mine = r < r1 ? n_pos[r] + CNN_PAD : 0u;
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (r < r1) row_off[r] = rows + incl - mine;
        rows += __shfl(incl, 63);
    }
    if (lane == 0) *live = (int)((rows + 255u) / 256u * 256u);
}

// ---------------------------------------------------------------------------------------------------------
// v_exp_f32 / v_rcp_f32 are 1-ulp instructions: |error| of the gates ~2e-7, far inside the 1e-4 bar on the probabilities
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

// GRU weights are wave-uniform: read them through the constant address space so they arrive as SGPR operands of the FMAs
// (s_load_dwordx16 rows) instead of occupying ~2.5k vector registers per lane.
typedef const float __attribute__((address_space(4))) *cfptr_t;

// h[16] (x) W[16][48] accumulated into z r g (16 each).  The 768 weights arrive as SGPR operands of the FMAs; left to the compiler
// all 48 s_load_dwordx16 of a product were hoisted to its top, which needs 768 live SGPRs of the ~100 there are: the loop body held
// 878 v_readlane + 876 v_writelane + 660 s_mov of SGPR spill traffic beside its 2 400 arithmetic instructions.  Here the weights
// come in chunks of 32 (two s_load_dwordx16 in asm), double-buffered: the chunk after next is requested before the 32 FMAs of the
// current one, and waited for (SMEM returns out of order: lgkmcnt(0)) after them -- 64 SGPRs live, nothing spilled.
typedef float sf32x16 __attribute__((ext_vector_type(16)));
struct WChunk { sf32x16 a, b; };
template <int CHUNK> __device__ __forceinline__ void gru_wload(WChunk &w, cfptr_t W) {
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=&s"(w.a), "=&s"(w.b) : "s"(W), "n"(CHUNK * 128), "n"(CHUNK * 128 + 64));
}
__device__ __forceinline__ void gru_wwait(WChunk &w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w.a), "+s"(w.b)); }
typedef float gf32x2 __attribute__((ext_vector_type(2)));
// the 32 weights of a chunk are 16 adjacent (output v, v + 1) pairs of ONE input j or of two consecutive ones (48 outputs per input):
// v_pk_fma_f32 with the SGPR pair as one operand and h[j] broadcast to both halves -- half the vector instructions of v_fmac_f32
template <int CHUNK> __device__ __forceinline__ void gru_wfma(const WChunk &w, const gf32x2 (&hh)[16], gf32x2 (&acc)[24]) {
#pragma unroll
    for (int e = 0; e < 32; e += 2) {
        const int idx = CHUNK * 32 + e, j = idx / 48, v = idx % 48;          // weights W[j][v], W[j][v + 1]: input j, outputs v, v + 1 (z 0-15, r 16-31, g 32-47)
        const gf32x2 wv = e < 16 ? gf32x2{w.a[e], w.a[e + 1]} : gf32x2{w.b[e - 16], w.b[e - 15]};
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[v / 2]) : "s"(wv), "v"(hh[j]));
    }
}
template <int C> struct GruSteps {
    static __device__ __forceinline__ void run(WChunk &cur, WChunk &nxt, cfptr_t W, const gf32x2 (&hh)[16], gf32x2 (&acc)[24]) {
        if (C + 1 < 24) gru_wload<(C + 1 < 24 ? C + 1 : 23)>(nxt, W);
        gru_wfma<C>(cur, hh, acc);
        if (C + 1 < 24) { gru_wwait(nxt); GruSteps<C + 1>::run(nxt, cur, W, hh, acc); }
    }
};
template <> struct GruSteps<24> {
    static __device__ __forceinline__ void run(WChunk &, WChunk &, cfptr_t, const gf32x2 (&)[16], gf32x2 (&)[24]) {}
};
__device__ __forceinline__ void gru_matvec(const float (&h)[16], cfptr_t W, float (&z)[16], float (&r)[16], float (&g)[16]) {
    gf32x2 hh[16], acc[24];
#pragma unroll
    for (int j = 0; j < 16; j++) hh[j] = gf32x2{h[j], h[j]};
#pragma unroll
    for (int q = 0; q < 8; q++) { acc[q] = gf32x2{z[2 * q], z[2 * q + 1]}; acc[8 + q] = gf32x2{r[2 * q], r[2 * q + 1]}; acc[16 + q] = gf32x2{g[2 * q], g[2 * q + 1]}; }
    WChunk w0, w1;
    gru_wload<0>(w0, W);
    gru_wwait(w0);
    GruSteps<0>::run(w0, w1, W, hh, acc);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        z[2 * q] = acc[q][0]; z[2 * q + 1] = acc[q][1]; r[2 * q] = acc[8 + q][0]; r[2 * q + 1] = acc[8 + q][1]; g[2 * q] = acc[16 + q][0]; g[2 * q + 1] = acc[16 + q][1];
    }
}

// Positions are visited in order of DESCENDING signal length (k3_encode_len / k3_encode_perm below): a wavefront runs as many
// time steps as its longest position, and the lengths are spread from 3 to the cap of 20 (mean ~10), so sorted wavefronts run
// half the steps of wavefronts of 64 consecutive positions.  Any order gives the same numbers: positions are independent.
__global__ __launch_bounds__(64) void k3_encode(const float *core, const float *resid, const float *sig, const uint64_t *perm_src,
                                                const unsigned *perm_row, const unsigned *hist, uint8_t *valid_out, float *out,
                                                const float *wts, dn_cnn_op op) {
    // the grid covers the BOUND of the pass's positions (known on the host); how many there really are is the total of the
    // length histogram the counting sort built (wave-uniform loads)
    unsigned n_total = 0;
    for (int b = 0; b < DN_RAWDEPTH_DEV + 1; b++) n_total += hist[b];
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= n_total) return;
    const bool live = i < n_total;
    cfptr_t K1 = (cfptr_t)(wts + op.aux[0]), R1 = (cfptr_t)(wts + op.aux[1]), b1 = (cfptr_t)(wts + op.aux[2]);
    cfptr_t K2 = (cfptr_t)(wts + op.aux[3]), R2 = (cfptr_t)(wts + op.aux[4]), b2 = (cfptr_t)(wts + op.aux[5]);
    const uint64_t src = perm_src[live ? i : 0];
    float h1[16], h2[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { h1[u] = 0.f; h2[u] = 0.f; }
    for (int t = 0; t < DN_RAWDEPTH_DEV; t++) {
        const float x = live ? sig[src * DN_RAWDEPTH_DEV + t] : 0.0f;
        const bool on = x != 0.0f;                        // masked time step: both layers keep their state (reads.h:161)
        if (__ballot(on) == 0) continue;
        float z[16], rr[16], g[16];
        // layer 1: recurrent part (+ recurrent bias), then the 1-wide input part
#pragma unroll
        for (int u = 0; u < 16; u++) { z[u] = b1[48 + u]; rr[u] = b1[64 + u]; g[u] = b1[80 + u]; }
        gru_matvec(h1, R1, z, rr, g);
        float n1[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float zz = sigmoidf_(x * K1[u] + b1[u] + z[u]);
            const float rg = sigmoidf_(x * K1[16 + u] + b1[16 + u] + rr[u]);
            const float c = tanhf_(x * K1[32 + u] + b1[32 + u] + rg * g[u]);
            n1[u] = zz * h1[u] + (1.0f - zz) * c;
        }
        // layer 2: input part into (xz, xr, xh), recurrent part into (hz, hr, hh)
        float xz[16], xr[16], xh[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { xz[u] = b2[u]; xr[u] = b2[16 + u]; xh[u] = b2[32 + u]; z[u] = b2[48 + u]; rr[u] = b2[64 + u]; g[u] = b2[80 + u]; }
        gru_matvec(n1, K2, xz, xr, xh);
        gru_matvec(h2, R2, z, rr, g);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float zz = sigmoidf_(xz[u] + z[u]);
            const float rg = sigmoidf_(xr[u] + rr[u]);
            const float c = tanhf_(xh[u] + rg * g[u]);
            const float n2 = zz * h2[u] + (1.0f - zz) * c;
            h1[u] = on ? n1[u] : h1[u];
            h2[u] = on ? n2 : h2[u];
        }
    }
    if (!live) return;
    const unsigned row = perm_row[i];
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)row * 64);
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = make_float4(h2[q * 4], h2[q * 4 + 1], h2[q * 4 + 2], h2[q * 4 + 3]);
    const unsigned ci = (unsigned)core[src] - 1u, ri = (unsigned)resid[src] - 1u;     // reads.h:112-138 indices are 1-based
#pragma unroll
    for (int j = 0; j < 5; j++) { const unsigned d = (ci >> (2 * (4 - j))) & 3u; o[4 + j] = make_float4(d == 0, d == 1, d == 2, d == 3); }
#pragma unroll
    for (int j = 0; j < 4; j++) { const unsigned d = (ri >> (2 * (3 - j))) & 3u; o[9 + j] = make_float4(d == 0, d == 1, d == 2, d == 3); }
#pragma unroll
    for (int q = 13; q < 16; q++) o[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    valid_out[row] = 1;
}

// ---------------------------------------------------------------------------------------------------------
// k3_encode_mfma (round 3): the same encoder with the GRU's matrix-vector products on the MATRIX cores.  k3_encode spends ~2 200 vector
// instructions per time step on three 16 x 48 products per position (wave-uniform weights as SGPR operands of v_pk_fma_f32); batched
// over the 32 positions of a wavefront they are 32 x 16 x 48 GEMMs -- v_mfma_f32_32x32x16_f16 with K = the 16 units exactly:
//     D'[output (32 rows)][position (32 columns)] += W^T[output][unit] * h^T[unit][position]
//   A operand = W^T: lane (i = lane & 31, hh = lane >> 5) holds the 8 inputs k' = 8 hh .. 8 hh + 7 of output row i
//   B operand = h^T: lane (p = lane & 31, hh) holds 8 units of position p
//   D'              : lane (p, hh), register q: row (q & 3) + 8 (q >> 2) + 4 hh
// With rows 0-15 = one 16-unit block and rows 16-31 = another, lane (p, hh) ends up with exactly the units (e & 3) + 8 (e >> 2) + 4 hh,
// e = 0 .. 7, of both blocks (registers e and 8 + e) -- and if k' = 8 hh + e is DEFINED to be that same unit (the contraction index may be
// permuted freely as long as A and B agree), the lane's results are the lane's next B operand: no shuffle between time steps at all.
// Tiles: layer 1 [z; r] and [g_rec; 0] from R1; layer 2 [z; r] from K2 (input n1) + R2 (input h2), [xh; hh_rec] from [K2g; 0] and [0; R2g].
// fp32 accuracy as in the convolutions: every operand split into two fp16 pieces, three products (dropped < 2^-22); the states are in
// [-1, 1] and the weights O(1), so fp16's range is not an issue here.  18 MFMAs + ~400 vector instructions per step and 32 positions
// (k3_encode: ~1 100 per 32).  Biases and the 1-wide input kernel of layer 1 sit in LDS as per-lane tiles (two variants, by hh).
// Not bit-identical to k3_encode (other summation order): checked against the PyTorch fp32 rendering at 1e-4 like everything in K3;
// the exact-fp32 mode (DN_CNN_MATH_FP32) keeps k3_encode.
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 gh16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void enc_split8(const float (&v)[8], gh16x8 &h, gh16x8 &l) {
#pragma unroll
    for (int e = 0; e < 8; e++) { const _Float16 hh = (_Float16)v[e]; h[e] = hh; l[e] = (_Float16)(v[e] - (float)hh); }
}
__device__ __forceinline__ f32x16 enc_mma3(const gh16x8 ah, const gh16x8 al, const gh16x8 bh, const gh16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);      // smallest terms first, as in the convolutions
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}
// The order by signal length is LOCAL here: a workgroup takes 256 consecutive positions of one sequence, counting-sorts them by descending
// length in LDS and hands 32 of them to each of its 8 wavefronts -- within 256 positions the lengths (3 .. 20, mean ~10) already fill
// every wavefront with one or two adjacent values, and the two global sort kernels of k3_encode (histogram + atomically placed
// permutation: 0.27 ms per 1.2 M positions beside 0.58 for this kernel behind them; 0.65 ms like this) are not launched.
#ifndef ENC_WG
#define ENC_WG 512                                         // threads of a k3_encode_mfma workgroup: 8 wavefronts of 32 positions (two lanes each)
#endif
__global__ __launch_bounds__(ENC_WG, 2) void k3_encode_mfma(const float *core, const float *resid, const float *sig, CnnRows R, uint8_t *valid_out, float *out,
                                                         const float *wts, dn_cnn_op op) {
    // constant tiles (float, [tile][16 registers][lane half]: a lane's values depend on its half only, so the reads are broadcasts):
    // accumulator initial values of the four tiles, layer 1's input kernel (z, r, g) and its candidate gate's input bias
    __shared__ float ctile[6][16][2];
    __shared__ unsigned lh[DN_RAWDEPTH_DEV + 1], lbase[DN_RAWDEPTH_DEV + 1];
    __shared__ unsigned short order[ENC_WG / 2];
    const int r = R.r0 + blockIdx.y;
    const unsigned npos = R.n_pos[r], p0 = blockIdx.x * (ENC_WG / 2);
    if (p0 >= npos) return;                                // workgroup-uniform: the grid covers the bound of the positions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, hh = lane >> 5;
    const uint64_t seq0 = R.io_off[r];
    {   // ---- thread tid < ENC_WG / 2 looks at position p0 + tid: its signal length (index of the last non-zero sample + 1; 0 beyond the sequence) ----
        if (tid < DN_RAWDEPTH_DEV + 1) lh[tid] = 0;
        __syncthreads();
        unsigned len = 0, rank = 0;
        if (tid < ENC_WG / 2) {
            if (p0 + tid < npos) {
                const float4 *x = reinterpret_cast<const float4 *>(sig + (seq0 + p0 + tid) * DN_RAWDEPTH_DEV);   // 80-byte rows: 16-byte aligned
#pragma unroll
                for (int q = 0; q < DN_RAWDEPTH_DEV / 4; q++) {
                    const float4 v = x[q];
                    if (v.x != 0.0f) len = 4 * q + 1;
                    if (v.y != 0.0f) len = 4 * q + 2;
                    if (v.z != 0.0f) len = 4 * q + 3;
                    if (v.w != 0.0f) len = 4 * q + 4;
                }
            }
            rank = atomicAdd(&lh[len], 1u);
        }
        __syncthreads();
        if (tid < DN_RAWDEPTH_DEV + 1) { unsigned longer = 0; for (int b = tid + 1; b < DN_RAWDEPTH_DEV + 1; b++) longer += lh[b]; lbase[tid] = longer; }
        __syncthreads();
        if (tid < ENC_WG / 2) order[lbase[len] + rank] = (unsigned short)tid;   // longest first; positions beyond the sequence (length 0, with the masked ones) last
        __syncthreads();
    }
    const unsigned pp = p0 + order[32 * wave + p];        // the position this lane pair works on
    const bool live = pp < npos;
    const bool idle = __ballot(live) == 0;                 // a wavefront with nothing but positions beyond the sequence: it only joins the barrier below
    const float *K1 = wts + op.aux[0], *R1 = wts + op.aux[1], *b1 = wts + op.aux[2];
    const float *K2 = wts + op.aux[3], *R2 = wts + op.aux[4], *b2 = wts + op.aux[5];
    auto unit = [&](int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; };
    // ---- constant tiles: register q of lane (p, hh) belongs to block q >> 3, unit(q & 7, hh) ----
    if (tid < 32) {                                        // 32 (register, half) slots, one thread each (wavefront 0, idle or not)
        const int q = tid >> 1, hf = tid & 1;
        const int u = unit(q & 7, hf), blk = q >> 3;
        ctile[0][q][hf] = blk == 0 ? b1[u] + b1[48 + u] : b1[16 + u] + b1[64 + u];       // layer 1 [z; r]: input bias + recurrent bias
        ctile[1][q][hf] = blk == 0 ? b1[80 + u] : 0.0f;                                    // layer 1 [g_rec; -]
        ctile[2][q][hf] = blk == 0 ? b2[u] + b2[48 + u] : b2[16 + u] + b2[64 + u];       // layer 2 [z; r]
        ctile[3][q][hf] = blk == 0 ? b2[32 + u] : b2[80 + u];                              // layer 2 [xh; hh_rec]
        ctile[4][q][hf] = blk == 0 ? K1[u] : K1[16 + u];                                   // layer 1 input kernel: z | r
        ctile[5][q][hf] = blk == 0 ? K1[32 + u] : b1[32 + u];                              // ... g | the candidate gate's input bias
    }
    __syncthreads();
    if (idle) return;
    // ---- A fragments: lane = output row i_ = lane & 31 of the tile, k' = 8 hh + kk <-> input unit(kk, hh) ----
    auto afrag = [&](const float *Wlo, int col_lo, const float *Whi, int col_hi, gh16x8 &ah, gh16x8 &al) {
        // rows 0-15 of the tile: column col_lo + row of matrix Wlo ([16][48]); rows 16-31: column col_hi + (row - 16) of Whi; nullptr = zero rows
        float v[8];
        const int r_ = p;
        const float *W = r_ < 16 ? Wlo : Whi;
        const int col = r_ < 16 ? col_lo + r_ : col_hi + r_ - 16;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) v[kk] = W ? W[unit(kk, hh) * 48 + col] : 0.0f;
        enc_split8(v, ah, al);
    };
    gh16x8 aR1a_h, aR1a_l, aR1b_h, aR1b_l, aK2a_h, aK2a_l, aK2b_h, aK2b_l, aR2a_h, aR2a_l, aR2b_h, aR2b_l;
    afrag(R1, 0, R1, 16, aR1a_h, aR1a_l);                 // layer 1 [z; r]
    afrag(R1, 32, nullptr, 0, aR1b_h, aR1b_l);            // layer 1 [g_rec; 0]
    afrag(K2, 0, K2, 16, aK2a_h, aK2a_l);                 // layer 2 [z; r], input part
    afrag(R2, 0, R2, 16, aR2a_h, aR2a_l);                 // ... recurrent part
    afrag(K2, 32, nullptr, 0, aK2b_h, aK2b_l);            // layer 2 [xh; 0] from n1
    afrag(nullptr, 0, R2, 32, aR2b_h, aR2b_l);            // layer 2 [0; hh_rec] from h2
    auto tile = [&](int t) { f32x16 c;
#pragma unroll
        for (int q = 0; q < 16; q++) c[q] = ctile[t][q][hh];
        return c; };
    const uint64_t src = seq0 + (live ? pp : p0);
    float h1[8], h2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { h1[e] = 0.f; h2[e] = 0.f; }
    gh16x8 h1h, h1l, h2h, h2l;
    enc_split8(h1, h1h, h1l); enc_split8(h2, h2h, h2l);
    for (int t = 0; t < DN_RAWDEPTH_DEV; t++) {
        const float x = live ? sig[src * DN_RAWDEPTH_DEV + t] : 0.0f;
        const bool on = x != 0.0f;                        // masked time step: both layers keep their state (reads.h:161)
        if (__ballot(on) == 0) continue;
        // ---- layer 1 ----
        f32x16 zr = enc_mma3(aR1a_h, aR1a_l, h1h, h1l, tile(0));
        f32x16 gr = enc_mma3(aR1b_h, aR1b_l, h1h, h1l, tile(1));
        const f32x16 k1a = tile(4), k1b = tile(5);
        float n1[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float zz = sigmoidf_(__builtin_fmaf(x, k1a[e], zr[e]));
            const float rg = sigmoidf_(__builtin_fmaf(x, k1a[8 + e], zr[8 + e]));
            const float c = tanhf_(__builtin_fmaf(x, k1b[e], k1b[8 + e]) + rg * gr[e]);
            n1[e] = zz * h1[e] + (1.0f - zz) * c;
        }
        gh16x8 n1h, n1l;
        enc_split8(n1, n1h, n1l);
        // ---- layer 2 ----
        zr = enc_mma3(aK2a_h, aK2a_l, n1h, n1l, tile(2));
        zr = enc_mma3(aR2a_h, aR2a_l, h2h, h2l, zr);
        gr = enc_mma3(aK2b_h, aK2b_l, n1h, n1l, tile(3));
        gr = enc_mma3(aR2b_h, aR2b_l, h2h, h2l, gr);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float zz = sigmoidf_(zr[e]);
            const float rg = sigmoidf_(zr[8 + e]);
            const float c = tanhf_(gr[e] + rg * gr[8 + e]);
            const float n2 = zz * h2[e] + (1.0f - zz) * c;
            h1[e] = on ? n1[e] : h1[e];
            h2[e] = on ? n2 : h2[e];
        }
        enc_split8(h1, h1h, h1l); enc_split8(h2, h2h, h2l);
    }
    if (!live) return;
    const unsigned row = R.row_off[r] + pp;
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)row * 64);
    o[hh] = make_float4(h2[0], h2[1], h2[2], h2[3]);              // units 4 hh .. 4 hh + 3
    o[2 + hh] = make_float4(h2[4], h2[5], h2[6], h2[7]);          // units 8 + 4 hh ..
    const unsigned ci = (unsigned)core[src] - 1u, ri = (unsigned)resid[src] - 1u;     // reads.h:112-138 indices are 1-based
    if (hh == 0) {
#pragma unroll
        for (int j = 0; j < 5; j++) { const unsigned d = (ci >> (2 * (4 - j))) & 3u; o[4 + j] = make_float4(d == 0, d == 1, d == 2, d == 3); }
        valid_out[row] = 1;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned d = (ri >> (2 * (3 - j))) & 3u; o[9 + j] = make_float4(d == 0, d == 1, d == 2, d == 3); }
#pragma unroll
        for (int q = 13; q < 16; q++) o[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// signal length of a position = index of its last non-zero sample + 1 (reads.h:147 pads with zeros; a masked step is x == 0)
#define ENC_BINS (DN_RAWDEPTH_DEV + 1)
__global__ __launch_bounds__(256) void k3_encode_len(const float *sig, CnnRows R, uint8_t *len_by_row, unsigned *hist) {
    __shared__ unsigned h[ENC_BINS];
    const int r = R.r0 + blockIdx.y;
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < R.n_pos[r];
    if (threadIdx.x < ENC_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    if (live) {
        const float4 *x = reinterpret_cast<const float4 *>(sig + (R.io_off[r] + p) * DN_RAWDEPTH_DEV);   // 80-byte rows: 16-byte aligned
        unsigned len = 0;
#pragma unroll
        for (int q = 0; q < DN_RAWDEPTH_DEV / 4; q++) {
            const float4 v = x[q];
            if (v.x != 0.0f) len = 4 * q + 1;
            if (v.y != 0.0f) len = 4 * q + 2;
            if (v.z != 0.0f) len = 4 * q + 3;
            if (v.w != 0.0f) len = 4 * q + 4;
        }
        len_by_row[R.row_off[r] + p] = (uint8_t)len;
        atomicAdd(&h[len], 1u);
    }
    __syncthreads();
    if (threadIdx.x < ENC_BINS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// counting sort, longest first: slot = (positions with a longer signal) + rank inside the bin (workgroup-aggregated cursors)
__global__ __launch_bounds__(256) void k3_encode_perm(CnnRows R, const uint8_t *len_by_row, const unsigned *hist, unsigned *cursor,
                                                      uint64_t *perm_src, unsigned *perm_row) {
    __shared__ unsigned h[ENC_BINS], base[ENC_BINS];
    const int r = R.r0 + blockIdx.y;
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < R.n_pos[r];
    if (threadIdx.x < ENC_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    unsigned len = 0, rank = 0, row = 0;
    if (live) { row = R.row_off[r] + p; len = len_by_row[row]; rank = atomicAdd(&h[len], 1u); }
    __syncthreads();
    if (threadIdx.x < ENC_BINS && h[threadIdx.x]) {
        unsigned longer = 0;
        for (int b = threadIdx.x + 1; b < ENC_BINS; b++) longer += hist[b];
        base[threadIdx.x] = longer + atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    }
    __syncthreads();
    if (live) { const unsigned slot = base[len] + rank; perm_src[slot] = R.io_off[r] + p; perm_row[slot] = row; }
}

// ---------------------------------------------------------------------------------------------------------
// implicit-GEMM Conv1D on the fp32 matrix cores
//   workgroup = 256 threads = 2 x 2 wavefronts, tile 128 rows x BN columns, K walked in 32-deep steps (one tap, 32 input
//   channels).  Both operands sit in LDS row-major with a 36-float pitch ([m][k] and [n][k]) so that every fragment is ONE
//   ds_read_b128 per lane: lane (m = lane & 31, h = lane >> 5) reads k = 16 h + 4 j .. 4 j + 3 and feeds MFMA (j, q) with
//   k = 16 h + 4 j + q -- a permutation of the 32 k's that A and B share, so the sum is unchanged.  Weights are
//   re-laid at load time (dn_load_cnn) as [tap][cin / 32][cout][32] so the B tile is a straight float4 copy.
//   The next step's global loads are issued before the current step's MFMAs (register prefetch) and land in the other
//   LDS buffer: one barrier per step.
// ---------------------------------------------------------------------------------------------------------
#define CNN_PITCH 36
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (Round 3: the TRANSPOSED epilogue -- weights as the A operand of every MFMA, so that a lane holds one row and 4 x 4 consecutive channels and
// the residual loads and stores are 16-byte accesses, a quarter of the instructions of this issue-bound epilogue -- was built for all conv
// kernels and is bit-identical, but the network took 19.2 ms against 17.0 per 1.2 M positions in one session (gpurun_out/r3t2; 5-tap
// separable layers +30 %, 17-tap +17 %, 3-tap convolutions +6-23 %): an instruction then writes 32 bytes to each of 32 rows, and four of them
// complete a 128-byte line, where the layout below writes whole 128-byte row segments.  The memory side does care.  Not kept.)
// epilogue shared by the conv kernels: C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) +
// 4 * (lane >> 5).  Row validity of the wavefront's 64 rows is one ballot; the residual values are fetched 16 at a time
// (no load -> wait -> load chains).
// PAIR (K3_EPI_PAIR; needs two column tiles per wavefront): lane n of column tile j holds column 2 n + j instead of 32 j + n (the kernel fetches its B fragments in that
// order: a different address, nothing else), so a lane's two tiles are ADJACENT columns and a row leaves as one 8-byte store per lane (a lane half = 256 contiguous bytes)
// instead of two 4-byte ones -- the epilogues are bound by the number of store instructions they issue (k3_sep_ws's stamps: 8 350 of a tile's 40 000 ticks for 128 stores
// per lane, during which its producers wait at the barrier), and BatchNorm runs packed over the column pair.  Same expression per element: bit-identical.
#ifndef K3_EPI_PAIR
#define K3_EPI_PAIR 1
#endif
template <int BN, bool ADD, bool PAIR = false>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[2][BN / 64], float *__restrict__ Y, const float *__restrict__ scale,
                                              const float *__restrict__ shift, const float *__restrict__ Add,
                                              const uint8_t *__restrict__ valid, int m0, int n0, int wm, int wn, int lane, int cout, int relu,
                                              float post = 1.0f, int wave_rows = 64, int vbyte = -1) {
    // y = max(acc * scale + shift (+ residual), floor), zero on padding rows.  An accumulator register q of row block i holds row
    // 32 i + 8 (q >> 2) + (q & 3) + 4 (lane >> 5): the row is wave-uniform up to the lane half, so per (i, q) the row's base
    // address and its padding mask are SCALAR (one SGPR pair each: the select is a v_cndmask on that pair, the address an SGPR
    // base + one lane offset computed once); the NJ column tiles of the row are 128 bytes apart (immediate offsets).  The first
    // version computed a 64-bit address and a shifted mask bit per element in the vector unit: ~10 vector instructions per
    // element, 6.8 k ticks per 128 x 256 tile -- a seventh of the 17-tap layer's time and the whole of the short layers' tail.
    constexpr int NJ = BN / 64;
    // vbyte >= 0: the caller fetched this lane's validity byte (valid[m0 + wm * 64 + lane]) long before -- K3_VB_AHEAD: fetched here, the epilogue of every tile starts with a
    // load it waits for (k3_pair128's stamps showed such a wait at ~2 000 ticks inside a busy kernel)
    const unsigned long long vmask = __ballot((vbyte >= 0 ? vbyte : (int)valid[m0 + wm * 64 + lane]) != 0);
    const bool all_valid = vmask == ~0ull;                 // wave-uniform
    const float floor_ = relu ? 0.0f : -3.402823466e38f;
    if constexpr (PAIR) {
        static_assert(BN == 128, "the column-pair epilogue needs exactly two column tiles per wavefront");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const int colp = n0 + wn * (BN / 2) + 2 * (lane & 31);
        const f32x2 sc2 = {scale[colp] * post, scale[colp + 1] * post}, sh2 = {shift[colp], shift[colp + 1]};
        auto uniform_ptr = [](const void *p) {
            const unsigned long long v = (unsigned long long)p;
            return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
        };
        const size_t wbase = (size_t)(m0 + wm * 64) * cout;
        const int wbytes = max(min(wave_rows, 64), 0) * cout * 4;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Y + wbase), 0, wbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(ADD ? Add + wbase : Y + wbase)), 0, wbytes, 0x00020000);
        const int voff = (4 * (lane >> 5) * cout + colp) * 4;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            f32x2 addv[16];
            if (ADD) {                                      // all residual loads of the row block first: the stores below may not pass them
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int soff = __builtin_amdgcn_readfirstlane((i * 32 + (q & 3) + 8 * (q >> 2)) * cout * 4);
                    addv[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, voff, soff, 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = i * 32 + (q & 3) + 8 * (q >> 2);                   // + 4 for the upper lane half
                const int soff = __builtin_amdgcn_readfirstlane(row * cout * 4);
                f32x2 y = __builtin_elementwise_fma(f32x2{acc[i][0][q], acc[i][1][q]}, sc2, sh2);
                if (ADD) y += addv[q];
                float y0 = fmaxf(y[0], floor_), y1 = fmaxf(y[1], floor_);
                if (!all_valid) {
                    const unsigned long long k0 = (((vmask >> row) & 1ull) ? 0x00000000ffffffffull : 0ull) | (((vmask >> (row + 4)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y0) : "v"(y0), "s"(k0));
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y1) : "v"(y1), "s"(k0));
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{y0, y1}), ry, voff, soff, 0);
            }
        }
        return;
    }
    const int colb = n0 + wn * (BN / 2) + (lane & 31);
    float sc[NJ], sh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { sc[j] = scale[colb + j * 32] * post; sh[j] = shift[colb + j * 32]; }   // post: a power of two (exact), 1 except on the fp16 path
    // buffer addressing: descriptor = the wavefront's 64 rows of Y (and of the residual), voffset = the lane's part (bytes),
    // soffset = the row's part (scalar), immediate = the column tile
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const size_t wbase = (size_t)(m0 + wm * 64) * cout;
    const int wbytes = max(min(wave_rows, 64), 0) * cout * 4;     // wave_rows < 64: the wavefront's last rows are not stored (they fall outside the descriptor)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Y + wbase), 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(ADD ? Add + wbase : Y + wbase)), 0, wbytes, 0x00020000);
    const int voff = (4 * (lane >> 5) * cout + colb) * 4;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float addv[16][NJ];
        if (ADD) {                                          // all residual loads of the row block first: the stores below may not pass them
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int soff = __builtin_amdgcn_readfirstlane((i * 32 + (q & 3) + 8 * (q >> 2)) * cout * 4);
#pragma unroll
                for (int j = 0; j < NJ; j++) addv[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voff + j * 128, soff, 0));
            }
        }
        // rows in pairs (q, q + 1): packed FMA and packed add; the padding select only where the wavefront's rows have padding at all
        // (a tile in 140 on the 20 kb workload)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int row = i * 32 + (q & 3) + 8 * (q >> 2);                   // + 4 for the upper lane half; q + 1: the next row
            const int soff0 = __builtin_amdgcn_readfirstlane(row * cout * 4), soff1 = __builtin_amdgcn_readfirstlane((row + 1) * cout * 4);
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                f32x2 y = __builtin_elementwise_fma(f32x2{acc[i][j][q], acc[i][j][q + 1]}, f32x2{sc[j], sc[j]}, f32x2{sh[j], sh[j]});
                if (ADD) y += f32x2{addv[q][j], addv[q + 1][j]};
                float y0 = fmaxf(y[0], floor_), y1 = fmaxf(y[1], floor_);
                if (!all_valid) {
                    const unsigned long long k0 = (((vmask >> row) & 1ull) ? 0x00000000ffffffffull : 0ull) | (((vmask >> (row + 4)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                    const unsigned long long k1 = (((vmask >> (row + 1)) & 1ull) ? 0x00000000ffffffffull : 0ull) | (((vmask >> (row + 5)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y0) : "v"(y0), "s"(k0));
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y1) : "v"(y1), "s"(k1));
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), ry, voff + j * 128, soff0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), ry, voff + j * 128, soff1, 0);
            }
        }
    }
}

// Workgroup numbering of the conv kernels (1-D grid): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own
// L2.  Id L runs on XCD L % 8; on one XCD consecutive ids walk the COLUMN tiles of one row tile before moving to the next row
// tile, so the second (third, fourth) read of the same activation rows hits that XCD's L2 instead of HBM.
__device__ __forceinline__ bool conv_tile(int cout, int BN, int rows, int &m0, int &n0, int BM = CNN_BM) {
    const int nct = cout / BN;
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    m0 = ((slot / nct) * 8 + xcd) * BM; n0 = (slot % nct) * BN;
    return m0 < rows;
}
static inline unsigned conv_grid(unsigned rows, int cout, int BN, int BM = CNN_BM) { return ((rows / BM + 7) / 8) * 8 * (unsigned)(cout / BN); }

template <int BN, int NBUF, bool ADD>
__global__ __launch_bounds__(256) void k3_conv(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wt,
                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                  const float *__restrict__ Add, const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int k,
                                                  int cin, int cout, int relu) {
    rows = min(rows, *live);                              // the grid covers the BOUND of the pass's rows; the live count is on the device
    __shared__ __attribute__((aligned(16))) float As[NBUF][CNN_BM * CNN_PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * CNN_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;              // each wavefront: 64 rows x BN/2 columns
    int m0, n0;
    if (!conv_tile(cout, BN, rows, m0, n0)) return;
    constexpr int NJ = BN / 64;                           // 32-wide column tiles per wavefront
    constexpr int NB = BN / 32;                           // float4 B loads per thread per step
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    const int half = (k - 1) / 2;
    const int cblocks = cin >> 5;
    const int steps = k * cblocks;
    const int l_r = tid >> 3, l_k = (tid & 7) * 4;        // loader: 32 rows x 8 float4 per pass
    f32x4 ra[4], rb[NB];
    bool pin[4];
    auto gload = [&](int s) {
        const int tap = s / cblocks, c0 = (s - tap * cblocks) << 5;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int src = m0 + p * 32 + l_r + tap - half;
            const bool in = src >= 0 && src < rows;       // clamped address + select: no branch around the load
            ra[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + c0 + l_k);
            pin[p] = in;                                  // applied at the LDS store: nothing may wait on the load before the MFMAs
        }
        const float *wb = Wt + ((size_t)s * cout + n0) * 32;
#pragma unroll
        for (int p = 0; p < NB; p++) rb[p] = *reinterpret_cast<const f32x4 *>(wb + (size_t)(p * 32 + l_r) * 32 + l_k);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; p++) *reinterpret_cast<f32x4 *>(&As[buf][(p * 32 + l_r) * CNN_PITCH + l_k]) = pin[p] ? ra[p] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NB; p++) *reinterpret_cast<f32x4 *>(&Bs[buf][(p * 32 + l_r) * CNN_PITCH + l_k]) = rb[p];
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int fm = lane & 31, fh = (lane >> 5) * 16;
    for (int s = 0; s < steps; s++) {
        const int cur = s & (NBUF - 1), nxt = (s + 1) & (NBUF - 1);
        gload(min(s + 1, steps - 1));                     // the last step reloads itself: unconditional keeps the prefetch in registers
        __builtin_amdgcn_sched_barrier(0);                // keep the prefetch ABOVE the MFMAs (the scheduler otherwise sinks it)
        const float *Ab = &As[cur][(wm * 64 + fm) * CNN_PITCH + fh];
        const float *Bb = &Bs[cur][(wn * (BN / 2) + fm) * CNN_PITCH + fh];
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++) {
            f32x4 a[2], b[NJ];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * CNN_PITCH + j4 * 4);
#pragma unroll
            for (int j = 0; j < NJ; j++) b[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * CNN_PITCH + j4 * 4);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                // ... and everything that consumes it BELOW them
        if (NBUF == 1) __syncthreads();                   // single buffer (narrow layers: more workgroups per CU instead)
        lstore(nxt);
        __syncthreads();
    }
    conv_epilogue<BN, ADD>(acc, Y, scale, shift, Add, valid, m0, n0, wm, wn, lane, cout, relu);
}

// ---------------------------------------------------------------------------------------------------------
// the same convolution on the 16-bit matrix cores at fp32-level accuracy: every fp32 operand is split into 16-bit pieces and
// the products that matter are accumulated in fp32.  Two splits are built (template parameter NP = pieces per operand):
//
//   NP = 3, bf16 ("bf16x6"): x = h + m + l EXACTLY (3 x 8 = the 24 significant bits); six products
//       x w ~= h h' + (h m' + m h') + (h l' + l h' + m m')                 (dropped: m l', l m', l l' < 2^-24 |x w|)
//     v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, so six of them are 2.67x faster than the fp32 instruction
//     (417 vs 157 TFLOP/s peak).  bf16 has the fp32 exponent range: no range caveat at all.
//   NP = 2, fp16 ("f16x3"): x = h + l + O(2^-22 |x|) (2 x 11 bits); three products h h' + h l' + l h' -- HALF the matrix work
//     of bf16x6 (peak 834 TFLOP/s of fp32-equivalent work) and two planes instead of three through LDS.  The price is fp16's
//     exponent range: (a) |x| > 65504 does not convert -- the loader tracks max |x| and raises a device flag, and the host then
//     repeats the pass in bf16x6 (dn_capi.hip: cnn_execute), so the result is never silently wrong; (b) below 2^-3 the low piece
//     is subnormal and the split keeps an ABSOLUTE 2^-25 instead of a relative 2^-22 -- the matrix cores honour fp16 subnormals
//     (tools/ubench_f16_denorm.hip, measured), and 3e-8 absolute on an activation is below fp32 rounding of the O(1) sums it
//     feeds.  Weights are pre-scaled per layer by a power of two into [2^13, 2^14) so their low pieces stay normal; the
//     epilogue multiplies the folded BatchNorm scale by the inverse power (exact).
//
// Weights are split once at dn_load_cnn ([channel block][tap][piece][cout][32]); activations stay fp32 in HBM and are split by
// the loader on their way into LDS (NP planes, pitch 40 elements: one ds_read_b128 per fragment).
// tools/cnn_split_precision.py: probabilities move by 2e-6 for bf16x6 (fp32 MFMA vs CPU: 2e-6).
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CNN_BP 40

__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float x = q < 4 ? lo4[q] : hi4[q - 4];
        const __bf16 hh = (__bf16)x;
        const float r1 = x - (float)hh;                    // exact
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;                   // exact
        h[q] = hh; m[q] = mm; l[q] = (__bf16)r2;
    }
}
// One channel PAIR split into its two fp16 pieces (+ the range report's running maximum), the way every f16x3 kernel below does it.  K3_SPLIT_MIX:
//   0  element by element as first written: per pair v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32 (or 2 x v_sub_f32), v_cvt_pk_f16_f32, and 2-3 v_max_f32
//      for the maximum (fmaxf (amax, fabsf (x)) per element: the compiler quiets a possible NaN of every operand first);
//   2  the maximum as ONE v_max3_f32 (amax, |x0|, |x1|) per pair;
//   1  that, and the rest x - (float) h as ONE v_fma_mix_f32 per element, which reads the fp16 half in place (h x -1.0 + x: exact, the same value).
// Same values in all three (tools/variant_check.py); what differs is the producers' instruction count, which is what the separable layers wait for.
#ifndef K3_VB_AHEAD
#define K3_VB_AHEAD 1                                       /* the persistent separable kernels request a tile's validity bytes at the top of the tile instead of inside its epilogue */
#endif
#ifndef K3_SPLIT_MIX
#define K3_SPLIT_MIX 1
#endif
typedef _Float16 sp16x2 __attribute__((ext_vector_type(2)));
typedef float spf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(const float x0, const float x1, sp16x2 &h, sp16x2 &l, float &amax) {
    if (K3_SPLIT_MIX == 0) {
        amax = fmaxf(amax, fabsf(x0)); amax = fmaxf(amax, fabsf(x1));
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
        h = sp16x2{h0, h1}; l = sp16x2{(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1)};
        return;
    }
    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x0)), __builtin_fabsf(x1));
    h = __builtin_convertvector(spf32x2{x0, x1}, sp16x2);
    if (K3_SPLIT_MIX == 2) { l = __builtin_convertvector(spf32x2{x0, x1} - __builtin_convertvector(h, spf32x2), sp16x2); return; }
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    l = __builtin_convertvector(spf32x2{r0, r1}, sp16x2);
}
__device__ __forceinline__ void split2(const f32x4 lo4, const f32x4 hi4, f16x8 &h, f16x8 &l, float &amax) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const float x0 = q < 4 ? lo4[q] : hi4[q - 4], x1 = q < 4 ? lo4[q + 1] : hi4[q - 3];
        sp16x2 hp, lp;
        split_pair(x0, x1, hp, lp, amax);                  // the difference is exact while |x| <= 65504
        h[q] = hp[0]; h[q + 1] = hp[1]; l[q] = lp[0]; l[q + 1] = lp[1];
    }
}
template <int NP> __device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// What a split-mode kernel reports about the values it split, once per wavefront: rf[0] |= 1 when one of them does not fit fp16 (> 65 504),
// rf[1] = max(rf[1], largest |value|) (non-negative floats order like their bit patterns; the slot is read first, so after the first
// workgroups almost no wavefront issues the atomic).  rf = the op's pair of words in the pass's report block; k3_range_check folds the
// block into the one word the host looks at: bit 0 overflow, bit 1 UNDERFLOW -- a layer whose largest split value is below 2^-6.  Below
// 2^-3 the low fp16 piece is subnormal and the split keeps an absolute 2^-25 instead of a relative 2^-22; that is harmless while the
// layer's values are O(1) (3e-8 against sums of O(1)) and catastrophic when the whole layer is tiny and its weights correspondingly large
// (round 4's precision fuzz, family tiny_act: activations ~1e-5 -> P off by 0.26).  The host then repeats the pass with bf16 pieces
// (fp32's exponent range), exactly as for an overflow.
__device__ __forceinline__ void range_report(float amax, unsigned *rf, int lane) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    if (lane == 0) {
        if (amax > 65504.0f) atomicOr(rf, 1u);
        const unsigned b = __float_as_uint(amax);
        if (b > *(volatile unsigned *)(rf + 1)) atomicMax(rf + 1, b);
    }
}
#define CNN_RANGE_WORDS(n_ops) (2 + 2 * (n_ops))
__global__ __launch_bounds__(64) void k3_range_check(unsigned *R, int n_ops) {
    for (int op = threadIdx.x; op < n_ops; op += 64) {
        const unsigned f = R[2 + 2 * op], a = R[3 + 2 * op];
        if (f) atomicOr(R, 1u);
        if (a != 0u && a < 0x3c800000u) atomicOr(R, 2u);                  // 0 < largest |value| < 2^-6 (an all-zero layer -- no live rows -- says nothing)
        R[2 + 2 * op] = 0u; R[3 + 2 * op] = 0u;                           // clean for the next pass
    }
}

// Loop order: input-channel block outermost, taps inside.  The A tile of a channel block (128 + k - 1 rows) is split and
// staged ONCE and every tap reads it at a row offset, so a k-tap layer converts each activation once instead of k times;
// only the B tile changes per step.  Weights are laid out [channel block][tap][piece][cout][32] to match.
// BM = 256 (8 wavefronts) halves the weight traffic per flop: every step streams a fresh B tile from L2, and with 128-row
// workgroups that stream alone takes ~2/3 of the L2 bandwidth on the long-K layers.  Measured: 17 taps x 128 channels -8 %,
// 9 x 128 -6 %, but the 3-tap layers +8 % (two 512-thread workgroups per CU need <= 128 VGPRs: 44 bytes of scratch), so only
// layers with >= 9 taps and >= 128 input channels take it.
// (Round 2: a ping-pong form -- the workgroup's two halves half a step out of phase, one feeding the matrix pipe while the other does
// the memory half-step -- is in tools/k3_conv_pp_experiment.hip: bit-identical, 17 x 128 -> 256 in 3.64 ms against 3.05.  Not kept.)
// (Round 2: the persistent form that pays for the separable kernels -- a workgroup walking the tile ids of its XCD -- makes this one
// slower: 256-row layers +14 % alone, full pipeline 692 -> 668 Msamples/s, A/B in one session.  Its tiles are long (13 us and more) and
// plentiful; the dispatcher balances them better than a static share.  Not kept.)
// (Round 2: B fragments straight from L2 into registers -- no B tile in LDS, no barrier inside a channel block, half the LDS fragment
// traffic -- measured on every 128-column layer: 17 x 128 -> 256 3.39 ms against 3.10 for the 256-row form below, the others within 2 %:
// neither the barriers nor the LDS pipe is what the long-K layers wait for; halving the weight stream per flop (256 rows) is what pays.)
// (Round 2: a double-buffered B tile with ONE barrier per step -- stores and next loads issued before the step's MFMAs -- was measured A/B
// in one session: 3 x 256 -> 256 -6 %, but the 256-row form drops to one workgroup per CU (84 KB of LDS) and loses 3-6 %, the narrow
// layers 2-8 %; network 20.4 -> 20.6 ms.  With two workgroups per CU the other one fills the gap between the barriers.  Not kept.)
#define CNN_AROWS (CNN_BM + 16)
template <int BN, bool ADD, int NP, int BM = CNN_BM>
__global__ __launch_bounds__(BM * 2, BM == 256 ? 4 : 1) void k3_conv_split(const float *__restrict__ X, float *__restrict__ Y, const uint16_t *__restrict__ Wb,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     const float *__restrict__ Add, const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int k,
                                                     int cin, int cout, int relu, float post, unsigned *range_flag) {
    rows = min(rows, *live);
    __shared__ __attribute__((aligned(16))) uint16_t As[NP][(BM + 16) * CNN_BP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[NP][BN * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    if (!conv_tile(cout, BN, rows, m0, n0, BM)) return;
    // the epilogue's validity byte, requested before the whole tile -- in the 128-row form only: the 256-row form runs under a 128-register cap and spilled the one register
    const int vb_ahead = (K3_VB_AHEAD && BM == 128) ? (int)valid[m0 + ((int)threadIdx.x >> 7) * 64 + ((int)threadIdx.x & 63)] : -1;
    constexpr int NJ = BN / 64;
    constexpr int LR = BM / 2;                            // rows one loader pass covers (4 threads per row)
    constexpr int NBQ = BN / LR;                          // 16-byte B chunks per thread per piece
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    const int half = (k - 1) / 2;
    const int cblocks = cin >> 5;
    const int steps = k * cblocks;
    const int arows = BM + k - 1;
    const int l_r = tid >> 2, l_k = (tid & 3) * 8;        // loader: LR rows x 4 chunks of 8 elements per pass
    // The A tile is BM + 16 rows: two full loader passes (BM = 256: LR = 128) or three of 64 rows (BM = 128: 144 = 2 x 64 + 16) cover all
    // but its last 16 rows.  Those 16 rows x 32 channels are 128 float4: ONE float4 on the first 128 threads (two wavefronts, a wave-uniform
    // branch) instead of a third full pass -- which cost every thread 8 registers and two loads for rows only 16 of its LR rows have, and the
    // 256-row form lives under a 128-register cap (round 3: 13 / 18 registers spilled, one reload in every step: VERDICT r3 item 1a).
    constexpr int NPASS = BM / LR;                         // full loader passes: 2
    f32x4 ra[NPASS][2], rh = {0.f, 0.f, 0.f, 0.f};
    u32x4 rb[NP][NBQ];
    float amax = 0.0f;                                     // NP == 2: largest |activation| this thread has split
    // Buffer addressing, one path for every tile: descriptor = the rows of [0, rows) the tile needs, starting at row
    // max(m0 - half, 0); voffset = the lane's (row, chunk) minus the rows the first tile lacks -- a row before the pass wraps to
    // a huge unsigned offset, a row past its end lies beyond num_records: both read as the zeros 'same' padding wants, without a
    // select; soffset = the channel block / the step.  No 64-bit addresses in vector registers.
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const int row0 = max(m0 - half, 0), lack = row0 - (m0 - half), rows_here = min(rows, m0 + BM + half) - row0;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)row0 * cin)), 0, rows_here * cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<uint16_t *>(Wb)), 0, steps * NP * cout * 64, 0x00020000);
    const int aoff = ((l_r - lack) * cin + l_k) * 4;       // + p * LR rows
    const int boff = (l_r * 32 + l_k) * 2;                 // + q * LR rows of 64 bytes
    const bool halo = tid < 128;                           // wave-uniform: wavefronts 0 and 1 fetch the tile's last 16 rows, thread = (row tid >> 3, float4 tid & 7)
    auto gloadA = [&](int cb) {
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            ra[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * LR * cin * 4, cb << 7, 0));
            ra[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * LR * cin * 4 + 16, cb << 7, 0));
        }
        if (halo) {                                        // offset recomputed here (once per channel block) rather than held across the steps
            int t = (int)threadIdx.x;
            asm volatile("" : "+v"(t));
            rh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, ((BM + (t >> 3) - lack) * cin + (t & 7) * 4) * 4, cb << 7, 0));
        }
    };
    auto gloadB = [&](int s) {
#pragma unroll
        for (int pc = 0; pc < NP; pc++) {
            const int so = ((s * NP + pc) * cout + n0) * 64;
#pragma unroll
            for (int q = 0; q < NBQ; q++) rb[pc][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, boff + q * LR * 64, so, 0));
        }
    };
    auto lstoreA = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            const int o = (p * LR + l_r) * CNN_BP + l_k;
            if (NP == 3) {
                bf16x8 h, m, l;
                split3(ra[p][0], ra[p][1], h, m, l);
                *reinterpret_cast<bf16x8 *>(&As[0][o]) = h; *reinterpret_cast<bf16x8 *>(&As[1][o]) = m; *reinterpret_cast<bf16x8 *>(&As[NP - 1][o]) = l;
            } else {
                f16x8 h, l;
                split2(ra[p][0], ra[p][1], h, l, amax);
                *reinterpret_cast<f16x8 *>(&As[0][o]) = h; *reinterpret_cast<f16x8 *>(&As[1][o]) = l;
            }
        }
        if (halo) {                                        // the same split, four elements wide
            const int o = (BM + (tid >> 3)) * CNN_BP + (tid & 7) * 4;
            if (NP == 3) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 h, m, l;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float x = rh[e];
                    const __bf16 hh = (__bf16)x; const float r1 = x - (float)hh;
                    const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
                    h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
                }
                *reinterpret_cast<bf16x4 *>(&As[0][o]) = h; *reinterpret_cast<bf16x4 *>(&As[1][o]) = m; *reinterpret_cast<bf16x4 *>(&As[NP - 1][o]) = l;
            } else {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    sp16x2 hp, lp;
                    split_pair(rh[e], rh[e + 1], hp, lp, amax);
                    h[e] = hp[0]; h[e + 1] = hp[1]; l[e] = lp[0]; l[e + 1] = lp[1];
                }
                *reinterpret_cast<f16x4 *>(&As[0][o]) = h; *reinterpret_cast<f16x4 *>(&As[1][o]) = l;
            }
        }
    };
    auto lstoreB = [&]() {
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int q = 0; q < NBQ; q++) *reinterpret_cast<u32x4 *>(&Bs[pc][(q * LR + l_r) * CNN_BP + l_k]) = rb[pc][q];
    };
    gloadA(0); gloadB(0);
    lstoreA(); lstoreB();
    __syncthreads();
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    // fragment addresses: ONE per-thread base for each operand; the tap moves the A base by a row pitch per step (one vector add), planes,
    // row blocks and the k16 half are immediate offsets of the ds_read_b128 (the compiler used to rebuild the row index from the wavefront
    // number with a 64-bit multiply-add every step, and kept that number in scratch)
    const uint16_t *aF = &As[0][(wm * 64 + fm) * CNN_BP + fk];
    // column tile j of lane fm = column 2 fm + j (conv_epilogue's PAIR form: 8-byte stores) instead of 32 j + fm: measured SLOWER here (17 x 128 -> 256: 3 021-3 024 us against
    // 2 969-2 974; 9 x 128 -> 128: 828-842 against 816-826; gpurun_out/r6l) -- the epilogue is 2-4 % of these tiles and the fragment reads' new order costs more: -DK3_EPI_PAIR_CONV=1 only
#ifndef K3_EPI_PAIR_CONV
#define K3_EPI_PAIR_CONV 0
#endif
    constexpr bool EPAIR = K3_EPI_PAIR && K3_EPI_PAIR_CONV && BN == 128 && NP == 2;
    const uint16_t *bF = &Bs[0][(wn * (BN / 2) + (EPAIR ? 2 * fm : fm)) * CNN_BP + fk];
    constexpr int APL = (BM + 16) * CNN_BP, BPL = BN * CNN_BP;            // plane strides (elements)
    int tap = 0, cb = 0;
    for (int s = 0; s < steps; s++) {
        const bool lastTap = tap == k - 1;
        gloadB(min(s + 1, steps - 1));
        if (lastTap) gloadA(min(cb + 1, cblocks - 1));      // wave-uniform branch
        __builtin_amdgcn_sched_barrier(0);
        const uint16_t *aT = aF + tap * CNN_BP;
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++) {
            u32x4 a[2][NP], b[NJ][NP];
#pragma unroll
            for (int pc = 0; pc < NP; pc++) {
#pragma unroll
                for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(aT + pc * APL + i * 32 * CNN_BP + k16 * 16);
#pragma unroll
                for (int j = 0; j < NJ; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(bF + pc * BPL + (EPAIR ? j : j * 32) * CNN_BP + k16 * 16);
            }
            // smallest terms first, so the big h h' product meets an accumulator that already holds the corrections
            constexpr int NT = NP == 3 ? 6 : 3;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                constexpr int PA3[6] = {1, 2, 0, 1, 0, 0}, PB3[6] = {1, 0, 2, 0, 1, 0};    // m m', l h', h l', m h', h m', h h'
                constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};                      // l h', h l', h h'
                const int pa = NP == 3 ? PA3[t] : PA2[t % 3], pb = NP == 3 ? PB3[t] : PB2[t % 3];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        acc[i][j] = mfma16<NP>(a[i][pa], b[j][pb], acc[i][j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lstoreB();
        if (lastTap) lstoreA();
        __syncthreads();
        tap = lastTap ? 0 : tap + 1;
        cb += lastTap ? 1 : 0;
    }
    if (NP == 2) range_report(amax, range_flag, lane);     // out of fp16's range (either way): the host repeats the pass in bf16x6
    // the epilogue's thread coordinates are RE-DERIVED from the hardware id behind an opaque move: kept live across the step loop they
    // were three of the 256-row form's spilled registers (it runs under a 128-register cap)
    int tid_e = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    conv_epilogue<BN, ADD, EPAIR>(acc, Y, scale, shift, Add, valid, m0, n0, tid_e >> 7, (tid_e >> 6) & 1, tid_e & 63, cout, relu, post, 64, vb_ahead);
}

// (Round 3: k3_conv_dma -- the weight tile by LDS-DMA (global_load_lds_dwordx4, swizzled through the source address) into two buffers,
// three taps = 72 MFMAs per wavefront and ONE barrier per step, fragments one group ahead, 256 x 128 tiles, one workgroup per CU -- is in
// tools/k3_conv_dma_experiment.hip: bit-identical, and slower on every layer (17 x 128 -> 256: 3.59-3.66 ms against 3.09-3.19; 9 x 128 -> 128:
// 1.13-1.17 against 0.98-1.00; 3 x 256 -> 256: 1.51-1.53 against 1.47-1.49).  The reason is in tools/ubench_tick.hip: under a chip-wide
// MFMA load the clock settles at 1.5-1.8 GHz and the matrix pipe then delivers 1.5-1.9 PFLOP/s of dense fp16, not 2.5; the 17-tap layer's
// 1.30 PFLOP/s of issued products is already 70-85 % of that.  Removing barriers and staging instructions does not buy what a second
// workgroup per CU does)
#if defined(DN_WS_TRACE) || defined(DN_SPLIT_TRACE)       /* experiment builds only (tools/ws_trace.py, tools/split_trace.py): shader-clock stamps of one workgroup's phases */
#ifndef DN_WS_TRACE
#define DN_WS_TRACE DN_SPLIT_TRACE
#endif
__device__ unsigned long long ws_trace[16][64];
#define WS_T(i) do { if (blockIdx.x == DN_WS_TRACE && lane == 0) ws_trace[wave][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define WS_TRACE_TILE 2                                   /* the workgroup's third tile: steady state */
extern "C" int dn_debug_ws_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_trace), sizeof(ws_trace)); }
#else
#define WS_T(i) do { } while (0)
#define WS_TRACE_TILE (-1)
#endif
// ---------------------------------------------------------------------------------------------------------
// k3_sep_split: SeparableConv1D in ONE kernel -- the depthwise filter is applied while the A tile of the pointwise GEMM is
// staged, so its output never goes to HBM (as two kernels the pair moves 4 x rows x C x 4 bytes, fused 2 x; the 29 separable
// layers are memory-side, so that is most of their time).
//   per channel block (32 channels):
//     raw tile   (128 + KW - 1) rows x 32 channels of fp32 input, global -> registers (one step ahead) -> LDS, pitch 40 floats
//     depthwise  thread = 4 consecutive rows x 4 channels: a register window slides over KW + 3 input rows (each read once
//                from LDS), taps accumulate in ascending order with fmaf -- bit-identical to k3_dwconv -- and the four results
//                are split into 16-bit pieces and written to the A planes
//     pointwise  the 1-tap step of k3_conv_split on those planes
//   Two barriers per block: [A planes complete / raw tile free] -> next raw tile + taps to LDS, MFMAs -> [planes and B free] ->
//   next B tile to LDS, next depthwise.  A workgroup does not overlap its own depthwise with its own MFMAs; the 2 workgroups of a CU do.
// (Round 2: buffer-descriptor loaders alone, which made every k3_conv_split layer 3-10 % faster, make this kernel 4-7 % SLOWER (A/B in
// one session, twice); its loads are few and its address arithmetic is not what it waits for.  Not kept.)
// (Round 2: the vector-unit diet that helped k3_sep_ws -- packed FMAs in a fixed interleaved order, packed conversions, buffer
// addressing, selects only on edge tiles -- was applied here too and measured A/B in one session: 5-tap layers unchanged, 9-tap layers
// 3.5 % slower (240 VGPRs instead of 232); with two or three workgroups per CU this kernel is not short of vector issue.  Not kept.)
// (Round 2: taking the B fragments straight from L2 instead of an LDS tile cuts the kernel's LDS from 64 to 43 KB for 128 channels,
// enough for three workgroups per CU -- but the fragments' registers push it to 244 VGPRs (two workgroups again, 3.35 ms against 3.19
// for the eleven 9-tap layers), and capping the registers at 168 spills: 10.8 ms.  Not kept.)
// The raw-tile pitch (160 B) makes 4 rows advance the bank window by half (640 mod 256 = 128): the four 16-lane groups of a
// ds_read_b128 (MI355X_MICROARCH.md, LDS) then cover all 64 banks once.
// ---------------------------------------------------------------------------------------------------------
#define SEP_XP 40                                           // floats per raw-tile row
#define SEP_XPW 36                                          // ... of the wave-specialised kernel's raw slices (see its depthwise)
// (the bf16x6 form of the 128-column tile holds 83 KB of LDS and ~300 registers: ONE workgroup per CU, and it says so -- round 2 asked for
// two and the compiler reported "desired occupancy 2, final 1"; the persistent grid below follows the same number)
template <int BN, int KW, bool ADD, int NP>
__global__ __launch_bounds__(256, (NP == 3 && BN == 128) ? 1 : 2) void k3_sep_split(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                    const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                    const float *__restrict__ shift, const float *__restrict__ Add,
                                                    const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                    unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int XROWS = CNN_BM + KW - 1;
    constexpr int NLD = (XROWS * 8 + 255) / 256;           // float4 loads per thread for one raw tile
    __shared__ __attribute__((aligned(16))) float Xr[XROWS * SEP_XP];
    __shared__ __attribute__((aligned(16))) float Wl[KW * 32];
    __shared__ __attribute__((aligned(16))) uint16_t As[NP][CNN_BM * CNN_BP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[NP][BN * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // PERSISTENT where the layer has one column tile (cout == BN: every layer that takes this kernel by default): the grid is two
    // workgroups per CU and a workgroup takes the row tiles blockIdx.x + k gridDim.x, running their channel blocks as ONE stream of
    // steps.  The raw tile of step g + 2 is requested during step g ACROSS tile boundaries, so only a workgroup's first tile pays the
    // two exposed load latencies of the prologue, and a tile's epilogue runs while the next tile's loads are in flight.  Phase trace
    // of the one-tile-per-workgroup form (9 x 128 -> 128): 30 k cycles per tile = 7.0 k prologue + 4 x 3.9 k + 2.0 k drain + 5.0 k
    // epilogue; its MFMAs are 4 k of that.  With several column tiles (DN_CNN_SEP_WS=0 on the 256-wide layers) the old numbering stays.
    int m0f = 0, n0 = 0, t0 = 0, tstride = 1, my_tiles = 1;
    if (cout == BN) {
        const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
        t0 = (int)blockIdx.x; tstride = (int)gridDim.x;
        my_tiles = t0 < ntiles ? (ntiles - t0 + tstride - 1) / tstride : 0;
        if (my_tiles == 0) return;
    } else if (!conv_tile(cout, BN, rows, m0f, n0)) return;
    auto tile_m0 = [&](int it) { return cout == BN ? (t0 + it * tstride) * CNN_BM : m0f; };
    constexpr int NJ = BN / 64;
    constexpr bool SPAIR = K3_EPI_PAIR && K3_EPI_PAIR_CONV && BN == 128 && NP == 2;      // adjacent columns per lane (conv_epilogue's PAIR form): off with the convolutions' (see k3_conv_split)
    constexpr int NBQ = BN / 64;
    constexpr int half = (KW - 1) / 2;
    f32x16 acc[2][NJ];
    const int cblocks = cin >> 5;
    const int nb = my_tiles * cblocks;                     // steps of this workgroup
    const int l_r = tid >> 2, l_k = (tid & 3) * 8;        // B loader: 64 rows x 4 chunks of 8 elements per pass
    const int dq = tid & 7, dr = (tid >> 3) * 4;          // depthwise: channels 4 dq .. 4 dq + 3, output rows dr .. dr + 3
    // SEP_NSET register sets of raw rows + taps.  With ONE set a raw tile is stored one step after its loads were issued: a step cannot be shorter
    // than the memory latency under load, and a CU holds one tile per workgroup in flight (35-50 KB: Little's law gives the 4.5-5.2 TB/s these
    // layers reach, from HBM and from the Infinity Cache alike -- profiles/r04_infinity_cache_experiment.txt).  With TWO sets (experiment build
    // -DSEP_NSET=2) a tile is stored two steps after its loads.
#ifndef SEP_NSET
#define SEP_NSET 1
#endif
    constexpr int NSET = (SEP_NSET == 2 && BN == 128 && NP == 2) ? 2 : 1;     // the 64-column form would drop from three workgroups per CU to two (179 registers)
    struct RawT { f32x4 rx[NLD]; unsigned pin; f32x4 rw; };       // pin: bit p = load p was inside the pass (one register instead of NLD predicates)
    RawT T0, T1;
    T0.rw = f32x4{0.f, 0.f, 0.f, 0.f}; T1.rw = T0.rw;
    u32x4 rb[NP][NBQ];
    float amax = 0.0f;
    int ld_it = 0, ld_cb = 0;                              // the raw tiles are requested in step order: position of that stream
    auto gloadX = [&](RawT &T) {                            // past the last step it stays on the last one (harmless)
        const int cb = ld_cb, m0 = tile_m0(ld_it);
        if (ld_cb + 1 < cblocks) ld_cb++; else if (ld_it + 1 < my_tiles) { ld_cb = 0; ld_it++; }
        T.pin = 0u;
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 256 * p, rr = f >> 3, q = f & 7;
            const int src = m0 - half + rr;
            const bool in = rr < XROWS && src >= 0 && src < rows;
            T.rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + (cb << 5) + q * 4);
            T.pin |= in ? (1u << p) : 0u;
        }
        if (tid < KW * 8) T.rw = *reinterpret_cast<const f32x4 *>(Wd + (size_t)(tid >> 3) * cin + (cb << 5) + (tid & 7) * 4);
    };
    auto lstoreX = [&](RawT &T) {                           // raw tile + the depthwise taps of the same channel block
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 256 * p, rr = f >> 3, q = f & 7;
            if (rr < XROWS) *reinterpret_cast<f32x4 *>(&Xr[rr * SEP_XP + q * 4]) = ((T.pin >> p) & 1u) ? T.rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (tid < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[(tid >> 3) * 32 + (tid & 7) * 4]) = T.rw;
    };
    auto gloadB = [&](int cb) {
#pragma unroll
        for (int pc = 0; pc < NP; pc++) {
            const uint16_t *wb = Wb + ((size_t)(cb * NP + pc) * cout + n0) * 32;
#pragma unroll
            for (int q = 0; q < NBQ; q++) rb[pc][q] = *reinterpret_cast<const u32x4 *>(wb + (size_t)(q * 64 + l_r) * 32 + l_k);
        }
    };
    auto lstoreB = [&]() {
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int q = 0; q < NBQ; q++) *reinterpret_cast<u32x4 *>(&Bs[pc][(q * 64 + l_r) * CNN_BP + l_k]) = rb[pc][q];
    };
    auto depthwise = [&]() {
        f32x4 o[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KW + 3; j++) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(&Xr[(dr + j) * SEP_XP + dq * 4]);
            if (j < KW) w[j & 3] = *reinterpret_cast<const f32x4 *>(&Wl[j * 32 + dq * 4]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = j - i;                       // tap of output row dr + i that input row dr + j meets
                if (t >= 0 && t < KW) {
#pragma unroll
                    for (int e = 0; e < 4; e++) o[i][e] = __builtin_fmaf(x[e], w[t & 3][e], o[i][e]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int off = (dr + i) * CNN_BP + dq * 4;
            if (NP == 3) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 h, m, l;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float x = o[i][e];
                    const __bf16 hh = (__bf16)x; const float r1 = x - (float)hh;
                    const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
                    h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
                }
                *reinterpret_cast<bf16x4 *>(&As[0][off]) = h; *reinterpret_cast<bf16x4 *>(&As[1][off]) = m; *reinterpret_cast<bf16x4 *>(&As[NP - 1][off]) = l;
            } else {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    sp16x2 hp, lp;
                    split_pair(o[i][e], o[i][e + 1], hp, lp, amax);
                    h[e] = hp[0]; h[e + 1] = hp[1]; l[e] = lp[0]; l[e + 1] = lp[1];
                }
                *reinterpret_cast<f16x4 *>(&As[0][off]) = h; *reinterpret_cast<f16x4 *>(&As[1][off]) = l;
            }
        }
    };
    gloadX(T0); gloadB(0);
    lstoreX(T0); lstoreB();
    __syncthreads();
    gloadX(T0);                                            // tile 1
    if (NSET == 2) gloadX(T1);                             // tile 2: tile k travels in T0 for odd k, T1 for even k (cblocks is even: parity of k = parity of its block)
    gloadB(1 % cblocks);
    depthwise();
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    for (int it = 0; it < my_tiles; it++) {
        const int vb_ahead = K3_VB_AHEAD ? (int)valid[tile_m0(it) + wm * 64 + lane] : -1;      // the epilogue's validity byte, requested a whole tile before it is voted on
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        for (int cb = 0; cb < cblocks; cb++) {
            const int g = it * cblocks + cb;
#ifdef DN_SPLIT_TRACE
#define SP_T(i) do { if (BN == 128 && KW == 9 && it == WS_TRACE_TILE) WS_T(i); } while (0)
#else
#define SP_T(i) do { } while (0)
#endif
            SP_T(2 + 6 * cb);
            __syncthreads();                               // A planes of step g complete; the raw tile is free
            SP_T(3 + 6 * cb);
            if (NSET == 2) {                               // raw tile of step g + 1 (requested two steps ago), then the request for step g + 3 into the same set
                if (cb & 1) { lstoreX(T1); gloadX(T1); } else { lstoreX(T0); gloadX(T0); }      // wave-uniform
            } else {
                lstoreX(T0);                               // raw tile of step g + 1 (loaded during the previous step)
                gloadX(T0);                                // ... of step g + 2
            }
            SP_T(4 + 6 * cb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k16 = 0; k16 < 2; k16++) {
                u32x4 a[2][NP], b[NJ][NP];
#pragma unroll
                for (int pc = 0; pc < NP; pc++) {
#pragma unroll
                    for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[pc][(wm * 64 + i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
#pragma unroll
                    for (int j = 0; j < NJ; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(&Bs[pc][(wn * (BN / 2) + (SPAIR ? 2 * fm + j : j * 32 + fm)) * CNN_BP + k16 * 16 + fk]);
                }
                constexpr int NT = NP == 3 ? 6 : 3;
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    constexpr int PA3[6] = {1, 2, 0, 1, 0, 0}, PB3[6] = {1, 0, 2, 0, 1, 0};
                    constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
                    const int pa = NP == 3 ? PA3[t] : PA2[t % 3], pb = NP == 3 ? PB3[t] : PB2[t % 3];
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++)
                            acc[i][j] = mfma16<NP>(a[i][pa], b[j][pb], acc[i][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            SP_T(5 + 6 * cb);
            if (g + 1 < nb) {                              // wave-uniform
                __syncthreads();                           // every wavefront is done with the planes and the B tile; raw tile of g + 1 visible
                SP_T(6 + 6 * cb);
                lstoreB();
                gloadB((g + 2) % cblocks);
                depthwise();                               // A planes of step g + 1 (the next tile's first block after a tile's last)
                SP_T(7 + 6 * cb);
            }
        }
        conv_epilogue<BN, ADD, SPAIR>(acc, Y, scale, shift, Add, valid, tile_m0(it), n0, wm, wn, lane, cout, relu, post, 64, vb_ahead);
        SP_T(40);
    }
    if (NP == 2) range_report(amax, range_flag, lane);
}

// (Round 3: k3_sep_pair -- two consecutive 9-tap 128 -> 128 separable layers in one launch, the intermediate activations in LDS, 120-row
// tiles, HBM bytes per pair 0.52 x -- is in tools/k3_sep_pair_experiment.hip: bit-identical, and slower (746 us per pair against 2 x 262):
// one phase-locked 512-thread workgroup per CU loses to two independent k3_sep_split workgroups that overlap each other's phases.)
// ---------------------------------------------------------------------------------------------------------
// k3_sep_ws: the fused SeparableConv1D with WAVE SPECIALISATION.  In k3_sep_split a workgroup alternates between its
// depthwise phase (vector unit + LDS) and its pointwise phase (matrix cores); the two never overlap inside the workgroup, and
// the measured time is close to the sum of both (17-tap layers: slower than the two-kernel path).  Here a workgroup has 8
// wavefronts with fixed roles, two per SIMD:
//     producers (4)  raw rows global -> registers -> their own slice of LDS (32 + KW - 1 rows each: no cross-wave dependency),
//                    depthwise filter out of that slice, 16-bit pieces into the A planes of the NEXT channel block
//     consumers (4)  the MFMAs of the CURRENT channel block on its A planes; each owns ALL 128 rows x 64 columns (round 3: as 2 x 2 wavefronts
//                    of 64 x 128 every weight fragment was fetched from L2 by two of them), B fragments straight from L2
// A planes, B tile and tap table are double-buffered: ONE barrier per channel block, and the vector work of block cb + 1 runs
// beside the matrix work of block cb on every SIMD (the matrix pipe and the vector pipe issue from different wavefronts).
// ---------------------------------------------------------------------------------------------------------
template <int BN, int KW, bool ADD, int NP>
__global__ __launch_bounds__(512) void k3_sep_ws(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                 const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, const float *__restrict__ Add,
                                                 const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                 unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int SROWS = 32 + KW - 1;                     // raw rows a producer wave needs for its 32 output rows
    constexpr int NLD = (SROWS * 8 + 63) / 64;             // float4 loads per lane for one raw slice
    __shared__ __attribute__((aligned(16))) float Xr[4][SROWS * SEP_XPW];
    __shared__ __attribute__((aligned(16))) float Wl[2][KW * 32];
    __shared__ __attribute__((aligned(16))) uint16_t As[2][NP][CNN_BM * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;                       // wave-uniform
    // PERSISTENT workgroup: it takes the row tiles blockIdx.x, + gridDim.x, ... (BN == cout: one column tile) and runs their channel
    // blocks as ONE stream of `nb` steps.  The producers are always one step ahead, so while the consumers write a tile's results
    // (6.8 k ticks of the 47 k a one-tile workgroup took) the producers already filter the next tile's first block, and only the
    // first tile of a workgroup waits for its first planes (8.3 k ticks) -- tools/ws_trace.py.
    const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_tiles == 0) return;
#ifndef K3_WS_STAGGER
#define K3_WS_STAGGER 0
#endif
    // experiment: the persistent workgroups run their equal tiles in lockstep, so all 256 CUs store their 128 KB of results in the same ~3 us (10 TB/s asked of the memory)
    // and all load in the same phases; K3_WS_STAGGER = n delays workgroup b by ((37 b) mod 256) / 256 of n x ~4 us before its first tile
    if (K3_WS_STAGGER) {
        const int units = (int)(((unsigned)blockIdx.x * 37u) & 255u) * K3_WS_STAGGER * 127 / 256;      // s_sleep units of 64 clocks
        for (int u = 0; u < units; u += 127) __builtin_amdgcn_s_sleep(127);
    }
    const int n0 = 0;
    constexpr int NJ = BN / 64;
    constexpr int NBQ = BN / 64;
    constexpr int half = (KW - 1) / 2;
    const int cblocks = cin >> 5;
    const int nb = my_tiles * cblocks;                     // steps of this workgroup (even: cblocks is)
    auto tile_m0 = [&](int it) { return ((int)blockIdx.x + it * (int)gridDim.x) * CNN_BM; };
    // ---- consumer state ----
    const int cw = wave & 3;
    const int ct = tid & 255;
    // ---- producer state ----
    const int pw = wave & 3;                               // slice: output rows 32 pw .. 32 pw + 31 of the tile
    const int cp = (lane & 15) * 2, dr = (lane >> 4) * 8;  // channel pair cp, cp + 1; output rows dr .. dr + 7 of the slice
    // Two register sets of raw rows + taps in flight: a slice is stored to LDS TWO iterations after its loads were issued.  With one
    // set the loads had exactly one iteration to land, so an iteration could not be shorter than the HBM latency under load (~3 us
    // against ~0.7 us of MFMA work per channel block: the kernel ran at the memory LATENCY, not at any bandwidth).
    struct RawSet { f32x4 rx[NLD]; bool pin[NLD]; f32x4 rw; bool edge; };
    RawSet S0, S1;
    S0.rw = f32x4{0.f, 0.f, 0.f, 0.f}; S1.rw = S0.rw;
    float amax = 0.0f;
    float *Xs = Xr[pw];
    // The raw slices are fetched in step order, one call per step: the position of the load stream (tile, channel block) is kept
    // incrementally (no division per call); past the last step it stays on the last one (loads past the end are harmless).
    // Interior tiles use buffer addressing: descriptor = the wavefront's SROWS rows, voffset = the lane's (row, float4) inside
    // them (six lane constants), soffset = the channel block -- no 64-bit address arithmetic in the vector unit, which is what
    // this kernel is short of.  The taps likewise; lanes beyond the KW x 8 float4 of a block fall outside the descriptor and
    // read zeros, so there is no branch around that load and the count of loads in flight is the same on every path.
    int ld_cb = 0, ld_it = 0;
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    int xoff[NLD];
#pragma unroll
    for (int p = 0; p < NLD; p++) { const int f = lane + 64 * p; xoff[p] = ((f >> 3) * cin + (f & 7) * 4) * 4; }
    const int woff = ((ct >> 3) * cin + (ct & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t rtap = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(Wd)), 0, KW * cin * 4, 0x00020000);
    auto gloadX = [&](RawSet &S) {
        const int cb = ld_cb, m0 = tile_m0(ld_it);
        if (ld_cb + 1 < cblocks) ld_cb++; else if (ld_it + 1 < my_tiles) { ld_cb = 0; ld_it++; }
        // rows outside [0, rows) read as zeros ('same' padding at the ends of the pass); only the first and the last row tile have
        // any: they take the clamped-address path and 24 selects per slice (lstoreX), behind wave-uniform branches
        S.edge = m0 - half < 0 || m0 + CNN_BM + half > rows;
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) {
                const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
                const int src = m0 + 32 * pw - half + rr;
                const bool in = rr < SROWS && src >= 0 && src < rows;
                S.rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + (cb << 5) + q * 4);
                S.pin[p] = in;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)(m0 + 32 * pw - half) * cin)), 0, SROWS * cin * 4, 0x00020000);
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, xoff[p], cb << 7, 0));
        }
        S.rw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtap, woff, cb << 7, 0));
    };
    auto lstoreX = [&](RawSet &S, int wbuf) {
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = S.pin[p] ? S.rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
            if (rr < SROWS) *reinterpret_cast<f32x4 *>(&Xs[rr * SEP_XPW + q * 4]) = S.rx[p];
        }
        if (ct < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[wbuf][(ct >> 3) * 32 + (ct & 7) * 4]) = S.rw;
    };
    // Depthwise filter of one producer wavefront: 32 output rows x 32 channels per channel block.  A lane owns a PAIR of adjacent
    // channels and 8 consecutive output rows: its 24 input rows and its 17 taps are each read ONCE from LDS as 8-byte pairs
    // (ds_read_b64; the first version re-read every input row 5 times and the taps per 4-row strip as 16-byte quads: 37 KB of LDS
    // reads per wavefront and channel block, now 21 KB) and the filter runs as packed fp32 FMAs (v_pk_fma_f32: the two channels of
    // the pair in one instruction, half the vector instructions), taps in ascending order -- bit-identical to k3_dwconv.
    // Raw-slice pitch 36 floats: the two 16-lane row groups of a 32-lane LDS pass start 8 rows = 1152 B = half a bank window apart.
    auto depthwise = [&](int abuf, int wbuf) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 o[8], w[KW];
#pragma unroll
        for (int t = 0; t < KW; t++) w[t] = *reinterpret_cast<const f32x2 *>(&Wl[wbuf][t * 32 + cp]);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = f32x2{0.f, 0.f};
        // All input rows first, then per input row one GROUP of up to 8 independent FMAs (one per output row it meets), in exactly
        // this order: the FMAs are volatile asm statements, which keep their relative order.  Left to the compiler this became 8
        // serial chains of 17 dependent v_pk_fma_f32 with a hazard nop after each (a dependent vector instruction issues every
        // ~8.5 ticks, an independent one every ~5.5: tools/ubench_coissue.hip, profiles/r01_valu_issue_microbench.txt);
        // __builtin_amdgcn_sched_barrier between the groups did not help, the chains are formed before the scheduler sees them.
        // The phase trace (tools/ws_trace.py) shows the producers' filter, not the matrix work, setting the pace of the workgroup.
        f32x2 x[KW + 7];
#pragma unroll
        for (int j = 0; j < KW + 7; j++) x[j] = *reinterpret_cast<const f32x2 *>(&Xs[(dr + j) * SEP_XPW + cp]);
#pragma unroll
        for (int j = 0; j < KW + 7; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int t = j - i;                       // tap of output row dr + i that input row dr + j meets (ascending per output)
                if (t >= 0 && t < KW) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(w[t]));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int off = (32 * pw + dr + i) * CNN_BP + cp;
            if (NP == 3) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 h, m, l;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float x = o[i][e];
                    const __bf16 hh = (__bf16)x; const float r1 = x - (float)hh;
                    const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
                    h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
                }
                *reinterpret_cast<bf16x2 *>(&As[abuf][0][off]) = h; *reinterpret_cast<bf16x2 *>(&As[abuf][1][off]) = m;
                *reinterpret_cast<bf16x2 *>(&As[abuf][NP - 1][off]) = l;
            } else {
                // the pair at once (split_pair: packed round-to-nearest conversions; same values as element by element)
                sp16x2 h, l;
                split_pair(o[i][0], o[i][1], h, l, amax);
                *reinterpret_cast<sp16x2 *>(&As[abuf][0][off]) = h; *reinterpret_cast<sp16x2 *>(&As[abuf][1][off]) = l;
            }
        }
    };
    // The two roles run their own loops (the accumulators exist only on the consumer side, the filter window only on the producer
    // side: the register allocation is the larger of the two, not the sum).  Every wavefront executes the same number of barriers.
    if (producer) {
        // the producers are the second-dispatched half of the workgroup and set its pace: between two wavefronts of a SIMD the vector
        // issue goes by priority, then age (MI355X_MICROARCH.md), so they take priority 1 once, before their loop (17-tap layers -2.7 %)
        __builtin_amdgcn_s_setprio(1);
        // block b's raw slice travels in set S1 for even b >= 2, S0 for odd b (and for blocks 0, 1 of the prologue); cblocks is even
        // step b's raw slice travels in set S1 for even b >= 2, S0 for odd b (and for steps 0, 1 of the prologue)
        gloadX(S0); lstoreX(S0, 0); gloadX(S0); gloadX(S1);
        __syncthreads();
        depthwise(0, 0); lstoreX(S0, 1); gloadX(S0);
        __syncthreads();
        // The steady-state loop has NO conditional around its loads: with `if (b + 2 < nb)` around the second half, the two paths
        // into the loop head carried different numbers of outstanding loads, the compiler's wait insertion took the conservative
        // one and put s_waitcnt vmcnt(0) in front of the first use of the OLDER set.  The last pair of steps is peeled instead.
        for (int b = 0; b + 2 < nb; b += 2) {
            const bool tr = WS_TRACE_TILE >= 0 && b / cblocks == WS_TRACE_TILE; const int c4 = WS_TRACE_TILE >= 0 ? 4 * (b % cblocks) : 0; (void)tr; (void)c4;
            // even step: filter step b + 1 (stored during step b - 1), store step b + 2 (set S1, loaded two steps ago)
            if (tr) WS_T(3 + c4);
            depthwise(1, 1); if (tr) WS_T(4 + c4); lstoreX(S1, 0); gloadX(S1);
            if (tr) WS_T(5 + c4);
            __syncthreads();
            if (tr) WS_T(6 + c4);
            // odd step b + 1: filter step b + 2, store step b + 3 (set S0)
            depthwise(0, 0); if (tr) WS_T(7 + c4); lstoreX(S0, 1); gloadX(S0);
            if (tr) WS_T(8 + c4);
            __syncthreads();
            if (tr) WS_T(9 + c4);
        }
        depthwise(1, 1);                                   // the last step (nb - 1, odd)
        __syncthreads();
        __syncthreads();
        if (NP == 2) range_report(amax, range_flag, lane);
        return;
    }
    constexpr int CJ = BN / 128;                           // consumer = ALL 128 rows x BN / 4 columns (CJ column blocks of 32): a weight fragment is
    f32x16 acc[4][CJ];                                     // fetched by ONE wavefront of the CU (as 2 x 2 wavefronts of 64 x BN / 2 each was fetched by two)
    // B fragments come STRAIGHT from L2 into registers (pre-split weights [channel block][piece][cout][32]: a fragment is one
    // 16-byte load, 64-byte rows of consecutive lanes coalesce): no B tile in LDS -- that tile was 82 of the kernel's 155 KB, which
    // kept this workgroup off every CU where a per-read stage of another batch held some LDS, and half of its LDS traffic.
    // Double-buffered per k16 step: the loads of step s + 1 are in flight during the MFMAs of step s.
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    constexpr bool EPAIR = K3_EPI_PAIR && CJ == 2;         // column tile j of lane fm = column 2 fm + j (conv_epilogue's PAIR form) instead of 32 j + fm
    const uint16_t *wlane = Wb + ((size_t)(n0 + cw * (BN / 4) + (EPAIR ? 2 * fm : fm))) * 32 + fk;
    auto loadB = [&](u32x4 (&b)[CJ][NP], int step) {            // step = 2 * cb + k16
        const int cb = (step >> 1) % cblocks, k16 = step & 1;          // the weights of a step depend on its channel block only
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int j = 0; j < CJ; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(wlane + ((size_t)(cb * NP + pc) * cout + (EPAIR ? j : j * 32)) * 32 + k16 * 16);
    };
#ifndef K3_WS_BDEPTH
#define K3_WS_BDEPTH 1                                     // 2: the weight fragments of a whole channel block (both k16 steps) are requested a channel block ahead
#endif
    constexpr bool BDEEP = K3_WS_BDEPTH == 2 && NP == 2;
    u32x4 b0[CJ][NP], b1[CJ][NP], b2[BDEEP ? CJ : 1][BDEEP ? NP : 1], b3[BDEEP ? CJ : 1][BDEEP ? NP : 1];
    loadB(b0, 0);
    if constexpr (BDEEP) loadB(b1, 1);
    __syncthreads();
    __syncthreads();
    auto mma = [&](int cur, int k16, u32x4 (&b)[CJ][NP]) {
#ifdef DN_WS_NOMMA
        return;
#endif
        u32x4 a[4][NP];
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int i = 0; i < 4; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[cur][pc][(i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
        constexpr int NT = NP == 3 ? 6 : 3;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            constexpr int PA3[6] = {1, 2, 0, 1, 0, 0}, PB3[6] = {1, 0, 2, 0, 1, 0};
            constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
            const int pa = NP == 3 ? PA3[t] : PA2[t % 3], pb = NP == 3 ? PB3[t] : PB2[t % 3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < CJ; j++)
                    acc[i][j] = mfma16<NP>(a[i][pa], b[j][pb], acc[i][j]);
        }
    };
    for (int it = 0; it < my_tiles; it++) {
        const bool tr = it == WS_TRACE_TILE; (void)tr;
        const int vb_a0 = K3_VB_AHEAD ? (int)valid[tile_m0(it) + lane] : -1, vb_a1 = K3_VB_AHEAD ? (int)valid[tile_m0(it) + 64 + lane] : -1;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < CJ; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        if (tr) WS_T(3);
        if constexpr (BDEEP) {
            // a channel block's eight fragments arrive while the 48 MFMAs of the block before it run (one k16 step ahead, 24 MFMAs = 770 ticks, is less than a
            // load from L2 takes under this kernel's own traffic: the stamps had the 48 MFMAs of a block at 3.35 k ticks for 1.5 k of matrix time)
            for (int cb = 0; cb < cblocks; cb += 2) {
                const int step = it * cblocks + cb;
                loadB(*reinterpret_cast<u32x4 (*)[CJ][NP]>(&b2), 2 * step + 2); loadB(*reinterpret_cast<u32x4 (*)[CJ][NP]>(&b3), 2 * step + 3);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1, b1);
                if (tr) WS_T(5 + 3 * cb);
                __syncthreads();
                if (tr) WS_T(6 + 3 * cb);
                loadB(b0, 2 * step + 4); loadB(b1, 2 * step + 5);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0, *reinterpret_cast<u32x4 (*)[CJ][NP]>(&b2));
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 1, *reinterpret_cast<u32x4 (*)[CJ][NP]>(&b3));
                if (tr) WS_T(5 + 3 * (cb + 1));
                __syncthreads();
                if (tr) WS_T(6 + 3 * (cb + 1));
            }
        } else
        for (int cb = 0; cb < cblocks; cb++) {
            const int cur = cb & 1, step = it * cblocks + cb;
            loadB(b1, 2 * step + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (tr) WS_T(4 + 3 * cb);
            loadB(b0, 2 * step + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 1, b1);
            if (tr) WS_T(5 + 3 * cb);
            __syncthreads();
            if (tr) WS_T(6 + 3 * cb);
        }
        // two row halves, each the epilogue of a 64 x (BN / 4)-column wavefront tile of a BN / 2-wide workgroup tile
        conv_epilogue<BN / 2, ADD, EPAIR>(*reinterpret_cast<f32x16 (*)[2][CJ]>(&acc[0]), Y, scale, shift, Add, valid, tile_m0(it), n0 + (cw >> 1) * (BN / 2), 0, cw & 1, lane, cout, relu, post, 64, vb_a0);
        conv_epilogue<BN / 2, ADD, EPAIR>(*reinterpret_cast<f32x16 (*)[2][CJ]>(&acc[2]), Y, scale, shift, Add, valid, tile_m0(it), n0 + (cw >> 1) * (BN / 2), 1, cw & 1, lane, cout, relu, post, 64, vb_a1);
        if (tr) WS_T(40);
    }
}

// (Round 3: k3_sep_ts -- the same layer TIME-SLICED instead of wave-specialised: all eight wavefronts filter 128 input channels (rows straight
// from global memory into registers, taps in LDS, no MFMA in flight so the vector pipe runs at its full rate), then all eight multiply, four
// barriers per tile -- is in tools/k3_sep_ts_experiment.hip with its phase traces: bit-identical, 4 018 us against 3 810 for the five
// 256 -> 256 layers alone and 717-721 against 733-736 Msamples/s in the pipeline.  The filter phases do run at the predicted 2.7-4.3 k ticks
// (3.4 k per 32 channels beside the MFMAs here), but the multiply phases take 12-18 k where the MFMAs need 6 k: every latency this kernel
// overlaps across its two roles is exposed there.)
// (Round 3: k3_sep_ring -- the A planes in a ring of 2-4 stages with LDS counters (written / consumed, polled) instead of the workgroup
// barrier, so that the producers run ahead through the consumers' epilogue -- is in tools/k3_sep_ring_experiment.hip: bit-identical, and
// SLOWER at every depth (five 256 -> 256 layers: 4 197-4 397 us against 3 822 in the same session).  A polled hand-over costs more than
// s_barrier's, and the consumers' own chain -- 8 x 3.35 k ticks of MFMA phases + 8.5 k of epilogue -- is nearly the tile's 41 k already:
// the run-ahead only moves producer instructions into the issue-bound epilogue.)
// (Round 3: k3_sep_ws16 -- SIXTEEN wavefronts, two producers (row slice x channel half) and two consumers (64 x 64 outputs) per SIMD,
// every wavefront under 128 registers -- is in tools/k3_sep_ws16_experiment.hip with its phase traces: bit-identical, and SLOWER in both
// of its forms: roles overlapping 4 065 us for the five 256 -> 256 layers, roles alternating (a second barrier per step) 4 302 us, this
// kernel 3 802 us in the same session.  What the traces say: alone (alternating form) a producer's 68-FMA filter takes 1.5 k ticks and
// the consumers' 2 x 12 MFMAs 2.5 k -- three times what the matrix pipe needs; overlapping, the producers crawl at ~22 ticks per
// instruction.  More wavefronts per SIMD do not buy issue slots here)
// depthwise part of SeparableConv1D: each thread owns 4 channels of DW_ROWS consecutive rows and slides a register window
// over them, so every input row is read once per thread instead of k times
#define DW_ROWS 16
template <int KW>
__global__ __launch_bounds__(256) void k3_dwconv(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wt,
                                                 const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int c) {
    rows = min(rows, *live);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = c >> 2;
    const long r0 = (long)(idx / c4) * DW_ROWS;
    const int ch = (int)(idx % c4) * 4;
    if (r0 >= rows) return;
    constexpr int half = (KW - 1) / 2;
    float4 w[KW], x[KW];
#pragma unroll
    for (int t = 0; t < KW; t++) w[t] = *reinterpret_cast<const float4 *>(Wt + (size_t)t * c + ch);
    auto ld = [&](long r) { return (r >= 0 && r < rows) ? *reinterpret_cast<const float4 *>(X + (size_t)r * c + ch) : make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
    for (int t = 0; t < KW - 1; t++) x[t + 1] = ld(r0 - half + t);
#pragma unroll
    for (int i = 0; i < DW_ROWS; i++) {
#pragma unroll
        for (int t = 0; t < KW - 1; t++) x[t] = x[t + 1];
        x[KW - 1] = ld(r0 + i + half);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < KW; t++) {
            acc.x = __builtin_fmaf(x[t].x, w[t].x, acc.x); acc.y = __builtin_fmaf(x[t].y, w[t].y, acc.y);
            acc.z = __builtin_fmaf(x[t].z, w[t].z, acc.z); acc.w = __builtin_fmaf(x[t].w, w[t].w, acc.w);
        }
        const long row = r0 + i;
        if (row < rows) {
            if (!valid[row]) acc = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(Y + (size_t)row * c + ch) = acc;
        }
    }
}

__global__ __launch_bounds__(256) void k3_add_relu(const float *__restrict__ A, const float *__restrict__ Bv, float *__restrict__ Y, size_t n4,
                                                   const int *__restrict__ live, int c4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4 || i >= (size_t)*live * (size_t)c4) return;
    const float4 a = reinterpret_cast<const float4 *>(A)[i], b = reinterpret_cast<const float4 *>(Bv)[i];
    reinterpret_cast<float4 *>(Y)[i] = make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}

// Dense(cin -> 3) + softmax per position; writes the probabilities next to the other per-position outputs (at ref_off).
// 16 lanes share one position (each a float4 of the row: coalesced 256-byte rows), partial dots meet by xor-shuffles.
__global__ __launch_bounds__(256) void k3_dense_softmax(const float *__restrict__ X, const float *__restrict__ Wt,
                                                       const float *__restrict__ bias, CnnRows R, int cin, float *probs) {
    const int r = R.r0 + blockIdx.y;
    const unsigned np = R.n_pos[r];
    if (np == 0) return;
    const unsigned sub = threadIdx.x & 15;
    const unsigned p = blockIdx.x * 64 + (threadIdx.x >> 4) * 4;          // 4 consecutive positions per 16-lane group
    const unsigned row0 = R.row_off[r];
    for (unsigned i = 0; i < 4; i++) {
        const unsigned pp = p + i;
        const bool live = pp < np;
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        if (live) {
            const float *x = X + (size_t)(row0 + pp) * cin;
            for (int c = sub * 4; c < cin; c += 64) {
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                const float *w = Wt + c * 3;
                z0 += v.x * w[0] + v.y * w[3] + v.z * w[6] + v.w * w[9];
                z1 += v.x * w[1] + v.y * w[4] + v.z * w[7] + v.w * w[10];
                z2 += v.x * w[2] + v.y * w[5] + v.z * w[8] + v.w * w[11];
            }
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { z0 += __shfl_xor(z0, d); z1 += __shfl_xor(z1, d); z2 += __shfl_xor(z2, d); }
        if (live && sub == 0) {
            z0 += bias[0]; z1 += bias[1]; z2 += bias[2];
            const float m = fmaxf(z0, fmaxf(z1, z2));
            const float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
            const float s = e0 + e1 + e2;
            float *o = probs + (R.io_off[r] + pp) * 3;
            o[0] = e0 / s; o[1] = e1 / s; o[2] = e2 / s;
        }
    }
}

// The canary's comparison (dn_capi.hip cnn_execute): probabilities of sequences [r0, r1) as the fp16 pass wrote them against the same sequences' from bf16
// pieces; a difference above tol anywhere (or a NaN on either side) raises bit 2 of the report word the host reads.
__global__ __launch_bounds__(256) void k3_canary_compare(const float *__restrict__ a, const float *__restrict__ b, CnnRows R, float tol, unsigned *flag) {
    const unsigned r = R.r0 + blockIdx.y;
    const unsigned np = R.n_pos[r];
    const uint64_t o = R.io_off[r] * 3;
    bool bad = false;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < np * 3; i += gridDim.x * 256) bad = bad || !(fabsf(a[o + i] - b[o + i]) <= tol);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 4u);
}
void k3_launch_canary_compare(const float *probs, const float *canary, const CnnRows &rows, unsigned max_pos, float tol, unsigned *flag, hipStream_t st) {
    hipLaunchKernelGGL(k3_canary_compare, dim3((max_pos * 3 + 255) / 256 < 64u ? (max_pos * 3 + 255) / 256 : 64u, rows.r1 - rows.r0), dim3(256), 0, st, probs, canary, rows, tol, flag);
}

#include "k3_block64.h"
#include "k3_pair128.h"
#include <type_traits>

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
    const uint16_t *wts_split;        // device: split conv weights (null = fp32 MFMA path)
    const int64_t *wb_off;            // host: per op offset into wts_split (in elements)
    int pieces;                       // 3: bf16x6, 2: f16x3
    const float *post;                // host, per op: inverse of the power of two the fp16 weights were scaled by (1 for bf16)
    unsigned *range_flag;             // device: the pass's range report block (CNN_RANGE_WORDS(n_ops) words; word 0 is what the host reads)
    unsigned n_pass_pos;              // positions of the sequences [r0, r1)
    uint8_t *enc_len; unsigned *enc_hist; uint64_t *perm_src; unsigned *perm_row;   // device scratch of the encoder's counting sort
    // profiling (null = off): HIP event pairs around every launch of the network's dominant kernel, the 17-tap separable layer
    void (*mark)(void *who, int begin, int kind, hipStream_t st); void *mark_who;
    unsigned *row_off_w; int *live;   // device: this pass's row offsets (written by k3_layout) and its live row count
};

static unsigned k3_cu_count() {                            // persistent kernels: one workgroup per CU of the CURRENT device (DN_CNN_WS_WGS overrides)
    static const unsigned env = (getenv("DN_CNN_WS_WGS") && atoi(getenv("DN_CNN_WS_WGS")) > 0) ? (unsigned)atoi(getenv("DN_CNN_WS_WGS")) : 0u;
    if (env) return env;
    static unsigned per_dev[64] = { 0 };                   // cached per device: one process may drive devices with different CU counts
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!per_dev[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        per_dev[dev] = (unsigned)v;
    }
    return per_dev[dev];
}
static unsigned k3_sep_wgs(int bn, int np = 2) {                       // persistent k3_sep_split: workgroups per launch = what a CU holds (two of the 128-column
    static const unsigned env = (getenv("DN_CNN_SEP_WGS") && atoi(getenv("DN_CNN_SEP_WGS")) > 0) ? (unsigned)atoi(getenv("DN_CNN_SEP_WGS")) : 0u;   // form, three of the 64-column one)
    return env ? env : (bn == 64 ? 3u : (np == 3 ? 1u : 2u)) * k3_cu_count();
}
// the GRU products on the matrix cores whenever the convolutions are (split modes); DN_CNN_ENC_MFMA=0: the vector-unit encoder everywhere
static bool k3_encode_mfma_enabled(const CnnRun &c) { static const bool on = !(getenv("DN_CNN_ENC_MFMA") && atoi(getenv("DN_CNN_ENC_MFMA")) == 0); return on && c.wts_split != nullptr; }
static bool k3_sep_ws_enabled() { static const bool on = !(getenv("DN_CNN_SEP_WS") && atoi(getenv("DN_CNN_SEP_WS")) == 0); return on; }
static bool k3_bm256_enabled() { static const bool on = !(getenv("DN_CNN_BM256") && atoi(getenv("DN_CNN_BM256")) == 0); return on; }
static bool k3_fuse_enabled() { static const bool on = !(getenv("DN_CNN_FUSE") && atoi(getenv("DN_CNN_FUSE")) == 0); return on; }

// a depthwise op can be folded into the pointwise convolution that follows it when nothing else reads its output
static bool k3_can_fuse(const CnnRun &c, int i) {
    if (!c.wts_split || !k3_fuse_enabled() || i + 1 >= c.n_ops) return false;
    const dn_cnn_op &d = c.ops[i], &p = c.ops[i + 1];
    if (d.op != DN_CNN_DWCONV || (p.op != DN_CNN_CONV && p.op != DN_CNN_CONV_ADD) || p.k != 1 || p.src != d.dst || p.cin != d.cin) return false;
    if (p.op == DN_CNN_CONV_ADD && p.a == d.dst) return false;
    if (d.cin % 32 || p.cout % 64 || (d.k != 3 && d.k != 5 && d.k != 9 && d.k != 17)) return false;
    for (int j = i + 2; j < c.n_ops; j++) {                // is the depthwise output read again before it is overwritten?
        const dn_cnn_op &o = c.ops[j];
        const bool reads = (o.op != DN_CNN_ENCODE_GRU && o.src == d.dst) || ((o.op == DN_CNN_ADD_RELU || o.op == DN_CNN_CONV_ADD) && o.a == d.dst) ||
                           (o.op == DN_CNN_ADD_RELU && o.b == d.dst);
        if (reads) return false;
        if (o.dst == d.dst) break;
    }
    return true;
}

static int k3_env_int(const char *name, int dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
// ---- a whole 64-channel residual block in one launch (k3_block64.h) ----
// DN_CNN_BLOCK64: 2 (default) = six separable layers + shortcut convolution + join in one launch, 1 = the six separable layers only (the shortcut stays
// its own launch), 0 = layer by layer.  k3_block64_force (>= 0) overrides the environment: tools/k3_block64_check.hip runs both paths in one process.
int k3_block64_force = -1;
static int k3_block64_mode() { static const int env = k3_env_int("DN_CNN_BLOCK64", 2); return k3_block64_force >= 0 ? k3_block64_force : env; }
// Do the ops at i form such a block?  Returns the number of ops ONE launch covers (13: the whole block, 12: its separable layers), 0 if not:
//   i + 2 j, i + 2 j + 1 (j = 0 .. 5): DWCONV 5 x 64 -> CONV 1 x 64 -> 64, each a fusable pair (k3_can_fuse: nothing else reads the depthwise output), chained,
//   i + 12: CONV_ADD 5 x 64 -> 64 reading the block's input and adding the chain's result;
//   no intermediate buffer is read after the block before it is overwritten (the one launch never materialises them).
static int k3_block64_span(const CnnRun &c, int i) {
    const int mode = k3_block64_mode();
    if (mode <= 0 || c.pieces != 2 || !c.wts_split || !k3_fuse_enabled() || i < 0 || i + 12 >= c.n_ops) return 0;
    if (c.rows.rows % 256 || (unsigned long long)c.rows.rows * 256ull >= (1ull << 31)) return 0;          // byte offsets of the buffer descriptors stay below 2^31
    const int bX = c.ops[i].src;
    int prev = bX;
    for (int j = 0; j < 6; j++) {
        if (!k3_can_fuse(c, i + 2 * j)) return 0;
        const dn_cnn_op &d = c.ops[i + 2 * j], &p = c.ops[i + 2 * j + 1];
        if (d.k != 5 || d.cin != 64 || p.op != DN_CNN_CONV || p.cin != 64 || p.cout != 64 || d.src != prev || d.dst == bX || p.dst == bX) return 0;
        prev = p.dst;
    }
    const dn_cnn_op &sc = c.ops[i + 12];
    if (sc.op != DN_CNN_CONV_ADD || sc.k != 5 || sc.cin != 64 || sc.cout != 64 || sc.src != bX || sc.a != prev || sc.dst == bX) return 0;
    // Liveness of what the layer-by-layer path would have written on the way.  The 12-op launch materialises ONLY the chain's result (`prev`, the last pointwise
    // output), the 13-op launch only the shortcut's destination: an intermediate buffer that a later op reads before it is overwritten rules the fused form
    // out altogether (0: layer by layer), and the chain's result being read later keeps the shortcut its own launch (12).  (Round-5 advisor: the check used to
    // answer 12 for ANY live buffer and was skipped in mode 1 -- a description with distinct buffers per layer would have read stale data.)
    bool prev_live = false;
    for (int j = 0; j < 12; j++) {
        const int b = c.ops[i + j].dst;
        if (b == sc.dst) continue;
        for (int q = i + 13; q < c.n_ops; q++) {
            const dn_cnn_op &o = c.ops[q];
            const bool reads = (o.op != DN_CNN_ENCODE_GRU && o.src == b) || ((o.op == DN_CNN_ADD_RELU || o.op == DN_CNN_CONV_ADD) && o.a == b) ||
                               (o.op == DN_CNN_ADD_RELU && o.b == b);
            if (reads) { if (b != prev) return 0; prev_live = true; break; }
            if (o.dst == b) break;
        }
    }
    // (`prev` is also written by earlier layers of a ping-pong chain; what a later reader sees is the chain's result, which the 12-op launch does write)
    return (mode == 1 || prev_live) ? 12 : 13;
}
static void k3_launch_block64(const CnnRun &c, int i, int span, const float *x, float *y, hipStream_t st) {
    B64Args a{};
    a.X = x; a.Y = y; a.valid = c.valid; a.live = c.live; a.rows = (int)c.rows.rows;
    for (int j = 0; j < 6; j++) {
        const dn_cnn_op &d = c.ops[i + 2 * j], &p = c.ops[i + 2 * j + 1];
        a.L[j].wd = c.wts + d.w; a.L[j].wb = c.wts_split + c.wb_off[i + 2 * j + 1]; a.L[j].scale = c.wts + p.scale; a.L[j].shift = c.wts + p.shift;
        a.L[j].range = c.range_flag + 2 + 2 * (i + 2 * j); a.L[j].post = c.post[i + 2 * j + 1]; a.L[j].relu = p.relu;
    }
    const dn_cnn_op &sc = c.ops[i + 12];
    a.wc = c.wts_split + c.wb_off[i + 12]; a.cscale = c.wts + sc.scale; a.cshift = c.wts + sc.shift; a.crange = c.range_flag + 2 + 2 * (i + 12);
    a.cpost = c.post[i + 12]; a.crelu = sc.relu;
    const unsigned grid = std::max(1u, std::min(k3_cu_count(), c.rows.rows / 32u));
    if (span == 13) hipLaunchKernelGGL((k3_block64<true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k3_block64<false>), dim3(grid), dim3(512), 0, st, a);
}

// ---- two consecutive 9-tap separable layers of the 128-channel stage in one launch (k3_pair128.h) ----
// DN_CNN_PAIR128: 1 = on, 0 = layer by layer.  k3_pair128_force (>= 0) overrides the environment (tools/k3_pair128_check.hip).
int k3_pair128_force = -1;
#ifndef K3_PAIR128_DEFAULT
#define K3_PAIR128_DEFAULT 1
#endif
static int k3_pair128_mode() { static const int env = k3_env_int("DN_CNN_PAIR128", K3_PAIR128_DEFAULT); return k3_pair128_force >= 0 ? k3_pair128_force : env; }
// Do ops i .. i + 3 form such a pair?  DWCONV 9 x cin0 -> CONV 1 x cin0 -> 128 -> DWCONV 9 x 128 -> CONV 1 x 128 -> 128 (cin0 = 64 or 128), each a fusable pair, chained,
// no residual add on either, and the first layer's output read by nothing but the second layer's filter.
static bool k3_takes_pair128(const CnnRun &c, int i) {
    if (k3_pair128_mode() <= 0 || c.pieces != 2 || !c.wts_split || i < 0 || i + 3 >= c.n_ops) return false;
    if (c.rows.rows % 256) return false;
    if (!k3_can_fuse(c, i) || !k3_can_fuse(c, i + 2)) return false;
    const dn_cnn_op &d0 = c.ops[i], &p0 = c.ops[i + 1], &d1 = c.ops[i + 2], &p1 = c.ops[i + 3];
    if (d0.k != 9 || d1.k != 9 || (d0.cin != 64 && d0.cin != 128) || p0.cout != 128 || d1.cin != 128 || p1.cout != 128) return false;
    if (p0.op != DN_CNN_CONV || p1.op != DN_CNN_CONV || d1.src != p0.dst) return false;
    for (int j = i + 3; j < c.n_ops; j++) {                // is the first layer's output read again before it is overwritten (by p1 itself, usually)?
        const dn_cnn_op &o = c.ops[j];
        if (j > i + 3) {
            const bool reads = (o.op != DN_CNN_ENCODE_GRU && o.src == p0.dst) || ((o.op == DN_CNN_ADD_RELU || o.op == DN_CNN_CONV_ADD) && o.a == p0.dst) || (o.op == DN_CNN_ADD_RELU && o.b == p0.dst);
            if (reads) return false;
        }
        if (o.dst == p0.dst) break;
    }
    return true;
}
static void k3_launch_pair128(const CnnRun &c, int i, const float *x, float *y, hipStream_t st) {
    P128Args a{};
    a.X = x; a.Y = y; a.valid = c.valid; a.live = c.live; a.rows = (int)c.rows.rows;
    for (int j = 0; j < 2; j++) {
        const dn_cnn_op &d = c.ops[i + 2 * j], &p = c.ops[i + 2 * j + 1];
        a.L[j].wd = c.wts + d.w; a.L[j].wb = c.wts_split + c.wb_off[i + 2 * j + 1]; a.L[j].scale = c.wts + p.scale; a.L[j].shift = c.wts + p.shift;
        a.L[j].range = c.range_flag + 2 + 2 * (i + 2 * j); a.L[j].post = c.post[i + 2 * j + 1]; a.L[j].relu = p.relu;
    }
    const unsigned grid = std::max(1u, std::min(k3_cu_count(), c.rows.rows / 32u));
    if (c.ops[i].cin == 64) hipLaunchKernelGGL((k3_pair128<64>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k3_pair128<128>), dim3(grid), dim3(256), 0, st, a);
}

// Which fused kernel: the wave-specialised one pays off where the depthwise filter is long and the layer wide (17 taps, 256
// output channels: one workgroup covers ALL 256 columns, so the filter is applied once per row tile); the short filters of the
// narrow layers are memory-side and run better as k3_sep_split with 2-3 workgroups per CU.
static bool k3_takes_ws(int np, const dn_cnn_op &d, const dn_cnn_op &o) { return np == 2 && d.k == 17 && o.cout == 256 && o.cin % 64 == 0 && k3_sep_ws_enabled(); }
// which convolutions take the 256-row form: >= DN_CNN_BM256_MINK taps and >= DN_CNN_BM256_MINCIN input channels.  Round 2 measured the 3-tap layers 8 % SLOWER in
// this form (it spilled 18 registers under its 128-register cap) and kept it to >= 9 taps x >= 128 channels; without the spills (round 4) 3 x 256 -> 256 runs 1 206
// against 1 439 us, 3 x 256 -> 128 618 against 714, 9 x 64 -> 128 509 against 582 (gpurun_out/r4l): every 128-column convolution takes it now.
static bool k3_takes_bm256(int pieces, const dn_cnn_op &o, unsigned rows) {
    static const int mink = k3_env_int("DN_CNN_BM256_MINK", 3), mincin = k3_env_int("DN_CNN_BM256_MINCIN", 64);
    return pieces == 2 && o.cout % 128 == 0 && o.k >= mink && o.cin >= mincin && rows % 256 == 0 && k3_bm256_enabled();
}

template <int BN, bool ADD, int NP>
static int k3_launch_sep(const CnnRun &c, int i, const float *in, float *out, const float *add, hipStream_t st) {
    const dn_cnn_op &d = c.ops[i], &o = c.ops[i + 1];
    const unsigned rows = c.rows.rows;
#define SEP_ARGS in, out, c.wts + d.w, c.wts_split + c.wb_off[i + 1], c.wts + o.scale, c.wts + o.shift, add, c.valid, (int)rows, c.live, \
        o.cin, o.cout, o.relu, c.post[i + 1], c.range_flag + 2 + 2 * i
#define SEP_GO(KW_) hipLaunchKernelGGL((k3_sep_split<BN, KW_, ADD, NP>), dim3(o.cout == BN ? min(conv_grid(rows, o.cout, BN), k3_sep_wgs(BN, NP)) : conv_grid(rows, o.cout, BN)), dim3(256), 0, st, SEP_ARGS)
    if (k3_takes_ws(NP, d, o)) {                            // BN == cout: one column tile
        hipLaunchKernelGGL((k3_sep_ws<256, 17, ADD, 2>), dim3(min(conv_grid(rows, o.cout, 256), k3_cu_count())), dim3(512), 0, st, SEP_ARGS);
        return 0;
    }
    switch (d.k) { case 3: SEP_GO(3); break; case 5: SEP_GO(5); break; case 9: SEP_GO(9); break; case 17: SEP_GO(17); break; default: return -1; }
#undef SEP_GO
#undef SEP_ARGS
    return 0;
}

int k3_run(const CnnRun &c, hipStream_t st) {
    const unsigned rows = c.rows.rows;
    // logical buffer index -> device pointer.  A fused separable layer reads the depthwise INPUT while it writes the pointwise
    // OUTPUT; model descriptions ping-pong (pointwise dst == depthwise src), and a workgroup's halo rows belong to its
    // neighbours' tiles, so writing in place would race.  The fused kernel therefore writes into the (dead) buffer the
    // depthwise op would have written, and the two logical buffers swap their pointers.
    float *pb[8];
    for (int b = 0; b < 8; b++) pb[b] = c.buf[b];
    // row offsets and the live row count of THIS pass: every kernel below reads them, whatever the description's first op is
    hipLaunchKernelGGL(k3_layout, dim3(1), dim3(64), 0, st, c.row_off_w, c.rows.n_pos, c.rows.r0, c.rows.r1, c.live);
    for (int i = 0; i < c.n_ops; i++) {
        const dn_cnn_op &o = c.ops[i];
        // profiling: one HIP event pair around EVERY op of the description (a fused separable layer counts under its depthwise op), so
        // that the run itself says which kernel family dominates (bench.py); k3_describe names the kernel an op takes
        struct Mark { const CnnRun &c; int i; hipStream_t st; Mark(const CnnRun &c_, int i_, hipStream_t s_) : c(c_), i(i_), st(s_) { if (c.mark) c.mark(c.mark_who, 1, i, st); }
                      ~Mark() { if (c.mark) c.mark(c.mark_who, 0, i, st); } } mark_op(c, i, st);
        if (const int span = k3_block64_span(c, i)) {
            // a whole residual block (13 ops), or its six separable layers (12), in one launch: reads the block's input, writes the shortcut's destination
            // (resp. the last pointwise op's); the buffers in between are never touched, so no logical buffers swap
            k3_launch_block64(c, i, span, pb[o.src], span == 13 ? pb[c.ops[i + 12].dst] : pb[c.ops[i + 11].dst], st);
            i += span - 1;
            continue;
        }
        if (k3_takes_pair128(c, i)) {                      // two separable layers, one launch: into the first depthwise op's own destination (never an input of the pair)
            const dn_cnn_op &p1 = c.ops[i + 3];
            k3_launch_pair128(c, i, pb[o.src], pb[o.dst], st);
            { float *t = pb[p1.dst]; pb[p1.dst] = pb[o.dst]; pb[o.dst] = t; }       // the result now IS the second pointwise op's destination
            i += 3;
            continue;
        }
        if (k3_can_fuse(c, i)) {
            const dn_cnn_op &pw = c.ops[i + 1];
            const float *add = pw.op == DN_CNN_CONV_ADD ? pb[pw.a] : nullptr;
            const float *in = pb[o.src];
            float *out = pb[o.dst];                        // the depthwise op's own destination: never an input of this layer
            int rc;
            if (c.pieces == 3) {
                if (pw.cout % 128 == 0) rc = add ? k3_launch_sep<128, true, 3>(c, i, in, out, add, st) : k3_launch_sep<128, false, 3>(c, i, in, out, add, st);
                else rc = add ? k3_launch_sep<64, true, 3>(c, i, in, out, add, st) : k3_launch_sep<64, false, 3>(c, i, in, out, add, st);
            } else {
                if (pw.cout % 128 == 0) rc = add ? k3_launch_sep<128, true, 2>(c, i, in, out, add, st) : k3_launch_sep<128, false, 2>(c, i, in, out, add, st);
                else rc = add ? k3_launch_sep<64, true, 2>(c, i, in, out, add, st) : k3_launch_sep<64, false, 2>(c, i, in, out, add, st);
            }
            if (rc) return rc;
            { float *t = pb[pw.dst]; pb[pw.dst] = pb[o.dst]; pb[o.dst] = t; }     // the result now IS the pointwise op's destination
            i++;                                           // the pointwise op is done too
            continue;
        }
        switch (o.op) {
            case DN_CNN_ENCODE_GRU:
                hipMemsetAsync(pb[o.dst], 0, (size_t)rows * 64 * sizeof(float), st);
                if (c.n_pass_pos && k3_encode_mfma_enabled(c)) {       // sorts by signal length inside its workgroups
                    hipLaunchKernelGGL(k3_encode_mfma, dim3((c.max_pos + ENC_WG / 2 - 1) / (ENC_WG / 2), c.rows.r1 - c.rows.r0), dim3(ENC_WG), 0, st, c.core, c.resid, c.sig,
                                       c.rows, c.valid, pb[o.dst], c.wts, o);
                    break;
                }
                hipMemsetAsync(c.enc_hist, 0, 2 * ENC_BINS * sizeof(unsigned), st);
                hipLaunchKernelGGL(k3_encode_len, dim3((c.max_pos + 255) / 256, c.rows.r1 - c.rows.r0), dim3(256), 0, st, c.sig, c.rows, c.enc_len, c.enc_hist);
                hipLaunchKernelGGL(k3_encode_perm, dim3((c.max_pos + 255) / 256, c.rows.r1 - c.rows.r0), dim3(256), 0, st, c.rows, c.enc_len, c.enc_hist,
                                   c.enc_hist + ENC_BINS, c.perm_src, c.perm_row);
                if (c.n_pass_pos)
                    hipLaunchKernelGGL(k3_encode, dim3((c.n_pass_pos + 63) / 64), dim3(64), 0, st, c.core, c.resid, c.sig, c.perm_src, c.perm_row, c.enc_hist,
                                       c.valid, pb[o.dst], c.wts, o);
                break;
            case DN_CNN_CONV:
            case DN_CNN_CONV_ADD:
                if (o.cin % 32 || o.cout % 64) return -1;
            {
                const float *add = o.op == DN_CNN_CONV_ADD ? pb[o.a] : nullptr;          // fused residual join: y = act(conv + buf[a])
#define CONV_GO(BN_, NBUF_, ADD_) hipLaunchKernelGGL((k3_conv<BN_, NBUF_, ADD_>), dim3(conv_grid(rows, o.cout, BN_)), dim3(256), 0, st, \
        pb[o.src], pb[o.dst], c.wts + o.w, c.wts + o.scale, c.wts + o.shift, add, c.valid, (int)rows, c.live, o.k, o.cin, o.cout, o.relu)
#define CONV_GO_SP(BN_, ADD_, NP_) hipLaunchKernelGGL((k3_conv_split<BN_, ADD_, NP_>), dim3(conv_grid(rows, o.cout, BN_)), dim3(256), 0, st, \
        pb[o.src], pb[o.dst], c.wts_split + c.wb_off[i], c.wts + o.scale, c.wts + o.shift, add, c.valid, (int)rows, c.live, o.k, o.cin, o.cout, o.relu, \
        c.post[i], c.range_flag + 2 + 2 * i)
#define CONV_GO_BM(BN_, ADD_) hipLaunchKernelGGL((k3_conv_split<BN_, ADD_, 2, 256>), dim3(conv_grid(rows, o.cout, BN_, 256)), dim3(512), 0, st, \
        pb[o.src], pb[o.dst], c.wts_split + c.wb_off[i], c.wts + o.scale, c.wts + o.shift, add, c.valid, (int)rows, c.live, o.k, o.cin, o.cout, o.relu, \
        c.post[i], c.range_flag + 2 + 2 * i)
#define CONV_GO_BF(BN_, ADD_) do { if (c.pieces == 3) CONV_GO_SP(BN_, ADD_, 3); else if (BN_ == 128 && k3_takes_bm256(c.pieces, o, rows)) CONV_GO_BM(128, ADD_); \
        else CONV_GO_SP(BN_, ADD_, 2); } while (0)
                if (c.wts_split) {
                    // (256-column workgroups -- each wavefront 64 rows x 128 columns, 12 LDS fragment reads per 24 MFMAs instead of 8 per 12 --
                    // were measured: 308-356 VGPRs, one workgroup per CU, 17 x 128 -> 256: 4.37 ms against 3.58; not kept)
                    if (o.cout % 128 == 0) { if (add) CONV_GO_BF(128, true); else CONV_GO_BF(128, false); }
                    else { if (add) CONV_GO_BF(64, true); else CONV_GO_BF(64, false); }
                } else if (o.cout % 128 == 0) { if (add) CONV_GO(128, 2, true); else CONV_GO(128, 2, false); }
                else { if (add) CONV_GO(64, 1, true); else CONV_GO(64, 1, false); }
#undef CONV_GO
#undef CONV_GO_BF
#undef CONV_GO_BM
#undef CONV_GO_SP
                break;
            }
            case DN_CNN_DWCONV: {
                const size_t n = (size_t)((rows + DW_ROWS - 1) / DW_ROWS) * (o.cin / 4);
                const dim3 g((unsigned)((n + 255) / 256));
#define DW_CASE(KW) case KW: hipLaunchKernelGGL(k3_dwconv<KW>, g, dim3(256), 0, st, pb[o.src], pb[o.dst], c.wts + o.w, c.valid, (int)rows, c.live, o.cin); break;
                switch (o.k) { DW_CASE(3) DW_CASE(5) DW_CASE(7) DW_CASE(9) DW_CASE(17) default: return -1; }
#undef DW_CASE
                break;
            }
            case DN_CNN_ADD_RELU: {
                const size_t n4 = (size_t)rows * o.cin / 4;
                hipLaunchKernelGGL(k3_add_relu, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, pb[o.a], pb[o.b], pb[o.dst], n4, c.live, o.cin / 4);
                break;
            }
            case DN_CNN_DENSE_SOFTMAX:
                if (o.cout != 3) return -1;
                hipLaunchKernelGGL(k3_dense_softmax, dim3((c.max_pos + 63) / 64, c.rows.r1 - c.rows.r0), dim3(256), 0, st, pb[o.src], c.wts + o.w,
                                   c.wts + o.shift, c.rows, o.cin, c.probs);
                break;
            default: return -1;
        }
    }
#ifndef K3_NO_RANGE_CHECK                                  /* tools/k3_block64_check.hip reads the per-op words itself */
    if (c.pieces == 2 && c.range_flag) hipLaunchKernelGGL(k3_range_check, dim3(1), dim3(64), 0, st, c.range_flag, c.n_ops);
#endif
    return 0;
}

// The kernel op i of the description takes, named as rocprofv3 prints it (template arguments included) -- the same predicates k3_run
// dispatches on.  Returns 0 for an op that is covered by the previous one (the pointwise half of a fused separable layer).
int k3_describe(const CnnRun &c, int i, char *buf, size_t cap) {
    if (i < 0 || i >= c.n_ops || !buf || !cap) return -1;
    buf[0] = 0;
    const dn_cnn_op &o = c.ops[i];
    const int np = c.pieces;
    for (int j = 0; j <= i;) {                             // k3_run's own walk up to op i: is it inside a block that one launch covers?
        const int span = k3_block64_span(c, j);
        if (span && i < j + span) {
            if (j < i) return 0;
            snprintf(buf, cap, "k3_block64<%s>", span == 13 ? "true" : "false");
            return 1;
        }
        if (!span && k3_takes_pair128(c, j)) {              // four ops, one launch: reported under the first
            if (i < j + 4) {
                if (j < i) return 0;
                snprintf(buf, cap, "k3_pair128<%d>", o.cin);
                return 1;
            }
            j += 4;
            continue;
        }
        j += span ? span : (k3_can_fuse(c, j) ? 2 : 1);
    }
    if (i > 0 && k3_can_fuse(c, i - 1)) return 0;
    if (k3_can_fuse(c, i)) {
        const dn_cnn_op &pw = c.ops[i + 1];
        const char *add = pw.op == DN_CNN_CONV_ADD ? "true" : "false";
        if (k3_takes_ws(np, o, pw)) snprintf(buf, cap, "k3_sep_ws<256, 17, %s, 2>", add);
        else snprintf(buf, cap, "k3_sep_split<%d, %d, %s, %d>", pw.cout % 128 == 0 ? 128 : 64, o.k, add, np);
        return 1;
    }
    switch (o.op) {
        case DN_CNN_ENCODE_GRU: snprintf(buf, cap, k3_encode_mfma_enabled(c) ? "k3_encode_mfma" : "k3_encode"); return 1;
        case DN_CNN_CONV: case DN_CNN_CONV_ADD: {
            const char *add = o.op == DN_CNN_CONV_ADD ? "true" : "false";
            const int bn = o.cout % 128 == 0 ? 128 : 64;
            if (!c.wts_split) snprintf(buf, cap, "k3_conv<%d, %d, %s>", bn, bn == 128 ? 2 : 1, add);
            else if (bn == 128 && k3_takes_bm256(np, o, c.rows.rows)) snprintf(buf, cap, "k3_conv_split<128, %s, 2, 256>", add);
            else snprintf(buf, cap, "k3_conv_split<%d, %s, %d, 128>", bn, add, np);
            return 1;
        }
        case DN_CNN_DWCONV: snprintf(buf, cap, "k3_dwconv<%d>", o.k); return 1;
        case DN_CNN_ADD_RELU: snprintf(buf, cap, "k3_add_relu"); return 1;
        case DN_CNN_DENSE_SOFTMAX: snprintf(buf, cap, "k3_dense_softmax"); return 1;
    }
    return -1;
}
