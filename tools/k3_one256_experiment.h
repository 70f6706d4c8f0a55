// k3_one256_experiment.h -- EXPERIMENT (round 5), not part of libdnascent_hip.so: included by tools/k3_one256_check.hip behind k3_cnn.hip.
//
// k3_one256: ONE 17-tap SeparableConv1D of the 256-channel stage (CIN -> 256, CIN = 128 or 256) with NO producer / consumer split: four wavefronts per CU, one per
// SIMD, each with the whole 512-register file, each doing a quarter of everything -- it FILTERS a quarter of the input channels (rows straight from global memory into
// registers, requested a step before they are used; 17 taps as packed FMAs; split into the chunk's fp16 planes in LDS, which all four read) and MULTIPLIES the previous
// chunk's planes by ITS 64 columns of the pointwise matrix, resident in registers for the whole launch (256 registers at CIN = 256), and stores them (8 bytes per lane
// and row).  One LDS-only barrier per 32-row step.  The idea: k3_sep_ws splits the CU into four filtering and four multiplying wavefronts, its filter goes global ->
// registers -> LDS -> registers, its weights stream from L2 every step, and a SIMD's time is the sum of its wavefronts' vector and matrix streams either way.
//
// RESULT (gpurun_out/r6n, 1.2 M rows): bit-identical to k3_sep_ws on its first run (values and range reports; 4 096 / 25 600 / 1.2 M rows, CIN 128 and 256) and SLOWER:
// 883 us against 703 (CIN = 256), 565 against 426 (CIN = 128).  Stamps: a step is 12 800 ticks -- filter of two channel blocks 5 500-8 500 (its ~450 instructions are
// worth 2 500), multiply 3 400 (96 MFMAs at 35 each: fine), store 800.  The filter's first instruction is s_waitcnt vmcnt(0): on gfx9 loads and stores share vmcnt and the
// compiler waits for zero when a load is needed while stores are pending, so every step begins by waiting for the previous step's stores to land; peeling the loop's
// conditionals (this version) spilled 26 registers and kept the wait.  The same reordering idea in k3_pair128 (stores mid-step, one register set) measured slower (453 us
// against 433): not what bounds it.  Not pursued: k3_sep_ws's split hides exactly this latency behind the other role.
#pragma once

struct O256Args {
    const float *X; float *Y;                               // [row][CIN], [row][256]
    const uint8_t *valid; const int *live; int rows; int pad_;
    const float *wd;                                        // depthwise taps [17][CIN]
    const uint16_t *wb;                                     // pointwise weights, pre-split fp16 pieces [channel block][piece][256][32]
    const float *scale, *shift; unsigned *range; float post; int relu;
};

#ifdef O256_TRACE
__device__ unsigned long long o256_trace[4][8];
#define O256_T(i) do { if (blockIdx.x == O256_TRACE && s == 40) { __builtin_amdgcn_sched_barrier(0); if (lane == 0) o256_trace[wave][i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define O256_T(i) do { } while (0)
#endif

// the MFMAs of one chunk for CB channel blocks: as p128_multiply, with the B fragments of CB blocks
template <int CB>
__device__ __forceinline__ void o256_multiply(const uint16_t *Pb, const u32x4 (&bw)[CB][2][2][2], const int n, const int hh, f32x16 (&acc)[2]) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[j][q] = 0.0f;
    u32x4 fa[2][2];                                         // [buffer][piece: 0 = hi, 1 = lo]
    const uint16_t *ab = Pb + n * CNN_BP + 8 * hh;
    auto frags = [&](int g, u32x4 (&f)[2]) {
        const int cb = g >> 1, k16 = g & 1;
        f[0] = *reinterpret_cast<const u32x4 *>(ab + cb * (2 * B64_APL) + k16 * 16);
        f[1] = *reinterpret_cast<const u32x4 *>(ab + cb * (2 * B64_APL) + B64_APL + k16 * 16);
    };
    frags(0, fa[0]);
#pragma unroll
    for (int g = 0; g < 2 * CB; g++) {
        if (g + 1 < 2 * CB) frags(g + 1, fa[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int cb = g >> 1, k16 = g & 1;
        const u32x4 ah = fa[g & 1][0], al = fa[g & 1][1];
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(al, bw[cb][k16][0][j], acc[j]);
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][1][j], acc[j]);
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][0][j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// epilogue -> global memory, [row][256] fp32 (b64_epilogue's expressions; 1 024-byte rows)
template <bool MASKED>
__device__ __forceinline__ void o256_store(const f32x16 (&acc)[2], const b64f2 sc, const b64f2 sh, const float floor_, const unsigned vml, const __amdgpu_buffer_rsrc_t rY, const int ybase) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        const int rowq = (q & 3) + 8 * (q >> 2);
        b64f2 y0 = __builtin_elementwise_fma(b64f2{acc[0][q], acc[0][q + 1]}, b64f2{sc[0], sc[0]}, b64f2{sh[0], sh[0]});
        b64f2 y1 = __builtin_elementwise_fma(b64f2{acc[1][q], acc[1][q + 1]}, b64f2{sc[1], sc[1]}, b64f2{sh[1], sh[1]});
        float a0 = __builtin_fmaxf(y0[0], floor_), a1 = __builtin_fmaxf(y1[0], floor_), b0 = __builtin_fmaxf(y0[1], floor_), b1 = __builtin_fmaxf(y1[1], floor_);
        if (MASKED) {
            const bool oka = (vml >> rowq) & 1u, okb = (vml >> (rowq + 1)) & 1u;
            a0 = oka ? a0 : 0.0f; a1 = oka ? a1 : 0.0f; b0 = okb ? b0 : 0.0f; b1 = okb ? b1 : 0.0f;
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{a0, a1}), rY, ybase + rowq * 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{b0, b1}), rY, ybase + (rowq + 1) * 1024, 0, 0);
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void k3_one256(const O256Args A) {
    constexpr int CB = CIN / 32, NCBF = CB / 4;             // channel blocks in all / filtered by one wavefront
    __shared__ __attribute__((aligned(16))) uint16_t Pl[2][CB][2 * B64_APL];       // the chunk's planes: [chunk parity][channel block][piece][32 x CNN_BP]
    __shared__ __attribute__((aligned(16))) float Tw[17 * CIN];                    // the depthwise taps (read per step: they would be 68 more registers)
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = min(A.rows, *A.live);
    const int nct = rows >> 5, per = (nct + (int)gridDim.x - 1) / (int)gridDim.x, c_lo = (int)blockIdx.x * per, nch = min(per, nct - c_lo);
    if (nch <= 0) return;
    const int S0 = c_lo * 32;
    for (int i = tid; i < 17 * CIN; i += 256) Tw[i] = A.wd[i];
    // ---- resident: this wavefront's 64 columns (64 wave + 2 n + j) of the pointwise matrix ----
    u32x4 bw[CB][2][2][2];
#pragma unroll
    for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    bw[cb][k16][pc][j] = *reinterpret_cast<const u32x4 *>(A.wb + ((size_t)((cb * 2 + pc) * 256 + 64 * wave + 2 * n + j)) * 32 + k16 * 16 + 8 * hh);
    const int colp = 64 * wave + 2 * n;
    const b64f2 sc = {A.scale[colp] * A.post, A.scale[colp + 1] * A.post}, sh = {A.shift[colp], A.shift[colp + 1]};
    const float floor_ = A.relu ? 0.0f : -3.402823466e38f;
    // descriptors over THIS STRIPE's rows; rows outside the pass fall outside them: zeros ('same' padding) / dropped stores
    const int xb = max(0, S0 - 64), xrows = min(rows, S0 + 32 * nch + 96) - xb;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X + (size_t)xb * CIN)), 0, xrows * CIN * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.Y + (size_t)S0 * 256)), 0, min(rows - S0, 32 * nch) * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid + S0)), 0, min(rows - S0, 32 * nch), 0x00020000);
    // ---- filter role: lane (cpl, rq) of channel block f: channel pair cpl, output rows 8 rq .. 8 rq + 7, input rows 8 rq - 8 .. 8 rq + 15 of the chunk ----
    const int cpl = lane & 15, rq = lane >> 4;
    const int ch0 = wave * (CIN / 4);                       // first channel this wavefront filters
    b64u2 xp[NCBF][24];
    const int xlane = ((8 * rq - 8) * CIN + ch0 + 2 * cpl) * 4;
    auto gloadX = [&](int c, int f) {                       // chunk c's rows of channel block f (past the stripe: loaded, never used)
        const int base = xlane + (S0 + 32 * c - xb) * CIN * 4;
#pragma unroll
        for (int j = 0; j < 24; j++) xp[f][j] = __builtin_bit_cast(b64u2, __builtin_amdgcn_raw_buffer_load_b64(rX, base, f * 128 + j * CIN * 4, 0));
    };
#pragma unroll
    for (int f = 0; f < NCBF; f++) gloadX(0, f);
    float amax = 0.0f;
    __syncthreads();                                        // the taps are in LDS
    // a step's two halves; the loop below runs them UNCONDITIONALLY (first and last step peeled): with `if (s < nch)` / `if (c1 >= 0)` around them the paths into the
    // loop head carried different numbers of outstanding loads and stores, the compiler's wait insertion took the conservative one -- s_waitcnt vmcnt(0) at the top of
    // every step, i.e. the previous step's STORES had to land before the filter could touch rows that arrived long ago (first version: 12 800 ticks per step)
    auto filter = [&](const int s) {                       // chunk s: this wavefront's channel blocks -> planes
        float am = 0.0f;                                    // (a chunk's 32 output rows are always inside the pass: nothing to mask in the range report)
#pragma unroll
        for (int f = 0; f < NCBF; f++) {
            b64f2 x[24], tw[17];
#pragma unroll
            for (int j = 0; j < 24; j++) x[j] = __builtin_bit_cast(b64f2, xp[f][j]);
            gloadX(s + 1, f);                               // the same rows of the next chunk, into the registers just read
#pragma unroll
            for (int t = 0; t < 17; t++) tw[t] = *reinterpret_cast<const b64f2 *>(&Tw[t * CIN + ch0 + f * 32 + 2 * cpl]);
            b64f2 o[8];
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = b64f2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 24; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int t = j - i;
                    if (t >= 0 && t < 17) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(tw[t]));
                }
            }
            uint16_t *ap = &Pl[s & 1][wave * NCBF + f][0] + 8 * rq * CNN_BP + 2 * cpl;
            b64_split_store<false>(o, am, ap, 0, 0);
        }
        amax = __builtin_fmaxf(amax, am);
    };
    auto multiply = [&](const int c1, const unsigned vb) { // chunk c1: planes x this wavefront's columns -> global memory
        f32x16 acc[2];
        o256_multiply<CB>(&Pl[c1 & 1][0][0], bw, n, hh, acc);
        const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
        const unsigned vml = hh ? vm >> 4 : vm;
        const int ybase = ((32 * c1 + 4 * hh) * 256 + colp) * 4;
        if (vm == 0xffffffffu) o256_store<false>(acc, sc, sh, floor_, vml, rY, ybase);
        else { asm volatile("; chunk with padding rows" ::: "memory"); o256_store<true>(acc, sc, sh, floor_, vml, rY, ybase); }
    };
    // step s filters chunk s and multiplies chunk s - 1, whose validity bytes were requested a step earlier
    unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, n, 0, 0);
    filter(0);
    b64_barrier();
    for (int s = 1; s < nch; s++) {
        O256_T(0);
        const unsigned vbn = __builtin_amdgcn_raw_buffer_load_b8(rV, 32 * s + n, 0, 0);
        filter(s);
        O256_T(1);
        multiply(s - 1, vb);
        vb = vbn;
        O256_T(3);
        b64_barrier();
        O256_T(4);
    }
    multiply(nch - 1, vb);
    range_report(amax, A.range, lane);
}
