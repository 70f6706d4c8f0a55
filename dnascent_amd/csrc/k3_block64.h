// k3_block64.h -- included by k3_cnn.hip (needs its f32x16 / u32x4 / mfma16 / range_report / CNN_BP).
//
// k3_block64: a WHOLE 64-channel residual block of the network (SURVEY s2.3 layers 4-17 and 18-31; runCNN's graph, detect.cpp:577-675) in ONE launch:
//     six SeparableConv1D (5 taps, 64 -> 64, folded BatchNorm, ReLU between them) + the shortcut Conv1D (5 taps, 64 -> 64, folded BatchNorm) + Add + ReLU.
// Layer by layer these are 13 launches that each stream the whole pass through HBM (12 x 512 B + 768 B per position at 4.4-5.4 TB/s: the stage is
// HBM-bound); here a position's activations enter the chip once and leave it once (HBM: 256 B in, 256 B out, + the shortcut's re-read of the input,
// which comes from L2 / the Infinity Cache), and every value is computed ONCE -- no halo rows recomputed by neighbouring tiles, no weights re-fetched.
//
// Formulation: a DATAFLOW PIPELINE over 32-row chunks, one persistent 8-wavefront workgroup per CU walking a contiguous stripe of the pass's rows.
//   wavefront = one LAYER (a pipeline stage), its weights resident in REGISTERS for the whole launch:
//       stage l (6 of them): the 64 x 64 pointwise matrix as MFMA B fragments (2 channel blocks x 2 k16 x {hi, lo} x 2 column tiles = 64 registers),
//                            its 5 depthwise taps, its folded BatchNorm;
//       conv  c (2 of them): one 32-column tile of the 5 x 64 x 64 shortcut kernel (160 registers).
//   step s: stage l works on chunk s - l, the shortcut on chunk s - 6; ONE workgroup barrier per step.  Between two stages the activations cross as
//   fp32 rows in a double-buffered LDS ring (2 x 32 rows x 272 B per stage boundary): written by stage l's epilogue during step s, read by stage l + 1's
//   depthwise filter during step s + 1.  Every layer's 5-tap filter needs 2 rows either side, so layer l's chunk grid is shifted up by 2 l rows: a stage
//   can always produce its whole chunk from the chunk it was handed plus the LAST FOUR ROWS of the previous one, which it keeps itself (a private 4-row
//   halo in LDS).  The fp16 hi / lo planes the matrix cores read (A operand) are private to a stage and hold one 32-channel block at a time (5 KB).
//   A stripe starts with one warm-up chunk (outputs discarded) that fills the halos; the rows it gets wrong are exactly the ones the previous stripe's
//   last chunk computes.
//
// Arithmetic, operand splits, K order, epilogue expressions and masks are those of k3_sep_split / k3_conv_split: results are bit-identical to the
// layer-by-layer path (tools/k3_block64_check.hip compares the two on the device, tools/variant_check.py the whole pipeline).  f16x3 only; the other
// arithmetics (and any description that does not have this block shape) run layer by layer.
//
// LDS (163 648 B of the CU's 163 840): rings 6 x 2 x 32 x 272, halos 5 x 4 x 272, A planes 6 x 2 x 32 x 80, shortcut planes 2 x 2 x 2 x 36 x 80.
// One workgroup per CU, two wavefronts per SIMD: SIMD pairs (stage 0, stage 1), (stage 2, stage 3), (stage 4, shortcut tile 0), (stage 5, shortcut tile 1).
#pragma once

#define B64_RP 68                                           // ring pitch in floats: rows 8 apart start half a bank window apart (ds_read_b64 of two row quarters)
#define B64_RING (32 * B64_RP)                              // floats of one ring buffer
#define B64_APL (32 * CNN_BP)                               // elements of one stage A plane (one piece, one channel block)
#define B64_CPL (36 * CNN_BP)                               // ... of one shortcut A plane (32 rows + 2 either side)

struct B64Layer {
    const float *wd;                                        // depthwise taps [5][64]
    const uint16_t *wb;                                     // pointwise weights, pre-split fp16 pieces [channel block][piece][cout][32]
    const float *scale, *shift;                             // folded BatchNorm + bias
    unsigned *range;                                        // the layer's pair of words in the pass's range report block
    float post; int relu;
};
struct B64Args {
    const float *X; float *Y;                               // block input / output, [row][64] fp32
    const uint8_t *valid; const int *live; int rows; int pad_;
    B64Layer L[6];
    const uint16_t *wc;                                     // shortcut kernel, pre-split [channel block][tap][piece][cout][32]
    const float *cscale, *cshift; unsigned *crange; float cpost; int crelu;
};

#ifndef B64_ABL
#define B64_ABL 0                                           /* experiment builds of tools/k3_block64_check.hip (TIMING ONLY, wrong results): 1 = no ring reads, 2 = no ring writes, 4 = the A planes' LDS round trip replaced by a register dependency; third version: 8 = wavefronts 0-5 idle (first filter: 32), 16 = the shortcut wavefronts idle, 64 / 128 = the odd / even ones of wavefronts 0-4 idle */
#endif
#ifdef B64_TRACE                                            /* experiment builds only (tools/k3_block64_check.hip -DB64_TRACE=<workgroup>): shader-clock stamps of one step's phases */
#ifndef B64_TRACE_STEP
#define B64_TRACE_STEP 40
#endif
__device__ unsigned long long b64_trace[8][16];
#define B64_T(role, i) do { if (blockIdx.x == B64_TRACE && s == B64_TRACE_STEP) { __builtin_amdgcn_sched_barrier(0); if (lane == 0) b64_trace[role][i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } \
                            if (blockIdx.x == B64_TRACE && (i) == 10 && (s == 0 || s == nsteps - 1) && lane == 0) { b64_trace[role][s == 0 ? 12 : 13] = __builtin_amdgcn_s_memtime(); b64_trace[role][14] = (unsigned long long)nsteps; } } while (0)
#else
#define B64_T(role, i) do { } while (0)
#endif

// The step barrier.  __syncthreads() is fence + barrier, and the fence waits for EVERY outstanding memory operation of the wavefront (s_waitcnt vmcnt(0)):
// the rows requested a chunk ahead and the results just stored would have to finish before every barrier -- the prefetch would hide nothing.  What the
// pipeline needs ordered across a step is LDS traffic only (rings, written before the barrier, read after it): wait for that, then s_barrier.  Global loads
// stay in flight across it (MI355X_MICROARCH.md: barriers do not drain VMEM); the compiler still waits for each load before its first use.
__device__ __forceinline__ void b64_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef float b64f2 __attribute__((ext_vector_type(2)));
typedef _Float16 b64h2 __attribute__((ext_vector_type(2)));
typedef _Float16 b64h4 __attribute__((ext_vector_type(4)));
typedef unsigned b64u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const void *b64_uniform_ptr(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    return (const void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

// Where the pipeline's stages start.  Wavefronts w and w + 4 share a SIMD and run the same program between the same barriers: left alone both filter at the
// same time (vector pipe contended, matrix pipe idle) and then both multiply (the reverse).  -DB64_STAGGER=1 STAGGERS the second wavefront of a pair
// (MI355X_MICROARCH.md, two waves per SIMD, item 9): stages 1 and 3 (wavefronts 4, 5) DEFER a chunk's epilogue to the start of the next step (its sums stay in
// registers across the barrier), so that they filter while their partner multiplies and multiply while it filters; a deferred stage hands its chunk over one step
// later, which the next stage's start absorbs.  Bit-identical, and no faster (see the switch): not the default.
#ifndef B64_STAGGER
#define B64_STAGGER 0                                       /* measured (gpurun_out/r5c, r5d): 520-544 us staggered against 512-542 us plain for 1.2 M rows: the phases that cost are vector + LDS on both partners, not matrix against vector; kept as a build switch */
#endif
#define B64_START(st) (B64_STAGGER ? ((st) + ((st) >= 2) + ((st) >= 4)) : (st))      /* 0 1 3 4 6 7 | 0 1 2 3 4 5 */
#define B64_CONV_START (B64_START(5) + 1)
#define B64_DEFERS(st) (B64_STAGGER && ((st) == 1 || (st) == 3))

// the split of 8 filtered rows x 2 channels into the fp16 planes (+ the range report's largest |value|; MASKED: rows outside the pass do not count).
// B64_SPLIT_WIDE (default): the five steps of the split run ACROSS the eight rows -- eight independent instructions per step -- instead of row by row (the
// compiler interleaved two rows: dependent vector instructions of one wavefront issue ~8 ticks apart, and the phase took 800-1 700 ticks for ~70 instructions).
#ifndef B64_SPLIT_WIDE
#define B64_SPLIT_WIDE 1
#endif
// what the fp16 rounding of a channel pair left over, o - (float) h, in ONE instruction per channel: v_fma_mix_f32 reads the fp16 half directly
// (h x -1.0 + o, exact: the difference is representable), where conversion + subtraction are 3 instructions per pair.  B64_MIX=0: the plain expression.
#ifndef B64_V3
#define B64_V3 0                                            /* 1: the third version below (filter in the accumulator layout); bit-identical, 480 us per block against 467: not the default */
#endif
#ifndef B64_MIX
#define B64_MIX (B64_V3 ? 0 : 1)                            /* measured in one session (gpurun_out/r5n, 1.2 M rows): second version 473 / 467 / 477 us with 0 / 1 / 2, third version 480 / 507 / 489 */
#endif
__device__ __forceinline__ b64f2 b64_rest(const b64h2 h, const b64f2 o) {
    if (!B64_MIX) return o - __builtin_convertvector(h, b64f2);
    float r0, r1;
    const float o0 = o[0], o1 = o[1];
    if (B64_MIX == 2) {                                    // two plain subtractions instead of the packed one the compiler forms
        const b64f2 f = __builtin_convertvector(h, b64f2);
        const float f0 = f[0], f1 = f[1];
        asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r0) : "v"(o0), "v"(f0));
        asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r1) : "v"(o1), "v"(f1));
        return b64f2{r0, r1};
    }
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(o0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(o1));
    return b64f2{r0, r1};
}

template <bool MASKED>
__device__ __forceinline__ void b64_split_store(const b64f2 (&o)[8], float &am, uint16_t *ap, const int g0, const int rows) {
    b64h2 h[8], l[8]; b64f2 rest[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = __builtin_convertvector(o[i], b64h2);
    if (B64_SPLIT_WIDE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) rest[i] = b64_rest(h[i], o[i]);
    if (B64_SPLIT_WIDE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = __builtin_convertvector(rest[i], b64h2);
    if (B64_SPLIT_WIDE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) { *reinterpret_cast<b64h2 *>(ap + i * CNN_BP) = h[i]; *reinterpret_cast<b64h2 *>(ap + B64_APL + i * CNN_BP) = l[i]; }
    float m0 = am, m1 = 0.0f;                              // two chains of v_max3_f32 (m, |a|, |b|) for the range report's largest |value|
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float &m = (i & 1) ? m1 : m0;
        if (MASKED) {
            const int g = g0 + i;
            if (g >= 0 && g < rows) m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(o[i][0])), __builtin_fabsf(o[i][1]));
        } else m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(o[i][0])), __builtin_fabsf(o[i][1]));
    }
    am = __builtin_fmaxf(m0, m1);
}

// a stage's epilogue: folded BatchNorm, ReLU, padding mask (MASKED: some row of the chunk is padding); accumulator register q of lane (n, hh) is row
// (q & 3) + 8 (q >> 2) + 4 hh, column tile j is channel 2 n + j.  Rows go in pairs (q, q + 1): adjacent registers for the packed FMA.
template <bool MASKED, bool TO_GLOBAL>
__device__ __forceinline__ void b64_epilogue(const f32x16 (&acc)[2], const b64f2 sc, const b64f2 sh, const float floor_, const unsigned vml, float *wrow,
                                             const __amdgpu_buffer_rsrc_t rY, const int ybase) {
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
        if (TO_GLOBAL) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{a0, a1}), rY, ybase + rowq * 256, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{b0, b1}), rY, ybase + (rowq + 1) * 256, 0, 0);
        } else {
            float *r0 = wrow + rowq * B64_RP;
            r0[0] = a0; r0[1] = a1; r0[B64_RP] = b0; r0[B64_RP + 1] = b1;
        }
    }
}

// One separable layer as a pipeline stage.  KIND 0: the block's first layer (its input comes from global memory, a chunk ahead in registers),
// 1: a middle layer (LDS ring in, LDS ring out), 2: the last layer (ring in; out to the ring the shortcut joins from, or to global memory when the
// shortcut runs as its own launch: CONV == false).  DEFER: the epilogue of a chunk runs at the start of the next step (see B64_START).
template <int KIND, bool CONV, bool DEFER>
__device__ __forceinline__ void b64_stage(const B64Args &A, const int st, const int t0, const int lane, const int S0, const int nch, const int nsteps, const int rows,
                                          const float *rin, float *rout, float *hal, uint16_t *Ap) {
    const B64Layer &P = A.L[st];
    const int n = lane & 31, hh = lane >> 5;               // GEMM role: accumulator column pair n (channels 2 n, 2 n + 1), lane half
    const int cpl = lane & 15, rq = lane >> 4;             // filter role: channel pair cpl of the channel block, row quarter rq (8 output rows)
    // ---- resident operands ----
    u32x4 bw[2][2][2][2];                                  // [channel block][k16][piece][column tile]: column tile j of lane n is output channel 2 n + j
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    bw[cb][k16][pc][j] = *reinterpret_cast<const u32x4 *>(P.wb + ((size_t)((cb * 2 + pc) * 64 + 2 * n + j)) * 32 + k16 * 16 + 8 * hh);
    b64f2 tw[2][5];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int t = 0; t < 5; t++) tw[cb][t] = *reinterpret_cast<const b64f2 *>(P.wd + t * 64 + cb * 32 + 2 * cpl);
    const b64f2 sc = {P.scale[2 * n] * P.post, P.scale[2 * n + 1] * P.post}, sh = {P.shift[2 * n], P.shift[2 * n + 1]};
    const float floor_ = P.relu ? 0.0f : -3.402823466e38f;
    float amax = 0.0f;
#ifdef B64_YOUNG_PRIO
    if (st == 1 || st == 3) __builtin_amdgcn_s_setprio(B64_YOUNG_PRIO);      // experiment: wavefronts 4, 5 are the second-dispatched partners of stages 0, 2 and lose the issue arbitration by age (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#endif
    // ---- first stage: the input chunk travels global -> registers, requested one chunk ahead.  Lane (cpl, rq) of channel block cb needs rows
    //      XC - 4 + 8 rq + j, j = 0 .. 11 (XC = first row of the input chunk): 8 bytes each, a 16-lane group reads 128 contiguous bytes of a row.
    //      Rows outside [0, rows) fall outside the descriptor and read as the zeros 'same' padding wants.
    b64u2 xp[2][12];
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X)), 0, KIND == 0 ? rows * 256 : 0, 0x00020000);
    const int xlane = ((8 * rq - 4) * 64 + 2 * cpl) * 4;
    auto gloadX = [&](int c, int cb) {
        const int base = xlane + ((S0 - 20 + 32 * c) * 64 + cb * 32) * 4;
#pragma unroll
        for (int j = 0; j < 12; j++) xp[cb][j] = __builtin_bit_cast(b64u2, __builtin_amdgcn_raw_buffer_load_b64(rX, base + j * 256, 0, 0));
    };
    if (KIND == 0) { gloadX(0, 0); gloadX(0, 1); }
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.Y)), 0, (KIND == 2 && !CONV) ? rows * 256 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid)), 0, rows, 0x00020000);
    constexpr bool TO_GLOBAL = KIND == 2 && !CONV;
    f32x16 acc[2];
    // what a deferred epilogue needs of its chunk: the votes of its rows and where it goes
    unsigned d_vm = 0; int d_c = -1;
    auto epilogue = [&](const unsigned vm, const int c) {
        const unsigned vml = hh ? vm >> 4 : vm;
        float *wrow = rout + (c & 1) * B64_RING + 4 * hh * B64_RP + 2 * n;
        const int ybase = ((S0 - 20 - 2 * (st + 1) + 32 * c + 4 * hh) * 64 + 2 * n) * 4;
        if (TO_GLOBAL && c == 0) return;                   // the warm-up chunk's rows belong to the previous stripe
        if ((B64_ABL & 2) && !TO_GLOBAL) {                  // keep the sums alive without the ring stores
            float keep = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; q++) keep += acc[0][q] * sc[0] + acc[1][q] * sc[1];
            if (keep == 1234.5f) wrow[0] = keep;
            return;
        }
        if (vm == 0xffffffffu) b64_epilogue<false, TO_GLOBAL>(acc, sc, sh, floor_, vml, wrow, rY, ybase);
        else { asm volatile("; chunk with padding rows" ::: "memory"); b64_epilogue<true, TO_GLOBAL>(acc, sc, sh, floor_, vml, wrow, rY, ybase); }
    };
    for (int s = 0; s < nsteps; s++) {
        const int c = s - t0;
        B64_T(st, 0);
        if (DEFER && d_c >= 0) { epilogue(d_vm, d_c); d_c = -1; }
        if (c >= 0 && c < nch) {                           // wave-uniform
            const int og0 = S0 - 20 - 2 * (st + 1) + 32 * c;          // first global row of this stage's output chunk
            // validity of the 32 output rows: fetched now (both lane halves the same 32 bytes; rows outside [0, rows) fall outside the descriptor and read 0),
            // voted on by the epilogue.  NO branch around the load: behind `if (in range)` the compiler waited for it (vmcnt(0): this byte and every row
            // requested ahead) right where it was issued -- ~1 000 ticks at the top of every step of every stage in the first version's stamps
            const unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, og0 + n, 0, 0);
            const bool edge = og0 < 0 || og0 + 32 > rows;  // the chunk reaches outside the pass: those rows are masked out of the range report
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[j][q] = 0.0f;
            float am = 0.0f;
            const float *rb = rin + (c & 1) * B64_RING;
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                // ---- the filter's 12 input rows: in[j], j = 8 rq .. 8 rq + 11, where in[0 .. 3] = the halo (last four rows of the previous chunk), in[4 .. 35] = the chunk ----
                b64f2 x[12];
                if (KIND == 0) {
#pragma unroll
                    for (int j = 0; j < 12; j++) x[j] = __builtin_bit_cast(b64f2, xp[cb][j]);
                } else {
                    if (B64_ABL & 1) {
#pragma unroll
                        for (int j = 0; j < 12; j++) x[j] = tw[cb][j % 5] + b64f2{(float)s, (float)j};
                    } else {
                    const float *p4 = rb + 8 * rq * B64_RP + cb * 32 + 2 * cpl;            // in[8 rq + 4] = row 8 rq of the chunk
                    const float *pa = rq == 0 ? hal + cb * 32 + 2 * cpl : p4 - 4 * B64_RP; // in[8 rq]: the halo for row quarter 0, else four rows further up in the chunk
#pragma unroll
                    for (int j = 0; j < 4; j++) x[j] = *reinterpret_cast<const b64f2 *>(pa + j * B64_RP);
#pragma unroll
                    for (int j = 4; j < 12; j++) x[j] = *reinterpret_cast<const b64f2 *>(p4 + (j - 4) * B64_RP);
                    if (rq == 3) {                         // the chunk's last four rows are the next chunk's halo (the reads above were issued first: LDS keeps a wavefront's order)
#pragma unroll
                        for (int r = 0; r < 4; r++) *reinterpret_cast<b64f2 *>(hal + r * B64_RP + cb * 32 + 2 * cpl) = x[8 + r];
                    }
                    }
                }
                B64_T(st, 1 + 4 * cb);
                // ---- depthwise: output row 8 rq + i = sum over taps t of in[8 rq + i + t] w[t], taps ascending (k3_dwconv's order), packed over the channel pair ----
                b64f2 o[8];
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = b64f2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 12; j++) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int t = j - i;
                        if (t >= 0 && t < 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(tw[cb][t]));
                    }
                }
                B64_T(st, 2 + 4 * cb);
                if (KIND == 0) gloadX(c + 1, cb);           // the same rows of the next chunk (past the stripe: loaded, never used)
                // ---- split into the two fp16 pieces -> this stage's A planes ----
                uint16_t *ap = Ap + 8 * rq * CNN_BP + 2 * cpl;
                if (B64_ABL & 4) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {              // the split's arithmetic without the planes: the pieces stay in o[] for the register "fragments"
                        am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(o[i][0]), __builtin_fabsf(o[i][1])));
                        const b64h2 h = __builtin_convertvector(o[i], b64h2);
                        const b64f2 rest = o[i] - __builtin_convertvector(h, b64f2);
                        const b64h2 l = __builtin_convertvector(rest, b64h2);
                        o[i] = b64f2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
                    }
                } else
                if (!edge) b64_split_store<false>(o, am, ap, 0, 0);
                else { asm volatile("; chunk at an end of the pass" ::: "memory"); b64_split_store<true>(o, am, ap, og0 + 8 * rq, rows); }
                B64_T(st, 3 + 4 * cb);
                // ---- pointwise: 32 rows x 64 columns, K = this channel block; pieces l h', h l', h h' per k16 as in k3_sep_split ----
#pragma unroll
                for (int k16 = 0; k16 < 2; k16++) {
                    u32x4 ah, al;
                    if (B64_ABL & 4) {
                        ah = u32x4{__builtin_bit_cast(unsigned, o[4 * k16][0]), __builtin_bit_cast(unsigned, o[4 * k16 + 1][0]), __builtin_bit_cast(unsigned, o[4 * k16 + 2][0]), __builtin_bit_cast(unsigned, o[4 * k16 + 3][0])};
                        al = u32x4{__builtin_bit_cast(unsigned, o[4 * k16][1]), __builtin_bit_cast(unsigned, o[4 * k16 + 1][1]), __builtin_bit_cast(unsigned, o[4 * k16 + 2][1]), __builtin_bit_cast(unsigned, o[4 * k16 + 3][1])};
                    } else {
                    ah = *reinterpret_cast<const u32x4 *>(Ap + n * CNN_BP + k16 * 16 + 8 * hh);
                    al = *reinterpret_cast<const u32x4 *>(Ap + B64_APL + n * CNN_BP + k16 * 16 + 8 * hh);
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(al, bw[cb][k16][0][j], acc[j]);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][1][j], acc[j]);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][0][j], acc[j]);
                }
                B64_T(st, 4 + 4 * cb);
            }
            if (c > 0 || S0 == 0) amax = __builtin_fmaxf(amax, am);   // a later stripe's warm-up chunk filters rows it cannot know (the previous stripe reports them)
            const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
            if (DEFER) { d_vm = vm; d_c = c; }
            else epilogue(vm, c);
        }
        B64_T(st, 9);
        b64_barrier();
        B64_T(st, 10);
    }
    range_report(amax, P.range, lane);
}

// The shortcut convolution + Add + ReLU as the pipeline's last stage: column tile ct (32 output channels) of chunk s - B64_CONV_START.  The two wavefronts
// SHARE the fp16 planes of the 36 input rows (round 5, second version): wavefront ct splits channel block ct of the NEXT chunk's rows into the planes (double-
// buffered by chunk parity; the step barrier orders the hand-over), and both multiply the current chunk out of the planes written a step earlier.  In the
// first version each split all 64 channels for itself: its stamps showed the shortcut wavefronts at 6 300 ticks per step with every LDS round trip of the
// separable stages ablated away -- they, not the stages, set the step -- and 2 x ~1 000 of those ticks were the split.
__device__ __forceinline__ void b64_conv(const B64Args &A, const int ct, const int t0, const int lane, const int S0, const int nch, const int nsteps, const int rows,
                                         const float *r5, uint16_t *Acv) {
    const int n = lane & 31, hh = lane >> 5;
    u32x4 wc[2][5][2][2];                                  // [channel block][tap][k16][piece] of output channel 32 ct + n: 160 registers
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int tap = 0; tap < 5; tap++)
#pragma unroll
            for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
                for (int pc = 0; pc < 2; pc++)
                    wc[cb][tap][k16][pc] = *reinterpret_cast<const u32x4 *>(A.wc + ((size_t)(((cb * 5 + tap) * 2 + pc) * 64 + ct * 32 + n)) * 32 + k16 * 16 + 8 * hh);
    const float sc = A.cscale[ct * 32 + n] * A.cpost, sh = A.cshift[ct * 32 + n];
    const float floor_ = A.crelu ? 0.0f : -3.402823466e38f;
    float amax = 0.0f;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X)), 0, rows * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.Y)), 0, rows * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid)), 0, rows, 0x00020000);
    // the 36 input rows x 32 channels of this wavefront's channel block are 288 float4: lane takes f = lane + 64 p, p = 0 .. 4 (the fifth only on lanes 0-31),
    // requested a whole step before they are split (two steps before they are multiplied)
    f32x4 xr[5];
    auto gloadX = [&](int c) {
        const int base = ((S0 - 32 + 32 * c - 2) * 64 + ct * 32) * 4;
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int f = lane + 64 * p;
            xr[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, base + ((f >> 3) * 64 + (f & 7) * 4) * 4, 0, 0));
        }
    };
    gloadX(1);
#ifdef B64_CONV_PRIO
    __builtin_amdgcn_s_setprio(B64_CONV_PRIO);             // experiment: the shortcut wavefronts set the step; let them win the issue arbitration against their stage partner
#endif
    for (int s = 0; s < nsteps; s++) {
        const int c = s - t0;
        B64_T(6 + ct, 0);
        if (!(B64_ABL & 16) && c >= 0 && c + 1 < nch) {    // wave-uniform: channel block ct of chunk c + 1 -> planes [(c + 1) & 1][ct]
            uint16_t *Ap = Acv + (((c + 1) & 1) * 2 + ct) * (2 * B64_CPL);
#pragma unroll
            for (int p = 0; p < 5; p++) {
                const int f = lane + 64 * p;
                if (p < 4 || lane < 32) {
                    const b64f2 xa = {xr[p][0], xr[p][1]}, xb = {xr[p][2], xr[p][3]};
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(xa[0])), __builtin_fabsf(xa[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(xb[0])), __builtin_fabsf(xb[1]));
                    const b64h2 ha = __builtin_convertvector(xa, b64h2), hb = __builtin_convertvector(xb, b64h2);
                    const b64h2 la = __builtin_convertvector(b64_rest(ha, xa), b64h2), lb = __builtin_convertvector(b64_rest(hb, xb), b64h2);
                    const b64h4 h = {ha[0], ha[1], hb[0], hb[1]}, l = {la[0], la[1], lb[0], lb[1]};
                    const int off = (f >> 3) * CNN_BP + (f & 7) * 4;
                    *reinterpret_cast<b64h4 *>(Ap + off) = h; *reinterpret_cast<b64h4 *>(Ap + B64_CPL + off) = l;
                }
            }
            gloadX(c + 2);                                  // (past the stripe: loaded, never used)
        }
        B64_T(6 + ct, 1);
        if (!(B64_ABL & 16) && c >= 1 && c < nch) {        // wave-uniform; chunk 0 is the stripe's warm-up chunk: nothing to join
            const int G0 = S0 - 32 + 32 * c;
            const unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, G0 + n, 0, 0);
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const uint16_t *Ac = Acv + ((c & 1) * 2 + cb) * (2 * B64_CPL);
#pragma unroll
                for (int tap = 0; tap < 5; tap++)
#pragma unroll
                    for (int k16 = 0; k16 < 2; k16++) {
                        const u32x4 ah = *reinterpret_cast<const u32x4 *>(Ac + (n + tap) * CNN_BP + k16 * 16 + 8 * hh);
                        const u32x4 al = *reinterpret_cast<const u32x4 *>(Ac + B64_CPL + (n + tap) * CNN_BP + k16 * 16 + 8 * hh);
                        acc = mfma16<2>(al, wc[cb][tap][k16][0], acc);
                        acc = mfma16<2>(ah, wc[cb][tap][k16][1], acc);
                        acc = mfma16<2>(ah, wc[cb][tap][k16][0], acc);
                    }
                B64_T(6 + ct, 4 + 4 * cb);
            }
            const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
            const unsigned vml = hh ? vm >> 4 : vm;
            const float *rb = r5 + (c & 1) * B64_RING + 4 * hh * B64_RP + ct * 32 + n;
            const int ybase = ((G0 + 4 * hh) * 64 + ct * 32 + n) * 4;
            if (vm == 0xffffffffu) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int rowq = (q & 3) + 8 * (q >> 2);
                    float y = __builtin_fmaf(acc[q], sc, sh);
                    y += rb[rowq * B64_RP];
                    y = __builtin_fmaxf(y, floor_);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rY, ybase + rowq * 256, 0, 0);
                }
            } else {
                asm volatile("; chunk with padding rows" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int rowq = (q & 3) + 8 * (q >> 2);
                    float y = __builtin_fmaf(acc[q], sc, sh);
                    y += rb[rowq * B64_RP];
                    y = __builtin_fmaxf(y, floor_);
                    y = ((vml >> rowq) & 1u) ? y : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rY, ybase + rowq * 256, 0, 0);
                }
            }
        }
        B64_T(6 + ct, 9);
        b64_barrier();
        B64_T(6 + ct, 10);
    }
    range_report(amax, A.crange, lane);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Round 5, third version (-DB64_V3=1; bit-identical, NOT the default: it is no faster, see the end of this comment): THE FILTER RUNS IN THE ACCUMULATOR LAYOUT.
// In the version above a chunk crosses LDS twice per layer: the epilogue writes fp32 rows to a ring, the next stage's filter reads them back in ITS lane
// layout (lane = channel pair x row quarter), filters, splits and writes the fp16 planes, and the multiply reads those.  Here the wavefront that produced
// layer l's sums ALSO filters them for layer l + 1, in the registers they are already in: after the epilogue lane (n, hh) holds channels (2 n, 2 n + 1) of
// rows {0-3, 8-11, 16-19, 24-27} + 4 hh -- a 5-tap filter down the rows needs its neighbours' rows, and four rounds of v_permlane32_swap (the lane halves
// exchange register halves; no LDS) turn that into 20 CONSECUTIVE rows per lane: rows -4 .. 15 in the lower half, 12 .. 31 in the upper, the first four of
// the lower half being the chunk before's last four, which the wavefront keeps in registers.  Each half then filters 16 output rows (80 packed FMAs, taps
// ascending as everywhere), splits them and stores layer l + 1's fp16 planes.  LDS per layer boundary: ONE crossing (planes written, planes read) instead of
// two, 40 instructions instead of ~72; the fp32 rings and the halo rows are gone (only the last layer still hands fp32 rows to the shortcut's join).
//
//   wavefront l = 0 .. 4:   planes of layer l (chunk c)  -> 24 MFMAs -> BatchNorm / ReLU / mask -> exchange -> filter of layer l + 1 -> split -> planes of layer l + 1
//   wavefront 5 ("ends"):   the block's FIRST filter (input rows from global memory, a chunk ahead in registers, in the lane layout of the version above)
//                           and its LAST multiply (layer 5 -> the fp32 ring the shortcut joins from)
//   wavefronts 6, 7:        the shortcut, as above
//   step s: the first filter works on chunk s, wavefront l on chunk s - 1 - l, the last multiply on chunk s - 6, the shortcut on chunk s - 7.
// The row grids are those of the version above (layer l's chunk c starts at row S0 - 20 - 2 (l + 1) + 32 c), so are the sums' order and every expression:
// bit-identical again.  LDS: planes 6 x 2 x (2 channel blocks x 2 pieces x 32 x 80 B) + ring 2 x 32 x 272 + shortcut planes = 163 328 B.
//
// Measured (gpurun_out/r5m, r5n; 1.2 M rows, one session): 480 us per block against 473 for the second version with the same split, 467 with its default one.
// The LDS instructions it removes were not what a step costs.  Its stamps and ablations (tools/k3_block64_check.hip -DB64_TRACE / -DB64_ABL): a step is ~5 800
// ticks in both versions; with the shortcut wavefronts idle 389-422 us, with wavefronts 0-5 idle 304; ONE filtering wavefront per SIMD (its partner idle) takes
// 3 900 ticks for a chunk -- 24 MFMAs (990 with the fragment reads) + ~320 vector instructions at 7-8 ticks each -- and two of them on a SIMD 5 500.  Cutting
// the vector instructions (50 fewer per wavefront: b64_rest, v_max3_f32 for the range report) moved the block by 1 % in either direction depending on the
// instruction chosen; the average clock under this kernel is 1.9 GHz against 2.2-2.3 for its lighter ablations.  NOTES.md has the table.
#define B64_V3_CONV_START 7

// v_permlane32_swap on a channel pair: a's upper lane half <-> b's lower lane half.  (Components are copied to scalars first: __builtin_bit_cast straight
// from a vector element, `bit_cast(unsigned, a[e])`, read element 0 for every e with this compiler -- the first build multiplied one column tile only.)
__device__ __forceinline__ void b64_swap(b64f2 &a, b64f2 &b) {
    const float ax = a[0], ay = a[1], bx = b[0], by = b[1];
    const b64u2 r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
    const b64u2 r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
    a = b64f2{__uint_as_float(r0[0]), __uint_as_float(r1[0])};
    b = b64f2{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
}

// the epilogue into registers: Y[q] = (channel 2 n, channel 2 n + 1) of row (q & 3) + 8 (q >> 2) + 4 hh
template <bool MASKED>
__device__ __forceinline__ void b64_epilogue_regs(const f32x16 (&acc)[2], const b64f2 sc, const b64f2 sh, const float floor_, const unsigned vml, b64f2 (&Y)[16]) {
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
        Y[q] = b64f2{a0, a1}; Y[q + 1] = b64f2{b0, b1};
    }
}

// the 24 MFMAs of one chunk out of a layer's planes [channel block][piece][32 x CNN_BP]
__device__ __forceinline__ void b64_multiply(const uint16_t *Pb, const u32x4 (&bw)[2][2][2][2], const int n, const int hh, f32x16 (&acc)[2]) {
    u32x4 ah[2][2], al[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++) {
            ah[cb][k16] = *reinterpret_cast<const u32x4 *>(Pb + cb * (2 * B64_APL) + n * CNN_BP + k16 * 16 + 8 * hh);
            al[cb][k16] = *reinterpret_cast<const u32x4 *>(Pb + cb * (2 * B64_APL) + B64_APL + n * CNN_BP + k16 * 16 + 8 * hh);
        }
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[j][q] = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++) {
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(al[cb][k16], bw[cb][k16][0][j], acc[j]);
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah[cb][k16], bw[cb][k16][1][j], acc[j]);
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah[cb][k16], bw[cb][k16][0][j], acc[j]);
        }
}

__device__ __forceinline__ void b64_load_bw(const B64Layer &P, const int n, const int hh, u32x4 (&bw)[2][2][2][2]) {
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    bw[cb][k16][pc][j] = *reinterpret_cast<const u32x4 *>(P.wb + ((size_t)((cb * 2 + pc) * 64 + 2 * n + j)) * 32 + k16 * 16 + 8 * hh);
}

// wavefront l = 0 .. 4: layer l's multiply + epilogue, layer l + 1's filter + split
__device__ __forceinline__ void b64_v3_stage(const B64Args &A, const int st, const int lane, const int S0, const int nch, const int nsteps, const int rows,
                                             const uint16_t *Pin, uint16_t *Pout) {
    const B64Layer &P = A.L[st], &Q = A.L[st + 1];
    const int n = lane & 31, hh = lane >> 5;
    const int t0 = st + 1;
    u32x4 bw[2][2][2][2];
    b64_load_bw(P, n, hh, bw);
    b64f2 tw[5];
#pragma unroll
    for (int t = 0; t < 5; t++) tw[t] = *reinterpret_cast<const b64f2 *>(Q.wd + t * 64 + 2 * n);
    const b64f2 sc = {P.scale[2 * n] * P.post, P.scale[2 * n + 1] * P.post}, sh = {P.shift[2 * n], P.shift[2 * n + 1]};
    const float floor_ = P.relu ? 0.0f : -3.402823466e38f;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid)), 0, rows, 0x00020000);
    float amax = 0.0f;
    b64f2 carry[4];                                        // this layer's rows 28 .. 31 of the chunk before (upper lane half; the lower half's copy is never used)
#pragma unroll
    for (int i = 0; i < 4; i++) carry[i] = b64f2{0.f, 0.f};
    uint16_t *const apl = Pout + (n >> 4) * (2 * B64_APL) + 16 * hh * CNN_BP + 2 * (n & 15);
    for (int s = 0; s < nsteps; s++) {
        const int c = s - t0;
        B64_T(st, 0);
        if (!(B64_ABL & 8) && !((B64_ABL & 64) && (st & 1)) && !((B64_ABL & 128) && !(st & 1)) && c >= 0 && c < nch) {         // wave-uniform
            const int og0 = S0 - 20 - 2 * (st + 1) + 32 * c;          // first row of layer st's chunk; layer st + 1's chunk starts two rows earlier
            const unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, og0 + n, 0, 0);
            const bool edge = og0 - 2 < 0 || og0 + 30 > rows;
            f32x16 acc[2];
            b64_multiply(Pin + (c & 1) * (4 * B64_APL), bw, n, hh, acc);
            B64_T(st, 1);
            const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
            const unsigned vml = hh ? vm >> 4 : vm;
            b64f2 Y[16];
            if (vm == 0xffffffffu) b64_epilogue_regs<false>(acc, sc, sh, floor_, vml, Y);
            else { asm volatile("; chunk with padding rows" ::: "memory"); b64_epilogue_regs<true>(acc, sc, sh, floor_, vml, Y); }
            B64_T(st, 2);
            // ---- 20 consecutive rows per lane: W[w] = row w - 4 (lower half) / row w + 12 (upper half) of the chunk ----
            b64f2 W[20];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                b64f2 a = Y[i], b = Y[8 + i]; b64_swap(a, b); W[4 + i] = a; W[8 + i] = b;              // a: rows i | 16 + i,      b: rows 4 + i | 20 + i
                b64f2 a2 = Y[4 + i], b2 = Y[12 + i]; b64_swap(a2, b2); W[12 + i] = a2; W[16 + i] = b2; // a2: rows 8 + i | 24 + i, b2: rows 12 + i | 28 + i
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                b64f2 p = carry[i], d = W[16 + i];
                b64_swap(p, d);                             // p's upper half <- this chunk's rows 12 + i; d's lower half <- the previous chunk's rows 28 + i
                W[i] = hh ? p : d;
                carry[i] = W[16 + i];
            }
            B64_T(st, 3);
            // ---- depthwise, taps ascending: output k = sum_t W[k + t] w[t] = row k - 2 (lower half) / k + 14 (upper half) of layer st's grid = plane row k / 16 + k ----
            b64f2 o[2][8];
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int i = 0; i < 8; i++) o[g][i] = b64f2{0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 20; w++) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int t = w - k;
                    if (t >= 0 && t < 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[k >> 3][k & 7]) : "v"(W[w]), "v"(tw[t]));
                }
            }
            B64_T(st, 4);
            float am = 0.0f;
            uint16_t *ap = apl + (c & 1) * (4 * B64_APL);
            if (!edge) { b64_split_store<false>(o[0], am, ap, 0, 0); b64_split_store<false>(o[1], am, ap + 8 * CNN_BP, 0, 0); }
            else {
                asm volatile("; chunk at an end of the pass" ::: "memory");
                b64_split_store<true>(o[0], am, ap, og0 - 2 + 16 * hh, rows); b64_split_store<true>(o[1], am, ap + 8 * CNN_BP, og0 - 2 + 16 * hh + 8, rows);
            }
            if (c > 0 || S0 == 0) amax = __builtin_fmaxf(amax, am);
            B64_T(st, 5);
        }
        B64_T(st, 9);
        b64_barrier();
        B64_T(st, 10);
    }
    range_report(amax, Q.range, lane);
}

// wavefront 5: the block's first filter (chunk s) and its last multiply (chunk s - 6)
__device__ __forceinline__ void b64_v3_ends(const B64Args &A, const int lane, const int S0, const int nch, const int nsteps, const int rows,
                                            uint16_t *P0, const uint16_t *P5, float *r5) {
    const B64Layer &F = A.L[0], &P = A.L[5];
    const int n = lane & 31, hh = lane >> 5;
    const int cpl = lane & 15, rq = lane >> 4;
    u32x4 bw[2][2][2][2];
    b64_load_bw(P, n, hh, bw);
    b64f2 tw[2][5];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int t = 0; t < 5; t++) tw[cb][t] = *reinterpret_cast<const b64f2 *>(F.wd + t * 64 + cb * 32 + 2 * cpl);
    const b64f2 sc = {P.scale[2 * n] * P.post, P.scale[2 * n + 1] * P.post}, sh = {P.shift[2 * n], P.shift[2 * n + 1]};
    const float floor_ = P.relu ? 0.0f : -3.402823466e38f;
    float amax = 0.0f;
    b64u2 xp[2][12];
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X)), 0, rows * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid)), 0, rows, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    const int xlane = ((8 * rq - 4) * 64 + 2 * cpl) * 4;
    auto gloadX = [&](int c, int cb) {
        const int base = xlane + ((S0 - 20 + 32 * c) * 64 + cb * 32) * 4;
#pragma unroll
        for (int j = 0; j < 12; j++) xp[cb][j] = __builtin_bit_cast(b64u2, __builtin_amdgcn_raw_buffer_load_b64(rX, base + j * 256, 0, 0));
    };
    gloadX(0, 0); gloadX(0, 1);
    for (int s = 0; s < nsteps; s++) {
        B64_T(5, 0);
        const int c5 = s - 6;
        const int g5 = S0 - 32 + 32 * c5;                  // first row of layer 5's chunk
        const unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, g5 + n, 0, 0);      // NO branch around it (see b64_stage); outside the pass it reads 0 and nobody looks
        if (!(B64_ABL & 32) && s < nch) {                  // wave-uniform: the first filter, chunk s
            const int og0 = S0 - 22 + 32 * s;
            const bool edge = og0 < 0 || og0 + 32 > rows;
            float am = 0.0f;
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                b64f2 x[12];
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = __builtin_bit_cast(b64f2, xp[cb][j]);
                b64f2 o[8];
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = b64f2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 12; j++) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int t = j - i;
                        if (t >= 0 && t < 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(tw[cb][t]));
                    }
                }
                gloadX(s + 1, cb);                          // (past the stripe: loaded, never used)
                uint16_t *ap = P0 + (s & 1) * (4 * B64_APL) + cb * (2 * B64_APL) + 8 * rq * CNN_BP + 2 * cpl;
                if (!edge) b64_split_store<false>(o, am, ap, 0, 0);
                else { asm volatile("; chunk at an end of the pass" ::: "memory"); b64_split_store<true>(o, am, ap, og0 + 8 * rq, rows); }
            }
            if (s > 0 || S0 == 0) amax = __builtin_fmaxf(amax, am);
        }
        B64_T(5, 1);
        if (!(B64_ABL & 8) && c5 >= 0 && c5 < nch) {       // wave-uniform: the last multiply, chunk s - 6 -> the ring the shortcut joins from
            f32x16 acc[2];
            b64_multiply(P5 + (c5 & 1) * (4 * B64_APL), bw, n, hh, acc);
            const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
            const unsigned vml = hh ? vm >> 4 : vm;
            float *wrow = r5 + (c5 & 1) * B64_RING + 4 * hh * B64_RP + 2 * n;
            if (vm == 0xffffffffu) b64_epilogue<false, false>(acc, sc, sh, floor_, vml, wrow, rY, 0);
            else { asm volatile("; chunk with padding rows" ::: "memory"); b64_epilogue<true, false>(acc, sc, sh, floor_, vml, wrow, rY, 0); }
        }
        B64_T(5, 9);
        b64_barrier();
        B64_T(5, 10);
    }
    range_report(amax, F.range, lane);
}

template <bool CONV>
__global__ __launch_bounds__(512, 2) void k3_block64(const B64Args A) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = min(A.rows, *A.live);
    // stripes: the pass's 32-row chunks dealt to the workgroups in contiguous runs
    const int nct = rows >> 5;
    const int per = (nct + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c_lo = (int)blockIdx.x * per;
    const int mych = min(per, nct - c_lo);
    if (mych <= 0) return;
    const int S0 = c_lo * 32, nch = mych + 1;
    // wavefronts w and w + 4 share a SIMD: roles (0, 1), (2, 3), (4, shortcut 0), (5, shortcut 1)
    const int role = (int)((0x76315420u >> (4 * wave)) & 15u);
    if constexpr (CONV && B64_V3) {
        __shared__ __attribute__((aligned(16))) uint16_t Pl[6][2][4 * B64_APL];        // layer l's planes: [chunk parity][channel block][piece][32 x CNN_BP]
        __shared__ __attribute__((aligned(16))) float ring5[2][B64_RING];              // layer 5's rows, for the shortcut's join
        __shared__ __attribute__((aligned(16))) uint16_t Acv[2][2][2 * B64_CPL];       // the shortcut's planes: [chunk parity][channel block][piece]
        const int nsteps = nch + B64_V3_CONV_START;
        if (role < 5) b64_v3_stage(A, role, lane, S0, nch, nsteps, rows, &Pl[role][0][0], &Pl[role + 1][0][0]);
        else if (role == 5) b64_v3_ends(A, lane, S0, nch, nsteps, rows, &Pl[0][0][0], &Pl[5][0][0], &ring5[0][0]);
        else b64_conv(A, role - 6, B64_V3_CONV_START, lane, S0, nch, nsteps, rows, &ring5[0][0], &Acv[0][0][0]);
    } else {
        __shared__ __attribute__((aligned(16))) float ring[6][2][B64_RING];
        __shared__ __attribute__((aligned(16))) float halo[5][4 * B64_RP];            // stages 1 .. 5 (stage 0 takes its rows from global memory)
        __shared__ __attribute__((aligned(16))) uint16_t Apl[6][2 * B64_APL];
        __shared__ __attribute__((aligned(16))) uint16_t Acv[2][2][2 * B64_CPL];       // the shortcut's planes: [chunk parity][channel block][piece]
        const int nsteps = nch + (CONV ? B64_CONV_START : B64_START(5) + 1);             // (a deferred stage finishes a step after its last chunk: covered)
        for (int i = tid; i < 5 * 4 * B64_RP; i += 512) (&halo[0][0])[i] = 0.0f;
        __syncthreads();
        if (role < 6) {
            const float *rin = &ring[role ? role - 1 : 0][0][0];
            float *rout = &ring[role][0][0];
            if (role == 0) b64_stage<0, CONV, false>(A, 0, B64_START(0), lane, S0, nch, nsteps, rows, rin, rout, &halo[0][0] /* unused */, &Apl[0][0]);
            else if (role == 5) b64_stage<2, CONV, false>(A, 5, B64_START(5), lane, S0, nch, nsteps, rows, rin, rout, &halo[4][0], &Apl[5][0]);
            else if (B64_DEFERS(1) && (role == 1 || role == 3)) b64_stage<1, CONV, true>(A, role, B64_START(role), lane, S0, nch, nsteps, rows, rin, rout, &halo[role - 1][0], &Apl[role][0]);
            else b64_stage<1, CONV, false>(A, role, B64_START(role), lane, S0, nch, nsteps, rows, rin, rout, &halo[role - 1][0], &Apl[role][0]);
        } else if (CONV) {
            b64_conv(A, role - 6, B64_CONV_START, lane, S0, nch, nsteps, rows, &ring[5][0][0], &Acv[0][0][0]);
        } else {
            for (int s = 0; s < nsteps; s++) b64_barrier();
        }
    }
}
