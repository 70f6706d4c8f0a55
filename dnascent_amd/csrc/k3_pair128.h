// k3_pair128.h -- included by k3_cnn.hip behind k3_block64.h (uses its b64_* helpers: barrier, swap, epilogue-into-registers, split_store).
//
// k3_pair128: TWO consecutive SeparableConv1D layers of the 128-channel stage (9 taps; CIN0 -> 128 -> 128, folded BatchNorm, ReLU; SURVEY s2.3 layers 32-55: the
// six + six separable layers of the two 128-channel residual blocks) in ONE launch.  Layer by layer each of them streams the pass through HBM at 4.7 TB/s
// (k3_sep_split: 1 KB per position and layer, the bound of those kernels); here the first layer's output never leaves the chip.  It is k3_block64's third version
// (the depthwise filter in the accumulator layout) at 128 channels -- where it pays, because what it replaces is HBM-bound, not already fused:
//   * one persistent 256-thread workgroup per CU -- FOUR wavefronts, one per SIMD, each with the whole 512-register file: a wavefront keeps HALF a layer's
//     pointwise matrix (128 x 64 as fp16 hi / lo B fragments = 128 registers) resident for the whole launch;
//   * wavefronts 0, 1 = layer 0, columns 0-63 / 64-127: the chunk's 32 rows x CIN0 channels (fp16 planes in LDS) x their 64 columns -> BatchNorm / ReLU / mask ->
//     v_permlane32_swap exchange (each lane half gets 24 consecutive rows of its channel pair: 16 of its own, 8 from the other half or from the chunk before) ->
//     layer 1's 9-tap filter on 16 output rows per half -> split -> layer 1's planes (their 64 channels of them);
//   * wavefronts 2, 3 = layer 1, columns 0-63 / 64-127: planes -> 48 MFMAs -> BatchNorm / ReLU / mask -> global memory; and layer 0's FILTER for half of the input
//     channels each: rows from global memory (requested two chunks ahead, in registers) -> 9 taps -> split -> layer 0's planes;
//   * step s: layer 0's filter works on chunk s, layer 0's matrix on chunk s - 1, layer 1's on chunk s - 2; one LDS-only barrier per step; layer l's chunk grid is
//     shifted up by 4 (l + 1) rows; a stripe pays one warm-up chunk.
// Arithmetic, operand splits, K order, epilogue expressions and masks are k3_sep_split's: bit-identical to the two launches it replaces (tools/k3_pair128_check.hip).
#pragma once

struct P128Layer {
    const float *wd;                                        // depthwise taps [9][cin]
    const uint16_t *wb;                                     // pointwise weights, pre-split fp16 pieces [channel block][piece][128][32]
    const float *scale, *shift;
    unsigned *range;
    float post; int relu;
};
struct P128Args {
    const float *X; float *Y;                               // [row][CIN0], [row][128]
    const uint8_t *valid; const int *live; int rows; int pad_;
    P128Layer L[2];
};

#ifdef P128_TRACE
__device__ unsigned long long p128_trace[4][8];
#define P128_T(i) do { if (blockIdx.x == P128_TRACE && s == 40) { __builtin_amdgcn_sched_barrier(0); if (lane == 0) p128_trace[wave][i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define P128_T(i) do { } while (0)
#endif

// the MFMAs of one chunk: planes [CB][piece][32 x CNN_BP] x resident B fragments [CB][k16][piece][column tile].  One wavefront per SIMD: nobody covers an LDS wait,
// so the two A fragments of group g + 1 (a group = one k16 half of a channel block: 2 reads, 6 MFMAs) are requested BEFORE group g's MFMAs issue (P128_PIN: pinned
// with sched_barrier; left to the scheduler the reads sit right in front of the MFMA that waits for them: 42 cycles per MFMA in the first version's stamps)
#ifndef P128_PIN
#define P128_PIN 1
#endif
template <int CB>
__device__ __forceinline__ void p128_multiply(const uint16_t *Pb, const u32x4 (&bw)[4][2][2][2], const int n, const int hh, f32x16 (&acc)[2]) {
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
        if (P128_PIN) __builtin_amdgcn_sched_barrier(0);
        const int cb = g >> 1, k16 = g & 1;
        const u32x4 ah = fa[g & 1][0], al = fa[g & 1][1];
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(al, bw[cb][k16][0][j], acc[j]);
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][1][j], acc[j]);
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = mfma16<2>(ah, bw[cb][k16][0][j], acc[j]);
        if (P128_PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

// the last layer's epilogue: folded BatchNorm, ReLU, padding mask -> global memory, [row][128] fp32 (b64_epilogue's expressions; 512-byte rows)
template <bool MASKED>
__device__ __forceinline__ void p128_store(const f32x16 (&acc)[2], const b64f2 sc, const b64f2 sh, const float floor_, const unsigned vml, const __amdgpu_buffer_rsrc_t rY, const int ybase) {
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
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{a0, a1}), rY, ybase + rowq * 512, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, b64f2{b0, b1}), rY, ybase + (rowq + 1) * 512, 0, 0);
    }
}

template <int CIN0>
__global__ __launch_bounds__(256) void k3_pair128(const P128Args A) {
    constexpr int CB0 = CIN0 / 32;
    __shared__ __attribute__((aligned(16))) uint16_t P0[2][CB0][2 * B64_APL];      // layer 0's planes: [chunk parity][channel block][piece][32 x CNN_BP]
    __shared__ __attribute__((aligned(16))) uint16_t P1[2][4][2 * B64_APL];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = min(A.rows, *A.live);
    const int nct = rows >> 5, per = (nct + (int)gridDim.x - 1) / (int)gridDim.x, c_lo = (int)blockIdx.x * per, mych = min(per, nct - c_lo);
    if (mych <= 0) return;
    const int S0 = c_lo * 32, nch = mych + 1, nsteps = nch + 2;
    const int h = wave & 1;                                 // which 64 columns (layer wavefronts) / which half of the input channels (filter role)
    const int lyr = wave >> 1;
    const P128Layer &P = A.L[lyr];
    // ---- resident operands of this wavefront's layer: columns 64 h + 2 n + j ----
    u32x4 bw[4][2][2][2];
    const int CBL = lyr == 0 ? CB0 : 4;
#pragma unroll
    for (int cb = 0; cb < 4; cb++)
        if (cb < CBL) {
#pragma unroll
            for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
                for (int pc = 0; pc < 2; pc++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        bw[cb][k16][pc][j] = *reinterpret_cast<const u32x4 *>(P.wb + ((size_t)((cb * 2 + pc) * 128 + 64 * h + 2 * n + j)) * 32 + k16 * 16 + 8 * hh);
        }
    const int colp = 64 * h + 2 * n;
    const b64f2 sc = {P.scale[colp] * P.post, P.scale[colp + 1] * P.post}, sh = {P.shift[colp], P.shift[colp + 1]};
    const float floor_ = P.relu ? 0.0f : -3.402823466e38f;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid)), 0, rows, 0x00020000);
    float amax = 0.0f;
    if (lyr == 0) {
        // =========================== layer 0: matrix + epilogue, then layer 1's filter in the accumulator layout ===========================
        b64f2 tw[9];                                        // layer 1's taps of channels 64 h + 2 n, + 1
#pragma unroll
        for (int t = 0; t < 9; t++) tw[t] = *reinterpret_cast<const b64f2 *>(A.L[1].wd + t * 128 + colp);
        b64f2 carry[8];                                     // this layer's rows 24 .. 31 of the chunk before (upper lane half)
#pragma unroll
        for (int i = 0; i < 8; i++) carry[i] = b64f2{0.f, 0.f};
        // the validity bytes of a chunk's rows are requested a WHOLE STEP before the epilogue votes on them (asked for at the top of the chunk's own step they came back
        // ~2 000 ticks after the MFMAs had finished: the first version's stamps showed the epilogue "taking" 2 030 ticks)
        unsigned vb_next = __builtin_amdgcn_raw_buffer_load_b8(rV, S0 - 28 + n, 0, 0);
        for (int s = 0; s < nsteps; s++) {
            const int c = s - 1;
            P128_T(0);
            const unsigned vb = vb_next;
            vb_next = __builtin_amdgcn_raw_buffer_load_b8(rV, S0 - 28 + 32 * (c + 1) + n, 0, 0);      // (outside the pass it reads 0 and nobody looks)
            if (c >= 0 && c < nch) {                        // wave-uniform
                const int og0 = S0 - 28 + 32 * c;           // first row of layer 0's chunk; layer 1's starts four rows earlier
                const bool edge = og0 - 4 < 0 || og0 + 28 > rows;
                f32x16 acc[2];
                p128_multiply<CB0>(&P0[c & 1][0][0], bw, n, hh, acc);
                P128_T(1);
                const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
                const unsigned vml = hh ? vm >> 4 : vm;
                b64f2 Y[16];
                if (vm == 0xffffffffu) b64_epilogue_regs<false>(acc, sc, sh, floor_, vml, Y);
                else { asm volatile("; chunk with padding rows" ::: "memory"); b64_epilogue_regs<true>(acc, sc, sh, floor_, vml, Y); }
                P128_T(6);
                // ---- 24 consecutive rows per lane: W[w] = row w - 8 (lower half) / row w + 8 (upper half) of the chunk ----
                b64f2 W[24];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    b64f2 a = Y[i], b = Y[8 + i]; b64_swap(a, b); W[8 + i] = a; W[12 + i] = b;              // a: rows i | 16 + i,      b: rows 4 + i | 20 + i
                    b64f2 a2 = Y[4 + i], b2 = Y[12 + i]; b64_swap(a2, b2); W[16 + i] = a2; W[20 + i] = b2; // a2: rows 8 + i | 24 + i, b2: rows 12 + i | 28 + i
                }
                P128_T(7);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    b64f2 p = carry[i], d = W[16 + i];
                    b64_swap(p, d);                         // p's upper half <- this chunk's rows 8 + i; d's lower half <- the previous chunk's rows 24 + i
                    W[i] = hh ? p : d;
                    carry[i] = W[16 + i];
                }
                P128_T(2);
                // ---- layer 1's filter, taps ascending: output k = sum_t W[k + t] w[t] = row k - 4 (lower half) / k + 12 (upper half) of layer 0's grid = plane row k / 16 + k ----
                b64f2 o[2][8];
#pragma unroll
                for (int g = 0; g < 2; g++)
#pragma unroll
                    for (int i = 0; i < 8; i++) o[g][i] = b64f2{0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 24; w++) {
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int t = w - k;
                        if (t >= 0 && t < 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[k >> 3][k & 7]) : "v"(W[w]), "v"(tw[t]));
                    }
                }
                P128_T(3);
                float am = 0.0f;
                uint16_t *ap = &P1[c & 1][2 * h + (n >> 4)][0] + 16 * hh * CNN_BP + 2 * (n & 15);
                if (!edge) { b64_split_store<false>(o[0], am, ap, 0, 0); b64_split_store<false>(o[1], am, ap + 8 * CNN_BP, 0, 0); }
                else {
                    asm volatile("; chunk at an end of the pass" ::: "memory");
                    b64_split_store<true>(o[0], am, ap, og0 - 4 + 16 * hh, rows); b64_split_store<true>(o[1], am, ap + 8 * CNN_BP, og0 - 4 + 16 * hh + 8, rows);
                }
                if (c > 0 || S0 == 0) amax = __builtin_fmaxf(amax, am);
                P128_T(4);
            }
            b64_barrier();
            P128_T(5);
        }
        range_report(amax, A.L[1].range, lane);
    } else {
        // =========================== layer 1: matrix + epilogue -> global memory; and layer 0's filter for half of the input channels ===========================
        constexpr int NCBF = CB0 / 2;                       // channel blocks this wavefront filters: CB0 / 2 of them, from block h * NCBF
        const int cpl = lane & 15, rq = lane >> 4;          // filter role: channel pair cpl of the channel block, row quarter rq (8 output rows)
        b64f2 tw[NCBF][9];
#pragma unroll
        for (int f = 0; f < NCBF; f++)
#pragma unroll
            for (int t = 0; t < 9; t++) tw[f][t] = *reinterpret_cast<const b64f2 *>(A.L[0].wd + t * CIN0 + (h * NCBF + f) * 32 + 2 * cpl);
        // input rows: lane (cpl, rq) of channel block f needs rows XC - 8 + 8 rq + j, j = 0 .. 15 (XC = S0 - 28 + 32 c + 4 = first row of layer 0's chunk + 4 ... see below);
        // requested TWO chunks ahead (three register sets): ~40 KB per CU in flight, what the memory needs to stream (Little's law at ~2 us)
        // descriptors over THIS STRIPE's rows (byte offsets stay far below 2^31 whatever the pass size); rows outside the pass fall outside them: zeros / dropped stores
        const int xb = max(0, S0 - 64), xrows = min(rows, S0 + 32 * nch + 96) - xb;
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X + (size_t)xb * CIN0)), 0, xrows * CIN0 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.Y + (size_t)S0 * 128)), 0, min(rows - S0, 32 * mych) * 512, 0x00020000);
#ifndef P128_SETS
#define P128_SETS 2                                         /* register sets of input rows = chunks requested ahead + 1.  Three would cover more latency, but 2 x 32 newer loads + the stores exceed what s_waitcnt vmcnt can count (63): the wait for a set then also waits for the NEXT one's first load (measured: 440 us with 3, 432 with 2) */
#endif
        b64u2 xp[P128_SETS][NCBF][16];
        // dw_0's output chunk c = rows [S0 - 28 + 32 c, + 32) (layer 0's grid); output row 8 rq + i needs input rows 8 rq + i - 4 .. + 4: j = i + t, t = 0 .. 8, from row og - 4 + 8 rq
        const int xlane = ((8 * rq - 4) * CIN0 + 2 * cpl) * 4;
        auto gloadX = [&](int c, int set) {
            const int base = xlane + ((S0 - 28 + 32 * c - xb) * CIN0 + h * NCBF * 32) * 4;
#pragma unroll
            for (int f = 0; f < NCBF; f++)
#pragma unroll
                for (int j = 0; j < 16; j++) xp[set][f][j] = __builtin_bit_cast(b64u2, __builtin_amdgcn_raw_buffer_load_b64(rX, base, f * 128 + j * CIN0 * 4, 0));      // the row / block part as the SCALAR offset: one lane offset for the 32 loads (as immediates they do not fit 12 bits: 41 vector adds per step)
        };
        gloadX(0, 0); if (P128_SETS == 3) gloadX(1, 1);
        float amax0 = 0.0f;
        // one step; SET = s % 3 is a compile-time constant (the three register sets rotate; the loop below is unrolled by three)
        auto step = [&](const int s, auto SETC) {
            constexpr int SET = decltype(SETC)::value;
            P128_T(0);
            const int c1 = s - 2;
            const int g1 = S0 - 32 + 32 * c1;               // first row of layer 1's chunk
            const unsigned vb = __builtin_amdgcn_raw_buffer_load_b8(rV, g1 + n, 0, 0);      // no branch around it (see b64_stage); outside the pass it reads 0 and nobody looks
            if (s < nch) {                                  // wave-uniform: layer 0's filter, chunk s
                gloadX(s + P128_SETS - 1, (SET + P128_SETS - 1) % P128_SETS);      // (past the stripe: loaded, never used)
                const int og0 = S0 - 28 + 32 * s;
                const bool edge = og0 < 0 || og0 + 32 > rows;
                float am = 0.0f;
#pragma unroll
                for (int f = 0; f < NCBF; f++) {
                    b64f2 x[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) x[j] = __builtin_bit_cast(b64f2, xp[SET][f][j]);
                    b64f2 o[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) o[i] = b64f2{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 16; j++) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int t = j - i;
                            if (t >= 0 && t < 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(tw[f][t]));
                        }
                    }
                    uint16_t *ap = &P0[s & 1][h * NCBF + f][0] + 8 * rq * CNN_BP + 2 * cpl;
                    if (!edge) b64_split_store<false>(o, am, ap, 0, 0);
                    else { asm volatile("; chunk at an end of the pass" ::: "memory"); b64_split_store<true>(o, am, ap, og0 + 8 * rq, rows); }
                }
                if (s > 0 || S0 == 0) amax0 = __builtin_fmaxf(amax0, am);
            }
            P128_T(1);
            if (c1 >= 1 && c1 < nch) {                      // wave-uniform: layer 1's matrix, chunk s - 2 -> global memory (chunk 0 is the stripe's warm-up chunk)
                f32x16 acc[2];
                p128_multiply<4>(&P1[c1 & 1][0][0], bw, n, hh, acc);
                P128_T(2);
                const unsigned vm = (unsigned)__ballot((vb & 0xffu) != 0);
                const unsigned vml = hh ? vm >> 4 : vm;
                const int ybase = ((g1 - S0 + 4 * hh) * 128 + colp) * 4;
                if (vm == 0xffffffffu) p128_store<false>(acc, sc, sh, floor_, vml, rY, ybase);
                else { asm volatile("; chunk with padding rows" ::: "memory"); p128_store<true>(acc, sc, sh, floor_, vml, rY, ybase); }
            }
            P128_T(4);
            b64_barrier();
            P128_T(5);
        };
        for (int s = 0; s < nsteps; s += P128_SETS) {
            step(s, std::integral_constant<int, 0>{});
            if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>{});
            if (P128_SETS == 3 && s + 2 < nsteps) step(s + 2, std::integral_constant<int, 2 % P128_SETS>{});
        }
        range_report(amax0, A.L[0].range, lane);
    }
}
