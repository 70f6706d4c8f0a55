// k3_conv_ws_experiment.h -- EXPERIMENT (round 5), not part of libdnascent_hip.so: included by tools/k3_conv_ws_check.hip behind k3_cnn.hip (needs its f32x16 /
// u32x4 / mfma16 / range_report / split_pair / CNN_BP and k3_block64.h's b64_barrier / b64_uniform_ptr).
//
// k3_conv_ws: a Conv1D (KW taps, CIN -> cout, folded BatchNorm, optional residual add, ReLU) WEIGHT-STATIONARY.  k3_conv_split streams a fresh weight tile from
// L2 through LDS every step and reads two fragments from LDS per MFMA; its matrix pipe is 60-70 % busy.  Here, as for the shortcut tiles of k3_block64, the
// weights never move: a workgroup (one per CU, 8 wavefronts) owns ONE 32-column slice of the output and keeps that slice's whole kernel -- KW x CIN x 32 values as
// fp16 hi / lo B fragments -- in its REGISTERS, split along K: the (tap, channel block) pairs are dealt round-robin to the 8 wavefronts (17 x 128 -> 256: 68 pairs,
// 9 or 8 per wavefront, 144 registers).  The workgroup walks a stripe of the pass's rows in 32-row chunks, ONE barrier per step; in step s every wavefront
//   multiplies chunk s's 32 rows by ITS pairs (6 MFMAs per pair, A fragments from the fp16 planes of the input rows in LDS) into one 32 x 32 accumulator and
//     writes it to LDS as a partial sum (double-buffered by chunk parity);
//   adds the eight partial sums of rows 4 w .. 4 w + 3 of chunk s - 1 (fixed order 0 .. 7), applies BatchNorm / residual / ReLU / padding mask, stores them;
//   splits the 32 input rows chunk s + 1 adds to the window into the planes (a 128-row ring) and requests chunk s + 2's.
// cout / 32 workgroups share a stripe (one per column slice); they sit on the same XCD (blockIdx % 8), so the input rows they all read come from its L2.
// NOT bit-identical to k3_conv_split: a product's three MFMAs keep their order, but K is summed per wavefront and then across wavefronts (conv_split: one chain).
//
// RESULT (gpurun_out/r6a, 1.2 M rows): within 1.05e-5 of k3_conv_split on every shape, and SLOWER: 17 x 128 -> 256 + add 3 308-3 389 us against 3 113-3 240 (same
// process), 9 x 128 -> 128 1 156-1 201 against 852-1 020, 9 x 64 -> 128 723-766 against 506-511.  Its stamps (-DCW_TRACE): a step is 5 500-5 700 ticks for 3 264 of
// MFMAs per SIMD; the two wavefronts of a SIMD issue their 102 MFMAs in ~4 300-4 650 ticks (42-46 per MFMA) and the later one's reduction, split and partial-sum
// traffic (~1 100 ticks) is hidden behind nothing.  What was tried on that, all slower: the reduction as its own phase between two barriers (3 805 us with the
// residual's load inside it, 3 324 with that prefetched); the fragment reads of the next group pinned in front of a group's MFMAs (-DCW_PIN=1: 3 446 -- the ISA is
// exactly tools/ubench_mfma_lds.hip's loop, which runs at 35 cycles per MFMA bare, on one wavefront per SIMD or two); two barriers per step with the two wavefronts
// of a SIMD in antiphase, one multiplying while the other reduces / splits / stores (-DCW_ANTIPHASE=1: 3 649-3 706; stamps: the wavefront that multiplies ALONE on
// its SIMD needs 3 200-3 600 ticks for its 54 MFMAs, 60-66 each, pinned reads or not -- unexplained: not the fragment latency, not the LDS traffic of the others, which
// are waiting at the barrier by then).  Weight-stationary convolutions do not beat the streamed-weight kernel here; recorded, not pursued.
#pragma once

#define CW_RING 128                                         // rows of the plane ring: a chunk's window (32 + KW - 1 <= 48) + the 32 rows split for the next chunk while it is read
#define CW_PP 34                                            // pitch (floats) of a partial tile's row

#ifdef CW_TRACE                                              /* experiment builds of tools/k3_conv_ws_check.hip (-DCW_TRACE=<workgroup>): shader-clock stamps of one step's phases */
__device__ unsigned long long cw_trace[8][8];
#define CW_T(i) do { if (blockIdx.x == CW_TRACE && c == 40) { __builtin_amdgcn_sched_barrier(0); if (lane == 0) cw_trace[wave][i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CW_T(i) do { } while (0)
#endif

struct CwArgs {
    const float *X; float *Y; const float *Add;             // [row][CIN], [row][cout], [row][cout]
    const uint8_t *valid; const int *live; int rows; int cout;
    const uint16_t *w;                                      // pre-split [channel block][tap][piece][cout][32]
    const float *scale, *shift; unsigned *range; float post; int relu;
};

template <int KW, int CIN, bool ADD>
__global__ __launch_bounds__(512, 2) void k3_conv_ws(const CwArgs A) {
    constexpr int CB = CIN / 32, NPAIR = KW * CB, NPW = (NPAIR + 7) / 8, half = (KW - 1) / 2;
    constexpr int PLANE = CW_RING * CNN_BP;                 // elements of one (channel block, piece) plane
    constexpr int NLD = CIN / 64;                           // float4 per thread and 32 staged rows
    __shared__ __attribute__((aligned(16))) uint16_t Pl[CB][2][PLANE];
    __shared__ __attribute__((aligned(16))) float Ps[2][8][32 * CW_PP];          // partial sums, double-buffered by chunk parity
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = min(A.rows, *A.live), cout = A.cout, nsl = cout >> 5;
    // workgroup -> (stripe, column slice): the slices of a stripe on one XCD
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int slice = idx % nsl, stripe = xcd * (per_xcd / nsl) + idx / nsl, nstripes = 8 * (per_xcd / nsl);
    if (idx / nsl >= per_xcd / nsl) return;
    const int nct = rows >> 5, per = (nct + nstripes - 1) / nstripes, c_lo = stripe * per, nch = min(per, nct - c_lo);
    if (nch <= 0) return;
    const int S0 = c_lo * 32;
    // ---- this wavefront's share of the kernel, resident ----
    u32x4 bw[NPW][2][2];                                    // [pair][k16][piece]
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int p = wave + 8 * i;                         // pair = channel block * KW + tap (the layout's own order)
        if (p < NPAIR) {
#pragma unroll
            for (int k16 = 0; k16 < 2; k16++)
#pragma unroll
                for (int pc = 0; pc < 2; pc++)
                    bw[i][k16][pc] = *reinterpret_cast<const u32x4 *>(A.w + ((size_t)(p * 2 + pc) * cout + slice * 32 + n) * 32 + k16 * 16 + 8 * hh);
        }
    }
    const int rr = lane >> 4, c2 = (lane & 15) * 2;        // phase 2: row 4 wave + rr, columns c2, c2 + 1 of the slice
    const int col = slice * 32 + c2;
    const float sc0 = A.scale[col] * A.post, sc1 = A.scale[col + 1] * A.post, sh0 = A.shift[col], sh1 = A.shift[col + 1];
    const float floor_ = A.relu ? 0.0f : -3.402823466e38f;
    // buffer descriptors over THIS STRIPE's rows (a whole pass can exceed the 4 GB a descriptor spans: 4 Mi rows x 256 channels x 4 B); rows outside the pass
    // fall outside them and read as the zeros 'same' padding wants
    const int xb = max(0, S0 - CW_RING), xrows = min(rows, S0 + 32 * nch + CW_RING + 32) - xb;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.X + (size_t)xb * CIN)), 0, xrows * CIN * 4, 0x00020000);
    const int yrows = min(rows - S0, 32 * nch);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.Y + (size_t)S0 * cout)), 0, yrows * cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(ADD ? (const void *)(A.Add + (size_t)S0 * cout) : (const void *)(A.Y + (size_t)S0 * cout))), 0, yrows * cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b64_uniform_ptr(A.valid + S0)), 0, yrows, 0x00020000);
    float amax = 0.0f;
    // ---- staging: 32 rows x CIN floats = 8 CIN float4, thread t takes t + 512 j: row f / (CIN / 4), float4 f % (CIN / 4).  Rows outside the pass read as zeros. ----
    f32x4 xr[NLD];
    auto gload = [&](int r0) {                              // rows r0 .. r0 + 31
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int f = tid + 512 * j, r = f / (CIN / 4), q = f % (CIN / 4);
            xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, ((r0 + r - xb) * CIN + q * 4) * 4, 0, 0));
        }
    };
    auto lstore = [&](int r0, int nrows) {                  // the loaded rows -> planes (the first nrows of them)
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int f = tid + 512 * j, r = f / (CIN / 4), q = f % (CIN / 4);
            if (r < nrows) {
                sp16x2 ha, la, hb, lb;
                split_pair(xr[j][0], xr[j][1], ha, la, amax); split_pair(xr[j][2], xr[j][3], hb, lb, amax);
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const int slot = (r0 + r + CW_RING) & (CW_RING - 1), cb = q >> 3, k = (q & 7) * 4;
                *reinterpret_cast<h4 *>(&Pl[cb][0][slot * CNN_BP + k]) = h4{ha[0], ha[1], hb[0], hb[1]};
                *reinterpret_cast<h4 *>(&Pl[cb][1][slot * CNN_BP + k]) = h4{la[0], la[1], lb[0], lb[1]};
            }
        }
    };
    // prologue: the first chunk's window, rows S0 - half .. S0 + 32 + half
    gload(S0 - half); lstore(S0 - half, 32);
    gload(S0 - half + 32); lstore(S0 - half + 32, 2 * half);
    gload(S0 + 32 + half);                                  // the 32 rows chunk 1 needs beyond that (kept in registers until step 0 splits them)
    b64_barrier();
    // ONE barrier per step.  Step s: the MFMAs of chunk s; the reduction + epilogue of chunk s - 1 (its partial sums were written before the last barrier,
    // into the other buffer); the split of the rows chunk s + 1 adds to the window (a 128-row ring: they replace rows two chunks back); the request for
    // chunk s + 2's.  None of the three depends on this step's MFMAs, so they issue in their shadow -- the first version ran them as a second phase between two
    // barriers, with the matrix pipe idle: 5 700 cycles per step for 3 300 of MFMAs.
    typedef float f2 __attribute__((ext_vector_type(2)));
    unsigned vb = 0; f2 addv = {0.f, 0.f};
    for (int c = 0; c <= nch; c++) {
        const int G = S0 + 32 * c;
        CW_T(0);
        // ---- chunk c - 1: the eight partial sums of rows 4 wave .. 4 wave + 3 (reads first: they are long back when the adds issue) ----
        const int row = 4 * wave + rr;
        const unsigned vb_prev = vb; const f2 add_prev = addv;
        const int grow = 32 * c + row;                        // relative to the stripe (the descriptors start there)
        if (c < nch) {                                      // chunk c's validity byte and residual: requested now, used a step later
            vb = __builtin_amdgcn_raw_buffer_load_b8(rV, grow, 0, 0);
            if (ADD) addv = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rA, (grow * cout + col) * 4, 0, 0));
        }
        // ---- chunk c: this wavefront's pairs ----
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#ifndef CW_PIN
#define CW_PIN 0                                            /* (3 446 us pinned against 3 308-3 389 unpinned: see the header) 1: the four A fragments of pair i + 1 are requested before pair i's six MFMAs issue (pinned with sched_barrier); 0: left to the scheduler, which issues every pair of reads right in front of the MFMA that waits for them */
#endif
        // the A fragments of group g + 1 (a group = one k16 half of a pair: 2 fragment reads, 3 MFMAs) are requested BEFORE group g's MFMAs issue (CW_PIN: pinned with
        // sched_barrier).  Left to the scheduler every pair of reads sits right in front of the MFMA that waits for it: a wavefront alone on its SIMD then needs 66 cycles
        // per MFMA (stamps), where the bare loop of tools/ubench_mfma_lds.hip needs 35 with the reads one group ahead.
        u32x4 fa[2][2];                                     // [buffer][piece: 0 = hi, 1 = lo]
        auto frags = [&](int g, u32x4 (&f)[2]) {
            const int p = wave + 8 * (g >> 1), k16 = g & 1;
            const int cb = p / KW, tap = p % KW;
            const int slot = (G - half + tap + n + CW_RING) & (CW_RING - 1);
            const uint16_t *ap = &Pl[0][0][0] + cb * (2 * PLANE) + slot * CNN_BP + 8 * hh + k16 * 16;
            f[0] = *reinterpret_cast<const u32x4 *>(ap);
            f[1] = *reinterpret_cast<const u32x4 *>(ap + PLANE);
        };
        auto mul = [&](int g, const u32x4 (&f)[2]) {
            const int i = g >> 1, k16 = g & 1;
            acc = mfma16<2>(f[1], bw[i][k16][0], acc);
            acc = mfma16<2>(f[0], bw[i][k16][1], acc);
            acc = mfma16<2>(f[0], bw[i][k16][0], acc);
        };
        constexpr int NFULL = NPAIR / 8;                    // pairs every wavefront has; the first NPAIR % 8 wavefronts have one more
        const bool extra = NPAIR % 8 && wave < NPAIR % 8;
        auto multiply = [&]() {
            if (c < nch) {
                frags(0, fa[0]);
#pragma unroll
                for (int g = 0; g < 2 * NFULL; g++) {
                    if (g + 1 < 2 * NFULL) frags(g + 1, fa[(g + 1) & 1]);
                    else if (extra) frags(2 * NFULL, fa[0]);
                    if (CW_PIN) __builtin_amdgcn_sched_barrier(0);
                    mul(g, fa[g & 1]);
                    if (CW_PIN) __builtin_amdgcn_sched_barrier(0);
                }
                if (extra) { frags(2 * NFULL + 1, fa[1]); if (CW_PIN) __builtin_amdgcn_sched_barrier(0); mul(2 * NFULL, fa[0]); mul(2 * NFULL + 1, fa[1]); }
                float *pw = &Ps[c & 1][wave][(4 * hh) * CW_PP + n];
#pragma unroll
                for (int q = 0; q < 16; q++) pw[((q & 3) + 8 * (q >> 2)) * CW_PP] = acc[q];
            }
        };
        auto post = [&]() {
            if (c > 0) {                                    // chunk c - 1: sum in the order 0 .. 7, epilogue, store
                f2 pv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) pv[u] = *reinterpret_cast<const f2 *>(&Ps[(c - 1) & 1][u][row * CW_PP + c2]);
                f2 v = pv[0];
#pragma unroll
                for (int u = 1; u < 8; u++) v += pv[u];
                float y0 = __builtin_fmaf(v[0], sc0, sh0), y1 = __builtin_fmaf(v[1], sc1, sh1);
                if (ADD) { y0 += add_prev[0]; y1 += add_prev[1]; }
                y0 = __builtin_fmaxf(y0, floor_); y1 = __builtin_fmaxf(y1, floor_);
                const bool ok = (vb_prev & 0xffu) != 0;
                y0 = ok ? y0 : 0.0f; y1 = ok ? y1 : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64u2, f2{y0, y1}), rY, ((grow - 32) * cout + col) * 4, 0, 0);
            }
            if (c + 1 < nch) {                              // the next chunk's new rows: G + 32 + half .. G + 64 + half
                lstore(G + 32 + half, 32);
                gload(G + 64 + half);                       // (past the stripe: loaded, never used)
            }
        };
#ifndef CW_ANTIPHASE
#define CW_ANTIPHASE 0                                      /* (3 649-3 706 us: a wavefront alone on its SIMD multiplies at 60-66 cycles per MFMA, see the header) 1: TWO barriers per step, and the two wavefronts of a SIMD (w, w + 4) take its halves in opposite order -- one multiplies while the other reduces the previous chunk, stores it and splits the next rows; 0: one barrier, every wavefront multiplies, then does the rest */
#endif
        if (CW_ANTIPHASE) {
            if (wave < 4) { multiply(); CW_T(1); b64_barrier(); CW_T(2); post(); CW_T(3); }
            else { post(); CW_T(1); b64_barrier(); CW_T(2); multiply(); CW_T(3); }
        } else { multiply(); CW_T(1); post(); CW_T(3); }
        CW_T(4);
        b64_barrier();
        CW_T(5);
    }
    range_report(amax, A.range, lane);
}
