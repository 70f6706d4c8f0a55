// k3_sep_ts_experiment.hip -- NEGATIVE RESULT kept for the record (round 3).  Not built into the library: this is the kernel text as it
// sat in dnascent_amd/csrc/k3_cnn.hip behind k3_sep_ws (it uses that file's helpers: conv_epilogue, mfma16, CNN_BM, WS_T) together with
// its launch line:
//     hipLaunchKernelGGL((k3_sep_ts<17, ADD>), dim3(min(conv_grid(rows, o.cout, 256), k3_cu_count())), dim3(512), 0, st, SEP_ARGS);
// Measured on MI355X, 64 x 20 kb reads (1.2 M positions), the five 17-tap 256 -> 256 layers, every version bit-identical to k3_sep_ws
// (tools/variant_check.py digest); gpurun_out/r3u2, r3v2, r3w2, r3x2:
//     k3_sep_ws (wave-specialised, the product kernel)                                       3 810-3 940 us
//     rows requested BEFORE the weight fragments, taps prefetched in registers (9 spills)    5 365 us   (every weight fragment waited for 32 rows
//                                                                                                        from HBM: a wavefront's memory counter is in order)
//     rows requested last, taps in LDS, fragments 2 steps ahead, 64 x 64 wavefront tiles     4 464 us
//     fragments 4 steps ahead                                                                 4 520 us
//     128 x 32 wavefront tiles (each weight fragment fetched by ONE wavefront of the CU)      4 139 us
//     + MFMA order pinned (sched_barrier), row fragments one step ahead                       4 018 us   <- this file
//     in the pipeline (bench.py, 12 steps, A B A B): 717 / 721 Msamples/s against 733 / 736 with k3_sep_ws
// Phase trace of the last version (ticks; third tile of a workgroup): filter 2.7-4.3 k per 128 channels (the model said 2.3 k: the vector pipe
// does run at its full rate with no MFMA in flight), barrier waits 2-3 k, multiply 12-18 k per 128 channels where the MFMAs need 6.1 k,
// epilogue 5.5-8 k.  Ablations (wrong results, timing only): without the weight loads 3 498 us, without the LDS fragment reads 3 693, without
// both 3 283, without the filter and its row loads 2 446.  What the wave-specialised kernel hides by running its roles side by side --
// L2 latency of the weight fragments, HBM latency of the rows, LDS latency of the fragments -- is exposed here in every phase.
// ---------------------------------------------------------------------------------------------------------
// k3_sep_ts (round 3): the 17-tap 256-column separable layer TIME-SLICED instead of wave-specialised.  k3_sep_ws runs its filter beside
// its MFMAs, and beside a busy matrix pipe a wavefront issues one vector instruction every 16-25 ticks (tools/ubench_coissue.hip,
// tools/ws_trace.py): the producers' 1 088 packed FMAs per tile set the pace, 41 k ticks per tile where the matrix work needs 12 k
// and HBM ~28 k.  Here ALL eight wavefronts filter (no MFMA in flight: the vector pipe at its full rate, 2.8 ticks per instruction
// with two wavefronts per SIMD), then ALL eight multiply, in LONG phases -- 128 input channels at a time, four barriers per tile:
//   filter   wavefront w = output rows 16 w .. 16 w + 15 of the tile, lane = a channel PAIR of the 128: its 32 input rows come
//            straight from global memory as 8-byte loads (512 contiguous bytes per wavefront and row; no raw tile in LDS), requested
//            during the PREVIOUS phase; 272 v_pk_fma_f32 per lane, taps in ascending order (bit-identical to k3_dwconv), split into
//            the two fp16 planes [128 rows][128 channels] in LDS (pitch 136: conflict-free 16-byte fragment reads)
//   multiply 8 k16 steps over those planes; wavefront = 64 rows x 64 columns (the shape of the 256-row convolution), weight fragments
//            straight from L2 three steps ahead
// Same sums in the same order as k3_sep_ws: bit-identical (tools/variant_check.py).
// ---------------------------------------------------------------------------------------------------------
#ifndef TS_ABL
#define TS_ABL 0
#endif
#define TS_HK 128                                          // input channels per phase
#define TS_PP (TS_HK + 8)                                  // plane pitch (16-bit elements)
template <int KW, bool ADD>
__global__ __launch_bounds__(512) void k3_sep_ts(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                 const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, const float *__restrict__ Add,
                                                 const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                 unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int NP = 2, WR = 16, NX = WR + KW - 1;       // output rows / input rows of a wavefront's filter slice
    constexpr int half = (KW - 1) / 2;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) uint16_t As[NP][CNN_BM * TS_PP];
    __shared__ __attribute__((aligned(16))) float Wt[KW * 256];     // all taps of the layer (a persistent workgroup filters many tiles with them)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_tiles == 0) return;
    auto tile_m0 = [&](int it) { return ((int)blockIdx.x + it * (int)gridDim.x) * CNN_BM; };
    const int halves = cin / TS_HK;
    const int nph = my_tiles * halves;                     // filter phases of this workgroup
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    // ---- filter role ----
    const int orow0 = WR * wave;
    f32x2 xw[NX];                                          // the NEXT filter phase's input rows (in flight while the current phase multiplies)
    float amax = 0.0f;
    for (int e = tid; e < KW * cin; e += 512) Wt[e] = Wd[e];
    auto request = [&](int ph) {                           // phase ph = (tile, channel half); past the last one: the last one again (harmless)
        const int php = min(ph, nph - 1);
        const int it = php / halves, h = php - it * halves;
        const int first = tile_m0(it) + orow0 - half;      // first input row of the slice (may lie before the pass, or its end beyond it)
        const int lack = max(0, -first), base = max(first, 0);
        const int avail = max(0, min(NX - lack, rows - base));
        // descriptor = the rows of the slice that exist; a row before the pass wraps to a huge unsigned offset, a row past its end lies
        // beyond num_records: both read as the zeros 'same' padding wants
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)base * cin)), 0, avail * cin * 4, 0x00020000);
        const int vb = (2 * lane - lack * cin) * 4;
#pragma unroll
        for (int j = 0; j < NX; j++) xw[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, vb + j * cin * 4, h * TS_HK * 4, 0));
    };
    auto filter = [&](int h) {                             // xw -> the planes of this wavefront's 16 rows
        f32x2 o[WR], tw[KW];
#pragma unroll
        for (int t = 0; t < KW; t++) tw[t] = *reinterpret_cast<const f32x2 *>(&Wt[t * cin + h * TS_HK + 2 * lane]);
#pragma unroll
        for (int i = 0; i < WR; i++) o[i] = f32x2{0.f, 0.f};
        // per input row one group of independent FMAs (one per output row it meets), in exactly this order (volatile asm keeps it):
        // every output accumulates its taps in ascending order, and no FMA waits for the one before it
#pragma unroll
        for (int j = 0; j < NX; j++) {
#pragma unroll
            for (int i = 0; i < WR; i++) {
                const int t = j - i;
                if (t >= 0 && t < KW) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(xw[j]), "v"(tw[t]));
            }
        }
#pragma unroll
        for (int i = 0; i < WR; i++) {
            const int off = (orow0 + i) * TS_PP + 2 * lane;
            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(o[i][0]), __builtin_fabsf(o[i][1])));
            const f16x2 hpart = __builtin_convertvector(o[i], f16x2);
            const f32x2 rest = o[i] - __builtin_convertvector(hpart, f32x2);
            const f16x2 lpart = __builtin_convertvector(rest, f16x2);
            *reinterpret_cast<f16x2 *>(&As[0][off]) = hpart; *reinterpret_cast<f16x2 *>(&As[1][off]) = lpart;
        }
    };
    // ---- multiply role ----
    // wavefront = ALL 128 rows x 32 columns: a weight fragment is then fetched by exactly one wavefront of the CU (with 64 x 64 tiles two
    // wavefronts fetched each, and the multiply phases ran at the rate of the CU's L1 fill, 15-20 k ticks per phase where the MFMAs need 6 k)
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    f32x16 acc[4][1];
    const uint16_t *wlane = Wb + ((size_t)(wave * 32 + fm)) * 32 + fk;
    auto loadB = [&](u32x4 (&b)[NP], int h, int sidx) {         // sidx = k16 step inside the half: 32-channel block 4 h + sidx / 2
        const int cb = 4 * h + (sidx >> 1), k16 = sidx & 1;
#if TS_ABL & 1
        (void)cb; (void)k16; for (int pc = 0; pc < NP; pc++) asm volatile("" : "=v"(b[pc]));
#else
#pragma unroll
        for (int pc = 0; pc < NP; pc++) b[pc] = *reinterpret_cast<const u32x4 *>(wlane + ((size_t)(cb * NP + pc) * cout) * 32 + k16 * 16);
#endif
    };
    auto fragA = [&](u32x4 (&a)[4][NP], int sidx) {
#if TS_ABL & 2
        for (int pc = 0; pc < NP; pc++) for (int i = 0; i < 4; i++) asm volatile("" : "=v"(a[i][pc]));
        return;
#endif
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int i = 0; i < 4; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[pc][(i * 32 + fm) * TS_PP + sidx * 16 + fk]);
    };
    // one k16 step: the NEXT step's row fragments are requested first, then the 12 MFMAs in an order the scheduler may not change (left
    // alone it put three MFMAs on one accumulator back to back: each then waits out the 16 passes of the one before)
    auto mma = [&](u32x4 (&a)[4][NP], u32x4 (&an)[4][NP], int snext, u32x4 (&b)[NP]) {
        __builtin_amdgcn_sched_barrier(0);
        if (snext < 8) fragA(an, snext);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};                          // l h', h l', h h'
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][0] = mfma16<NP>(a[i][PA2[t]], b[PB2[t]], acc[i][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    request(0);
    __syncthreads();                                       // the taps are in LDS
    int ph = 0;
    for (int it = 0; it < my_tiles; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][0][q] = 0.0f;
        const bool tr = it == WS_TRACE_TILE; (void)tr;
        if (tr) WS_T(3);
        for (int h = 0; h < halves; h++, ph++) {
#if !(TS_ABL & 4)
            filter(h);
#endif
            if (tr) WS_T(4 + 8 * h);
            u32x4 b0[NP], b1[NP], b2[NP], b3[NP];     // weight fragments FOUR steps ahead: a step is 12 MFMAs, L2 under this load ~2 k ticks away
            loadB(b0, h, 0); loadB(b1, h, 1); loadB(b2, h, 2); loadB(b3, h, 3);
            if (tr) WS_T(5 + 8 * h);
            __syncthreads();                               // the planes are complete
            if (tr) WS_T(6 + 8 * h);
            u32x4 a0[4][NP], a1[4][NP];
            fragA(a0, 0);
            mma(a0, a1, 1, b0); loadB(b0, h, 4);
            mma(a1, a0, 2, b1); loadB(b1, h, 5);
            mma(a0, a1, 3, b2); loadB(b2, h, 6);
            mma(a1, a0, 4, b3); loadB(b3, h, 7);
            mma(a0, a1, 5, b0);
            mma(a1, a0, 6, b1);
            // the next filter phase's rows LAST: the memory counter of a wavefront is in order, so any load requested after them could
            // only be waited for together with them (the first version asked for them before the weight fragments: every fragment then
            // waited for 32 rows from HBM, 26 k ticks per phase)
            request(ph + 1);
            mma(a0, a1, 7, b2);
            mma(a1, a0, 8, b3);
            if (tr) WS_T(7 + 8 * h);
            __syncthreads();                               // every fragment of the planes has been read
            if (tr) WS_T(8 + 8 * h);
        }
        conv_epilogue<64, ADD>(*reinterpret_cast<f32x16 (*)[2][1]>(&acc[0]), Y, scale, shift, Add, valid, tile_m0(it), (wave >> 1) * 64, 0, wave & 1, lane, cout, relu, post);
        conv_epilogue<64, ADD>(*reinterpret_cast<f32x16 (*)[2][1]>(&acc[2]), Y, scale, shift, Add, valid, tile_m0(it), (wave >> 1) * 64, 1, wave & 1, lane, cout, relu, post);
        if (tr) WS_T(40);
    }
    if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
}

