// k3_sep_uni_experiment.hip -- NOT part of libdnascent_hip.so.  Round-4 experiment, kept as the record of a negative result (moved out of
// k3_cnn.hip in round 5: round-4 verdict item 7).  The fragment needs k3_cnn.hip around it (conv_epilogue, mfma16, range_report, CNN_BP, SEP_XPW);
// it was dispatched from k3_launch_sep behind DN_CNN_SEP_UNI=1:
//     hipLaunchKernelGGL((k3_sep_uni<ADD>), dim3(min(conv_grid(rows, o.cout, 256), k3_cu_count())), dim3(512), 0, st, SEP_ARGS);
// Result (DESIGN / NOTES, round 4): 4 962 us for the five 256 -> 256 layers against 3 800 for k3_sep_ws, NOT bit-identical (the compiler contracts the
// last fmaf of a chain with the fp16 conversion into v_fma_mixlo_f16), 5 spilled registers.

// ---------------------------------------------------------------------------------------------------------
// k3_sep_uni (round 4): the 17-tap 256-column separable layer with UNIFORM roles -- every wavefront filters AND multiplies, and the two
// instruction streams are interleaved INSIDE each wavefront.  What the round-3 micro-benchmarks say about this SIMD:
//   * a vector instruction of ANOTHER wavefront beside a dense MFMA stream issues every ~46 ticks (k3_sep_ws's producers: ~260 instructions
//     per step take 3.0-3.4 k ticks beside the consumers' MFMAs, 1.75 k alone);
//   * from the SAME wavefront, six plain v_fma_f32 / v_cvt and three or four LDS reads ride for free in an MFMA's 32-cycle slot
//     (tools/ubench_inwave.hip: 36 cycles per {MFMA + 6 v_fma_f32}, 45 with 8, 52 with 10; a v_pk_fma_f32 costs 17).
// So: 8 wavefronts, each owns a 64 x 64 tile of the 128 x 256 output (24 MFMAs per 32-channel step) and filters 16 rows x 32 channels of
// the NEXT step between them -- lane = (channel, group of 8 rows): 24 input rows and 17 taps as 4-byte LDS reads, 136 plain fmaf (taps in
// ascending order per output: bit-identical to k3_dwconv and to k3_sep_ws's packed form), split into the two fp16 planes.  The step is
// written as 24 SLOTS -- one MFMA, six or seven filter instructions, now and then an LDS access -- pinned with sched_barrier (left to the
// scheduler, also with sched_group_barrier, the MFMAs bunch at the top of the block and the filter trails behind them).  The output rows'
// chains are run in two halves (rows 0-3, then 4-7) so that the first half's split and plane stores sit under the second half's MFMAs.
// Raw rows travel global -> registers (two sets in flight, stored two steps after their loads) -> the wavefront's PRIVATE 32-row slice of
// LDS and are read back into registers before the step's barrier (the slice is private: no barrier between its store and its reads);
// weight fragments come straight from L2, one k16 half ahead; planes and taps are double-buffered: ONE barrier per step.  Persistent
// workgroups (one per CU) run their tiles' channel blocks as one stream of steps; the epilogue of a tile is the shared conv_epilogue.
// Same sums in the same order as k3_sep_ws: results are bit-identical (tools/variant_check.py).
// ---------------------------------------------------------------------------------------------------------
#define UNI_SROWS 32                                       // raw rows a wavefront's 16 output rows need (16 + 17 - 1)
// the filter's 136 FMAs in issue order: output rows 0-3 (input rows 0 .. 19), then 4-7 (input rows 4 .. 23); within an output row the taps ascend
struct UniFma { int j, i; };
static constexpr UniFma uni_fma(int p) {
    int q = 0;
    for (int h = 0; h < 2; h++)
        for (int j = 4 * h; j < 4 * h + 20; j++)
            for (int i = 4 * h; i < 4 * h + 4; i++) {
                const int t = j - i;
                if (t >= 0 && t < 17) { if (q == p) return UniFma{j, i}; q++; }
            }
    return UniFma{-1, -1};
}
template <int P0, int P1> __device__ __forceinline__ void uni_fmas(const float (&x)[24], const float (&w)[17], float (&o)[8]) {
    if constexpr (P0 < P1 && P0 < 136) {
        constexpr UniFma f = uni_fma(P0);
        o[f.i] = __builtin_fmaf(x[f.j], w[f.j - f.i], o[f.i]);
        uni_fmas<P0 + 1, P1>(x, w, o);
    }
}
template <bool ADD>
__global__ __launch_bounds__(512) void k3_sep_uni(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                  const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, const float *__restrict__ Add,
                                                  const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                  unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int KW = 17, NP = 2, half = 8, NLD = 4;      // float4 loads per lane for one raw slice (32 rows x 8)
    __shared__ __attribute__((aligned(16))) float Xr[8][UNI_SROWS * SEP_XPW];
    __shared__ __attribute__((aligned(16))) float Wl[2][KW * 32];
    __shared__ __attribute__((aligned(16))) uint16_t As[2][NP][CNN_BM * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_tiles == 0) return;
    const int cblocks = cin >> 5;
    auto tile_m0 = [&](int it) { return ((int)blockIdx.x + it * (int)gridDim.x) * CNN_BM; };
    const int wm = wave >> 2, wn = wave & 3;               // GEMM role: rows 64 wm .., columns 64 wn ..
    const int fch = lane & 31, frg = lane >> 5;            // filter role: channel fch, output rows 16 wave + 8 frg .. + 7
    struct RawSet { f32x4 rx[NLD]; bool pin[NLD]; f32x4 rw; bool edge; };
    RawSet S0, S1;
    S0.rw = f32x4{0.f, 0.f, 0.f, 0.f}; S1.rw = S0.rw;
    float amax = 0.0f;
    float *Xs = Xr[wave];
    int ld_cb = 0, ld_it = 0;
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    int xoff[NLD];
#pragma unroll
    for (int p = 0; p < NLD; p++) { const int f = lane + 64 * p; xoff[p] = ((f >> 3) * cin + (f & 7) * 4) * 4; }
    const int woff = ((tid >> 3) * cin + (tid & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t rtap = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(Wd)), 0, KW * cin * 4, 0x00020000);
    auto gloadX = [&](RawSet &S) {                          // the raw slice (and the taps) of the load stream's next step
        const int cb = ld_cb, m0 = tile_m0(ld_it);
        if (ld_cb + 1 < cblocks) ld_cb++; else if (ld_it + 1 < my_tiles) { ld_cb = 0; ld_it++; }
        S.edge = m0 - half < 0 || m0 + CNN_BM + half > rows;
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) {
                const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
                const int src = m0 + 16 * wave - half + rr;
                const bool in = src >= 0 && src < rows;
                S.rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + (cb << 5) + q * 4);
                S.pin[p] = in;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)(m0 + 16 * wave - half) * cin)), 0, UNI_SROWS * cin * 4, 0x00020000);
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, xoff[p], cb << 7, 0));
        }
        if (tid < 192) S.rw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtap, woff, cb << 7, 0));   // wavefronts 0-2: the 17 x 8 float4 of a block's taps (lanes past them read zeros)
    };
    auto lstoreX = [&](RawSet &S, int wbuf) {
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = S.pin[p] ? S.rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
            *reinterpret_cast<f32x4 *>(&Xs[rr * SEP_XPW + q * 4]) = S.rx[p];
        }
        if (tid < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[wbuf][(tid >> 3) * 32 + (tid & 7) * 4]) = S.rw;
    };
    float x[24], w[KW], o[8];
    auto readX = [&]() {                                    // this lane's 24 input rows of the slice (private to the wavefront: no barrier needed after lstoreX)
#pragma unroll
        for (int j = 0; j < 24; j++) x[j] = Xs[(8 * frg + j) * SEP_XPW + fch];
    };
    auto readW = [&](int wbuf) {
#pragma unroll
        for (int t = 0; t < KW; t++) w[t] = Wl[wbuf][t * 32 + fch];
    };
    auto splitOut = [&](int abuf, int i) {                  // output row i of the lane: the two fp16 pieces into the planes
        const int off = (16 * wave + 8 * frg + i) * CNN_BP + fch;
        amax = __builtin_fmaxf(amax, __builtin_fabsf(o[i]));
        const _Float16 h = (_Float16)o[i];
        const _Float16 l = (_Float16)(o[i] - (float)h);
        *reinterpret_cast<_Float16 *>(&As[abuf][0][off]) = h; *reinterpret_cast<_Float16 *>(&As[abuf][1][off]) = l;
    };
    // GEMM role
    constexpr int NJ = 2;
    f32x16 acc[2][NJ];
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t *wlane = Wb + ((size_t)(wn * 64 + fm)) * 32 + fk;
    auto loadB = [&](u32x4 (&b)[NJ][NP], int step2) {     // step2 = 2 * step + k16
        const int cb = (step2 >> 1) % cblocks, k16 = step2 & 1;
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int j = 0; j < NJ; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(wlane + ((size_t)(cb * NP + pc) * cout + j * 32) * 32 + k16 * 16);
    };
    auto loadA = [&](u32x4 (&a)[2][NP], int cur, int k16) {
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[cur][pc][(wm * 64 + i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
    };
    u32x4 b0[NJ][NP], b1[NJ][NP], a0[2][NP], a1[2][NP];
    // prologue: raw slices of steps 0 (stored at once), 1, 2 requested; the planes of step 0 filtered without any MFMA beside them
    gloadX(S0); lstoreX(S0, 0); gloadX(S0); gloadX(S1);
    loadB(b0, 0);
    __syncthreads();
    readX(); readW(0);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = 0.0f;
    uni_fmas<0, 136>(x, w, o);
#pragma unroll
    for (int i = 0; i < 8; i++) splitOut(0, i);
    lstoreX(S0, 1); readX(); gloadX(S0);
    __syncthreads();
    // step g (channel block cb of tile it): multiply planes[g & 1] and filter step g + 1 into planes[(g + 1) & 1] in 24 slots; then store the raw slice of
    // step g + 2 (set S1 for even g, S0 for odd g), read it back into x[], request step g + 4's into that set.  The workgroup's last step filters a repeat
    // of itself into planes nobody reads any more (the load stream stays on the last step): no branch inside the slots.
    constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
#ifndef UNI_ABL
#define UNI_ABL 0                                          /* experiment builds (timing only, wrong results): 1 = no filter FMAs, 2 = no MFMAs, 3 = neither */
#endif
#define UNI_MFMA(M, A, B) do { constexpr int r_ = (M) % 12, t_ = r_ / 4, i_ = (r_ % 4) / 2, j_ = r_ % 2; \
        if (!(UNI_ABL & 2)) acc[i_][j_] = mfma16<NP>(A[i_][PA2[t_]], B[j_][PB2[t_]], acc[i_][j_]); } while (0)
#define UNI_SLOT(M, A, B, P0, P1, EXTRA) do { UNI_MFMA(M, A, B); if (!(UNI_ABL & 1)) uni_fmas<P0, P1>(x, w, o); EXTRA; __builtin_amdgcn_sched_barrier(0); } while (0)
#define UNI_STEP(CUR, SET, G)                                                                                                            \
    do {                                                                                                                                  \
        readW((CUR) ^ 1); loadA(a0, CUR, 0); loadB(b1, 2 * (G) + 1);                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; i++) o[i] = 0.0f;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                                \
        UNI_SLOT(0, a0, b0, 0, 7, (void)0);      UNI_SLOT(1, a0, b0, 7, 14, (void)0);    UNI_SLOT(2, a0, b0, 14, 21, (void)0);            \
        UNI_SLOT(3, a0, b0, 21, 28, (void)0);    UNI_SLOT(4, a0, b0, 28, 35, loadA(a1, CUR, 1));                                          \
        UNI_SLOT(5, a0, b0, 35, 42, (void)0);    UNI_SLOT(6, a0, b0, 42, 49, (void)0);   UNI_SLOT(7, a0, b0, 49, 56, (void)0);            \
        UNI_SLOT(8, a0, b0, 56, 62, (void)0);    UNI_SLOT(9, a0, b0, 62, 68, (void)0);   UNI_SLOT(10, a0, b0, 68, 73, splitOut((CUR) ^ 1, 0)); \
        UNI_SLOT(11, a0, b0, 73, 78, splitOut((CUR) ^ 1, 1));                                                                             \
        loadB(b0, 2 * (G) + 2);                                                                                                           \
        UNI_SLOT(12, a1, b1, 78, 83, splitOut((CUR) ^ 1, 2));  UNI_SLOT(13, a1, b1, 83, 88, splitOut((CUR) ^ 1, 3));                      \
        UNI_SLOT(14, a1, b1, 88, 95, (void)0);   UNI_SLOT(15, a1, b1, 95, 102, (void)0);  UNI_SLOT(16, a1, b1, 102, 109, (void)0);        \
        UNI_SLOT(17, a1, b1, 109, 116, (void)0); UNI_SLOT(18, a1, b1, 116, 123, (void)0); UNI_SLOT(19, a1, b1, 123, 130, (void)0);        \
        UNI_SLOT(20, a1, b1, 130, 136, (void)0); UNI_SLOT(21, a1, b1, 136, 136, splitOut((CUR) ^ 1, 4); splitOut((CUR) ^ 1, 5));          \
        UNI_SLOT(22, a1, b1, 136, 136, splitOut((CUR) ^ 1, 6)); UNI_SLOT(23, a1, b1, 136, 136, splitOut((CUR) ^ 1, 7));                   \
        lstoreX(SET, CUR); readX(); gloadX(SET);                                                                                          \
        __syncthreads();                                                                                                                  \
    } while (0)
    for (int it = 0; it < my_tiles; it++) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        for (int cb = 0; cb < cblocks; cb += 2) {          // cblocks is even: a step's parity is its channel block's
            const int g = it * cblocks + cb;
            UNI_STEP(0, S1, g);
            UNI_STEP(1, S0, g + 1);
        }
        conv_epilogue<128, ADD>(acc, Y, scale, shift, Add, valid, tile_m0(it), (wn >> 1) * 128, wm, wn & 1, lane, cout, relu, post);
    }
#undef UNI_STEP
#undef UNI_SLOT
#undef UNI_MFMA
    range_report(amax, range_flag, lane);
}

