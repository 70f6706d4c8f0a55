// k3_sep_ws16_experiment.hip -- NEGATIVE RESULT kept for the record (round 3).  Not built into the library: this is the kernel text as it
// sat in dnascent_amd/csrc/k3_cnn.hip between k3_sep_ws and k3_dwconv (it uses that file's helpers: conv_epilogue, mfma16, CNN_BM, CNN_BP,
// WS_T) together with its launch line:
//     hipLaunchKernelGGL((k3_sep_ws16<17, ADD, ALT>), dim3(min(conv_grid(rows, o.cout, 256), k3_cu_count())), dim3(1024), 0, st, SEP_ARGS);
// Measured on MI355X, 64 x 20 kb reads (1.2 M positions), the five 17-tap 256 -> 256 layers, one session (gpurun_out/r3e):
//     k3_sep_ws (8 wavefronts, roles overlap)          3 802 us
//     k3_sep_ws16<ALT = false> (16, roles overlap)     4 065 us
//     k3_sep_ws16<ALT = true>  (16, roles alternate)   4 302 us
// all three bit-identical (tools/variant_check.py).  Phase traces (-DDN_WS_TRACE, tools/ws_trace.py; ticks of s_memtime):
//     8 wavefronts:   producer dw 3.2 k + store/load 0.7 k per step; consumer mma 2.0 k + 1.35 k, waits 1.0 k at the barrier
//     16 overlapping: producer dw 2.2-3.5 k for HALF the FMAs; consumer mma 1.0 k + 1.2 k, waits 2-3 k
//     16 alternating: producer dw 1.1-2.0 k + 0.57 k alone on the SIMDs; consumers 2.2-2.9 k for their 2 x 12 MFMAs (the pipe needs 0.8 k)
// The micro-benchmarks that motivated it (tools/ubench_coissue.hip: vector issue beside a busy matrix pipe scales with the number of vector
// wavefronts; tools/ubench_inwave.hip) do not carry over: with LDS fragment reads, L2 weight fragments and dependent FMA chains in the
// streams, neither role gets closer to its pipe's rate by having a sibling on the SIMD.
// ---------------------------------------------------------------------------------------------------------
// k3_sep_ws16: the same layer with SIXTEEN wavefronts per workgroup -- two producers and two consumers per SIMD.
// Why (round 3, tools/ubench_coissue.hip): beside a saturated matrix pipe ONE wavefront issues a packed-fp32 FMA every 16.2 cycles
// (5.5 alone) -- but that is a per-wavefront limit, not the pipe's: two vector wavefronts on the SIMD each keep their 16.2 (8.1 in
// aggregate), three reach 5.4.  k3_sep_ws's phase trace had the producers' 17-tap filter (136 v_pk_fma_f32 per channel block,
// 3.4 k cycles) setting the pace of the workgroup while the 48 MFMAs of a block need 1.5 k: the filter was short of ISSUE SLOTS per
// wavefront, not of vector throughput.  Here the filter of a channel block is cut in two by CHANNELS (producer = one of 4 row slices x
// one of 2 channel halves: 32 rows x 16 channels, lane = channel pair x 4 output rows, 68 FMAs) and the pointwise GEMM in two by
// columns (8 consumers as 2 x 4: 64 rows x 64 columns, 64 accumulator registers), so every wavefront fits 128 registers and four of
// them share a SIMD.  Same arithmetic in the same order per output element: bit-identical to k3_sep_ws (tools/variant_check.py).
// ---------------------------------------------------------------------------------------------------------
#define SEP_XP16 20                                         // floats per row of a 16-channel raw slice: 4 rows advance the bank window by a quarter
// ALT (round 3): the two roles ALTERNATE instead of overlapping -- a second barrier per step parks the consumers while the producers
// filter and the producers while the consumers multiply.  Reason: the vector issue of a wavefront drops to a third whenever ANOTHER
// wavefront of its SIMD has MFMAs in flight (ubench_coissue), and the consumers' MFMA phases are long and thin (a lone wavefront keeps
// the matrix pipe under half busy: tools/ubench_inwave.hip), so "overlap" means the producers crawl for three quarters of a step.
// Alternating, each phase has two wavefronts of ONE kind per SIMD sharing a pipe at its full rate.
template <int KW, bool ADD, bool ALT>
__global__ __launch_bounds__(1024) void k3_sep_ws16(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                    const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                    const float *__restrict__ shift, const float *__restrict__ Add,
                                                    const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                    unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int NP = 2, BN = 256;
    constexpr int SROWS = 32 + KW - 1;                     // raw rows a producer needs for its 32 output rows
    constexpr int NLD = (SROWS * 4 + 63) / 64;             // float4 loads per lane for one raw slice (16 channels = 4 float4 per row)
    __shared__ __attribute__((aligned(16))) float Xr[8][SROWS * SEP_XP16];
    __shared__ __attribute__((aligned(16))) float Wl[2][KW * 32];
    __shared__ __attribute__((aligned(16))) uint16_t As[2][NP][CNN_BM * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 8;                       // wave-uniform
    const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_tiles == 0) return;
    constexpr int half = (KW - 1) / 2;
    const int cblocks = cin >> 5;
    const int nb = my_tiles * cblocks;                     // steps of this workgroup (even: cblocks is)
    auto tile_m0 = [&](int it) { return ((int)blockIdx.x + it * (int)gridDim.x) * CNN_BM; };
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    if (producer) {
        const int pidx = wave - 8, pw = pidx & 3, ph = pidx >> 2;          // row slice (32 output rows), channel half (16 channels)
        const int ct = tid - 512;                                           // 0 .. 511: the first KW * 8 of them carry the block's taps
        const int cp = (lane & 7) * 2, dr = (lane >> 3) * 4;                // channel pair (of the half), output rows dr .. dr + 3 of the slice
        struct RawSet { f32x4 rx[NLD]; bool pin[NLD]; f32x4 rw; bool edge; };
        RawSet S0, S1;
        S0.rw = f32x4{0.f, 0.f, 0.f, 0.f}; S1.rw = S0.rw;
        float amax = 0.0f;
        float *Xs = Xr[pidx];
        int ld_cb = 0, ld_it = 0;
        int xoff[NLD];
#pragma unroll
        for (int p = 0; p < NLD; p++) { const int f = lane + 64 * p; xoff[p] = ((f >> 2) * cin + 16 * ph + (f & 3) * 4) * 4; }
        const int woff = ((ct >> 3) * cin + (ct & 7) * 4) * 4;
        const __amdgpu_buffer_rsrc_t rtap = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(Wd)), 0, KW * cin * 4, 0x00020000);
        auto gloadX = [&](RawSet &S) {
            const int cb = ld_cb, m0 = tile_m0(ld_it);
            if (ld_cb + 1 < cblocks) ld_cb++; else if (ld_it + 1 < my_tiles) { ld_cb = 0; ld_it++; }
            S.edge = m0 - half < 0 || m0 + CNN_BM + half > rows;
            if (S.edge) {
#pragma unroll
                for (int p = 0; p < NLD; p++) {
                    const int f = lane + 64 * p, rr = f >> 2, q = f & 3;
                    const int src = m0 + 32 * pw - half + rr;
                    const bool in = rr < SROWS && src >= 0 && src < rows;
                    S.rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + (cb << 5) + 16 * ph + q * 4);
                    S.pin[p] = in;
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)(m0 + 32 * pw - half) * cin)), 0, SROWS * cin * 4, 0x00020000);
#pragma unroll
                for (int p = 0; p < NLD; p++) S.rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, xoff[p], cb << 7, 0));
            }
            S.rw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtap, woff, cb << 7, 0));   // lanes beyond the KW x 8 float4 read zeros
        };
        auto lstoreX = [&](RawSet &S, int wbuf) {
            if (S.edge) {
#pragma unroll
                for (int p = 0; p < NLD; p++) S.rx[p] = S.pin[p] ? S.rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int p = 0; p < NLD; p++) {
                const int f = lane + 64 * p, rr = f >> 2, q = f & 3;
                if (rr < SROWS) *reinterpret_cast<f32x4 *>(&Xs[rr * SEP_XP16 + q * 4]) = S.rx[p];
            }
            if (ct < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[wbuf][(ct >> 3) * 32 + (ct & 7) * 4]) = S.rw;
        };
        // a lane: one channel pair, 4 consecutive output rows: its KW + 3 input rows and KW taps are read once (ds_read_b64), the FMAs of
        // an input row form one group of up to 4 independent v_pk_fma_f32; per output the taps accumulate in ascending order
        auto depthwise = [&](int abuf, int wbuf) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 o[4], w[KW];
#pragma unroll
            for (int t = 0; t < KW; t++) w[t] = *reinterpret_cast<const f32x2 *>(&Wl[wbuf][t * 32 + 16 * ph + cp]);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = f32x2{0.f, 0.f};
            // input rows in groups of four, the next group requested before the current group's FMAs: 16 registers of window instead
            // of 40 (the whole window at once spilled 35 registers under the 128 cap of four wavefronts per SIMD)
            constexpr int NG = (KW + 3 + 3) / 4;
            f32x2 xa[4], xb[4];
            auto ldg = [&](f32x2 (&x)[4], int g) {
#pragma unroll
                for (int q = 0; q < 4; q++) if (4 * g + q < KW + 3) x[q] = *reinterpret_cast<const f32x2 *>(&Xs[(dr + 4 * g + q) * SEP_XP16 + cp]);
            };
            auto fmag = [&](const f32x2 (&x)[4], int g) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int j = 4 * g + q;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int t = j - i;
                        if (j < KW + 3 && t >= 0 && t < KW) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[q]), "v"(w[t]));
                    }
                }
            };
            ldg(xa, 0);
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                if (g + 1 < NG) ldg(xb, g + 1);
                fmag(xa, g);
                if (g + 2 < NG) ldg(xa, g + 2);
                if (g + 1 < NG) fmag(xb, g + 1);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int off = (32 * pw + dr + i) * CNN_BP + 16 * ph + cp;
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(o[i][0]), __builtin_fabsf(o[i][1])));
                const f16x2 h = __builtin_convertvector(o[i], f16x2);
                const f32x2 rest = o[i] - __builtin_convertvector(h, f32x2);
                const f16x2 l = __builtin_convertvector(rest, f16x2);
                *reinterpret_cast<f16x2 *>(&As[abuf][0][off]) = h; *reinterpret_cast<f16x2 *>(&As[abuf][1][off]) = l;
            }
        };
        gloadX(S0); lstoreX(S0, 0); gloadX(S0); gloadX(S1);
        __syncthreads();
        depthwise(0, 0); lstoreX(S0, 1); gloadX(S0);
        __syncthreads();
        for (int b = 0; b + 2 < nb; b += 2) {                  // no conditional around the loads (see k3_sep_ws); the last pair is peeled
            const bool tr = WS_TRACE_TILE >= 0 && b / cblocks == WS_TRACE_TILE; const int c4 = WS_TRACE_TILE >= 0 ? 4 * (b % cblocks) : 0; (void)tr; (void)c4;
            if (tr) WS_T(3 + c4);
            depthwise(1, 1); if (tr) WS_T(4 + c4); lstoreX(S1, 0); gloadX(S1);
            if (tr) WS_T(5 + c4);
            if (ALT) __syncthreads();                          // ... the consumers' turn (step b)
            __syncthreads();
            if (tr) WS_T(6 + c4);
            depthwise(0, 0); if (tr) WS_T(7 + c4); lstoreX(S0, 1); gloadX(S0);
            if (tr) WS_T(8 + c4);
            if (ALT) __syncthreads();
            __syncthreads();
            if (tr) WS_T(9 + c4);
        }
        depthwise(1, 1);
        if (ALT) __syncthreads();
        __syncthreads();
        if (ALT) __syncthreads();
        __syncthreads();
        if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
        return;
    }
    // ---- consumers: 2 (rows) x 4 (columns) wavefronts, 64 x 64 outputs each ----
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 acc[2][2];
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t *wlane = Wb + ((size_t)(wn * 64 + fm)) * 32 + fk;
    auto loadB = [&](u32x4 (&b)[2][NP], int step) {            // step = 2 * (channel block of the stream) + k16
        const int cb = (step >> 1) % cblocks, k16 = step & 1;
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int j = 0; j < 2; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(wlane + ((size_t)(cb * NP + pc) * cout + j * 32) * 32 + k16 * 16);
    };
    u32x4 b0[2][NP], b1[2][NP];
    loadB(b0, 0);
    __syncthreads();
    __syncthreads();
    auto mma = [&](int cur, int k16, u32x4 (&b)[2][NP]) {
        u32x4 a[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[cur][pc][(wm * 64 + i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};      // l h', h l', h h': the order of every f16x3 kernel
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = mfma16<NP>(a[i][PA2[t]], b[j][PB2[t]], acc[i][j]);
        }
    };
    for (int it = 0; it < my_tiles; it++) {
        const bool tr = it == WS_TRACE_TILE; (void)tr;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        if (tr) WS_T(3);
        for (int cb = 0; cb < cblocks; cb++) {
            const int cur = cb & 1, step = it * cblocks + cb;
            loadB(b1, 2 * step + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (ALT) __syncthreads();                          // the producers' turn is over: both halves of B were requested before it
            mma(cur, 0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (tr) WS_T(4 + 3 * cb);
            if (!ALT) { loadB(b0, 2 * step + 2); __builtin_amdgcn_sched_barrier(0); }
            mma(cur, 1, b1);
            if (ALT) { __builtin_amdgcn_sched_barrier(0); loadB(b0, 2 * step + 2); }      // lands during the producers' next turn
            if (tr) WS_T(5 + 3 * cb);
            __syncthreads();
            if (tr) WS_T(6 + 3 * cb);
        }
        conv_epilogue<128, ADD>(acc, Y, scale, shift, Add, valid, tile_m0(it), (wn >> 1) * 128, wm, wn & 1, lane, cout, relu, post);
        if (tr) WS_T(40);
    }
}

