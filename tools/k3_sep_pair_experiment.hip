// k3_sep_pair_experiment.hip -- NEGATIVE RESULT kept for the record (round 3).  Not built into the library: the kernel text as it sat in
// dnascent_amd/csrc/k3_cnn.hip before the DN_WS_TRACE block (it uses that file's helpers: conv_epilogue with its wave_rows argument, mfma16,
// CNN_BM, CNN_BP, SEP_XP), dispatched by k3_run for two consecutive fused separable layers 128 -> 128 -> 128 with 9 taps whose
// intermediate nobody else reads (5 pairs + 1 single layer in the default network), grid = one workgroup per CU.
// Bit-identical to the unfused kernels (tools/variant_check.py), HBM bytes per pair 0.52 x -- and SLOWER: 746 us per pair against
// 2 x 262 us (64 x 20 kb reads, 1.2 M positions, one session: gpurun_out/r3n; network 18.63 ms against 17.47).  A 120-row tile takes
// ~42 k cycles: 8 channel-block steps of [stores + next loads | barrier | depthwise | barrier | 12 MFMAs per wavefront | barrier] plus two
// epilogues, every wavefront of the one workgroup a CU can hold (138 KB of LDS) in the same phase at the same time.  k3_sep_split's two
// independent workgroups per CU overlap each other's phases and stream at 4.7 TB/s; halving the bytes does not pay for losing that.
// ---------------------------------------------------------------------------------------------------------
// k3_sep_pair (round 3): TWO consecutive SeparableConv1D layers C -> C -> C (C = 128) in one launch, the intermediate activations in LDS.
// The 9-tap 128-channel separable layers are HBM-side (k3_sep_split: 4.6 TB/s of layer I/O, 264 us each): only fewer bytes help.  A
// workgroup produces 120 output rows of the SECOND layer: it needs the first layer's output on rows [m0 - 4, m0 + 124) -- exactly the
// 128-row tile geometry of k3_sep_split shifted by half a filter -- and for that the input on [m0 - 8, m0 + 128):
//   stage 1   k3_sep_split's tile computation (raw rows -> depthwise -> fp16 planes -> 128 x 128 x 32 MFMA steps over the channel blocks),
//             its epilogue (folded BatchNorm, ReLU, padding / out-of-pass rows zeroed) written to LDS (Y1, fp32) instead of HBM
//   stage 2   the same computation reading Y1 rows r .. r + 8 for output row r; rows 120-127 of the MFMA tile are not stored
// HBM bytes per pair: (136 read + 120 written) rows per 120 against 2 x (136 + 128) per 128: 0.52 x.  8 wavefronts (2 x 4: 64 rows x 32
// columns each), one workgroup per CU (139 KB of LDS), persistent over the row tiles.  Same depthwise order, same pieces, same MFMA order
// per output element as the unfused kernels: bit-identical (tools/variant_check.py, DN_CNN_PAIR=0).
// ---------------------------------------------------------------------------------------------------------
#define PAIR_ROWS 120
#define PAIR_YP 144                                         // Y1 pitch in floats: two rows advance the bank window by half (2 x 144 mod 64 = 32)
template <int KW>
__global__ __launch_bounds__(512) void k3_sep_pair(const float *__restrict__ X, float *__restrict__ Y,
                                                   const float *__restrict__ Wd1, const uint16_t *__restrict__ Wb1, const float *__restrict__ scale1, const float *__restrict__ shift1, int relu1, float post1,
                                                   const float *__restrict__ Wd2, const uint16_t *__restrict__ Wb2, const float *__restrict__ scale2, const float *__restrict__ shift2, int relu2, float post2,
                                                   const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int C = 128, NP = 2, half = (KW - 1) / 2;
    constexpr int XROWS = CNN_BM + KW - 1;                 // input rows of stage 1
    constexpr int NLD = (XROWS * 8 + 511) / 512;           // float4 loads per thread for one raw tile (32 channels)
    __shared__ __attribute__((aligned(16))) float Y1[CNN_BM * PAIR_YP];
    __shared__ __attribute__((aligned(16))) float Xr[XROWS * SEP_XP];
    __shared__ __attribute__((aligned(16))) float Wl[KW * 32];
    __shared__ __attribute__((aligned(16))) uint16_t As[NP][CNN_BM * CNN_BP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[NP][C * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;               // 64 rows x 32 columns per wavefront
    const int ntiles = (rows + PAIR_ROWS - 1) / PAIR_ROWS;
    constexpr int cblocks = C >> 5;
    const int dq = tid & 7, dr = (tid >> 3) * 2;          // depthwise: channels 4 dq .. 4 dq + 3 of the block, output rows dr, dr + 1
    const int l_r = tid >> 2, l_k = (tid & 3) * 8;        // B loader: 128 rows x 4 chunks of 8 elements, one piece per pass
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    f32x4 rx[NLD]; bool pin[NLD];
    f32x4 rw = {0.f, 0.f, 0.f, 0.f};
    u32x4 rb[NP];
    float amax = 0.0f;
    f32x16 acc[2];
    auto gloadX = [&](int m0, int cb) {                    // raw rows [m0 - 4 - half, ...) of stage 1, channel block cb
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 512 * p, rr = f >> 3, q = f & 7;
            const int src = m0 - 2 * half + rr;
            const bool in = rr < XROWS && src >= 0 && src < rows;
            rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : 0) * C + (cb << 5) + q * 4);
            pin[p] = in;
        }
    };
    auto gloadW = [&](const float *Wd, const uint16_t *Wb, int cb) {
        if (tid < KW * 8) rw = *reinterpret_cast<const f32x4 *>(Wd + (size_t)(tid >> 3) * C + (cb << 5) + (tid & 7) * 4);
#pragma unroll
        for (int pc = 0; pc < NP; pc++) rb[pc] = *reinterpret_cast<const u32x4 *>(Wb + ((size_t)(cb * NP + pc) * C + l_r) * 32 + l_k);
    };
    auto lstoreX = [&]() {
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 512 * p, rr = f >> 3, q = f & 7;
            if (rr < XROWS) *reinterpret_cast<f32x4 *>(&Xr[rr * SEP_XP + q * 4]) = pin[p] ? rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstoreW = [&]() {
        if (tid < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[(tid >> 3) * 32 + (tid & 7) * 4]) = rw;
#pragma unroll
        for (int pc = 0; pc < NP; pc++) *reinterpret_cast<u32x4 *>(&Bs[pc][l_r * CNN_BP + l_k]) = rb[pc];
    };
    // the depthwise filter of two output rows x four channels: taps in ascending order with fmaf, exactly as k3_sep_split / k3_dwconv.
    // src(row) -> pointer to the 4 floats of input row `row` (0 .. 127 + KW - 1) of this thread's channels
    auto depthwise = [&](auto src) {
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < KW + 1; j++) {
            const f32x4 x = src(dr + j);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int t = j - i;
                if (t >= 0 && t < KW) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(&Wl[t * 32 + dq * 4]);
#pragma unroll
                    for (int e = 0; e < 4; e++) o[i][e] = __builtin_fmaf(x[e], w[e], o[i][e]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float x = o[i][e];
                amax = fmaxf(amax, fabsf(x));
                const _Float16 hh = (_Float16)x;
                h[e] = hh; l[e] = (_Float16)(x - (float)hh);
            }
            const int off = (dr + i) * CNN_BP + dq * 4;
            *reinterpret_cast<f16x4 *>(&As[0][off]) = h; *reinterpret_cast<f16x4 *>(&As[1][off]) = l;
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++) {
            u32x4 a[2][NP], b[NP];
#pragma unroll
            for (int pc = 0; pc < NP; pc++) {
#pragma unroll
                for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[pc][(wm * 64 + i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
                b[pc] = *reinterpret_cast<const u32x4 *>(&Bs[pc][(wn * 32 + fm) * CNN_BP + k16 * 16 + fk]);
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};      // l h', h l', h h'
#pragma unroll
                for (int i = 0; i < 2; i++) acc[i] = mfma16<NP>(a[i][PA2[t]], b[PB2[t]], acc[i]);
            }
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][q] = 0.0f;
    };
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int m0 = tile * PAIR_ROWS;                   // first output row of the pair; stage 1 produces rows m0 - half .. m0 - half + 127
        // ---------------- stage 1 ----------------
        zero_acc();
        gloadX(m0, 0); gloadW(Wd1, Wb1, 0);
#pragma nounroll
        for (int cb = 0; cb < cblocks; cb++) {
            __syncthreads();                               // the previous MFMA phase / epilogue is done with Xr, Wl, As, Bs
            lstoreX(); lstoreW();
            if (cb + 1 < cblocks) { gloadX(m0, cb + 1); gloadW(Wd1, Wb1, cb + 1); } else gloadW(Wd2, Wb2, 0);
            __syncthreads();
            depthwise([&](int row) { return *reinterpret_cast<const f32x4 *>(&Xr[row * SEP_XP + dq * 4]); });
            __syncthreads();
            mma();
        }
        {   // epilogue 1 -> Y1: C/D layout of a 32 x 32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
            const int col = wn * 32 + fm;
            const float sc = scale1[col] * post1, sh = shift1[col], floor_ = relu1 ? 0.0f : -3.402823466e38f;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int r = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    const int g = m0 - half + r;                               // row of the pass
                    const bool live_row = g >= 0 && g < rows && valid[g] != 0;
                    const float y = fmaxf(__builtin_fmaf(acc[i][q], sc, sh), floor_);
                    Y1[r * PAIR_YP + col] = live_row ? y : 0.0f;
                }
        }
        // ---------------- stage 2 ----------------
        zero_acc();
#pragma nounroll
        for (int cb = 0; cb < cblocks; cb++) {
            __syncthreads();                               // Y1 complete (cb == 0) / the previous MFMA phase is done with Wl, As, Bs
            lstoreW();
            if (cb + 1 < cblocks) gloadW(Wd2, Wb2, cb + 1);
            __syncthreads();
            depthwise([&](int row) { return row < CNN_BM ? *reinterpret_cast<const f32x4 *>(&Y1[row * PAIR_YP + (cb << 5) + dq * 4]) : f32x4{0.f, 0.f, 0.f, 0.f}; });
            __syncthreads();
            mma();
        }
        {   // epilogue 2: rows m0 .. m0 + 119 (and inside the pass) leave for HBM
            f32x16 acc2[2][1];
            acc2[0][0] = acc[0]; acc2[1][0] = acc[1];
            const int lim = min(PAIR_ROWS, rows - m0) - wm * 64;
            conv_epilogue<64, false>(acc2, Y, scale2, shift2, nullptr, valid, m0, (wn >> 1) * 64, wm, wn & 1, lane, C, relu2, post2, lim);
        }
    }
    if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
}

