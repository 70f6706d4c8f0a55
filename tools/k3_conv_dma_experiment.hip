// k3_conv_dma_experiment.hip -- NEGATIVE RESULT kept for the record (round 3).  Not built into the library: the kernel text as it sat in
// dnascent_amd/csrc/k3_cnn.hip after k3_conv_split (it uses that file's helpers: conv_tile, conv_epilogue, split2, mfma16, CNN_BP), launched as
//     hipLaunchKernelGGL((k3_conv_dma<ADD>), dim3(conv_grid(rows, o.cout, 128, 256)), dim3(512), 0, st, <the arguments of k3_conv_split>)
// for f16x3 layers with cout % 128 == 0, rows % 256 == 0 and 3 / 9 / 17 taps.  tools/ubench_glds.hip is the check of the LDS-DMA semantics it
// relies on (lane i lands at base + 16 i; the swizzle goes on the source address).
// Measured on MI355X, 64 x 20 kb reads (1.2 M positions), one session each (gpurun_out/r3k), bit-identical to k3_conv_split (variant_check):
//     layer                 k3_conv_split     k3_conv_dma (reads, wait, MFMAs per group)     k3_conv_dma (fragments one group ahead, this file)
//     17 x 128 -> 256       3 091 / 3 188 us  3 589 us                                       3 664 us
//      9 x 128 -> 128         975 /   999       1 132                                        1 173
//      9 x  64 -> 128         570 /   602         686                                          698
//      3 x 256 -> 256       1 469 / 1 490       1 513                                        1 527
//      3 x 256 -> 128         717 /   732         781                                          804
// Why it cannot win (tools/ubench_tick.hip, same box): a chip-wide MFMA loop runs at 1.53 GHz with one wavefront per SIMD, 1.65 with two,
// 1.81 with four -- 1.52 / 1.66 / 1.86 PFLOP/s of dense fp16 by the wall clock, never the 2.5 of 2.4 GHz; and two MFMA wavefronts on a SIMD
// do not interleave, the older one keeps the pipe (each measures 32 cycles per MFMA, the pair takes twice as long).  The 17-tap layer issues
// 4.0 TFLOP of fp16 products in 3.09 ms = 1.30 PFLOP/s: 70-85 % of what the pipe delivers under load.  What is left is not in the barriers.
// ---------------------------------------------------------------------------------------------------------
// k3_conv_dma (round 3): the long convolutions with the WEIGHT tile brought in by LDS-DMA and three taps per step.
// What the round found out about the matrix pipe (tools/ubench_tick.hip, ubench_inwave.hip): a wavefront issues one
// v_mfma_f32_32x32x16_f16 per 32 cycles, the pipe of a SIMD takes TWO (two wavefronts each keep their 32), and under a chip-wide MFMA
// load the clock settles at 1.5-1.65 GHz.  k3_conv_split at MfmaUtil ~0.45 leaves more than half of that idle: a step is 24 MFMAs per
// wavefront between TWO barriers (the B tile is single-buffered: store after the barrier, barrier again), with the weight tile passing
// through registers (loads issued, ds_write_b128 later) -- its phase trace had 37 % of a tile in barriers and 19 % in those loads/stores.
// Here:  * the pre-split weights are a straight copy, so they go global -> LDS by global_load_lds_dwordx4 (no registers, no ds_write,
//          six instructions per wavefront and step), into TWO buffers: the DMA of step g + 1 runs under the MFMAs of step g;
//        * the LDS image is unpadded (an LDS-DMA writes lane i at base + 16 i) and XOR-swizzled through the per-lane SOURCE address
//          (chunk ^ ((row >> 2) & 3)): the fragments stay single conflict-free ds_read_b128;
//        * a step covers THREE taps of a channel block (72 MFMAs per wavefront) and ends in ONE barrier (__syncthreads: its vmcnt(0)
//          is what retires the DMA); 256 rows x 128 columns per workgroup, 8 wavefronts (64 x 64 each), 139.5 KB of LDS: one workgroup
//          per CU = the two wavefronts per SIMD the pipe can take.
// Same products in the same order as k3_conv_split (channel block, tap, k16, {l h', h l', h h'}): bit-identical (tools/variant_check.py).
// ---------------------------------------------------------------------------------------------------------
#define DMA_TS 3
template <bool ADD>
__global__ __launch_bounds__(512) void k3_conv_dma(const float *__restrict__ X, float *__restrict__ Y, const uint16_t *__restrict__ Wb,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   const float *__restrict__ Add, const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int k,
                                                   int cin, int cout, int relu, float post, unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int BM = 256, BN = 128, NP = 2;
    __shared__ __attribute__((aligned(16))) uint16_t As[NP][(BM + 16) * CNN_BP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][DMA_TS][NP][BN * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    if (!conv_tile(cout, BN, rows, m0, n0, BM)) return;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    const int half = (k - 1) / 2;
    const int cblocks = cin >> 5;
    const int spc = (k + DMA_TS - 1) / DMA_TS;              // steps per channel block
    const int nsteps = cblocks * spc;
    constexpr int LR = BM / 2;                              // rows one A-loader pass covers (4 threads per row)
    const int l_r = tid >> 2, l_k = (tid & 3) * 8;
    f32x4 ra[3][2];
    float amax = 0.0f;
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const int row0 = max(m0 - half, 0), lack = row0 - (m0 - half), rows_here = min(rows, m0 + BM + half) - row0;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)row0 * cin)), 0, rows_here * cin * 4, 0x00020000);
    const int aoff = ((l_r - lack) * cin + l_k) * 4;
    auto gloadA = [&](int cb) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            ra[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * LR * cin * 4, cb << 7, 0));
            ra[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * LR * cin * 4 + 16, cb << 7, 0));
        }
    };
    auto lstoreA = [&]() {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (p * LR + l_r < BM + 16) {
                const int o = (p * LR + l_r) * CNN_BP + l_k;
                f16x8 h, l;
                split2(ra[p][0], ra[p][1], h, l, amax);
                *reinterpret_cast<f16x8 *>(&As[0][o]) = h; *reinterpret_cast<f16x8 *>(&As[1][o]) = l;
            }
        }
    };
    // weight DMA: the (tap, piece) block of this workgroup's 128 columns is 8 KB contiguous = 8 pieces of 16 rows; wavefront w always brings
    // rows 16 w .. 16 w + 15 of every block of a step.  Lane: row 16 w + (lane >> 2), LDS chunk lane & 3 <- source chunk ^ ((row >> 2) & 3).
    const int d_row = 16 * wave + (lane >> 2), d_ch = (lane & 3) ^ ((d_row >> 2) & 3);
    const uint16_t *const d_src = Wb + (size_t)n0 * 32 + d_row * 32 + d_ch * 8;
    auto dma = [&](int g, int buf) {                       // step g = (cb, ts): taps ts * DMA_TS .. of channel block cb
        const int cb = g / spc, ts = g - cb * spc;
#pragma unroll
        for (int tt = 0; tt < DMA_TS; tt++) {
            const int tap = ts * DMA_TS + tt;
            if (tap < k) {                                 // wave-uniform
#pragma unroll
                for (int pc = 0; pc < NP; pc++)
                    __builtin_amdgcn_global_load_lds(d_src + ((size_t)((cb * k + tap) * NP + pc) * cout) * 32,
                                                     (__attribute__((address_space(3))) void *)(&Bs[buf][tt][pc][16 * wave * 32]), 16, 0, 0);
            }
        }
    };
    gloadA(0);
    dma(0, 0);
    lstoreA();
    __syncthreads();                                        // (vmcnt(0) of its fence retires the DMA)
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    const int sw = (fm >> 2) & 3;                          // the rows of a fragment differ from fm by multiples of 32: same swizzle term
    const int bch0 = ((0 + (lane >> 5)) ^ sw) * 8, bch1 = ((2 + (lane >> 5)) ^ sw) * 8;
    int cb = 0, ts = 0;
    for (int g = 0; g < nsteps; g++) {
        const int cur = g & 1;
        const bool lastOfBlock = ts == spc - 1;
        if (g + 1 < nsteps) dma(g + 1, cur ^ 1);            // the other buffer was last read in step g - 1: every wavefront has passed that step's barrier
        if (lastOfBlock && cb + 1 < cblocks) gloadA(cb + 1);
        __builtin_amdgcn_sched_barrier(0);
        // fragments one group ahead (a group = one tap x one k16 half = 8 ds_read_b128 for 12 MFMAs): with two wavefronts per SIMD nobody
        // else hides the LDS latency of a group, so the reads of the next group are issued before the MFMAs of the current one
        // (first version of this kernel: reads, wait, MFMAs per group -- 3.59 ms against 3.09 for the 17-tap layer)
        {
            const int ntaps = min(DMA_TS, k - ts * DMA_TS);
            u32x4 a0[2][NP], b0[2][NP], a1[2][NP], b1[2][NP];
            auto frags = [&](u32x4 (&a)[2][NP], u32x4 (&b)[2][NP], int tt, int k16) {
                const int tap = ts * DMA_TS + tt;
#pragma unroll
                for (int pc = 0; pc < NP; pc++) {
#pragma unroll
                    for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[pc][(wm * 64 + i * 32 + fm + tap) * CNN_BP + k16 * 16 + fk]);
#pragma unroll
                    for (int j = 0; j < 2; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(&Bs[cur][tt][pc][(wn * 64 + j * 32 + fm) * 32 + (k16 ? bch1 : bch0)]);
                }
            };
            auto mmas = [&](const u32x4 (&a)[2][NP], const u32x4 (&b)[2][NP]) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};      // l h', h l', h h'
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[i][j] = mfma16<NP>(a[i][PA2[t]], b[j][PB2[t]], acc[i][j]);
                }
            };
            frags(a0, b0, 0, 0);
            for (int tt = 0; tt < ntaps; tt++) {
                frags(a1, b1, tt, 1);
                __builtin_amdgcn_sched_barrier(0);
                mmas(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (tt + 1 < ntaps) frags(a0, b0, tt + 1, 0);       // wave-uniform
                __builtin_amdgcn_sched_barrier(0);
                mmas(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (lastOfBlock && cb + 1 < cblocks) {              // wave-uniform: next channel block's A planes
            __syncthreads();                                // every wavefront is done with the planes
            lstoreA();
        }
        __syncthreads();                                    // planes / the next weight buffer complete (its fence waits for the DMA)
        ts = lastOfBlock ? 0 : ts + 1;
        cb += lastOfBlock ? 1 : 0;
    }
    if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
    conv_epilogue<BN, ADD>(acc, Y, scale, shift, Add, valid, m0, n0, wm, wn, lane, cout, relu, post);
}

