// k3_conv_pp_experiment.hip -- NOT part of libdnascent_hip.so.  Round-2 experiment, kept as the record of a negative result.
//
// Idea: the phase trace of the 256-row k3_conv_split (17 x 128 -> 256: 68 steps, 275 k cycles per tile) showed its MfmaUtil of 0.5 as
// "two wavefronts per SIMD issue their 24 MFMAs one after the other, then BOTH do the step's loads, stores and barriers".  The kernel
// below runs the workgroup's 8 wavefronts as two groups half a step out of phase (one feeds the matrix pipe while the other does the
// memory half-step), B tile double-buffered and requested two steps ahead, A rows per group in its own LDS region.  It is bit-identical
// to k3_conv_split (tools/variant_check.py with DN_CNN_PP=3 / 9 routed to it) -- and SLOWER, 1.2 M positions, f16x3:
//     17 x 128 -> 256   3.64 ms  (k3_conv_split, 256 rows, two workgroups per CU: 3.05)
//      9 x 128 -> 128   1.16 ms  (0.97)         3 x 256 -> 256   1.40-1.57 ms  (1.47, 128 rows)
// Its own phase trace: 167 k cycles per tile (one workgroup per CU: 86 KB of LDS) against 2 x 275 k / 2 for the kept kernel.  The
// half-step in which the OLDER group feeds the matrix pipe and the younger one stores takes ~770 cycles -- the 24 MFMAs, as designed;
// the other half-step takes ~1.7 k: the younger group's MFMA phase stretches to 1.6 k while the older group's two LDS stores of the
// B tile take 750 (priority 1 for the younger half, requesting B two steps ahead instead of one: no change).  Not understood, not kept.
// The fragment needs k3_cnn.hip around it (conv_tile, conv_epilogue, split2, mfma16, CNN_BP) and, in the executor,
//     hipLaunchKernelGGL((k3_conv_pp<ADD>), dim3(conv_grid(rows, o.cout, 128, 256)), dim3(512), 0, st, <the arguments of k3_conv_split>)
// for layers with cout % 128 == 0, 3 <= k <= 17, rows % 256 == 0.

// ---------------------------------------------------------------------------------------------------------
// k3_conv_pp: the long-K convolution with two PING-PONG halves.  The phase trace of the 256-row k3_conv_split (17 x 128 -> 256) showed
// what its MfmaUtil of 0.5 is made of: the two wavefronts a workgroup has on a SIMD are released from the step's barrier together,
// issue their 24 MFMAs one after the other (2 x 768 cycles of matrix pipe) and then BOTH do the step's loads, LDS stores and barriers
// (~2.5 k cycles) -- only the second workgroup of the CU fills that.  Here the workgroup's 8 wavefronts are two groups of 4 (rows
// 0-127 and 128-255) half a step out of phase: while one group issues the MFMAs of step s, the other stores its half of the next B
// tile, requests the one after and (once per channel block) splits and stores its own A rows; a barrier, and they swap.  The matrix
// pipe of every SIMD is fed by one wavefront at any time, the vector / LDS / memory work of the other hides behind it.
//   B tile double-buffered (written during the two half-steps before it is read, into the buffer both groups left two half-steps ago);
//   A rows per group in its own LDS region (144 rows: a group refills it right after its own last tap of the channel block);
//   one workgroup per CU (86 KB of LDS, no 128-VGPR cap: nothing spills); same MFMA order per accumulator as k3_conv_split:
//   bit-identical results.
// ---------------------------------------------------------------------------------------------------------
template <bool ADD>
__global__ __launch_bounds__(512) void k3_conv_pp(const float *__restrict__ X, float *__restrict__ Y, const uint16_t *__restrict__ Wb,
                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                  const float *__restrict__ Add, const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int k,
                                                  int cin, int cout, int relu, float post, unsigned *range_flag) {
    constexpr int BN = 128, NP = 2, BM = 256, GR = 128 + 16;    // GR: A rows a group stages (128 + k - 1 <= 144)
    rows = min(rows, *live);
    __shared__ __attribute__((aligned(16))) uint16_t As[2][NP][GR * CNN_BP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][NP][BN * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, tg = tid & 255;             // group, thread within the group
    const int wm = (wave >> 1) & 1, wn = wave & 1;         // the wavefront's 64 rows inside its group, its 64 columns
    int m0, n0;
    if (!conv_tile(cout, BN, rows, m0, n0, BM)) return;
    constexpr int NJ = BN / 64;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    const int half = (k - 1) / 2, cblocks = cin >> 5, steps = k * cblocks, arows = 128 + k - 1;
    const int gm0 = m0 + grp * 128;                        // first output row of the group
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    // buffer addressing as in k3_conv_split: rows before the pass wrap to a huge offset, rows past it lie beyond num_records: zeros
    const int row0 = max(gm0 - half, 0), lack = row0 - (gm0 - half), rows_here = max(min(rows, gm0 + 128 + half) - row0, 0);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)row0 * cin)), 0, rows_here * cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<uint16_t *>(Wb)), 0, steps * NP * cout * 64, 0x00020000);
    const int l_r = tg >> 2, l_k = (tg & 3) * 8;           // loader: 64 rows x 4 chunks of 8 elements per pass
    const int aoff = ((l_r - lack) * cin + l_k) * 4;       // + p * 64 rows
    const int boff = ((grp * 64 + l_r) * 32 + l_k) * 2;    // the group's 64 columns of the B tile
    f32x4 ra[3][2];
    u32x4 rb[2][NP];                                       // B tiles in flight: set n & 1 carries tile n, requested TWO steps before its store
    float amax = 0.0f;
    auto gloadA = [&](int cb) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            ra[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * 64 * cin * 4, cb << 7, 0));
            ra[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + p * 64 * cin * 4 + 16, cb << 7, 0));
        }
    };
    auto gloadB = [&](u32x4 (&r)[NP], int s) {
#pragma unroll
        for (int pc = 0; pc < NP; pc++) r[pc] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, boff, ((s * NP + pc) * cout + n0) * 64, 0));
    };
    auto lstoreA = [&]() {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (p * 64 + l_r < GR) {
                const int o = (p * 64 + l_r) * CNN_BP + l_k;
                f16x8 h, l;
                split2(ra[p][0], ra[p][1], h, l, amax);
                *reinterpret_cast<f16x8 *>(&As[grp][0][o]) = h; *reinterpret_cast<f16x8 *>(&As[grp][1][o]) = l;
            }
        }
    };
    auto lstoreB = [&](const u32x4 (&r)[NP], int buf) {
#pragma unroll
        for (int pc = 0; pc < NP; pc++) *reinterpret_cast<u32x4 *>(&Bs[buf][pc][(grp * 64 + l_r) * CNN_BP + l_k]) = r[pc];
    };
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    auto mma = [&](int s, int tap) {                        // the MFMAs of step s: same order per accumulator as k3_conv_split
        const int buf = s & 1;
        u32x4 a[2][2][NP], b[2][NJ][NP];
        auto frags = [&](int k16) {
#pragma unroll
            for (int pc = 0; pc < NP; pc++) {
#pragma unroll
                for (int i = 0; i < 2; i++) a[k16][i][pc] = *reinterpret_cast<const u32x4 *>(&As[grp][pc][(wm * 64 + i * 32 + fm + tap) * CNN_BP + k16 * 16 + fk]);
#pragma unroll
                for (int j = 0; j < NJ; j++) b[k16][j][pc] = *reinterpret_cast<const u32x4 *>(&Bs[buf][pc][(wn * 64 + j * 32 + fm) * CNN_BP + k16 * 16 + fk]);
            }
        };
        frags(0); frags(1);                                 // the second half's fragments are in flight during the first half's MFMAs
#pragma unroll
        for (int k16 = 0; k16 < 2; k16++) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};                      // l h', h l', h h'
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        acc[i][j] = mfma16<NP>(a[k16][i][PA2[t]], b[k16][j][PB2[t]], acc[i][j]);
            }
        }
    };
    // memory half-step of a group whose latest MFMA step was `done` (-1: none yet): its half of B(next) into the free buffer, the
    // request for the tile after; after the last tap of a channel block its own A rows of the next block; after a block's first tap
    // the request for those rows
    // (the phase trace of the first version, with ONE step between a tile's request and its store, showed the older group waiting
    // 1.5 k cycles for it in every memory half-step: a B tile takes ~2.5 k cycles to arrive under load, a step is 1.5 k)
    auto mem = [&](u32x4 (&r)[NP], int done, int next) {
        if (next < steps) lstoreB(r, next & 1);
        gloadB(r, min(next + 2, steps - 1));
        if (done >= 0) {
            const int tap = done % k, cb = done / k;
            if (tap == k - 1 && cb + 1 < cblocks) lstoreA();
            if (tap == 0 && cb + 1 < cblocks) gloadA(cb + 1);
        }
    };
    // prologue: both groups stage block 0 of A and their half of B(0); B(1) requested
    gloadA(0); gloadB(rb[0], 0);
    lstoreA(); lstoreB(rb[0], 0);
    gloadB(rb[1], min(1, steps - 1)); gloadB(rb[0], min(2, steps - 1));
    __syncthreads();
    if (grp == 1) __builtin_amdgcn_s_setprio(1);          // the younger half loses every arbitration against the older one otherwise (MI355X_MICROARCH.md)
    if (grp == 0) {
        for (int s = 0; s < steps; s += 2) {                // two steps per trip: the register set of a tile is a compile-time choice
            mma(s, s % k);                                  // half-step 2 s
            __syncthreads();
            if (s + 1 < steps) { mem(rb[1], s, s + 1); __syncthreads(); }   // half-step 2 s + 1 (the last one has no barrier: nothing follows)
            if (s + 1 < steps) {
                mma(s + 1, (s + 1) % k);
                __syncthreads();
                if (s + 2 < steps) { mem(rb[0], s + 1, s + 2); __syncthreads(); }
            }
        }
    } else {
        for (int s = 0; s < steps; s += 2) {
            mem(rb[1], s - 1, s + 1);                       // half-step 2 s: B(s + 1), and the A rows after MFMA step s - 1
            __syncthreads();
            mma(s, s % k);                                  // half-step 2 s + 1
            if (s + 1 < steps) {
                __syncthreads();
                mem(rb[0], s, s + 2);
                __syncthreads();
                mma(s + 1, (s + 1) % k);
                if (s + 2 < steps) __syncthreads();
            }
        }
    }
    if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
    conv_epilogue<BN, ADD>(acc, Y, scale, shift, Add, valid, gm0, n0, wm, wn, lane, cout, relu, post);
}

