// k3_sep_ring_experiment.hip -- NEGATIVE RESULT kept for the record (round 3).  Not built into the library: this is the kernel text as it
// sat in dnascent_amd/csrc/k3_cnn.hip behind k3_sep_ws (it uses that file's helpers: conv_epilogue, mfma16, CNN_BM, CNN_BP, SEP_XPW)
// together with its launch line:
//     hipLaunchKernelGGL((k3_sep_ring<256, 17, ADD, 2>), dim3(min(conv_grid(rows, o.cout, 256), k3_cu_count())), dim3(512), 0, st, SEP_ARGS);
// Measured on MI355X, 64 x 20 kb reads (1.2 M positions), the five 17-tap 256 -> 256 layers (gpurun_out/r3r2, r3s2; all bit-identical to
// k3_sep_ws, tools/variant_check.py):
//     k3_sep_ws (one workgroup barrier per step, producers one step ahead)   3 822 us
//     k3_sep_ring, 4 stages, producers at priority 1, s_sleep 1              4 196 us       (s_sleep 8: 4 198; priority 0: 4 247)
//     k3_sep_ring, 2 stages (the barrier kernel's depth)                     4 397 us       (priority 0: 4 298)
// Why it loses: (1) a polled hand-over (ds_read -> readfirstlane -> compare -> s_sleep) is noticed 100-200 ticks late and its instructions
// compete with the working wavefront of the SIMD; s_barrier costs nothing while waiting -- at equal depth that alone is +15 %.
// (2) Running ahead does not shorten the tile: the consumers' own chain (8 x ~3.35 k ticks of MFMA phases -- twice what the matrix pipe
// needs, they wait for L2 weight fragments -- plus 8.5 k of issue-bound epilogue) is 35 k of the tile's 41 k; producer instructions
// moved into the epilogue window slow the epilogue down by what the producers gain.  What would have to come first: weight fragments that
// arrive in time (LDS-DMA two k16 steps ahead) and an epilogue with 16-byte accesses.
// ---------------------------------------------------------------------------------------------------------
// k3_sep_ring (round 3): k3_sep_ws with its A planes in a RING of RING_NS stages and LDS counters instead of the workgroup barrier.
// With one barrier per step the producers could be ONE step ahead: while the consumers wrote a tile's results (8.5 k of a tile's 41 k
// ticks, tools/ws_trace.py) the producers filtered one step and then waited out the rest of the epilogue at the barrier; and every
// step ended when the slower role of that step did.  Here a producer wavefront writes its 32 rows of a stage and adds 1 to the
// stage's `written` counter (an LDS atomic; LDS executes a wavefront's instructions in order, so the add lands after the stores --
// no s_waitcnt, which would also wait for the slices in flight from HBM); a consumer polls that counter before its first fragment
// read and adds 1 to the stage's `consumed` counter after its last; a producer polls THAT before it overwrites the stage.  Counters
// only grow (use g of a stage is complete at 4 (g + 1)).  Taps: every producer wavefront stages its own copy.  Same arithmetic in
// the same order as k3_sep_ws: bit-identical.
// ---------------------------------------------------------------------------------------------------------
#ifndef RING_NS
#define RING_NS 4
#endif
#ifndef RING_PRIO
#define RING_PRIO 1
#endif
#ifndef RING_SLEEP
#define RING_SLEEP 1
#endif
__device__ __forceinline__ void ring_wait(unsigned *flag, unsigned target) {
    int guard = 1 << 22;                                   // never reached; a protocol error shows as wrong results, not as a hung GPU
    while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target && --guard)
        __builtin_amdgcn_s_sleep(RING_SLEEP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ring_signal(unsigned *flag, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int BN, int KW, bool ADD, int NP>
__global__ __launch_bounds__(512) void k3_sep_ring(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ Wd,
                                                 const uint16_t *__restrict__ Wb, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, const float *__restrict__ Add,
                                                 const uint8_t *__restrict__ valid, int rows, const int *__restrict__ live, int cin, int cout, int relu, float post,
                                                 unsigned *range_flag) {
    rows = min(rows, *live);
    constexpr int SROWS = 32 + KW - 1;                     // raw rows a producer wave needs for its 32 output rows
    constexpr int NLD = (SROWS * 8 + 63) / 64;             // float4 loads per lane for one raw slice
    __shared__ __attribute__((aligned(16))) float Xr[4][SROWS * SEP_XPW];
    __shared__ __attribute__((aligned(16))) float Wl[4][2][KW * 32];       // every producer wavefront stages its OWN copy of a step's taps: no cross-wave hand-over
    __shared__ __attribute__((aligned(16))) uint16_t As[RING_NS][NP][CNN_BM * CNN_BP];
    __shared__ unsigned ring[2 * RING_NS];                 // [stage] planes written (4 per use: one per producer) | [RING_NS + stage] planes consumed (4 per use)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;                       // wave-uniform
    // PERSISTENT workgroup: it takes the row tiles blockIdx.x, + gridDim.x, ... (BN == cout: one column tile) and runs their channel
    // blocks as ONE stream of `nb` steps.  The producers are always one step ahead, so while the consumers write a tile's results
    // (6.8 k ticks of the 47 k a one-tile workgroup took) the producers already filter the next tile's first block, and only the
    // first tile of a workgroup waits for its first planes (8.3 k ticks) -- tools/ws_trace.py.
    const int ntiles = (rows + CNN_BM - 1) / CNN_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_tiles == 0) return;
    const int n0 = 0;
    constexpr int NJ = BN / 64;
    constexpr int NBQ = BN / 64;
    constexpr int half = (KW - 1) / 2;
    const int cblocks = cin >> 5;
    const int nb = my_tiles * cblocks;                     // steps of this workgroup (even: cblocks is)
    auto tile_m0 = [&](int it) { return ((int)blockIdx.x + it * (int)gridDim.x) * CNN_BM; };
    // ---- consumer state ----
    const int cw = wave & 3, wm = cw >> 1, wn = cw & 1;
    const int ct = tid & 255; (void)ct;
    // ---- producer state ----
    const int pw = wave & 3;                               // slice: output rows 32 pw .. 32 pw + 31 of the tile
    const int cp = (lane & 15) * 2, dr = (lane >> 4) * 8;  // channel pair cp, cp + 1; output rows dr .. dr + 7 of the slice
    // Two register sets of raw rows + taps in flight: a slice is stored to LDS TWO iterations after its loads were issued.  With one
    // set the loads had exactly one iteration to land, so an iteration could not be shorter than the HBM latency under load (~3 us
    // against ~0.7 us of MFMA work per channel block: the kernel ran at the memory LATENCY, not at any bandwidth).
    struct RawSet { f32x4 rx[NLD]; bool pin[NLD]; f32x4 rw[3]; bool edge; };
    RawSet S0, S1;
    float amax = 0.0f;
    float *Xs = Xr[pw];
    // The raw slices are fetched in step order, one call per step: the position of the load stream (tile, channel block) is kept
    // incrementally (no division per call); past the last step it stays on the last one (loads past the end are harmless).
    // Interior tiles use buffer addressing: descriptor = the wavefront's SROWS rows, voffset = the lane's (row, float4) inside
    // them (six lane constants), soffset = the channel block -- no 64-bit address arithmetic in the vector unit, which is what
    // this kernel is short of.  The taps likewise; lanes beyond the KW x 8 float4 of a block fall outside the descriptor and
    // read zeros, so there is no branch around that load and the count of loads in flight is the same on every path.
    int ld_cb = 0, ld_it = 0;
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    int xoff[NLD];
#pragma unroll
    for (int p = 0; p < NLD; p++) { const int f = lane + 64 * p; xoff[p] = ((f >> 3) * cin + (f & 7) * 4) * 4; }
    int woff[3];                                           // the KW x 8 float4 of a step's taps: chunk lane + 64 j (beyond them: outside the descriptor, zeros)
#pragma unroll
    for (int j = 0; j < 3; j++) { const int c = lane + 64 * j; woff[j] = ((c >> 3) * cin + (c & 7) * 4) * 4; }
    const __amdgpu_buffer_rsrc_t rtap = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(Wd)), 0, KW * cin * 4, 0x00020000);
    auto gloadX = [&](RawSet &S) {
        const int cb = ld_cb, m0 = tile_m0(ld_it);
        if (ld_cb + 1 < cblocks) ld_cb++; else if (ld_it + 1 < my_tiles) { ld_cb = 0; ld_it++; }
        // rows outside [0, rows) read as zeros ('same' padding at the ends of the pass); only the first and the last row tile have
        // any: they take the clamped-address path and 24 selects per slice (lstoreX), behind wave-uniform branches
        S.edge = m0 - half < 0 || m0 + CNN_BM + half > rows;
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) {
                const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
                const int src = m0 + 32 * pw - half + rr;
                const bool in = rr < SROWS && src >= 0 && src < rows;
                S.rx[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? src : m0) * cin + (cb << 5) + q * 4);
                S.pin[p] = in;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(X + (size_t)(m0 + 32 * pw - half) * cin)), 0, SROWS * cin * 4, 0x00020000);
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, xoff[p], cb << 7, 0));
        }
#pragma unroll
        for (int j = 0; j < 3; j++) S.rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtap, woff[j], cb << 7, 0));
    };
    auto lstoreX = [&](RawSet &S, int wbuf) {
        if (S.edge) {
#pragma unroll
            for (int p = 0; p < NLD; p++) S.rx[p] = S.pin[p] ? S.rx[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = lane + 64 * p, rr = f >> 3, q = f & 7;
            if (rr < SROWS) *reinterpret_cast<f32x4 *>(&Xs[rr * SEP_XPW + q * 4]) = S.rx[p];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) { const int c = lane + 64 * j; if (c < KW * 8) *reinterpret_cast<f32x4 *>(&Wl[pw][wbuf][(c >> 3) * 32 + (c & 7) * 4]) = S.rw[j]; }
    };
    // Depthwise filter of one producer wavefront: 32 output rows x 32 channels per channel block.  A lane owns a PAIR of adjacent
    // channels and 8 consecutive output rows: its 24 input rows and its 17 taps are each read ONCE from LDS as 8-byte pairs
    // (ds_read_b64; the first version re-read every input row 5 times and the taps per 4-row strip as 16-byte quads: 37 KB of LDS
    // reads per wavefront and channel block, now 21 KB) and the filter runs as packed fp32 FMAs (v_pk_fma_f32: the two channels of
    // the pair in one instruction, half the vector instructions), taps in ascending order -- bit-identical to k3_dwconv.
    // Raw-slice pitch 36 floats: the two 16-lane row groups of a 32-lane LDS pass start 8 rows = 1152 B = half a bank window apart.
    auto depthwise = [&](int abuf, int wbuf) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 o[8], w[KW];
#pragma unroll
        for (int t = 0; t < KW; t++) w[t] = *reinterpret_cast<const f32x2 *>(&Wl[pw][wbuf][t * 32 + cp]);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = f32x2{0.f, 0.f};
        // All input rows first, then per input row one GROUP of up to 8 independent FMAs (one per output row it meets), in exactly
        // this order: the FMAs are volatile asm statements, which keep their relative order.  Left to the compiler this became 8
        // serial chains of 17 dependent v_pk_fma_f32 with a hazard nop after each (a dependent vector instruction issues every
        // ~8.5 ticks, an independent one every ~5.5: tools/ubench_coissue.hip, profiles/r01_valu_issue_microbench.txt);
        // __builtin_amdgcn_sched_barrier between the groups did not help, the chains are formed before the scheduler sees them.
        // The phase trace (tools/ws_trace.py) shows the producers' filter, not the matrix work, setting the pace of the workgroup.
        f32x2 x[KW + 7];
#pragma unroll
        for (int j = 0; j < KW + 7; j++) x[j] = *reinterpret_cast<const f32x2 *>(&Xs[(dr + j) * SEP_XPW + cp]);
#pragma unroll
        for (int j = 0; j < KW + 7; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int t = j - i;                       // tap of output row dr + i that input row dr + j meets (ascending per output)
                if (t >= 0 && t < KW) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[i]) : "v"(x[j]), "v"(w[t]));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int off = (32 * pw + dr + i) * CNN_BP + cp;
            if (NP == 3) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 h, m, l;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float x = o[i][e];
                    const __bf16 hh = (__bf16)x; const float r1 = x - (float)hh;
                    const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
                    h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
                }
                *reinterpret_cast<bf16x2 *>(&As[abuf][0][off]) = h; *reinterpret_cast<bf16x2 *>(&As[abuf][1][off]) = m;
                *reinterpret_cast<bf16x2 *>(&As[abuf][NP - 1][off]) = l;
            } else {
                // the pair at once: packed round-to-nearest conversions, packed subtraction (same values as element by element)
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(o[i][0]), __builtin_fabsf(o[i][1])));
                const f16x2 h = __builtin_convertvector(o[i], f16x2);
                const f32x2 rest = o[i] - __builtin_convertvector(h, f32x2);
                const f16x2 l = __builtin_convertvector(rest, f16x2);
                *reinterpret_cast<f16x2 *>(&As[abuf][0][off]) = h; *reinterpret_cast<f16x2 *>(&As[abuf][1][off]) = l;
            }
        }
    };
    // The two roles run their own loops (the accumulators exist only on the consumer side, the filter window only on the producer
    // side: the register allocation is the larger of the two, not the sum) and meet only through the ring's counters.
    if (tid < 2 * RING_NS) ring[tid] = 0u;
    __syncthreads();
    if (producer) {
        __builtin_amdgcn_s_setprio(RING_PRIO);
        // step b's raw slice travels in set S0 for odd b (and step 0), S1 for even b >= 2; iteration b filters step b out of the slice in
        // LDS into stage b % RING_NS, then stores step b + 1's slice and requests step b + 3's
        gloadX(S0); lstoreX(S0, 0); gloadX(S0); gloadX(S1);
        auto iter = [&](int b, RawSet &S) {
            const int stg = b & (RING_NS - 1);
            if (b >= RING_NS) ring_wait(&ring[RING_NS + stg], 4u * (unsigned)(b / RING_NS));      // the consumers are done with the stage's last use
            depthwise(stg, b & 1);
            ring_signal(&ring[stg], lane);
            lstoreX(S, (b + 1) & 1); gloadX(S);
        };
        // (no conditional around the loads of the steady state: see k3_sep_ws; nb is even)
        for (int b = 0; b < nb; b += 2) { iter(b, S0); iter(b + 1, S1); }
        if (NP == 2 && __any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);
        return;
    }
    f32x16 acc[2][NJ];
    // B fragments come STRAIGHT from L2 into registers (pre-split weights [channel block][piece][cout][32]: a fragment is one
    // 16-byte load, 64-byte rows of consecutive lanes coalesce): no B tile in LDS -- that tile was 82 of the kernel's 155 KB, which
    // kept this workgroup off every CU where a per-read stage of another batch held some LDS, and half of its LDS traffic.
    // Double-buffered per k16 step: the loads of step s + 1 are in flight during the MFMAs of step s.
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t *wlane = Wb + ((size_t)(n0 + wn * (BN / 2) + fm)) * 32 + fk;
    auto loadB = [&](u32x4 (&b)[NJ][NP], int step) {            // step = 2 * cb + k16
        const int cb = (step >> 1) % cblocks, k16 = step & 1;          // the weights of a step depend on its channel block only
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int j = 0; j < NJ; j++) b[j][pc] = *reinterpret_cast<const u32x4 *>(wlane + ((size_t)(cb * NP + pc) * cout + j * 32) * 32 + k16 * 16);
    };
    u32x4 b0[NJ][NP], b1[NJ][NP];
    loadB(b0, 0);
    auto mma = [&](int cur, int k16, u32x4 (&b)[NJ][NP]) {
#ifdef DN_WS_NOMMA
        return;
#endif
        u32x4 a[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int i = 0; i < 2; i++) a[i][pc] = *reinterpret_cast<const u32x4 *>(&As[cur][pc][(wm * 64 + i * 32 + fm) * CNN_BP + k16 * 16 + fk]);
        constexpr int NT = NP == 3 ? 6 : 3;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            constexpr int PA3[6] = {1, 2, 0, 1, 0, 0}, PB3[6] = {1, 0, 2, 0, 1, 0};
            constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
            const int pa = NP == 3 ? PA3[t] : PA2[t % 3], pb = NP == 3 ? PB3[t] : PB2[t % 3];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    acc[i][j] = mfma16<NP>(a[i][pa], b[j][pb], acc[i][j]);
        }
    };
    for (int it = 0; it < my_tiles; it++) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        for (int cb = 0; cb < cblocks; cb++) {
            const int step = it * cblocks + cb, cur = step & (RING_NS - 1);
            ring_wait(&ring[cur], 4u * (unsigned)(step / RING_NS + 1));       // all four producers have written this use of the stage
            loadB(b1, 2 * step + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 0, b0);
            __builtin_amdgcn_sched_barrier(0);
            loadB(b0, 2 * step + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 1, b1);
            ring_signal(&ring[RING_NS + cur], lane);                     // every fragment of the stage has been read (the MFMAs that take them are issued)
        }
        conv_epilogue<BN, ADD>(acc, Y, scale, shift, Add, valid, tile_m0(it), n0, wm, wn, lane, cout, relu, post);
    }
}

