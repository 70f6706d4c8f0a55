// k3_chain_experiment.hip -- NOT part of libdnascent_hip.so.  Round-2 experiment, kept as the record of a negative result.
//
// Idea: every separable layer of the 64- and 128-channel blocks runs at HBM speed layer by layer (2 x C x 4 bytes of activation I/O
// per position against ~2 C^2 MACs), so run chains of up to three of them in ONE kernel: a workgroup loads BM rows + halo once,
// keeps the intermediate activations in LDS as fp32 and recomputes the halo rows its neighbours also compute.  The kernel below is
// bit-identical to the layer-by-layer path (tools/cnn_fuse_check.py compared them on 200 k positions) and cuts the HBM traffic of
// those layers to a third -- and is SLOWER (1.2 M positions, f16x3):
//     3 x k9 128->128   layer by layer 3 x 307 = 921 us     chained (BM 32) 1 454 us
//     3 x k5  64-> 64   layer by layer 3 x 130 = 390 us     chained (BM 64)   427 us
//     whole network     21.7 ms                              23.7 ms
// Why: LDS caps the tile at 32 (64) rows, so every workgroup re-fetches the layer's whole pre-split weight matrix from L2 per 32
// rows (6 KB per position, against 1 KB of activations it saves), the three layers' phases (depthwise / GEMM / epilogue) serialise
// inside a workgroup with two barriers each, and only two workgroups fit a CU.  A version that pays off needs the weights resident
// (persistent workgroups; 3 x 16 KB fits for 64 channels, 3 x 64 KB does not for 128) -- not done.
// The fragment needs k3_cnn.hip around it (conv_tile, mfma16, CNN_BP, the executor's CnnRun / k3_can_fuse).

// ---------------------------------------------------------------------------------------------------------
// k3_chain: NL consecutive SeparableConv1D layers (C -> C channels, KW taps, folded BatchNorm + ReLU between them) in ONE
// kernel.  Layer by layer every separable convolution of the 64- and 128-channel blocks moves its whole input and output through
// HBM (2 x C x 4 bytes per position against ~2 C^2 MACs: they run at HBM speed, profiles/r01_e_k3_layers.txt).  Here a workgroup
// owns BM output rows, loads them with a halo of NL x (KW - 1) / 2 rows on either side ONCE, and keeps the activations of the
// intermediate layers in LDS as fp32: layer l's depthwise filter reads them, the pointwise GEMM's epilogue writes layer l + 1's
// in place (the filter has consumed the whole tile into the 16-bit A planes before).  Halo rows are recomputed by the
// neighbouring workgroups -- the same arithmetic in the same order, so every value equals what the layer-by-layer path stores.
//   depthwise   lane = a PAIR of adjacent channels (packed fp32 FMAs), rows in strips of 4 over a register window, taps in
//               ascending order with fmaf: bit-identical to k3_dwconv; results split into two fp16 pieces -> A planes
//   pointwise   per wavefront one 32-column strip x all row tiles, v_mfma_f32_32x32x16_f16 x 3 (l h', h l', h h') per k16 in the
//               order of k3_sep_split; the B fragments of the whole layer come straight from L2 into registers (pre-split weights
//               [channel block][piece][cout][32]: a fragment is one 16-byte load), issued before the depthwise phase
//   epilogue    folded BatchNorm scale / shift (+ bias), ReLU, padding-row mask (valid[] of the GLOBAL row: halo rows included)
// f16x3 only (the other math modes run layer by layer).  LDS: (BM + NL (KW-1)) x C x 4 + 2 planes x C/32 x Rpad x 80 bytes:
// 69 KB for 128 channels (BM 32, NL 3, KW 9), 50 KB for 64 channels (BM 64, NL 3, KW 5): two / three workgroups per CU, so one's
// vector phase runs beside another's matrix phase.
// ---------------------------------------------------------------------------------------------------------
struct ChainArgs {
    const float *wd[3]; const uint16_t *wb[3]; const float *scale[3], *shift[3];
    int relu[3]; float post[3];
};

template <int C, int KW, int NL, int BM>
__global__ __launch_bounds__(256) void k3_chain(const float *__restrict__ X, float *__restrict__ Y, const ChainArgs A,
                                                const uint8_t *__restrict__ valid, int rows, unsigned *range_flag) {
    constexpr int HALF = (KW - 1) / 2;
    constexpr int R0 = BM + 2 * HALF * NL;                       // input rows of the tile
    constexpr int ROUT0 = R0 - 2 * HALF;                         // output rows of the first layer (the most)
    constexpr int RPAD = (ROUT0 + 31) / 32 * 32;
    constexpr int CB = C / 32, CP = C / 2, NG = 256 / CP;        // channel blocks, channel pairs, row groups of the depthwise phase
    constexpr int NS = C / 32, WPS = 4 / NS;                     // column strips; wavefronts that share a strip (C = 64: 2, C = 128: 1)
    constexpr int NRT = (RPAD / 32 + WPS - 1) / WPS;             // row tiles per wavefront
    __shared__ __attribute__((aligned(16))) float Xs[R0 * C];
    __shared__ __attribute__((aligned(16))) uint16_t As[2][CB][RPAD * CNN_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m0, n0;
    if (!conv_tile(C, C, rows, m0, n0, BM)) return;              // one column tile: every workgroup covers all C columns
    const int g0 = m0 - HALF * NL;                               // global row of Xs row 0
    // ---- input tile, fp32, rows outside the buffer as zeros: every load of the thread in flight at once (one HBM latency per tile,
    //      not one per load) ----
    {
        constexpr int NLD = (R0 * (C / 4) + 255) / 256;
        f32x4 v[NLD];
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 256 * p, rr = f / (C / 4), q = f % (C / 4);
            const int g = g0 + rr;
            const bool in = rr < R0 && g >= 0 && g < rows;
            v[p] = *reinterpret_cast<const f32x4 *>(X + (size_t)(in ? g : m0) * C + q * 4);
            if (!in) v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int f = tid + 256 * p, rr = f / (C / 4), q = f % (C / 4);
            if (rr < R0) *reinterpret_cast<f32x4 *>(&Xs[rr * C + q * 4]) = v[p];
        }
    }
    const int cp = (tid % CP) * 2, grp = tid / CP;               // depthwise: channels cp, cp + 1; row group grp
    const int strip = wave % NS, wrt = wave / NS;                // pointwise: column strip, first row tile (then + WPS)
    const int fm = lane & 31, fk = (lane >> 5) * 8;
    float amax = 0.0f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const int rout = R0 - 2 * HALF * (l + 1);                // rows this layer produces; its row a is Xs row a + HALF (l + 1) ... after the write
        // B fragments of the whole layer for this wavefront's strip: [channel block][k16][piece], one 16-byte load each
        u32x4 bf[CB][2][2];
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int k16 = 0; k16 < 2; k16++)
                    bf[cb][k16][pc] = *reinterpret_cast<const u32x4 *>(A.wb[l] + ((size_t)(cb * 2 + pc) * C + strip * 32 + fm) * 32 + k16 * 16 + fk);
        f32x2 w[KW];
#pragma unroll
        for (int t = 0; t < KW; t++) w[t] = *reinterpret_cast<const f32x2 *>(A.wd[l] + (size_t)t * C + cp);
        __syncthreads();                                         // Xs holds this layer's input (tile load / previous epilogue)
        // ---- depthwise: output row a (0 .. rout - 1) reads Xs rows a + HALF l .. + KW - 1 (its own row is a + HALF (l + 1)) ----
        {
            const int per = (rout + NG - 1) / NG;                // rows of one group, in strips of 4
            const int a_lo = grp * per, a_hi = min(rout, a_lo + per);
            uint16_t *ah = &As[0][cp >> 5][0], *al = &As[1][cp >> 5][0];
            for (int a = a_lo; a < a_hi; a += 4) {
                f32x2 o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = f32x2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < KW + 3; j++) {
                    const int xr = min(a + j + HALF * l, R0 - 1);           // rows past the strip's end are never used by a stored output
                    const f32x2 x = *reinterpret_cast<const f32x2 *>(&Xs[xr * C + cp]);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int t = j - i;                                // ascending taps per output row
                        if (t >= 0 && t < KW) o[i] = __builtin_elementwise_fma(x, w[t], o[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (a + i < a_hi) {
                        f16x2 h, lo;
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const float x = o[i][e];
                            amax = fmaxf(amax, fabsf(x));
                            const _Float16 hh = (_Float16)x;
                            h[e] = hh; lo[e] = (_Float16)(x - (float)hh);
                        }
                        const int off = (a + i) * CNN_BP + (cp & 31);
                        *reinterpret_cast<f16x2 *>(ah + off) = h; *reinterpret_cast<f16x2 *>(al + off) = lo;
                    }
                }
            }
        }
        __syncthreads();                                         // A planes complete; nobody reads Xs any more
        // ---- pointwise GEMM: this wavefront's strip x its NRT row tiles (independent accumulators interleave on the matrix pipe) ----
        f32x16 acc[NRT];
#pragma unroll
        for (int q = 0; q < NRT; q++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[q][e] = 0.0f;
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int k16 = 0; k16 < 2; k16++) {
                u32x4 ahf[NRT], alf[NRT];
#pragma unroll
                for (int q = 0; q < NRT; q++) {
                    const int rt = min(wrt + q * WPS, RPAD / 32 - 1);
                    ahf[q] = *reinterpret_cast<const u32x4 *>(&As[0][cb][(rt * 32 + fm) * CNN_BP + k16 * 16 + fk]);
                    alf[q] = *reinterpret_cast<const u32x4 *>(&As[1][cb][(rt * 32 + fm) * CNN_BP + k16 * 16 + fk]);
                }
                // smallest terms first, as in k3_sep_split: l h', h l', h h'
#pragma unroll
                for (int q = 0; q < NRT; q++) acc[q] = mfma16<2>(alf[q], bf[cb][k16][0], acc[q]);
#pragma unroll
                for (int q = 0; q < NRT; q++) acc[q] = mfma16<2>(ahf[q], bf[cb][k16][1], acc[q]);
#pragma unroll
                for (int q = 0; q < NRT; q++) acc[q] = mfma16<2>(ahf[q], bf[cb][k16][0], acc[q]);
            }
        // ---- epilogue: folded BatchNorm, ReLU, padding-row mask; into Xs for the next layer or out to HBM ----
        const int col = strip * 32 + fm;
        const float sc_ = A.scale[l][col] * A.post[l], sh_ = A.shift[l][col];
        const float floor_ = A.relu[l] ? 0.0f : -3.402823466e38f;
#pragma unroll
        for (int q = 0; q < NRT; q++) {
            const int rt = wrt + q * WPS;
            if (rt * 32 >= rout) continue;                        // wave-uniform: a tile of pure padding
            // validity of the tile's 32 rows (GLOBAL rows: padding rows between reads and rows outside the buffer stay zero)
            const int gl = g0 + HALF * (l + 1) + rt * 32 + fm;
            const unsigned vmask = (unsigned)__ballot((lane < 32) && (rt * 32 + fm < rout) && gl >= 0 && gl < rows && valid[min(max(gl, 0), rows - 1)] != 0);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ro = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int a = rt * 32 + ro;
                float y = __builtin_fmaf(acc[q][e], sc_, sh_);
                y = fmaxf(y, floor_);
                y = ((vmask >> ro) & 1u) ? y : 0.0f;
                if (a < rout) {
                    if (l + 1 < NL) Xs[(a + HALF * (l + 1)) * C + col] = y;
                    else { const int g = m0 + a; if (g < rows) Y[(size_t)g * C + col] = y; }
                }
            }
        }
    }
    if (__any(amax > 65504.0f) && lane == 0) atomicOr(range_flag, 1u);    // out of fp16 range: the host repeats the pass in bf16x6
}


// ---- executor side ----
static bool k3_chain_enabled() { static const bool on = !(getenv("DN_CNN_CHAIN") && atoi(getenv("DN_CNN_CHAIN")) == 0); return on; }

// how many consecutive separable layers starting at op i can run as one k3_chain launch (0: none).  They must all be fusable pairs
// (k3_can_fuse) of the same width C -> C and tap count, without a residual join, ping-ponging between the same two buffers, and the
// f16x3 split must be active.
static int k3_chain_length(const CnnRun &c, int i) {
    if (c.pieces != 2 || !k3_chain_enabled()) return 0;
    int n = 0;
    const dn_cnn_op &d0 = c.ops[i];
    for (int j = i; j + 1 < c.n_ops && n < 3; j += 2) {
        if (!k3_can_fuse(c, j)) break;
        const dn_cnn_op &d = c.ops[j], &p = c.ops[j + 1];
        if (p.op != DN_CNN_CONV || p.cin != p.cout || p.cin != d0.cin || d.k != d0.k || d.src != (j == i ? d0.src : c.ops[j - 1].dst) ||
            d.dst != d0.dst || p.dst != c.ops[i + 1].dst || d.src == d.dst) break;
        if (!((p.cin == 64 && d.k == 5) || (p.cin == 128 && d.k == 9))) break;
        n++;
    }
    return n >= 2 ? n : 0;
}

template <int C, int KW, int BM>
static void k3_launch_chain(const CnnRun &c, int i, int nl, const float *in, float *out, hipStream_t st) {
    ChainArgs A{};
    for (int l = 0; l < nl; l++) {
        const dn_cnn_op &d = c.ops[i + 2 * l], &p = c.ops[i + 2 * l + 1];
        A.wd[l] = c.wts + d.w; A.wb[l] = c.wts_split + c.wb_off[i + 2 * l + 1]; A.scale[l] = c.wts + p.scale; A.shift[l] = c.wts + p.shift;
        A.relu[l] = p.relu; A.post[l] = c.post[i + 2 * l + 1];
    }
    const unsigned rows = c.rows.rows;
    const dim3 grid(conv_grid(rows, C, C, BM));
    if (nl == 3) hipLaunchKernelGGL((k3_chain<C, KW, 3, BM>), grid, dim3(256), 0, st, in, out, A, c.valid, (int)rows, c.range_flag);
    else hipLaunchKernelGGL((k3_chain<C, KW, 2, BM>), grid, dim3(256), 0, st, in, out, A, c.valid, (int)rows, c.range_flag);
}


// ---- inside k3_run's op loop ----
        if (const int nl = k3_chain_length(c, i)) {
            // a run of separable layers in one launch: reads the first depthwise op's input, writes the (dead) depthwise destination;
            // that buffer then IS the last pointwise op's destination (same swap as the single fused layer below)
            const dn_cnn_op &pw = c.ops[i + 1];
            const float *in = pb[o.src]; float *out = pb[o.dst];
            if (pw.cin == 64) k3_launch_chain<64, 5, 64>(c, i, nl, in, out, st);
            else k3_launch_chain<128, 9, 32>(c, i, nl, in, out, st);
            { float *t = pb[pw.dst]; pb[pw.dst] = pb[o.dst]; pb[o.dst] = t; }
            i += 2 * nl - 1;
            continue;
        }
