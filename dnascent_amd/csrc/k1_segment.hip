// k1_segment.hip -- K1: raw int16 signal -> scrappie events -> DNAscent events, per read, on gfx950.
//
// Replaces detect_events (scrappie/event_detection.c:268-319) and the event build of normaliseEvents
// (event_handling.cpp:546-575) with results bit-identical to the reference:
//
//   k1_carry    serial fp64 prefix sums of x and x*x (event_detection.c:42-47).  Rounding of a running fp64 sum depends on the
//               order of the additions, so the order is kept: four reads per wavefront, the two dependent adds per sample run by 16
//               lanes with EXEC narrowed per sample.  Only the running sums at every 256th sample (and at the read's end) leave
//               the kernel: 16 bytes per 256 samples.  (Round 1 stored all of them: 16 bytes per sample, read back three times.)
//   k1_detect   ONE pass over the signal for everything else: a lane owns a 1024-sample chunk, restarts the exact chain from the
//               stored carry 256 samples before it (64 independent chains per wavefront, all lanes busy), forms both t-statistics
//               (event_detection.c:60-115; the mixed float/double expression order written out cast by cast) from a register
//               window of 28 prefix sums, and steps the short/long peak detector (event_detection.c:122-198) on them -- neither
//               the prefix sums nor the t-statistics ever touch HBM.  The detector is a serial state machine and is run
//               SPECULATIVELY: it starts 192 samples before the chunk from the default state; after a common emitted peak the
//               state no longer depends on history, so the state at the chunk start is almost always the true one.  Every peak
//               is recorded with the exact prefix sum at its position, which is all the event means need.
//   k1_events   verifies every chunk hand-off exactly (state in == previous state out), recomputes the rare chunk whose
//               speculation missed (so the result is exact, never approximate), then streams the recorded peaks ONCE through an
//               LDS tile: event means from the recorded sums (event_detection.c:213-266) and the event-build quirks of
//               event_handling.cpp:549-575 (first mean 0.0, last event dropped, mean <= 0 merged) in the same pass -- it reads
//               12 bytes per peak and writes 16 per event, the scrappie table itself never exists in HBM.
//   (taps)      k1_scan4<true> / k1_tstat write every prefix sum and both t-statistics, k1_events the scrappie table, to HBM when
//               dn_debug_keep_k1 asked for them (parity tests); same arithmetic functions / same registers as the product path.
#include "dn_dev.h"
#include <float.h>

#define SCAN_CHUNK 256
#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")   /* one wavefront per block: LDS ops are in order */

// ------------------------------------------------------------------------------------------------
// k1_scan4 (k1_carry): the order-exact running sums s += x, q += x * x of FOUR reads per wavefront.  The sums must be accumulated
// strictly left to right (their rounding depends on the order), so per sample there is an irreducible serial part of two dependent
// fp64 adds; everything else is 64 lanes wide: a chunk of 256 samples per read is loaded coalesced, converted int16 -> pA (fp32,
// pod5.cpp:60) -> fp64, squared and parked in LDS as {x, x * x}.  A single-lane LDS store costs ~50 cycles, so the chain is run by 16
// lanes at once (broadcast LDS reads; lane j stops adding after sample g + j, so it ends a group of 16 holding the running sums
// after ITS sample) -- and since those adds are whole-wavefront instructions whatever the number of active lanes, each 16-lane row
// of the wavefront runs the chain of its own read (EXEC narrowed with
// the 16-bit pattern replicated four times: two 32-bit scalar moves, s_mov_b64 takes no 64-bit literal), so a batch costs a quarter of the vector instructions for the same chain latency.
// Reads of different lengths: a finished (or absent) read keeps adding +0.0, which is exact (the running sums are never -0.0:
// they start at +0.0 and round-to-nearest never produces -0.0 from a sum with a +0.0 or non-zero operand), and writes nothing.
// The carry of a row (its lane 15) returns through LDS: the row stores its 16 results and reads the last one back.
// ------------------------------------------------------------------------------------------------
#define SCAN4_ADD8(v, o, m0, m1, m2, m3, m4, m5, m6, m7)                                                                         \
    {                                                                                                                             \
        unsigned long long sv_;                                                                                                   \
        asm volatile("s_mov_b64 %2, exec\n\t"                                                                                     \
                     "s_mov_b32 exec_lo, " m0 "\n\ts_mov_b32 exec_hi, " m0 "\n\tv_add_f64 %0, %0, %3\n\tv_add_f64 %1, %1, %4\n\t"                                \
                     "s_mov_b32 exec_lo, " m1 "\n\ts_mov_b32 exec_hi, " m1 "\n\tv_add_f64 %0, %0, %5\n\tv_add_f64 %1, %1, %6\n\t"                                \
                     "s_mov_b32 exec_lo, " m2 "\n\ts_mov_b32 exec_hi, " m2 "\n\tv_add_f64 %0, %0, %7\n\tv_add_f64 %1, %1, %8\n\t"                                \
                     "s_mov_b32 exec_lo, " m3 "\n\ts_mov_b32 exec_hi, " m3 "\n\tv_add_f64 %0, %0, %9\n\tv_add_f64 %1, %1, %10\n\t"                               \
                     "s_mov_b32 exec_lo, " m4 "\n\ts_mov_b32 exec_hi, " m4 "\n\tv_add_f64 %0, %0, %11\n\tv_add_f64 %1, %1, %12\n\t"                              \
                     "s_mov_b32 exec_lo, " m5 "\n\ts_mov_b32 exec_hi, " m5 "\n\tv_add_f64 %0, %0, %13\n\tv_add_f64 %1, %1, %14\n\t"                              \
                     "s_mov_b32 exec_lo, " m6 "\n\ts_mov_b32 exec_hi, " m6 "\n\tv_add_f64 %0, %0, %15\n\tv_add_f64 %1, %1, %16\n\t"                              \
                     "s_mov_b32 exec_lo, " m7 "\n\ts_mov_b32 exec_hi, " m7 "\n\tv_add_f64 %0, %0, %17\n\tv_add_f64 %1, %1, %18\n\t"                              \
                     "s_mov_b64 exec, %2"                                                                                         \
                     : "+v"(s), "+v"(q), "=&s"(sv_)                                                                               \
                     : "v"(v[o + 0].x), "v"(v[o + 0].y), "v"(v[o + 1].x), "v"(v[o + 1].y), "v"(v[o + 2].x), "v"(v[o + 2].y),        \
                       "v"(v[o + 3].x), "v"(v[o + 3].y), "v"(v[o + 4].x), "v"(v[o + 4].y), "v"(v[o + 5].x), "v"(v[o + 5].y),        \
                       "v"(v[o + 6].x), "v"(v[o + 6].y), "v"(v[o + 7].x), "v"(v[o + 7].y));                                        \
    }

template <bool FULL>
__global__ __launch_bounds__(64) void k1_scan4(BatchDev B) {
    __shared__ double2 buf[4][SCAN_CHUNK];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    const int r = blockIdx.x * 4 + g;
    const bool have = r < B.n_reads;
    const uint64_t s0 = have ? B.samp_off[r] : 0ull;
    const unsigned n = have ? (unsigned)(B.samp_off[r + 1] - s0) : 0u;
    const int16_t *a = B.adc + s0;
    double2 *out = FULL ? B.psum + s0 : nullptr;
    double2 *carry = B.carry + (have ? 4ull * B.chunk_off[r] + (unsigned)r : 0ull);      // {sum[256 j], sumsq[256 j]}, j = 0 .. ceil(n / 256)
    if (have && l == 0) carry[0] = make_double2(0.0, 0.0);
    const float off = have ? B.cal_off[r] : 0.0f, sc = have ? B.cal_scale[r] : 0.0f;
    unsigned nmax = n;
    nmax = max(nmax, (unsigned)__shfl_xor((int)nmax, 16)); nmax = max(nmax, (unsigned)__shfl_xor((int)nmax, 32));
    nmax = (unsigned)__builtin_amdgcn_readfirstlane((int)nmax);          // longest of the four reads (wave-uniform)
    double s = 0.0, q = 0.0;
    double2 *mybuf = buf[g];
    int16_t nxt[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const unsigned i = (unsigned)(j * 16 + l); nxt[j] = i < n ? a[i] : (int16_t)0; }
    for (unsigned base = 0; base < nmax; base += SCAN_CHUNK) {
        const unsigned cnt = base < n ? min((unsigned)SCAN_CHUNK, n - base) : 0u;    // samples of THIS row's read in the chunk
        // ---- parallel: convert + square, 16 samples per lane (lanes of a row take consecutive samples) ----
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned i = (unsigned)(j * 16 + l);
            const float v = ((float)nxt[j] + off) * sc;            // pod5.cpp:60
            const double x = (double)v;
            mybuf[i] = i < cnt ? make_double2(x, x * x) : make_double2(0.0, 0.0);
        }
        {
            const unsigned nb = base + SCAN_CHUNK;
#pragma unroll
            for (int j = 0; j < 16; j++) { const unsigned i = nb + (unsigned)(j * 16 + l); nxt[j] = i < n ? a[i] : (int16_t)0; }
        }
        LDS_FENCE();
        unsigned cmax = cnt;
        cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, 16)); cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, 32));
        const unsigned ngroups = ((unsigned)__builtin_amdgcn_readfirstlane((int)cmax) + 15u) >> 4;
        // ---- serial, order-exact (event_detection.c:45-46), four reads at once ----
        for (unsigned gq = 0; gq < ngroups; gq++) {
            double2 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = mybuf[gq * 16 + j];
            SCAN4_ADD8(v, 0, "0xffffffff", "0xfffefffe", "0xfffcfffc", "0xfff8fff8", "0xfff0fff0", "0xffe0ffe0", "0xffc0ffc0", "0xff80ff80")
            SCAN4_ADD8(v, 8, "0xff00ff00", "0xfe00fe00", "0xfc00fc00", "0xf800f800", "0xf000f000", "0xe000e000", "0xc000c000", "0x80008000")
            mybuf[gq * 16 + l] = make_double2(s, q);
            LDS_FENCE();
            const double2 c = mybuf[gq * 16 + 15];                 // the row's lane 15: running sums after the 16th sample
            s = c.x; q = c.y;
        }
        LDS_FENCE();
        // the running sums after this chunk (a finished read kept adding +0.0: they are sum[n]); the last carry of a read is its total
        if (cnt && l == 0) carry[base / SCAN_CHUNK + 1] = make_double2(s, q);
        if (FULL) {
            // ---- taps: parallel write-out of every prefix sum, 256 contiguous bytes per row and store ----
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const unsigned i = (unsigned)(j * 16 + l);
                if (i < cnt) out[base + i] = mybuf[i];
            }
        }
        LDS_FENCE();
    }
}

// ------------------------------------------------------------------------------------------------
// k1_tstat: one thread per sample
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 prefix_at(const double2 *P, unsigned j) {   // {sum[j], sumsq[j]}
    return j == 0 ? make_double2(0.0, 0.0) : P[j - 1];
}

// Division by the window length (3 or 6) as multiply + FMA correction: q = a r; q' = fma(fma(-q, d, a), r, q) with r = RN(1 / d)
// is the correctly rounded a / d (Markstein).  Checked against IEEE division for d = 3 and 6: fp32 EXHAUSTIVELY (every finite
// input; the only mismatches have |a / d| < 4 FLT_MIN, hence the guard), fp64 on 6.4e9 random + mantissa-edge cases with
// exponents within 2^+-200 (guarded likewise).  Outside the guards divq<false> raises `bad` and the wavefront repeats the
// statistic with the hardware division sequence (divq<true>).
// (float)(fabs((double)dm) / sqrt((double)vw)) of event_detection.c:111 -- an IEEE fp64 sqrt and an IEEE fp64 division (~55
// instructions) whose result is immediately rounded to fp32.  Fast path: y = |dm| * rsqrt(vw) from v_rsq_f64 + two Newton steps
// is within ~2^-50 relative (< 10 ulp64) of the reference's double q; (float)y == (float)q unless a rounding boundary of fp32
// (a midpoint between two floats: low 29 mantissa bits == 0x10000000) lies between them.  y within 1024 ulp64 of such a
// midpoint (64 would do for a 2^-26 v_rsq_f64; 1024 = 2^-43 relative is kept as slack) -- 4e-6 of all inputs -- or outside the
// normal fp32 range takes the exact path; everything else is bit-identical by construction.
// One t-statistic.  EXACT = false: every division runs its fast form unconditionally and `bad` collects the cases the fast forms
// are not proven for (operands outside the verified ranges of div_w, ratio near an fp32 rounding boundary); EXACT = true: the
// literal IEEE operations.  The kernel evaluates both windows fast and repeats a wavefront exactly only if any lane raised
// `bad` (4e-6 of the samples): ONE wave-level branch per thread instead of eleven data-dependent ones -- on this kernel the
// scalar unit (branch bookkeeping), not the vector unit, was the larger instruction stream.
template <bool EXACT> __device__ __forceinline__ float divq(float a, float d, float r, bool &bad) {
    if (EXACT) return a / d;
    bad = bad || !(fabsf(a) >= 1e-30f || a == 0.0f);            // tiny / NaN: outside the exhaustively verified range (+0 is fine)
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, a), r, q);
}
template <bool EXACT> __device__ __forceinline__ double divq(double a, double d, double r, bool &bad) {
    if (EXACT) return a / d;
    bad = bad || !((fabs(a) >= 1e-60 && fabs(a) <= 1e60) || a == 0.0);
    const double q = a * r;
    return fma(fma(-q, d, a), r, q);
}
template <bool EXACT> __device__ __forceinline__ float tstat_ratio(float dm, float vw, bool &bad) {
    const double a = fabs((double)dm), v = (double)vw;
    if (EXACT) return (float)(a / sqrt(v));
    double s = __builtin_amdgcn_rsq(v);
    s = s * fma(-0.5 * v * s, s, 1.5);
    s = s * fma(-0.5 * v * s, s, 1.5);
    const double y = a * s;
    const unsigned lo = (unsigned)__double_as_longlong(y) & 0x1FFFFFFFu;
    bad = bad || (lo - 0x0FFFFC00u) <= 0x800u || !((y > 0x1p-100 && y < 0x1p100) || y == 0.0);
    return (float)y;
}

// one t-statistic from the three prefix pairs it needs: a = {sum, sumsq}[i - w], b = [i], c = [i + w] (event_detection.c:89-111)
template <unsigned W, bool EXACT>
__device__ __forceinline__ float tstat_from(const double2 a, const double2 b, const double2 c, const bool valid, bool &bad) {
    constexpr float wf = (float)W;
    constexpr float rwf = 1.0f / wf;                             // RN(1 / w), folded by the compiler exactly as IEEE division
    constexpr double wd = (double)wf, rwd = 1.0 / wd;
    bool mybad = false;
    const double sum1 = b.x - a.x, sumsq1 = b.y - a.y;           // :90-95 (sum[0] == 0, so i == w is the same expression)
    const float sum2 = (float)(c.x - b.x);                       // :96
    const float sumsq2 = (float)(c.y - b.y);                     // :97
    const float mean1 = (float)divq<EXACT>(sum1, wd, rwd, mybad);            // :98
    const float mean2 = divq<EXACT>(sum2, wf, rwf, mybad);                   // :99
    const float m1sq = mean1 * mean1, m2sq = mean2 * mean2;
    const float s2w = divq<EXACT>(sumsq2, wf, rwf, mybad);
    float var = (float)(((divq<EXACT>(sumsq1, wd, rwd, mybad) - (double)m1sq) + (double)s2w) - (double)m2sq);   // :100-101
    var = fmaxf(var, FLT_MIN);                                   // :104
    const float dm = mean2 - mean1;                              // :110
    const float vw = divq<EXACT>(var, wf, rwf, mybad);
    const float t = tstat_ratio<EXACT>(dm, vw, mybad);           // :111
    bad = bad || (valid && mybad);
    return valid ? t : 0.0f;
}

template <unsigned W, bool EXACT>
__device__ __forceinline__ float tstat_at(const double2 *P, unsigned n, unsigned i, bool &bad) {
    constexpr unsigned w = W;
    const bool valid = n >= 2 * w && i >= w && i <= n - w;       // :76-86, loop bound :89 is inclusive; elsewhere the statistic is 0
    const unsigned ii = valid ? i : w;                           // a position whose loads are in range whenever n >= 2 w
    const double2 z = make_double2(0.0, 0.0);
    const double2 a = n >= 2 * w ? prefix_at(P, ii - w) : z, b = n >= 2 * w ? prefix_at(P, ii) : z, c = n >= 2 * w ? prefix_at(P, ii + w) : z;
    return tstat_from<W, EXACT>(a, b, c, valid, bad);
}

__global__ __launch_bounds__(256) void k1_tstat(BatchDev B) {
    const int r = blockIdx.y;
    const uint64_t s0 = B.samp_off[r];
    const unsigned n = (unsigned)(B.samp_off[r + 1] - s0);
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 *P = B.psum + s0;
    bool bad = false;
    float t1 = tstat_at<3, false>(P, n, i, bad);                  // event_detection.h:19-25
    float t2 = tstat_at<6, false>(P, n, i, bad);
    if (__builtin_expect(__any(bad), 0)) { bool x = false; t1 = tstat_at<3, true>(P, n, i, x); t2 = tstat_at<6, true>(P, n, i, x); }
    B.t1[s0 + i] = t1;
    B.t2[s0 + i] = t2;
}


// ------------------------------------------------------------------------------------------------
// peak detector state machine (event_detection.c:136-195), shared by the speculative pass and the exact redo
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ SegState seg_initial() {
    SegState st;
    st.s.masked_to = 0; st.s.peak_pos = -1; st.s.peak_val = FLT_MAX; st.s.valid = 0;
    st.l = st.s;
    return st;
}

// Where a lane's emitted peaks go: position + exact prefix sum.  HBM writes are write-through at 32-byte granularity, so a lane
// that stored every peak on its own (4 + 8 bytes) would write 64 bytes for 12; peaks are parked in the lane's column of a small
// LDS block and leave SINK_N at a time as full 32- / 64-byte pieces.  (16 at a time: WRITE_SIZE unchanged -- 583 against 602 MiB for
// 45 M peaks of 12 bytes, i.e. the pieces of eight already leave whole -- and k1_detect 3.5 -> 4.7 ms: eight it is.)
#define SINK_N 8
struct PeakSink {
    unsigned *pk; double *pks;           // the chunk's slots in HBM (DN_SEG_PEAKCAP each, 64-byte aligned)
    unsigned *bpos; double *bsum;        // this lane's column of the LDS block: entry k at [k * 64]
    __device__ __forceinline__ void put(unsigned pos, double sum, unsigned &npk) {
        const unsigned k = npk & (SINK_N - 1u);
        bpos[k * 64] = pos; bsum[k * 64] = sum;
        npk++;
        if ((npk & (SINK_N - 1u)) == 0u && npk <= DN_SEG_PEAKCAP) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            u32x4_ *dp = reinterpret_cast<u32x4_ *>(pk + (npk - SINK_N));
#pragma unroll
            for (int q = 0; q < SINK_N / 4; q++) { const u32x4_ a = { bpos[(4 * q) * 64], bpos[(4 * q + 1) * 64], bpos[(4 * q + 2) * 64], bpos[(4 * q + 3) * 64] }; dp[q] = a; }
            double2 *ds = reinterpret_cast<double2 *>(pks + (npk - SINK_N));
#pragma unroll
            for (int q = 0; q < SINK_N / 2; q++) ds[q] = make_double2(bsum[(2 * q) * 64], bsum[(2 * q + 1) * 64]);
        }
    }
    __device__ __forceinline__ void flush(unsigned npk) {                    // the last, partial group
        const unsigned m = min(npk, (unsigned)DN_SEG_PEAKCAP);
        for (unsigned j = m & ~(SINK_N - 1u); j < m; j++) { pk[j] = bpos[(j & (SINK_N - 1u)) * 64]; pks[j] = bsum[(j & (SINK_N - 1u)) * 64]; }
    }
};

// one sample of both detectors.  cur: the prefix sum sum[i]; a detector that moves its peak to i remembers it (ss / ls), and an
// emitted peak is recorded together with that sum -- create_event needs nothing else of the signal (event_detection.c:224-226)
template <bool EMIT>
__device__ __forceinline__ void seg_step(SegState &st, double &ss, double &ls, int i, float v1, float v2, double cur, PeakSink &sink, unsigned &npk) {
    const float peak_height = 0.2f;
    // short detector: window 3, threshold 1.4
    if (!(st.s.masked_to >= i)) {                                 // :140
        const float v = v1;
        if (st.s.peak_pos == -1) {
            if (v < st.s.peak_val) st.s.peak_val = v;
            else if (v - st.s.peak_val > peak_height) { st.s.peak_val = v; st.s.peak_pos = i; ss = cur; }
        } else {
            if (v > st.s.peak_val) { st.s.peak_val = v; st.s.peak_pos = i; ss = cur; }
            if (st.s.peak_val > 1.4f) {                           // :166-176 short dominates long
                st.l.masked_to = st.s.peak_pos + 3;
                st.l.peak_pos = -1; st.l.peak_val = FLT_MAX; st.l.valid = 0;
            }
            if (st.s.peak_val - v > peak_height && st.s.peak_val > 1.4f) st.s.valid = 1;
            if (st.s.valid && (i - st.s.peak_pos) > 1) {          // window_length / 2 == 1
                if (EMIT) sink.put((unsigned)st.s.peak_pos, ss, npk);
                st.s.peak_pos = -1; st.s.peak_val = v; st.s.valid = 0;
            }
        }
    }
    // long detector: window 6, threshold 9.0
    if (!(st.l.masked_to >= i)) {
        const float v = v2;
        if (st.l.peak_pos == -1) {
            if (v < st.l.peak_val) st.l.peak_val = v;
            else if (v - st.l.peak_val > peak_height) { st.l.peak_val = v; st.l.peak_pos = i; ls = cur; }
        } else {
            if (v > st.l.peak_val) { st.l.peak_val = v; st.l.peak_pos = i; ls = cur; }
            if (st.l.peak_val - v > peak_height && st.l.peak_val > 9.0f) st.l.valid = 1;
            if (st.l.valid && (i - st.l.peak_pos) > 3) {          // window_length / 2 == 3
                if (EMIT) sink.put((unsigned)st.l.peak_pos, ls, npk);
                st.l.peak_pos = -1; st.l.peak_val = v; st.l.valid = 0;
            }
        }
    }
}

// states compare equal if they behave identically from sample `at` on
__device__ __forceinline__ bool seg_equal(const SegState &a, const SegState &b, int at) {
    const int am = a.l.masked_to >= at ? a.l.masked_to : -1, bm = b.l.masked_to >= at ? b.l.masked_to : -1;
    return a.s.peak_pos == b.s.peak_pos && __float_as_int(a.s.peak_val) == __float_as_int(b.s.peak_val) &&
           a.s.valid == b.s.valid && am == bm && a.l.peak_pos == b.l.peak_pos &&
           __float_as_int(a.l.peak_val) == __float_as_int(b.l.peak_val) && a.l.valid == b.l.valid;
}

// ------------------------------------------------------------------------------------------------
// The chunk walker: chain -> t-statistics -> detector for one 1024-sample chunk per lane.
//   stream      sample index i = chunk * 1024 - 256 + p, p = 0 .. 1299 (25 tiles of 52): the chain restarts from the carry k1_carry
//               left at chunk * 1024 - 256 (a multiple of 256) -- for chunk 0 from 0 with the samples "before the read" as exact
//               zeros (adding +0.0 leaves the sums untouched), likewise beyond the read's end;
//   tile        64 rows (lanes) x 52 samples, int16 -> pA in fp32 (pod5.cpp:60) on the way into LDS, read back one row per lane
//               (row stride 53 words: conflict-free);
//   body        13 samples at a time, fully unrolled, over a RING of 13 prefix pairs held in registers: sample j's sums overwrite
//               slot j (the oldest), and the statistic of sample i0 = i - 5 finds the pairs at i0 - 6, - 3, 0, + 3, + 6 in slots
//               j + 1, + 4, + 7, + 10, + 13 (mod 13) -- static indices, no copying.  The detector so runs 5 samples behind the chain;
//   detector    speculative mode: default state from 192 samples before the chunk (peaks recorded from the chunk start on, the state
//               AT the chunk start is saved for k1_events' exact hand-off check); redo mode: every lane walks the SAME chunk from
//               the given true state (wave-uniform; every lane stores the same values).
// ------------------------------------------------------------------------------------------------
#define SEG_BODY 13
#define SEG_TILE (4 * SEG_BODY)
#define SEG_PITCH (SEG_TILE + 1)
#define SEG_PRE DN_SEG_CARRY                                  // chain warm-up = distance to the carry the stream starts from
#define SEG_TILES ((SEG_PRE + DN_SEG_CHUNK + 5 + SEG_TILE - 1) / SEG_TILE)

__device__ __forceinline__ void seg_body13(const float *row, const int A /* sample index of row[0] */, const int n, const int det_lo, const int det_hi,
                                           const int emit_lo, SegState *save_at_emit_lo, double &S, double &Q, double (&rs)[13], double (&rq)[13],
                                           SegState &st, double &ss, double &ls, PeakSink &sink, unsigned &npk) {
#pragma unroll
    for (int j = 0; j < SEG_BODY; j++) {
        const double x = (double)row[j];
        S = S + x; Q = Q + x * x;                             // the order-exact chain (event_detection.c:45-46): {sum, sumsq}[A + j + 1]
        rs[j] = S; rq[j] = Q;                                 // overwrites {sum, sumsq}[A + j - 12]: the ring now holds [A + j - 11, A + j + 1]
        const int i0 = A + j - 5;                             // the sample whose windows are complete: needs sum[i0 - 6 .. i0 + 6]
        const bool on = i0 >= det_lo && i0 < det_hi;
        if (!__any(on)) continue;
        const unsigned un = (unsigned)n, ui = (unsigned)i0;
        const bool v3 = un >= 6u && ui >= 3u && ui <= un - 3u, v6 = un >= 12u && ui >= 6u && ui <= un - 6u;   // :76-89
#define RING(d) make_double2(rs[(j + 1 + (d)) % SEG_BODY], rq[(j + 1 + (d)) % SEG_BODY])      /* {sum, sumsq}[i0 - 6 + d] */
        const double2 p0 = RING(0), p3 = RING(3), p6 = RING(6), p9 = RING(9), p12 = RING(12);
#undef RING
        bool bad = false;
        float t1 = tstat_from<3, false>(p3, p6, p9, v3 && on, bad);          // event_detection.h:19-25
        float t2 = tstat_from<6, false>(p0, p6, p12, v6 && on, bad);
        if (__builtin_expect(__any(bad), 0)) { bool x2 = false; t1 = tstat_from<3, true>(p3, p6, p9, v3, x2); t2 = tstat_from<6, true>(p0, p6, p12, v6, x2); }
        if (save_at_emit_lo && i0 == emit_lo) *save_at_emit_lo = st;         // wave-uniform position: the state at the chunk start
        if (on) {
            if (i0 >= emit_lo) seg_step<true>(st, ss, ls, i0, t1, t2, p6.x, sink, npk);
            else { unsigned none = 0; seg_step<false>(st, ss, ls, i0, t1, t2, p6.x, sink, none); }
        }
    }
}

// walks chunk `c` of read r on every lane that `have`s one.  REDO: all lanes walk the same chunk from state `st`.
template <bool REDO>
__device__ __forceinline__ void seg_walk(const BatchDev &B, float *tile /* [64 * SEG_PITCH] */, unsigned *lds_pos /* [SINK_N * 64] */, double *lds_sum /* [SINK_N * 64] */,
                                         const int r, const int cbase, const int c, const bool have, SegState &st, double &ss, double &ls, unsigned &npk) {
    const int lane = threadIdx.x;
    const uint64_t s0 = B.samp_off[r];
    const int n = (int)(B.samp_off[r + 1] - s0);
    const uint64_t c0 = B.chunk_off[r];
    const int16_t *adc = B.adc + s0;
    const float off = B.cal_off[r], sc = B.cal_scale[r];
    const int cc = have ? c : 0;
    const int my0 = cc * DN_SEG_CHUNK - SEG_PRE;                // sample index of stream position 0
    // carry {sum, sumsq} at my0 (k1_carry): index my0 / 256 of the read's carries; chunk 0 starts from the zeros before the read
    double S = 0.0, Q = 0.0;
    if (have && my0 > 0) { const double2 cy = B.carry[4ull * c0 + (unsigned)r + (unsigned)(my0 / DN_SEG_CARRY)]; S = cy.x; Q = cy.y; }
    double rs[13], rq[13];
#pragma unroll
    for (int k = 0; k < 13; k++) { rs[k] = S; rq[k] = Q; }       // never read by a live statistic: the detector starts >= 64 samples in
    PeakSink sink{ B.chunk_peaks + (c0 + cc) * DN_SEG_PEAKCAP, B.chunk_psum + (c0 + cc) * DN_SEG_PEAKCAP, lds_pos + lane, lds_sum + lane };
    const int chunk_lo = cc * DN_SEG_CHUNK;
    const int det_hi = have ? min(n, chunk_lo + DN_SEG_CHUNK) : 0;
    const int det_lo = REDO ? chunk_lo : max(1, chunk_lo - B.seg_warm);      // sample 0 is always masked (:140); seg_warm = DN_SEG_WARM unless a test shortened it
    SegState *save = (!REDO && have) ? &B.chunk_in[c0 + cc] : nullptr;
    for (int t = 0; t < SEG_TILES; t++) {
        __syncthreads();
        // tile: row = the lane that will read it; 52 samples per row = 13 groups of 4; thread handles (row, group) pairs round-robin
        for (int f = lane; f < 64 * (SEG_TILE / 4); f += 64) {
            const int row = f / (SEG_TILE / 4), g = f % (SEG_TILE / 4);
            const int rc = REDO ? c : cbase + row;                // the chunk this row's lane walks
            const int idx = rc * DN_SEG_CHUNK - SEG_PRE + t * SEG_TILE + 4 * g;
            float *d = tile + row * SEG_PITCH + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = idx + e;
                d[e] = (i >= 0 && i < n) ? ((float)adc[i] + off) * sc : 0.0f;    // pod5.cpp:60; outside the read: an exact zero
            }
        }
        __syncthreads();
        const float *rowp = tile + lane * SEG_PITCH;
#pragma unroll 1
        for (int h = 0; h < SEG_TILE / SEG_BODY; h++)
            seg_body13(rowp + SEG_BODY * h, my0 + t * SEG_TILE + SEG_BODY * h, n, det_lo, det_hi, chunk_lo, save, S, Q, rs, rq, st, ss, ls, sink, npk);
    }
    if (have) sink.flush(npk);
}

__global__ __launch_bounds__(64) void k1_detect(BatchDev B) {
    __shared__ float tile[64 * SEG_PITCH];
    __shared__ unsigned lds_pos[SINK_N * 64];
    __shared__ double lds_sum[SINK_N * 64];
    const int r = blockIdx.y;
    const int lane = threadIdx.x;
    const uint64_t c0 = B.chunk_off[r];
    const int nch = (int)(B.chunk_off[r + 1] - c0);
    const int cbase = blockIdx.x * 64;
    if (cbase >= nch) return;
    const int c = cbase + lane;
    const bool have = c < nch;
    SegState st = seg_initial();
    unsigned npk = 0;
    double ss = 0.0, ls = 0.0;          // sum[peak_pos] of the short / long detector's pending peak: set whenever a detector moves its peak
    seg_walk<false>(B, tile, lds_pos, lds_sum, r, cbase, c, have, st, ss, ls, npk);
    if (have) {
        B.chunk_npk[c0 + c] = npk;
        B.chunk_out[c0 + c] = st;
        // a peak still pending at the chunk's end was placed by THIS walk (it started from "no peak"), so its sum is the true one
        // whenever the state is -- k1_events hands both to the next chunk's redo if that chunk's speculation missed
        B.chunk_sums[c0 + c] = make_double2(ss, ls);
    }
}

// ------------------------------------------------------------------------------------------------
// k1_events: one wavefront per read
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

#define EV_TILE 512
__global__ __launch_bounds__(64) void k1_events(BatchDev B) {
    __shared__ float tile[64 * SEG_PITCH];                        // the exact redo's signal tile ...
    __shared__ unsigned lds_pos[SINK_N * 64];                          // ... and peak staging
    __shared__ double lds_sum[SINK_N * 64];
    __shared__ unsigned pre[4096];                                // peaks per chunk, cached for the first 4 096 chunks (4 M samples) of a read; beyond: read back from HBM
    __shared__ unsigned t_pos[EV_TILE + DN_SEG_PEAKCAP + 1];      // the streaming tile: start and prefix sum of consecutive scrappie events
    __shared__ double t_sum[EV_TILE + DN_SEG_PEAKCAP + 1];
    __shared__ float k_mean[65];                                  // kept events of one 64-wide step, slot 0 = the previous kept one
    __shared__ unsigned k_start[65];
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t s0 = B.samp_off[r];
    const int n = (int)(B.samp_off[r + 1] - s0);
    const uint64_t c0 = B.chunk_off[r];
    const int nch = (int)(B.chunk_off[r + 1] - c0);
    const uint64_t e0 = B.ev_off[r];
    const unsigned ecap = (unsigned)(B.ev_off[r + 1] - e0);
    ReadRes &R = B.res[r];

    // ---- 1. exact verification of the speculative hand-offs ----
    int any_bad = 0;
    for (int c = 1 + lane; c < nch; c += 64) {
        const SegState a = B.chunk_out[c0 + c - 1], b = B.chunk_in[c0 + c];
        if (!seg_equal(a, b, c * DN_SEG_CHUNK)) any_bad = 1;
    }
    unsigned rechecks = 0;
    if (__any(any_bad)) {
        // slow path (rare): walk the chain of chunks, redo every chunk whose assumed start state was wrong -- the whole wavefront
        // walks that one chunk from the true state (wave-uniform; every lane computes and stores the same values)
        SegState tru = B.chunk_out[c0];
        double2 trs = B.chunk_sums[c0];                           // {sum[short.peak_pos], sum[long.peak_pos]} that go with `tru`
        for (int c = 1; c < nch; c++) {
            const SegState in = B.chunk_in[c0 + c];
            if (seg_equal(tru, in, c * DN_SEG_CHUNK)) { tru = B.chunk_out[c0 + c]; trs = B.chunk_sums[c0 + c]; continue; }
            SegState st = tru;
            unsigned npk = 0;
            // a peak pending in `tru` lies BEFORE this chunk: the walk cannot recompute the prefix sum at its position, so it travels
            // with the state (without it the event on either side of such a peak would get a wrong mean)
            double ss = trs.x, ls = trs.y;
            seg_walk<true>(B, tile, lds_pos, lds_sum, r, 0, c, true, st, ss, ls, npk);
            if (lane == 0) { B.chunk_npk[c0 + c] = npk; B.chunk_out[c0 + c] = st; B.chunk_sums[c0 + c] = make_double2(ss, ls); }
            tru = st; trs = make_double2(ss, ls);
            rechecks++;
        }
        __threadfence_block();
    }
    __syncthreads();

    // ---- 2. one streaming pass over the peaks, chunk after chunk, through an LDS tile: scrappie events -> DNAscent events ----
    // scrappie event e (create_events :242-247): start[0] = 0, start[1 + j] = j-th peak (every recorded peak has 0 < p < n);
    // mean[e] = (float)(sum[start[e + 1]] - sum[start[e]]) / length (create_event :224-226), the last one ends at n.
    // DNAscent event build (event_handling.cpp:549-575): kept scrappie indices are e > 0 && mean[e] > 0; event j carries mean and
    // rawStart of the PREVIOUS kept index (0.0 / 0 for the first) and the raw span up to start[kept_j] - 1.  Nothing but the
    // recorded peaks is read and nothing but the events is written (the scrappie table itself only as a parity tap).
    unsigned running = 0;
    // (round 6: a read of more than 4 096 chunks -- 4.2 M samples, ~350 kb -- used to be an OVERFLOW of the whole batch; the reference has no such limit.  The counts
    // of the chunks beyond the cache are read back one by one in the streaming loop below: ultra-long reads pay a load per chunk there, nothing else changes)
    int overflow = 0;
    const int nchc = nch;
    for (int cb = 0; cb < nchc; cb += 64) {
        const int c = cb + lane;
        unsigned cnt = (c < nchc) ? B.chunk_npk[c0 + c] : 0u;
        if (cnt > DN_SEG_PEAKCAP) { overflow = 1; cnt = DN_SEG_PEAKCAP; }
        if (c < min(nchc, 4096)) pre[c] = cnt;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        running += cnt;
    }
    unsigned n_et = 1 + running;
    if (n_et > ecap) { overflow = 1; n_et = ecap; }
    overflow = __any(overflow);
    const double total = B.carry[4ull * c0 + (unsigned)r + (unsigned)((n + DN_SEG_CARRY - 1) / DN_SEG_CARRY)].x;     // sum[n]
    unsigned *tap_start = B.et_start ? B.et_start + e0 : nullptr;
    float *tap_mean = B.et_mean ? B.et_mean + e0 : nullptr;
    double *ev_mean = B.ev_mean + e0;
    unsigned *ev_start = B.ev_start + e0, *ev_len = B.ev_len + e0;
    if (lane == 0) { t_pos[0] = 0u; t_sum[0] = 0.0; }
    unsigned fill = 1;                      // entries in the tile: scrappie events E0 .. E0 + fill - 1
    unsigned E0 = 0, taken = 1;             // taken: scrappie events placed in the tile so far (of n_et)
    unsigned kept_total = 0;
    float prev_mean = 0.0f; unsigned prev_start = 0u;             // the previous kept scrappie event (wave-uniform)
    LDS_FENCE();
    __syncthreads();
    for (int c = 0; c < max(nchc, 1); c++) {
        if (c < nchc) {
            const unsigned cnt = min(c < 4096 ? pre[c] : min(B.chunk_npk[c0 + c], (unsigned)DN_SEG_PEAKCAP), n_et - taken);
            const unsigned *pk = B.chunk_peaks + (c0 + c) * DN_SEG_PEAKCAP;
            const double *pks = B.chunk_psum + (c0 + c) * DN_SEG_PEAKCAP;
            for (unsigned j = lane; j < cnt; j += 64) { t_pos[fill + j] = pk[j]; t_sum[fill + j] = pks[j]; }
            fill += cnt; taken += cnt;
            if (fill < EV_TILE && c + 1 < nchc) continue;
        }
        const bool last = c + 1 >= nchc;
        if (!last && fill < 2) continue;
        LDS_FENCE();
        __syncthreads();
        const unsigned proc = last ? fill : fill - 1;             // the newest entry waits for its successor
        for (unsigned ib = 0; ib < proc; ib += 64) {
            const unsigned i = ib + lane;
            const bool on = i < proc;
            unsigned st = 0; float mean = 0.0f;
            if (on) {
                st = t_pos[i];
                const bool more = i + 1 < fill;
                const unsigned en = more ? t_pos[i + 1] : (unsigned)n;
                const double a = t_sum[i], b = more ? t_sum[i + 1] : total;
                const float length = (float)(unsigned long long)((unsigned long long)en - (unsigned long long)st);
                mean = (float)(b - a) / length;
                if (tap_start) { tap_start[E0 + i] = st; tap_mean[E0 + i] = mean; }
            }
            const bool keep = on && (E0 + i > 0u) && ((double)mean > 0.);
            const unsigned long long m = __ballot(keep);
            if (m == 0ull) continue;
            const unsigned q = (unsigned)__popcll(m & ((1ull << lane) - 1ull)), k = (unsigned)__popcll(m);
            if (lane == 0) { k_mean[0] = prev_mean; k_start[0] = prev_start; }
            if (keep) { k_mean[q + 1] = mean; k_start[q + 1] = st; }
            LDS_FENCE();
            if (keep) {
                const unsigned j = kept_total + q;
                const double pm = (double)k_mean[q];
                const unsigned rs = k_start[q];
                unsigned lastS = st - 1u;                          // :563 (start >= 1 for e > 0)
                if (lastS > (unsigned)n - 1u) lastS = (unsigned)n - 1u;
                // (staging these in LDS to leave as 64-entry aligned blocks was measured: WRITE_SIZE identical -- the runs of one
                // step are contiguous and the L2 merges them -- and the kernel 12 % slower)
                ev_mean[j] = j > 0 ? pm : 0.;
                ev_start[j] = rs;
                ev_len[j] = (lastS >= rs) ? (lastS - rs + 1u) : 0u;
            }
            prev_mean = k_mean[k]; prev_start = k_start[k];
            kept_total += k;
            LDS_FENCE();
        }
        E0 += proc;
        if (last) break;
        {
            const unsigned kp = t_pos[fill - 1]; const double ks = t_sum[fill - 1];
            LDS_FENCE();
            __syncthreads();
            if (lane == 0) { t_pos[0] = kp; t_sum[0] = ks; }
            fill = 1;
            LDS_FENCE();
            __syncthreads();
        }
    }
    if (lane == 0) {
        R.n_samples = (unsigned)n;
        R.n_scrappie = n_et;
        R.n_events = kept_total;
        R.rechecks = rechecks;
        R.seg_overflow = overflow;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers (called from dn_capi.hip)
// ------------------------------------------------------------------------------------------------
void k1_launch_scan(const BatchDev &B, hipStream_t st) {
    // taps requested (B.psum != null): the same kernel also writes every prefix sum.  (A one-read-per-lane carry kernel -- no
    // EXEC narrowing, 64 chains per wavefront -- was tried: with 16 wavefronts for 1 000 reads its uncoalesced 2-byte tile loads
    // were fully exposed, 51 ms against 4.4.)
    if (B.psum) hipLaunchKernelGGL(k1_scan4<true>, dim3((B.n_reads + 3) / 4), dim3(64), 0, st, B);
    else hipLaunchKernelGGL(k1_scan4<false>, dim3((B.n_reads + 3) / 4), dim3(64), 0, st, B);
}
void k1_launch_tstat(const BatchDev &B, unsigned max_samples, hipStream_t st) {      // taps only
    if (B.psum && B.t1) hipLaunchKernelGGL(k1_tstat, dim3((max_samples + 255) / 256, B.n_reads), dim3(256), 0, st, B);
}
void k1_launch_detect(const BatchDev &B, unsigned max_chunks, hipStream_t st) {
    hipLaunchKernelGGL(k1_detect, dim3((max_chunks + 63) / 64, B.n_reads), dim3(64), 0, st, B);
}
void k1_launch_events(const BatchDev &B, hipStream_t st) {
    hipLaunchKernelGGL(k1_events, dim3(B.n_reads), dim3(64), 0, st, B);
}
