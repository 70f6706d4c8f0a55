// k2b_viterbi.hip -- K2b: eventalign (alignment.cpp:547-744) = window walk + builtinViterbi (alignment.cpp:193-516)
// + feature fill (reads.h:292-304) + tensor packing (reads.h:112-172), on gfx950.
//
// The windows of a read form a serial chain (the next one starts at the last match of the previous one,
// alignment.cpp:739-740), so ONE WAVEFRONT PER READ walks them.  Inside a window the Viterbi lattice
// (T observations x N <= 65 positions x {I, M, D}) is swept by ANTI-DIAGONALS: cell (t, i) needs (t-1, i-1), (t-1, i)
// and -- through the silent deletion chain D(t,i) <- D(t,i-1), alignment.cpp:405-427 -- (t, i-1), so all cells with
// t + i = const are independent.  Lane i owns position i and advances one observation per step; what it needs from
// lane i-1 arrives through wave-shift DPP.  This removes the reference's per-column serial pass over positions
// exactly: every value is produced by the same fp64 operations in the same order, only scheduled differently.
// NaN is log(0) as in probability.cpp; arg-max is "first wins, NaN never wins" (alignment.cpp:166-190).
// Backtrace codes live in LDS (1 byte per cell); the traceback and the feature fill are wave-uniform.
//
// Emission: eln(normalPDF(mu, 0.14, x)) with normalPDF as the reference BUILD computes it (GCC folds pow(v, 2.0) into
// v*v): (1/sqrt(2 s^2 pi)) * exp(-(x-mu)^2 / (2 s^2)), then log.  exp/log are the ROCm device-library fp64 functions
// (<= 1 ulp) where the reference calls glibc: scores can differ in the last bits (tolerance stated in the tests), the
// arg-max decisions are compared exactly.
#include "dn_dev.h"
#include <atomic>

#define VT_TMAX 512          // observations per window the lattice is sized for at most (a window spans <= 65 positions, ~2.2 events each)
#define VT_TFAST 224         // ... and in the first pass: a window of the synthetic workloads holds 110-160.  The backtrace block is
                             // (T + 1) x 66 bytes of LDS per wavefront: 34 KB at 512 kept the occupancy at 3 wavefronts per CU and
                             // the CNN's large workgroups off every CU an eventalign wavefront sat on; at 224 it is 15 KB (22 KB in
                             // all).  A read with a longer window is redone by a second launch of the 512 variant.
#define VT_THUGE 8192        // round 6: windows of up to this many observations (a stalled pore: hundreds to thousands of events rough-aligned to one k-mer) are walked with
                             // the lattice in GLOBAL memory (k2b_eventalign<VT_THUGE>, below); the reference has no limit, beyond this one a read fails
#define VT_HUGE_WGS 8        // ... by that many wavefronts per launch, each with its own 790 KB of scratch,
#define VT_HUGE_ROUNDS 4     // ... in that many launches: up to 32 such reads per batch (more: DN_READ_FAIL_WINDOW_EVENTS as before)
#define VT_NS 66             // backtrace row stride (positions per window <= 65)

struct VitConsts {           // alignment.cpp:199-204 (host libm), normalPDF constants, deletion chain before the first event
    double D2D, D2M, I2M, M2D, M2I, I2I;
    double c, d2, rd2, logc; // 1/sqrt(2 s^2 pi), 2 s^2, RN(1/(2 s^2)), log(c)
    double initD[VT_NS];     // D_prev[i] of alignment.cpp:241-251: M2D, then + D2D sequentially
};
struct VitRead { double iM2M, eM2M, eM2MorD, eOrI; int fail, pad; };   // alignment.cpp:207-210, per read (host libm); fail: NegativeLog

struct EaDev {               // outputs, all at ref_off[r] (capacity = reference length of the read)
    unsigned *coord, *qidx, *ridx; int *indel; unsigned *nsig; float *sig /* x20 */, *core, *resid;
    unsigned *win_ref, *win_len, *win_T; double *win_score;
    // `DNAscent align` table (alignment.cpp:697-733), optional (al_val == nullptr: not requested): one row per raw sample of
    // every event labelled M and of every event labelled I before the window's last match; rows of read r start at al_off[r]
    unsigned *al_coord, *al_rpos; double *al_val; unsigned char *al_kind; const unsigned long long *al_off; unsigned *al_n;
    unsigned char *redo;     // [n_reads] where a read stands between the passes of k2b_launch: 0 finished, 1 stopped at a window the 224 lattice cannot
                             // hold, 2 that window done by the 512 lattice, 3 stopped a second time
    unsigned *resume;        // [n_reads][8] the walk's state at the window it stopped at: ri, readHead, npos, nwin, al_rows
};

// log(0) is NaN in the reference (probability.cpp) and every use of it is one of: NaN + x = NaN, and lnGreaterThan
// (probability.cpp:107-131: NaN is never greater, anything is greater than NaN, first wins ties).  Writing log(0) as
// -infinity instead gives the identical lattice: -inf + x = -inf for the finite x that occur, and "a > b" on the
// extended reals IS lnGreaterThan.  So the kernel carries -inf, one compare per arg-max step, and converts back to NaN
// where a value leaves the kernel (the window score).
__device__ __forceinline__ double qnan() { return __longlong_as_double(0xfff0000000000000ll); }     // -inf, see above
__device__ __forceinline__ double real_nan() { return __longlong_as_double(0x7ff8000000000000ll); }
__device__ __forceinline__ bool ln_gt(double a, double b) { return a > b; }

__device__ __forceinline__ double shl_prev_d(double v, double fill, int lane) {    // lane l <- lane l-1, lane 0 <- fill
    long long b = __double_as_longlong(v), f = __double_as_longlong(fill);
    int lo = __builtin_amdgcn_update_dpp((int)(f & 0xffffffffll), (int)(b & 0xffffffffll), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// lane l <- lane l-1, lane 0 <- 0.0.  The lattice never USES what lane 0 receives: its copies of the transition constants that are
// added to a left neighbour's value are log(0) (VitHot of the lane, below), and 0.0 + log(0) is log(0) -- the value the reference has
// there.  With an explicit fill the compiler re-materialised the fill register before every DPP move: 6 of a step's ~65 instructions.
__device__ __forceinline__ double shl_prev_z(double v) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x138, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// The constants of the inner loop, held in VECTOR registers (every lane the same value).  As kernel-argument scalars they lost the
// fight for the ~100 SGPRs against the ~60 pointers of BatchDev / EaDev that stay live across the lattice loop: the compiler kept them
// spilled in VGPR lanes and fetched each one back with v_readlane before every use -- 65 v_readlane per lattice step beside its ~60
// fp64 instructions.  fp64 vector instructions take VGPR operands just as well.
struct VitHot { double I2I, M2I, I2M, M2D, D2D, D2M, eM2M, iM2M, rd2, d2, logc, c; };
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }

// one lattice cell (alignment.cpp:278-285/:351-356 insertion, :305-310/:372-381 match, :326-328/:408-413 deletion).
// Position 0 has no left neighbour and may come from START instead; it is expressed through the INPUTS so the cell has
// no per-lane special case: for lane 0 the shifted-in left values are log(0), `s0` carries start_prev (0 at t == 0, log(0)
// after; log(0) for every other lane) and tr3 is eOrI instead of D2M.
// ("+ insProb" with insProb == 0.0 is dropped: it can only change the sign of an exact zero.)
// What a cell records for the traceback is the DECODED move of each of its three states, so that the serial walk is a shift and a
// mask per step (round 3; it used to record the arg-max index and the walk re-derived the move, position 0's special cases included):
//   bits 0-2  from M: next state (0 D, 1 M, 2 I, 3 = START: the walk ends) | 4 if the move steps one position left
//   bits 3-4  from I: next state (1, 2 or 3)                 (an insertion stays at its position)
//   bits 5-6  from D: next state (0, 1 or 3)                 (a deletion always steps left, along the column)
// The values are the arg-max chain's own selects with other constants (VitCodes: per lane, position 0 differs), at no extra cost.
struct VitCodes { unsigned m0, m1, m2, m3, d0, d1; };
__device__ __forceinline__ VitCodes vit_codes(bool pos0) {
    VitCodes k;
    k.m0 = pos0 ? 1u : (2u | 4u);        // arg-max 0: from I at the left position            position 0: {M_0, START}: anything but 3 is M_0 one
    k.m1 = pos0 ? 1u : (1u | 4u);        //         1: from M at the left position                        column earlier, same position
    k.m2 = 1u;                           //         2: from M at the same position
    k.m3 = pos0 ? 3u : (0u | 4u);        //         3: from D at the left position            position 0: START
    k.d0 = pos0 ? (3u << 5) : (1u << 5); // deletion from M at the left position              position 0: D_0 always comes from START
    k.d1 = pos0 ? (3u << 5) : (0u << 5); //          from D at the left position
    return k;
}
__device__ __forceinline__ void vit_cell(const double s0, const double l3, const double tr3, const double Ip, const double Mp,
                                         const double lI2, const double lM2, const double lM1, const double lD1, const double e,
                                         const VitHot &vc, const VitHot &vr, const VitCodes &k, double &In, double &Mn, double &Dn, unsigned &code) {
    double bi = Ip + vc.I2I; unsigned ci = 2u << 3;                                        // from I: stays I
    { const double v = Mp + vc.M2I; if (ln_gt(v, bi)) { bi = v; ci = 1u << 3; } }         // from M
    { const double v = s0 + vc.M2I; if (ln_gt(v, bi)) { bi = v; ci = 3u << 3; } }         // from START
    double bm = lI2 + vc.I2M + e; unsigned cm = k.m0;
    { const double v = lM2 + vr.eM2M + e; if (ln_gt(v, bm)) { bm = v; cm = k.m1; } }
    { const double v = Mp + vr.iM2M + e;  if (ln_gt(v, bm)) { bm = v; cm = k.m2; } }
    { const double v = l3 + tr3 + e;      if (ln_gt(v, bm)) { bm = v; cm = k.m3; } }
    double bd = lM1 + vc.M2D; unsigned cd = k.d0;
    { const double v = lD1 + vc.D2D; if (ln_gt(v, bd)) { bd = v; cd = k.d1; } }
    In = bi; Mn = bm; Dn = bd;
    code = ci | cm | cd;
}

template <class C> __device__ __forceinline__ double emission(double x, double mu, const C &vc) {
    const double d = x - mu;
    const double sq = d * d;                              // pow(d, 2.0) as compiled in the reference
    const double n = -sq;
    const double q = n * vc.rd2;                          // exact n / d2 (FMA-corrected reciprocal, see k2_banded.hip)
    const double rem = fma(-q, vc.d2, n);
    const double arg = fma(rem, vc.rd2, q);
    // eln(c * exp(arg)) (probability.cpp:147, :35-47).  Where exp() is a normal number this equals log(c) + arg to
    // within the few ulps that separate any two correct implementations of exp/log (the reference's is glibc); where exp
    // goes subnormal (arg < -708.4) the reference's result loses precision and at arg < -745.13 it is log(0): there the
    // literal exp -> log chain is evaluated (rare: |x - mu| > 37 sigma).
    double e = vc.logc + arg;
    if (__any(arg < -708.0)) {
        const double p = vc.c * exp(arg);
        const double slow = (p == 0.0) ? qnan() : log(p);
        e = (arg < -708.0) ? slow : e;
    }
    return e;
}

// The windows of a read are walked by up to four launches (k2b_launch).  mode 0: every read from its first window in the 224-observation
// lattice; a window that does not fit stops the walk THERE (state to O.resume, redo = 1).  mode 1 (512 lattice): the reads with redo == 1
// do that ONE window and hand back (redo = 2).  mode 2 (224): they continue; a second oversized window stops them again (redo = 3).
// mode 3 (512): those go on to their end.  Round 2 redid such a read FROM ITS FIRST WINDOW in the 512 variant: a handful of wavefronts
// held the batch for up to 105 ms (rocprof max of k2b_eventalign<512>), longer than the whole first pass.
#ifdef DN_K2B_TRACE       /* experiment build only (tools/k2b_trace.py): shader-clock ticks per phase of the window walk, summed over all reads */
__device__ unsigned long long k2b_trace[12];
extern "C" int dn_debug_k2b_trace(unsigned long long *out, int reset) {
    if (reset) { unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(k2b_trace), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k2b_trace), sizeof(k2b_trace));
}
#define K2B_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define K2B_T(k) do { } while (0)
#endif
// K2B_W reads (= wavefronts; they share nothing) per WORKGROUP -- see K2_FILL_W in k2_banded.hip: packing the batch's long-lived wavefronts onto n / W
// CUs leaves the others to the network's kernels whole.  Every wavefront owns its slice of the shared arrays; the kernel's barriers only ever ordered
// ONE wavefront's LDS traffic, so they are wave-level fences here (the wavefronts of a workgroup walk different reads and leave at different times).
#ifndef K2B_W
#define K2B_W 4
#endif
#if K2B_W > 1
#define K2B_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define K2B_SYNC() __syncthreads()
#endif
template <int TMAX> struct K2bLds {
    double xs[TMAX];                                      // scaled observations of the window
    unsigned tk_start[TMAX], tk_len[TMAX];                // raw span of each taken event (event.raw, reads.h:68-72)
    unsigned ev_slot[TMAX], ev_cnt0[TMAX];                // label pass: position slot of an M-labelled event / samples before it
    unsigned ps_p[VT_NS], ps_cnt[VT_NS];                  // positions created by this window: lattice position, sample count
    unsigned ev_aoff[TMAX];                               // align table: first row of each printed event (0xffffffff: not printed)
    unsigned short evlab[TMAX];                           // label of the state that emitted observation t: state << 8 | position
    unsigned char bt[(TMAX + 1) * VT_NS];                 // backtrace codes: I 2 bits | M 3 bits | D 2 bits
};
// K2B_WAVES_EU (round-4 verdict item 4): a register budget for this kernel.  With its LDS declared statically (88 KB per four-read workgroup) the compiler
// knows one workgroup fits a CU, i.e. one wavefront per SIMD, and takes all the registers it likes (175; amdgpu_waves_per_eu / amdgpu_num_vgpr are ignored
// against that bound); a budget only binds when the LDS is DYNAMIC (the compiler then cannot bound the occupancy by it) and __launch_bounds__ names the
// wavefronts per SIMD.  Measured, bit-identical digests, one session (gpurun_out/r5g/ab.txt; 12 steps of the default bench, twice each):
//     0  static LDS, no budget     175 registers, 133 spilled scalars    kernel alone 70.4 ms   pipeline 789.6 / 788.8 Msamples/s
//     3  dynamic LDS, 168 budget   168 registers, 135 spilled scalars                 65.4                796.2 / 805.5   (+1.5 %)   <- default
//     4  dynamic LDS, 128 budget   128 registers + 33 spilled to scratch (136 B per lane)  87.7           778.5 / 781.9
// The verdict's 128-register form exists only with vector spills in the lattice loop: the kernel's real need is ~160 registers (12 constants + 15 lattice
// states + the traceback's 24 block registers as fp64 / dword pairs, beside ~60 live pointers of BatchDev / EaDev that the scalar file cannot hold).
#ifndef K2B_WAVES_EU
#define K2B_WAVES_EU 3
#endif
#if K2B_WAVES_EU > 0
#define K2B_BOUNDS(threads) __launch_bounds__(threads, K2B_WAVES_EU)
#else
#define K2B_BOUNDS(threads) __launch_bounds__(threads)
#endif
#define K2B_HUGE_HEAD 4096                                 /* bytes: the list of parked reads (count, then up to 1 023 read indices) in front of the lattices */
#define K2B_HUGE_STRIDE ((sizeof(K2bLds<VT_THUGE>) + 255) & ~(size_t)255)
static_assert(sizeof(K2bLds<VT_TFAST>) * K2B_W <= 160 * 1024 && sizeof(K2bLds<VT_TMAX>) <= 160 * 1024, "K2B_W wavefronts' lattices must fit gfx950's 160 KB of LDS per CU");
template <int TMAX>
__global__ K2B_BOUNDS(64 * (TMAX <= VT_TFAST ? K2B_W : 1)) void k2b_eventalign(BatchDev B, EaDev O, const VitRead *vrs, VitConsts vc, int mode, unsigned char *huge) {
    constexpr int WPB = TMAX <= VT_TFAST ? K2B_W : 1;     // the 512-observation lattice holds 50 KB: one per workgroup as before (it runs for a handful of windows)
    constexpr bool HUGE = TMAX > VT_TMAX;                 // round 6: the lattice in global memory (`huge`: a list of the parked reads, then VT_HUGE_WGS lattices)
#if K2B_WAVES_EU > 0
    extern __shared__ __attribute__((aligned(16))) unsigned char k2b_dyn_lds_[];
    K2bLds<TMAX> *lds_ = reinterpret_cast<K2bLds<TMAX> *>(HUGE ? huge + K2B_HUGE_HEAD + (size_t)blockIdx.x * K2B_HUGE_STRIDE : k2b_dyn_lds_);
#else
    __shared__ __attribute__((aligned(16))) K2bLds<HUGE ? 1 : TMAX> lds_s_[WPB];
    K2bLds<TMAX> *lds_ = reinterpret_cast<K2bLds<TMAX> *>(HUGE ? huge + K2B_HUGE_HEAD + (size_t)blockIdx.x * K2B_HUGE_STRIDE : (unsigned char *)lds_s_);
#endif
    K2bLds<TMAX> &L_ = lds_[WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0];
    double *xs = L_.xs; unsigned *tk_start = L_.tk_start, *tk_len = L_.tk_len, *ev_slot = L_.ev_slot, *ev_cnt0 = L_.ev_cnt0, *ps_p = L_.ps_p, *ps_cnt = L_.ps_cnt;
    unsigned short *evlab = L_.evlab; unsigned char *bt = L_.bt; unsigned *ev_aoff = L_.ev_aoff;
    int r_;
    if (HUGE) {                                           // mode = 5 + round: this wavefront's entry of the list k2b_huge_list made (count in word 0)
        const unsigned *list = reinterpret_cast<const unsigned *>(huge);
        const unsigned idx = (unsigned)(mode - 5) * VT_HUGE_WGS + blockIdx.x;
        if (idx >= list[0]) return;
        r_ = __builtin_amdgcn_readfirstlane((int)list[1 + idx]);
    } else r_ = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WPB + (threadIdx.x >> 6))) : (int)blockIdx.x;     // wave-uniform, and the compiler must know it
    const int r = r_;
    if (r >= B.n_reads) return;
    const int lane = threadIdx.x & 63;
    if (!HUGE && mode != 0 && O.redo[r] != mode) return;  // later passes: only the reads the pass before handed over
    if (HUGE && O.redo[r] != 5) return;
    ReadRes &R = B.res[r];
    const VitRead vr = vrs[r];
    if (R.status == 0 && vr.fail) {                       // eln() of a negative number: the reference throws NegativeLog (probability.cpp:45)
        if (lane == 0) { R.status = 4; R.n_positions = 0; R.n_windows = 0; }
        return;
    }
    if (R.status != 0) { if (lane == 0) { R.n_positions = 0; R.n_windows = 0; } return; }
    const double NaN = qnan();
    const uint64_t f0 = B.ref_off[r];
    const int n_ref = (int)(B.ref_off[r + 1] - f0);
    const char *ref = B.refseq + f0;
    const unsigned *r2q = B.ref2query + f0;
    const unsigned *rank_r = B.rank_r + f0;
    const uint64_t a0 = B.aln_off[r] + R.aln_begin;
    const unsigned *ae = B.aln_event + a0, *ak = B.aln_kmer + a0;
    const unsigned n_aln = R.n_aligned;
    const uint64_t e0 = B.ev_off[r];
    const double *ev_mean = B.ev_mean + e0;
    const unsigned *ev_start = B.ev_start + e0, *ev_len = B.ev_len + e0;
    const int16_t *adc = B.adc + B.samp_off[r];
    const float cal_off = B.cal_off[r], cal_sc = B.cal_scale[r];
    const double shift = R.shift, scale = R.scale;
    const int is_rev = B.is_rev[r], ref_start = B.ref_start[r], ref_end = B.ref_end[r];
    const double *model = B.model_mean;

    const VitHot hot = { in_vgpr(vc.I2I), in_vgpr(vc.M2I), in_vgpr(vc.I2M), in_vgpr(vc.M2D), in_vgpr(vc.D2D), in_vgpr(vc.D2M), in_vgpr(vr.eM2M), in_vgpr(vr.iM2M),
                         in_vgpr(vc.rd2), in_vgpr(vc.d2), in_vgpr(vc.logc), in_vgpr(vc.c) };
    unsigned readHead = 0; int ri = 0;
    unsigned npos = 0, nwin = 0;
    unsigned al_rows = 0;                                 // rows of the align table written so far
    int fail = 0;
    unsigned *const saved = O.resume + 8 * (size_t)r;
    if (mode != 0) { ri = (int)saved[0]; readHead = saved[1]; npos = saved[2]; nwin = saved[3]; al_rows = saved[4]; }   // wave-uniform loads (the huge passes: mode >= 5)
    auto park = [&](unsigned char next) {                 // stop here; the pass `next` picks the walk up at this window
        if (lane == 0) { saved[0] = (unsigned)ri; saved[1] = readHead; saved[2] = npos; saved[3] = nwin; saved[4] = al_rows; O.redo[r] = next; }
    };

#ifdef DN_K2B_TRACE
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    while (ri < n_ref - (DN_K - 1)) {                     // alignment.cpp:556
        K2B_T(6);
        // the walk's state is wave-uniform by construction; say so, or one value the compiler cannot prove uniform turns the whole window loop
        // (its branches, the traceback) into divergent code with execution masks
        ri = __builtin_amdgcn_readfirstlane(ri); readHead = (unsigned)__builtin_amdgcn_readfirstlane((int)readHead);
        npos = (unsigned)__builtin_amdgcn_readfirstlane((int)npos); nwin = (unsigned)__builtin_amdgcn_readfirstlane((int)nwin);
        al_rows = (unsigned)__builtin_amdgcn_readfirstlane((int)al_rows);
        const int toEnd = n_ref - ri;
        int W = toEnd < 50 ? toEnd : 50;
        if ((double)toEnd > 75.0) {                       // :564 break-point search in a 75-base look-ahead
            int bad = 0;
            for (int j = lane; j < 75; j += 64) { const char c = ref[ri + j]; bad |= !(c == 'A' || c == 'T' || c == 'G' || c == 'C'); }
            if (__any(bad)) { ri += W; continue; }        // :569-572
            bool hit = false;
            if (lane < 15) {                              // i = 50 .. 64 (:574)
                const int i = 50 + lane;
                const double m = model[rank_r[ri + i]], mb = model[rank_r[ri + i - 1]], mf = model[rank_r[ri + i + 1]];
                hit = (fabs(m - mf) > 0.75) && (fabs(m - mb) > 0.75);
            }
            const unsigned long long hm = __ballot(hit);
            if (hm) W = 50 + (__ffsll((long long)hm) - 1) + DN_K;   // :591
        }
        {
            int bad = 0;
            for (int j = lane; j < W; j += 64) { const char c = ref[ri + j]; bad |= !(c == 'A' || c == 'T' || c == 'G' || c == 'C'); }
            if (__any(bad)) { ri += W; continue; }        // :599-604
        }
        const int N = W - (DN_K - 1);
        K2B_T(0);
        const unsigned qlo = r2q[ri], qhi = r2q[ri + W - DN_K + 1];
        // ---- events rough-aligned to the window (:611-632), order-preserving compaction ----
        unsigned nt = 0; bool first = true;
        for (unsigned j = readHead; j < n_aln; j += 64) {
            const unsigned jj = j + lane;
            const bool in = jj < n_aln;
            const unsigned q = in ? ak[jj] : 0u;
            const unsigned long long stopm = __ballot(in && q >= qhi);
            const int limit = stopm ? (__ffsll((long long)stopm) - 1) : 64;
            const bool inw = in && lane < limit && qlo <= q && q < qhi;
            const unsigned long long wm = __ballot(inw);
            if (first && wm) { readHead = j + (unsigned)(__ffsll((long long)wm) - 1); first = false; }
            // two dependent rounds of loads, not four: the event index travels with the k-mer index, the event's span with its mean
            const unsigned e_idx = in ? ae[jj] : 0u;
            double mean = 0.; unsigned e_st = 0, e_ln = 0;
            if (inw) { mean = ev_mean[e_idx]; e_st = ev_start[e_idx]; e_ln = ev_len[e_idx]; }
            const bool take = inw && (0. < mean) && (mean < 250.);      // :624
            const unsigned long long tm = __ballot(take);
            const unsigned p = nt + (unsigned)__popcll(tm & ((1ull << lane) - 1ull));
            if (take && p < (unsigned)TMAX) { tk_start[p] = e_st; tk_len[p] = e_ln; xs[p] = (mean - shift) / scale; }
            nt += (unsigned)__popcll(tm);
            if (stopm) break;
        }
        const int indel = (int)(qhi - qlo) - (int)(W - DN_K + 1);       // :635-638
        if (nt < 2) { ri += W; continue; }                // :641
        if (nt > (unsigned)TMAX) {
            if (TMAX == VT_TFAST) { park(mode == 0 ? 1 : 3); return; }        // wave-uniform: THIS window goes to the large lattice (readHead already
            if (TMAX == VT_TMAX && huge) { park(5); return; }                 // points at its first event: gathering it again gives the same events);
            fail = 6; break;                                                  // beyond 512: the read goes on in global memory; beyond 8 192 (or no scratch): it fails
        }
        const int T = __builtin_amdgcn_readfirstlane((int)nt);
        K2B_SYNC();
        K2B_T(1);
        const int coord0 = is_rev ? (ref_end - ri - DN_K / 2) : (ref_start + ri + DN_K / 2);

        // ---- Viterbi, anti-diagonal sweep: lane i = position i, time t = d - i ----
        const bool is0 = lane == 0;
        const double mu = (lane < N) ? model[rank_r[ri + lane]] : 0.0;
        const double tr3 = is0 ? vr.eOrI : vc.D2M;
        const VitCodes kc = vit_codes(is0);
        VitHot hotl = hot;                                  // the lane's constants: position 0 has no left neighbour
        if (is0) { hotl.I2M = NaN; hotl.eM2M = NaN; hotl.M2D = NaN; hotl.D2D = NaN; }
        double I1 = NaN, M1 = NaN, D1 = (lane < VT_NS) ? vc.initD[lane] : NaN;       // own last results (init column, :234-251)
        double oI2 = I1, oM2 = M1, oD2 = D1;                                       // own results one step earlier (tail cell only)
        double sI1 = NaN, sM1 = NaN, sD1 = shl_prev_z(D1);                         // lane i-1's results of the last step (lane 0: never used, see shl_prev_z)
        double sI2 = sI1, sM2 = sM1, sD2 = sD1;                                    // ... and of the step before
        // tail cell: position 64 (only when N == 65), kept in lane 63
        const bool tail = N == 65;
        const double mu64 = tail ? model[rank_r[ri + 64]] : 0.0;
        double tI = NaN, tM = NaN, tD = vc.initD[64];
        const int nsteps = T + N - 1;                      // N == 65: T + 64 steps also cover the tail cell's last step
        // (Round 3, tools/k2b_trace.py: 95 k of a window's 158 k ticks are this loop, ~530 per step for ~60 vector instructions on a lone
        // wavefront.  Fewer instructions do not shorten it -- 6 fewer moves per step: nothing; v_max_f64 instead of compare + select:
        // slower; without the vote on the exp() underflow below: -9 %, but taking that vote one step ahead of its branch: -1 % -- the step is the issue
        // latency of a single wavefront on its SIMD, which other wavefronts fill in the pipeline.)
        // the observation of lane i at step d is xs[d - i]: what lane i - 1 held one step earlier.  It travels through the lanes like the
        // states do (DPP shift); only lane 0 reads LDS, one step ahead -- the read used to sit on every step's critical path.
        double xq = xs[0];
#pragma unroll 2
        for (int d = 0; d < nsteps; d++) {
            const double x0n = xs[min(d + 1, T - 1)];      // lane 0's next observation (wave-uniform address)
            if (tail) {
                const int t64 = d - 64;
                if (t64 >= 0 && t64 < T) {                 // uses lane 63's results of steps d-1 (time t64) and d-2 (time t64-1)
                    const double e = emission(xs[t64], mu64, hot);
                    double In, Mn, Dn; unsigned code;
                    vit_cell(NaN, oD2, hot.D2M, tI, tM, oI2, oM2, M1, D1, e, hot, hot, kc, In, Mn, Dn, code);   // (lane 63 stores it: not position 0's codes)
                    tI = In; tM = Mn; tD = Dn;
                    if (lane == 63) bt[(t64 + 1) * VT_NS + 64] = (unsigned char)code;
                }
                oI2 = I1; oM2 = M1; oD2 = D1;
            }
            const int t = d - lane;
            const bool act = (t >= 0) && (t < T) && (lane < N);
            const double e = emission(xq, mu, hot);
            const double s0 = (is0 && t == 0) ? 0.0 : NaN;                          // start_prev (:235, :432), position 0 only
            const double l3 = is0 ? s0 : sD2;
            double a, b, c2; unsigned code;
            vit_cell(s0, l3, tr3, I1, M1, sI2, sM2, sM1, sD1, e, hotl, hotl, kc, a, b, c2, code);
            if (act) { bt[(t + 1) * VT_NS + lane] = (unsigned char)code; I1 = a; M1 = b; D1 = c2; }
            sI2 = sI1; sM2 = sM1; sD2 = sD1;
            sI1 = shl_prev_z(I1); sM1 = shl_prev_z(M1); sD1 = shl_prev_z(D1);
            xq = shl_prev_d(xq, x0n, lane);
        }
        K2B_SYNC();
        K2B_T(2);
        // ---- termination (:446-476) ----
        double fD, fM, fI;
        if (tail) { fD = bcast_d(tD, 63); fM = bcast_d(tM, 63); fI = bcast_d(tI, 63); }
        else { fD = bcast_d(D1, N - 1); fM = bcast_d(M1, N - 1); fI = bcast_d(I1, N - 1); }
        double score = fD; int st = 0;
        { const double v = fM + vr.eM2MorD; if (ln_gt(v, score)) { score = v; st = 1; } }
        { const double v = fI + vc.I2M; if (ln_gt(v, score)) { score = v; st = 2; } }
        // ---- traceback (:460-509), wave-uniform.  Every emitting state (M or I) visited at column col emitted observation
        // col-1, so the walk writes one label per observation; silent D states only move along the column.
        // The walk is a serial chain of look-ups, one per observation.  Round 3 (tools/k2b_trace.py, ticks per window of the 500 x 50 kb
        // batch): its state lives in SCALAR registers (the window loop's state is provably wave-uniform now, see the top of the loop), the
        // bytes of a column sit in a vector register (lane = position; position 64 in the second byte), fetched a block of 8 columns ahead,
        // and a step is v_readlane + a shift and two masks on the decoded move: 77 k -> 34 k.  Before, the compiler had the whole window
        // loop as DIVERGENT code -- every `if` an execution-mask dance, ~600 ticks per step of this walk -- because one loop-carried value
        // (lastM_ref) came out of a shuffle; with only that repaired the walk took 57 k, and the branchy scalar form of it 58 k. ----
        {
            int i = N - 1, col = T;
            st = __builtin_amdgcn_readfirstlane(st);
            // labels are parked in registers (lane = observation & 63) and leave through ONE 64-lane LDS store per 64
            // observations: a single-lane LDS store per step costs ~50 cycles in this wave-uniform walk
            unsigned lab = 0u; int last_ob = -1;
            int done = 0;
            // codes of 8 columns per block: low byte = the lane's position, second byte = position 64 (wave-uniform); the next block's
            // bytes are requested while this one is walked, so no LDS latency sits between two steps
            unsigned nlo[8], nhi[8];
            auto fetch = [&](int c0) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int c = c0 - k, cc = c < 1 ? 1 : c;                      // columns that do not exist are never looked at
                    nlo[k] = bt[cc * VT_NS + lane]; nhi[k] = bt[cc * VT_NS + 64];
                }
            };
            fetch(T);
            for (int cb = T; !done; cb -= 8) {
                unsigned rows[8];
#pragma unroll
                for (int k = 0; k < 8; k++) rows[k] = nlo[k] | (nhi[k] << 8);
                fetch(cb - 8);
#pragma unroll
                for (int k = 0; k < 8; k++) {                // col == cb - k here unless the walk is done
                    auto code_at = [&]() -> unsigned {
                        const unsigned both = (unsigned)__builtin_amdgcn_readlane((int)rows[k], i & 63);
                        return ((i >> 6) ? (both >> 8) : both) & 0xffu;
                    };
                    while (st == 0 && !done) {               // silent deletions: along the column
                        const unsigned nx = (col > 0) ? ((code_at() >> 5) & 3u) : ((i == 0) ? 3u : 0u);   // column 0 (alignment.cpp:241-251): D all the way to D_0 <- START
                        if (nx == 3u) done = 1; else { st = (int)nx; i -= 1; }
                    }
                    if (!done) {
                        if (col <= 0) done = 1;              // only reachable in an all-log(0) lattice
                        else {
                            const unsigned code = code_at();
                            const int ob = col - 1;
                            lab = (lane == (ob & 63)) ? (unsigned)((st << 8) | i) : lab;
                            if ((ob & 63) == 0) evlab[ob + lane] = (unsigned short)lab;       // observations ob .. ob + 63 are complete
                            last_ob = ob;
                            const unsigned isM = (st == 1) ? 1u : 0u;
                            const unsigned f = code >> (isM ? 0u : 3u);                       // the state's field at bit 0
                            const unsigned nx = f & 3u;
                            i -= (int)((f >> 2) & isM);                                       // only a match can step left
                            done = (nx == 3u) ? 1 : 0;
                            st = done ? st : (int)nx;
                            col -= 1;
                        }
                    }
                }
            }
            // a walk that stopped inside a group of 64 (only in an all-log(0) lattice) still leaves what it labelled
            if (last_ob > 0 && (last_ob & 63) != 0 && lane >= (last_ob & 63)) evlab[(last_ob & ~63) + lane] = (unsigned short)lab;
        }
        K2B_SYNC();
        K2B_T(3);
        // ---- window log ----
        if (lane == 0) {
            O.win_ref[f0 + nwin] = (unsigned)ri; O.win_len[f0 + nwin] = (unsigned)W; O.win_T[f0 + nwin] = (unsigned)T;
            O.win_score[f0 + nwin] = (score == qnan()) ? real_nan() : score;   // log(0) leaves the kernel as NaN
        }
        nwin++;
        // ---- feature fill (:676-736 -> reads.h:292-304), parallel over the observations of the window.
        // Every raw sample of an event labelled M goes to the position of that label; consecutive M labels with the same
        // lattice position share one AlignedPosition (coordinates rise strictly, so a window never reopens an older one).
        // Per event: is it the first M of its position (-> slot number by prefix count) and how many samples of the same
        // position precede it (-> segmented prefix sum).  64 observations per round, carries in scalars. ----
        const unsigned slot0 = npos;
        int lastM_ev = 0, lastM_ref = 0;                    // :655-672 (0, 0 when the path holds no match)
        {
            int carry_pos = -1;                             // lattice position of the last M label seen so far
            unsigned carry_cnt = 0;                         // samples accumulated in that position
            unsigned carry_slots = 0;                       // positions created so far in this window
            for (int base = 0; base < T; base += 64) {
                const int e = base + lane;
                const bool in = e < T;
                const unsigned L = in ? evlab[e] : 0u;
                const bool isM = in && ((L >> 8) == 1u);
                const int p = (int)(L & 0xffu);
                const unsigned len = isM ? tk_len[e] : 0u;
                const unsigned long long mm = __ballot(isM);
                // position of the previous M label (this round or carried)
                const unsigned long long below = mm & ((1ull << lane) - 1ull);
                const int pl = below ? (63 - __clzll((long long)below)) : 0;
                const int pv = __shfl((int)L, pl) & 0xff;                          // (all lanes take part in the shuffle)
                const int prev_p = below ? pv : carry_pos;
                const bool head = isM && (p != prev_p);
                const unsigned long long hm = __ballot(head);
                const unsigned slot = slot0 + carry_slots + (unsigned)__popcll(hm & ((2ull << lane) - 1ull)) - 1u;   // valid for M lanes
                // samples before this event inside its position: inclusive scan of len restarted at heads
                unsigned run = len;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const unsigned up = __shfl_up(run, dlt);
                    // add the partial sum of the lanes [lane-2*dlt+1 .. lane-dlt] unless a head lies in (lane-dlt, lane]
                    const unsigned long long span = (lane >= dlt) ? (((2ull << lane) - 1ull) & ~((2ull << (lane - dlt)) - 1ull)) : ~0ull;
                    if (lane >= dlt && !(hm & span)) run += up;
                }
                const unsigned long long hb = hm & ((2ull << lane) - 1ull);       // heads at or below this lane
                const unsigned before = run - len + (hb ? 0u : carry_cnt);       // no head yet in this round: the carried position continues
                if (isM) { ev_slot[e] = slot; ev_cnt0[e] = before; }
                else if (in) ev_slot[e] = 0xffffffffu;
                if (head) ps_p[slot - slot0] = (unsigned)p;
                // last M of each position in this round records the running count (later rounds may overwrite with more)
                const unsigned long long above = mm & ~((2ull << lane) - 1ull);
                bool last_of_pos = false;
                if (isM) {
                    if (!above) last_of_pos = true;
                    else { const int nl = __ffsll((long long)above) - 1; last_of_pos = ((hm >> nl) & 1ull) != 0ull; }
                }
                if (last_of_pos) ps_cnt[slot - slot0] = before + len;
                // carries
                if (mm) {
                    const int ll = 63 - __clzll((long long)mm);
                    carry_pos = __builtin_amdgcn_readlane((int)L, ll) & 0xff;        // readlane, not a shuffle: the walk's state (ri, readHead through
                    carry_cnt = (unsigned)__builtin_amdgcn_readlane((int)(before + len), ll);   // lastM_*) must stay PROVABLY wave-uniform, or the whole window loop is compiled as divergent code
                    lastM_ev = base + ll; lastM_ref = carry_pos;
                }
                carry_slots += (unsigned)__popcll(hm);
            }
            npos += carry_slots;
            const int n_new = (int)carry_slots;
            K2B_SYNC();
            K2B_T(8);
            // samples: one lane per event.  14 k of a window's 162 k ticks, and it is the HBM latency of the raw samples (untouched since
            // K1): round 3 tried all of an event's samples requested first (20 loads in flight: 18.5 k, the fp64 division then runs for the
            // longest event of every 64) and one lane per sample through an LDS slot map (14.4 k, with or without interleaved divisions):
            // a lone wavefront pays two or three misses per window either way, and beside other wavefronts they are hidden.  Not kept.
            for (int e = lane; e < T; e += 64) {
                const unsigned slot = ev_slot[e];
                if (slot == 0xffffffffu) continue;
                const unsigned c0 = ev_cnt0[e], rs = tk_start[e], rl = tk_len[e];
                for (unsigned j = 0; j < rl && c0 + j < DN_RAWDEPTH_DEV; j++) {
                    const float v = ((float)adc[rs + j] + cal_off) * cal_sc;             // pod5.cpp:60
                    const double scaled = ((double)v - shift) / scale;                   // alignment.cpp:705
                    O.sig[(f0 + slot) * DN_RAWDEPTH_DEV + c0 + j] = (float)scaled;       // reads.h:156
                }
            }
            K2B_T(9);
            for (int q = lane; q < n_new; q += 64) {        // position records: one lane per new position
                const unsigned p = ps_p[q];
                const unsigned idxRef = (unsigned)(ri + (int)p + DN_K / 2);
                const unsigned slot = slot0 + (unsigned)q;
                O.coord[f0 + slot] = is_rev ? (unsigned)(coord0 - (int)p - 1) : (unsigned)(coord0 + (int)p);
                O.ridx[f0 + slot] = idxRef; O.qidx[f0 + slot] = r2q[idxRef];
                O.indel[f0 + slot] = indel; O.nsig[f0 + slot] = ps_cnt[q];
            }
        }
        K2B_T(4);
        // ---- `align` table rows of this window (:697-733): events in order, each with all of its raw samples ----
        if (O.al_val) {
            unsigned carry = 0;
            for (int base = 0; base < T; base += 64) {
                const int e = base + lane;
                const bool in = e < T;
                const unsigned L = in ? evlab[e] : 0u;
                const unsigned stt = L >> 8;
                const bool emit = in && (stt == 1u || (stt == 2u && e < lastM_ev));      // :728: insertions only before the last match
                const unsigned len = emit ? tk_len[e] : 0u;
                unsigned run = len;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) { const unsigned up = __shfl_up(run, dlt); if (lane >= dlt) run += up; }
                if (in) ev_aoff[e] = emit ? (al_rows + carry + run - len) : 0xffffffffu;
                carry += (unsigned)__builtin_amdgcn_readlane((int)run, 63);
            }
            K2B_SYNC();
            const unsigned long long A0 = O.al_off[r];
            for (int e = lane; e < T; e += 64) {
                const unsigned off = ev_aoff[e];
                if (off == 0xffffffffu) continue;
                const unsigned L = evlab[e];
                const int p = (int)(L & 0xffu);
                const unsigned char kind = ((L >> 8) == 1u) ? 0 : 1;
                const unsigned coord = is_rev ? (unsigned)(coord0 - p - 1) : (unsigned)(coord0 + p);
                const unsigned rs = tk_start[e], rl = tk_len[e];
                for (unsigned j = 0; j < rl; j++) {
                    const float v = ((float)adc[rs + j] + cal_off) * cal_sc;             // pod5.cpp:60
                    const unsigned long long row = A0 + off + j;
                    O.al_coord[row] = coord; O.al_rpos[row] = (unsigned)(ri + p); O.al_kind[row] = kind;
                    O.al_val[row] = ((double)v - shift) / scale;                         // :705
                }
            }
            al_rows += carry;
        }
        K2B_T(5);
#ifdef DN_K2B_TRACE
        tacc[7] += 1;
#endif
        readHead += (unsigned)lastM_ev + 1u;                // :739-740
        ri += lastM_ref + 1;
        K2B_SYNC();
        if (mode == 1) { park(2); return; }                 // the one oversized window is done: back to the small lattice
    }
#ifdef DN_K2B_TRACE
    if (lane == 0) for (int k = 0; k < 12; k++) atomicAdd(&k2b_trace[k], tacc[k]);
#endif
    if (lane == 0) {
        if (O.al_n) O.al_n[r] = fail ? 0u : al_rows;
        R.n_positions = fail ? 0u : npos;
        R.n_windows = nwin;
        if (fail) R.status = fail;
        O.redo[r] = 0;
    }
}

// Round 6: the reads that a window of more than 512 observations stopped (redo == 5), in read order: list[0] = how many, list[1 ..] = which.  A stalled pore leaves
// hundreds to thousands of events rough-aligned to ONE k-mer (a 6 000-sample noisy stall: ~1 100 events in one window; found by tests/test_segmentation_adversarial.py:
// the device failed such a read where the reference, which allocates per window, passes it).  They are rare: VT_HUGE_ROUNDS launches of VT_HUGE_WGS wavefronts walk
// them to their ends with the SAME code (k2b_eventalign<VT_THUGE>), the lattice (96 bytes per observation) in global memory -- no LDS is asked for, so these launches
// never wait for a CU to drain.  A batch with more such reads than the rounds cover fails the surplus as before (DN_READ_FAIL_WINDOW_EVENTS).
__global__ __launch_bounds__(64) void k2b_huge_list(BatchDev B, EaDev O, unsigned char *huge) {
    unsigned *list = reinterpret_cast<unsigned *>(huge);
    const int lane = threadIdx.x;
    unsigned n = 0;
    for (int base = 0; base < B.n_reads; base += 64) {
        const int r = base + lane;
        const bool p = r < B.n_reads && O.redo[r] == 5;
        const unsigned long long m = __ballot(p);
        const unsigned at = n + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (p) {
            if (at < VT_HUGE_WGS * VT_HUGE_ROUNDS) list[1 + at] = (unsigned)r;
            else { B.res[r].status = 6; B.res[r].n_positions = 0; if (O.al_n) O.al_n[r] = 0u; O.redo[r] = 0; }
        }
        n += (unsigned)__popcll(m);
    }
    if (lane == 0) list[0] = min(n, (unsigned)(VT_HUGE_WGS * VT_HUGE_ROUNDS));
}
size_t k2b_huge_scratch_bytes() { return K2B_HUGE_HEAD + (size_t)VT_HUGE_WGS * K2B_HUGE_STRIDE; }

// core / residual indices of every aligned position (reads.h:112-138, +1), one thread per position
__global__ __launch_bounds__(256) void k2b_features(BatchDev B, EaDev O) {
    const int r = blockIdx.y;
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (B.res[r].status != 0 || p >= B.res[r].n_positions) return;
    const uint64_t f0 = B.ref_off[r];
    const char *km = B.refseq + f0 + O.ridx[f0 + p] - DN_K / 2;
    unsigned core = 0, res = 0;
#pragma unroll
    for (int j = 2; j < 7; j++) core = core * 4u + base_code(km[j]);
    res = ((base_code(km[0]) * 4u + base_code(km[1])) * 4u + base_code(km[7])) * 4u + base_code(km[8]);
    O.core[f0 + p] = (float)(core + 1u);
    O.resid[f0 + p] = (float)(res + 1u);
}

// upper bound of a read's align-table rows: an event is printed once per rough-alignment pair it is part of, with all its samples
__global__ __launch_bounds__(256) void k2b_rowcap(BatchDev B, unsigned long long *cap) {
    __shared__ unsigned long long part[4];
    const int r = blockIdx.x;
    const ReadRes &R = B.res[r];
    unsigned long long s = 0;
    if (R.status == 0) {
        const uint64_t a0 = B.aln_off[r] + R.aln_begin;
        const unsigned *len = B.ev_len + B.ev_off[r];
        for (unsigned j = threadIdx.x; j < R.n_aligned; j += 256) s += len[B.aln_event[a0 + j]];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) cap[r] = part[0] + part[1] + part[2] + part[3];
}
void k2b_rowcap_launch(const BatchDev &B, unsigned long long *cap, hipStream_t st) {
    hipLaunchKernelGGL(k2b_rowcap, dim3(B.n_reads), dim3(256), 0, st, B, cap);
}

// tap: the emission term exactly as the lattice evaluates it, for a list of (observation, level) pairs (dn_debug_emission)
__global__ __launch_bounds__(64) void k2b_emission_tap(const double *x, const double *mu, double *out, unsigned n, VitConsts vc) {
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < n;
    const double e = emission(live ? x[i] : 0.0, live ? mu[i] : 0.0, vc);     // every lane calls it: it holds a wave-level vote
    if (live) out[i] = (e == qnan()) ? real_nan() : e;                         // log 0 leaves the kernel as NaN, like the window score
}
void k2b_emission_tap_launch(const double *x, const double *mu, double *out, unsigned n, const void *vc, hipStream_t st) {
    hipLaunchKernelGGL(k2b_emission_tap, dim3((n + 63) / 64), dim3(64), 0, st, x, mu, out, n, *reinterpret_cast<const VitConsts *>(vc));
}

// Returns hipSuccess, or the error of the dynamic-LDS opt-in (a launch that needs 88 KB of LDS without it fails opaquely later).
hipError_t k2b_launch(const BatchDev &B, const void *ea, const void *vr, const void *vc, unsigned max_ref, void *huge_scratch, hipStream_t st) {
    const EaDev O = *reinterpret_cast<const EaDev *>(ea);
    const VitConsts V = *reinterpret_cast<const VitConsts *>(vc);
    hipMemsetAsync(O.redo, 0, (size_t)B.n_reads, st);
    const dim3 gf((B.n_reads + K2B_W - 1) / K2B_W), bf(64 * K2B_W);
    const size_t lf = K2B_WAVES_EU > 0 ? sizeof(K2bLds<VT_TFAST>) * K2B_W : 0, lm = K2B_WAVES_EU > 0 ? sizeof(K2bLds<VT_TMAX>) : 0;     // dynamic LDS of the budgeted build
    if (K2B_WAVES_EU > 0) {
        // the opt-in is a property of the function ON A DEVICE: one process may hold contexts on several (dn_ctx_create(device = k)), so it is made once per
        // device, under the device the caller's context made current (round-5 advisor: a process-wide flag covered only the first device, return codes were dropped)
        static std::atomic<unsigned char> done[64];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k2b_eventalign<VT_TFAST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(K2bLds<VT_TFAST>) * K2B_W))) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k2b_eventalign<VT_TMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K2bLds<VT_TMAX>))) != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);
        }
    }
#ifndef K2B_HUGE
#define K2B_HUGE 1                                          /* experiment builds: -DK2B_HUGE=0 = round 5's behaviour (a window beyond 512 observations fails the read) */
#endif
    unsigned char *hs = (K2B_WAVES_EU > 0 && K2B_HUGE) ? (unsigned char *)huge_scratch : nullptr;     // (the static-LDS experiment build has no huge variant)
    hipLaunchKernelGGL(k2b_eventalign<VT_TFAST>, gf, bf, lf, st, B, O, (const VitRead *)vr, V, 0, hs);
    hipLaunchKernelGGL(k2b_eventalign<VT_TMAX>, dim3(B.n_reads), dim3(64), lm, st, B, O, (const VitRead *)vr, V, 1, hs);
    hipLaunchKernelGGL(k2b_eventalign<VT_TFAST>, gf, bf, lf, st, B, O, (const VitRead *)vr, V, 2, hs);
    hipLaunchKernelGGL(k2b_eventalign<VT_TMAX>, dim3(B.n_reads), dim3(64), lm, st, B, O, (const VitRead *)vr, V, 3, hs);
    if (hs) {
        // the reads a window of more than 512 observations stopped (in pass 1 or 3): listed, then walked to their ends with the lattice in global memory
        hipLaunchKernelGGL(k2b_huge_list, dim3(1), dim3(64), 0, st, B, O, hs);
        for (int round = 0; round < VT_HUGE_ROUNDS; round++)
            hipLaunchKernelGGL(k2b_eventalign<VT_THUGE>, dim3(VT_HUGE_WGS), dim3(64), 0, st, B, O, (const VitRead *)vr, V, 5 + round, hs);
    }
    hipLaunchKernelGGL(k2b_features, dim3((max_ref + 255) / 256, B.n_reads), dim3(256), 0, st, B, O);
    return hipSuccess;
}
