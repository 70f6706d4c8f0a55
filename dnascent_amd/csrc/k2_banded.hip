// k2_banded.hip -- K2: adaptive banded event-to-9mer alignment (event_handling.cpp:148-448) on gfx950.
//
//   k2_fill   one wavefront per read.  A band is 100 cells; lane l (l < 50) owns cells 2l and 2l+1, so a band is
//             two VGPRs of scores and the whole recurrence lives in registers: the three neighbours of a cell sit
//             in the same lane or one lane away (wave-shift DPP), depending on the last two Suzuki moves.  The
//             event value x_e and the k-mer level mu_k of a cell also stay in registers and are shifted by one
//             cell per band (only ONE of them moves per band: x on a "down" move, mu on a "right" move); the one
//             new value per band is wave-uniform and is fetched one band ahead through the scalar unit (s_load),
//             so the loop body holds no vector load and never waits on vmcnt.  Per band the kernel stores one
//             128-byte row: 100 trace bytes + the band's lower-left event index, i.e. one full cache line per
//             wavefront store.  No LDS, no atomics.
//             Arithmetic is the reference's, cast by cast (float scores, candidates evaluated in fp64 and
//             rounded back; event_handling.cpp:116-137, :296-306); the fp64 division by sigma is done exactly
//             with an FMA-corrected reciprocal (3 ops, brute-force verified against IEEE division).
//   k2_chase  backtrack (event_handling.cpp:356-412), one wavefront per read: trace rows are staged through a
//             double-buffered 8-KB LDS tile (coalesced 16-B loads), the walk itself is wave-uniform and touches
//             LDS once per step.  It only records the path.
//   k2_post   per-read block: emission log-probabilities of the path, the QC triple (:420-441), and the
//             cleaned (signal, rank) pairs for Theil-Sen, with every order-dependent fp64 sum accumulated in the
//             reference's order.
#include "dn_dev.h"

#define LOG_NEG_INF neg_inf()

// ---- wave shifts.  wave_shl:1 : lane l <- lane l+1 ; wave_shr:1 : lane l <- lane l-1 ; invalid source keeps `fill`
template <bool DPP>
__device__ __forceinline__ float from_next(float v, float fill, int lane) {
    if (DPP) return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
    float t = __shfl_down(v, 1);
    return lane == 63 ? fill : t;
}
template <bool DPP>
__device__ __forceinline__ float from_prev(float v, float fill, int lane) {
    if (DPP) return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
    float t = __shfl_up(v, 1);
    return lane == 0 ? fill : t;
}
template <bool DPP>
__device__ __forceinline__ double from_next_d(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    if (DPP) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    } else {
        lo = __shfl_down(lo, 1); hi = __shfl_down(hi, 1);
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <bool DPP>
__device__ __forceinline__ double from_prev_d(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    if (DPP) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    } else {
        lo = __shfl_up(lo, 1); hi = __shfl_up(hi, 1);
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double set_lane_d(double v, double nv, int which, int lane) { return lane == which ? nv : v; }

// self-test of the shift primitives (dn_ctx_create refuses to run if this fails)
__global__ void k2_selftest(int *out) {
    const int lane = threadIdx.x;
    const float v = (float)(lane * 3 + 1);
    const float a = from_next<true>(v, -7.0f, lane), b = from_next<false>(v, -7.0f, lane);
    const float c = from_prev<true>(v, -9.0f, lane), d = from_prev<false>(v, -9.0f, lane);
    const double x = (double)lane * 1.25e100 + 3.0;
    const double e = from_next_d<true>(x, lane), f = from_next_d<false>(x, lane);
    const double g = from_prev_d<true>(x, lane), h = from_prev_d<false>(x, lane);
    int ok = (a == b) && (c == d);
    if (lane < 63) ok = ok && (e == f) && (a == (float)((lane + 1) * 3 + 1));
    if (lane > 0) ok = ok && (g == h) && (c == (float)((lane - 1) * 3 + 1));
    if (lane == 63) ok = ok && (a == -7.0f);
    if (lane == 0) ok = ok && (c == -9.0f);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) out[0] = (m == ~0ull) ? 1 : 0;
}

struct BandConsts {          // per read, computed on the host with the host's libm (identical to the reference's calls)
    double lp_stay, lp_step; // event_handling.cpp:174-182
};

struct FillConsts {
    double lp_skip, lp_trim; // log(1e-30), log(0.01)
    double C;                // (double)(float)log(0.3989422804014327) - log(sigma)   (event_handling.cpp:134-135)
    double sigma, rsigma;    // rsigma = RN(1/sigma)
};

// one cell of the recurrence (event_handling.cpp:280-311 + :116-137)
__device__ __forceinline__ void cell(float diag, float up, float left, double x, double mu, const FillConsts &fc,
                                     double lp_step, double lp_stay, float &score, unsigned &from) {
    const double d = x - mu;
    const double q = d * fc.rsigma;                       // exact (x - mu) / sigma via FMA-corrected reciprocal
    const double rem = fma(-q, fc.sigma, d);
    const double ad = fma(rem, fc.rsigma, q);
    const float a = (float)ad;                            // :133
    float t = -0.5f * a;                                  // :135 (-0.5f * a) * a in float
    t = t * a;
    const float em = (float)(fc.C + (double)t);           // :135-136
    const double emd = (double)em;
    const float sd = (float)(((double)diag + lp_step) + emd);   // :296
    const float su = (float)(((double)up + lp_stay) + emd);     // :297
    const float sl = (float)((double)left + fc.lp_skip);        // :298
    // :300-306  max = d; if (u > max) max = u; from = (max == u) ? U : D; same for l.  For non-NaN scores
    // "max == u after the update" is exactly "u >= max before it", so one compare drives both selects.
    const bool ge_u = su >= sd;
    float mx = ge_u ? su : sd; unsigned f = ge_u ? 1u : 0u;
    const bool ge_l = sl >= mx;
    mx = ge_l ? sl : mx; f = ge_l ? 2u : f;
    score = mx; from = f;
}

// uniform (wave-wide) loads of the one new x / mu value a band needs go through the scalar unit (s_load, counted
// on lgkmcnt): the loop then holds no vector load, so its trace stores are never waited for (loads and stores share
// the in-order vmcnt on gfx950).  The arrays were written by earlier kernels and are read-only here, which is what
// the constant address space promises.
typedef const double __attribute__((address_space(4))) *cdptr_t;

template <bool DPP>
__global__ __launch_bounds__(64) void k2_fill(BatchDev B, const BandConsts *bc, FillConsts fc) {
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = (int)R.n_events, K = (int)R.n_kq;
    const int n_bands = E + K + 2;
    const double lp_stay = bc[r].lp_stay, lp_step = bc[r].lp_step;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    const cdptr_t xs_c = (cdptr_t)(uintptr_t)xs;
    const cdptr_t mu_c = (cdptr_t)(uintptr_t)mus;
    uint8_t *rows = B.trace + B.trace_off[r] * DN_TROW;
    const float NINF = neg_inf();
    const bool inb = lane < 50;
    // cell offsets of this lane; lanes >= 50 hold no cell: an offset that fails every range test
    const unsigned o0 = inb ? (unsigned)(2 * lane) : 0x40000000u, o1 = inb ? (unsigned)(2 * lane + 1) : 0x40000000u;
    // row bytes 104..107 carry the band's lower-left event index (lanes 52, 53 hold its two halves)
    const unsigned meta_shift = (lane == 53) ? 16u : 0u;
    const unsigned meta_mask = (lane == 52 || lane == 53) ? 0xffffu : 0u;

    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    int ev = 50, km = -51;                                // lower-left of band 1; band 0 is (49, -51)
    float Q0 = (o0 == 50u) ? 0.0f : NINF, Q1 = NINF;      // band 0: score 0 at the cell with kmer == -1 (offset 50)
    float P0 = (o0 == 50u) ? (float)fc.lp_trim : NINF, P1 = NINF;   // band 1: first event trimmed (offset 50)
    {
        const unsigned short w1 = (o0 == 50u) ? 1u : 0u;  // trace[1][50] = FROM_U
        reinterpret_cast<unsigned short *>(rows)[lane] = inb ? (unsigned short)0 : (unsigned short)((49u >> meta_shift) & meta_mask);
        reinterpret_cast<unsigned short *>(rows + DN_TROW)[lane] = inb ? w1 : (unsigned short)((50u >> meta_shift) & meta_mask);
    }
    // x / mu of this lane's two cells in band 1: event index ev - o, kmer index km + o
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };
    double X0 = ldx(ev - (int)o0), X1 = ldx(ev - (int)o1);
    double M0 = ldm(km + (int)o0), M1 = ldm(km + (int)o1);
    // value entering the band on the next "down" (x[ev+1] -> cell 0) / "right" (mu[km+100] -> cell 99) move
    double nx = xs_c[min(ev + 1, E - 1)];
    double nm = mu_c[max(min(km + 100, K - 1), 0)];

    float best = NINF; int best_e = 0; int found = 0;
    int prev_right = 0;                                   // band 0 -> 1 was a "down" move

    for (int b = 2; b < n_bands; b++) {
        // ---- Suzuki-Kasahara move (:237-253) ----
        const float lo = bcast_f(P0, 0), hi = bcast_f(P1, 49);
        int right;
        if (lo == NINF && hi == NINF) right = (b & 1);
        else right = lo < hi;

        float up0, up1, lf0, lf1, dg0, dg1;
        if (right) {
            km += 1;
            const double t = from_next_d<DPP>(M0, lane);
            M0 = M1; M1 = (lane == 49) ? nm : t;
            up0 = P1; lf0 = P0; up1 = from_next<DPP>(P0, NINF, lane); lf1 = P1;
            if (prev_right) { dg0 = Q1; dg1 = from_next<DPP>(Q0, NINF, lane); }
            else { dg0 = Q0; dg1 = Q1; }
        } else {
            ev += 1;
            const double t = from_prev_d<DPP>(X1, lane);
            X1 = X0; X0 = (lane == 0) ? nx : t;
            up0 = P0; lf0 = from_prev<DPP>(P1, NINF, lane); up1 = P1; lf1 = P0;
            if (prev_right) { dg0 = Q0; dg1 = Q1; }
            else { dg0 = from_prev<DPP>(Q1, NINF, lane); dg1 = Q0; }
        }
        // prefetch for the next band (consumed one iteration later)
        nx = xs_c[min(ev + 1, E - 1)];
        nm = mu_c[max(min(km + 100, K - 1), 0)];

        // ---- the two cells of this lane ----
        float S0, S1; unsigned F0, F1;
        cell(dg0, up0, lf0, X0, M0, fc, lp_step, lp_stay, S0, F0);
        cell(dg1, up1, lf1, X1, M1, fc, lp_step, lp_stay, S1, F1);
        // in range: 0 <= kmer < K and 0 <= event < E (:269-278), one unsigned compare each
        const unsigned e0 = (unsigned)ev - o0, e1 = (unsigned)ev - o1;
        const bool ok0 = ((unsigned)km + o0 < (unsigned)K) && (e0 < (unsigned)E);
        const bool ok1 = ((unsigned)km + o1 < (unsigned)K) && (e1 < (unsigned)E);
        S0 = ok0 ? S0 : NINF; F0 = ok0 ? F0 : 0u;
        S1 = ok1 ? S1 : NINF; F1 = ok1 ? F1 : 0u;
        if (km <= -1) {
            // trim column kmer == -1 is still inside the band (:256-265); only the first ~100 bands get here
            const bool t0 = ((unsigned)km + o0 == 0xffffffffu) && (e0 < (unsigned)E);
            const bool t1 = ((unsigned)km + o1 == 0xffffffffu) && (e1 < (unsigned)E);
            if (t0) { S0 = (float)(fc.lp_trim * (double)(e0 + 1u)); F0 = 1; }
            if (t1) { S1 = (float)(fc.lp_trim * (double)(e1 + 1u)); F1 = 1; }
        }

        // ---- one 128-byte row: 100 trace bytes (lanes 0..49) + the band's lower-left event index ----
        const unsigned meta = ((unsigned)ev >> meta_shift) & meta_mask;
        const unsigned short w = (unsigned short)(inb ? (F0 | (F1 << 8)) : meta);
        reinterpret_cast<unsigned short *>(rows + (size_t)b * DN_TROW)[lane] = w;

        // ---- end cell: best score on the last k-mer column after trimming the remaining events (:329-340) ----
        const int oe = K - 1 - km;
        if (oe >= 0 && oe < DN_W) {
            const int ee = ev - oe;
            if (ee >= 0 && ee < E) {
                const float sv = bcast_f((oe & 1) ? S1 : S0, oe >> 1);
                const float s = (float)((double)sv + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                if (s > best) { best = s; best_e = ee; found = 1; }
            }
        }
        Q0 = P0; Q1 = P1; P0 = S0; P1 = S1;
        prev_right = right;
    }
    if (lane == 0) {
        R.n_bands = (unsigned)n_bands;
        R.end_event = best_e;
        R.end_score = best;
        if (!found) R.status = 3;
    }
}

// ------------------------------------------------------------------------------------------------
// k2_chase: record the backtrack path.  aln arrays are filled from the back so they end up in forward order.
// ------------------------------------------------------------------------------------------------
#define CH_ROWS 64

__global__ __launch_bounds__(64) void k2_chase(BatchDev B, uint8_t *path_from) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][CH_ROWS * DN_TROW];
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = (int)R.n_events, K = (int)R.n_kq;
    const uint8_t *rows = B.trace + B.trace_off[r] * DN_TROW;
    const uint64_t a0 = B.aln_off[r];
    const unsigned cap = (unsigned)(B.aln_off[r + 1] - a0);
    unsigned *ae = B.aln_event + a0, *ak = B.aln_kmer + a0;
    uint8_t *pf = path_from + a0;

    int e = R.end_event, k = K - 1;
    int b = e + k + 2;
    // tile t covers bands [lo, lo + 63]; cur tile index 0/1
    int lo = b - (CH_ROWS - 1); if (lo < 0) lo = 0;
    auto load_tile = [&](int tlo, int4 (&regs)[8]) {
        // 64 rows * 128 B = 512 pieces of 16 B; lane handles pieces lane, lane+64, ...
        const int4 *src = reinterpret_cast<const int4 *>(rows + (size_t)tlo * DN_TROW);
#pragma unroll
        for (int i = 0; i < 8; i++) regs[i] = src[lane + 64 * i];
    };
    auto store_tile = [&](int which, const int4 (&regs)[8]) {
        int4 *dst = reinterpret_cast<int4 *>(tile[which]);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[lane + 64 * i] = regs[i];
    };
    int4 regs[8];
    // note: rows below band 0 do not exist; tiles are clamped at 0 and always hold 64 rows starting at `lo`
    // (rows above the read's last band are never addressed).  The trace allocation is padded by 64 rows.
    load_tile(lo, regs);
    store_tile(0, regs);
    int cur = 0;
    int nlo = lo - CH_ROWS; if (nlo < 0) nlo = 0;
    const bool have_next0 = lo > 0;
    if (have_next0) load_tile(nlo, regs);
    __syncthreads();
    // per-row lower-left event index of the current tile, one row per lane
    auto row_ev = [&](int which) -> int {
        const uint8_t *p = tile[which] + lane * DN_TROW + 104;
        return (int)(*reinterpret_cast<const unsigned *>(p));
    };
    int evrow = row_ev(cur);

    unsigned step = 0;
    unsigned rec_e = 0, rec_k = 0, rec_f = 0;
    int bad = 0;
    while (k >= 0 && e >= 0) {
        if (b < lo) {
            // switch to the prefetched tile
            cur ^= 1;
            store_tile(cur, regs);
            lo = nlo;
            nlo = lo - CH_ROWS; if (nlo < 0) nlo = 0;
            if (lo > 0) load_tile(nlo, regs);
            __syncthreads();
            evrow = row_ev(cur);
        }
        const int bi = b - lo;
        const int ev_b = __builtin_amdgcn_readlane(evrow, bi);
        const int off = ev_b - e;
        if (off < 0 || off >= DN_W || step >= cap) { bad = 1; break; }    // reference: out-of-bounds read (UB)
        const unsigned from = tile[cur][bi * DN_TROW + off];
        // stash step in lane (step & 63); flush 64 steps at a time, back to front
        const int slot = step & 63;
        if (lane == slot) { rec_e = (unsigned)e; rec_k = (unsigned)k; rec_f = from; }
        step++;
        if ((step & 63u) == 0u) {
            const unsigned idx = cap - (step - 64u) - 1u - (unsigned)lane;     // step-64+lane -> slot cap-1-(step-64+lane)
            ae[idx] = rec_e; ak[idx] = rec_k; pf[idx] = (uint8_t)rec_f;
        }
        if (from == 0) { e -= 1; k -= 1; b -= 2; }
        else if (from == 1) { e -= 1; b -= 1; }
        else { k -= 1; b -= 1; }
    }
    const unsigned rem = step & 63u;
    if (!bad && rem && (unsigned)lane < rem) {
        const unsigned base = step - rem;
        const unsigned idx = cap - (base + (unsigned)lane) - 1u;
        ae[idx] = rec_e; ak[idx] = rec_k; pf[idx] = (uint8_t)rec_f;
    }
    if (lane == 0) {
        if (bad) { R.status = 3; R.n_aligned = 0; R.aln_begin = cap; }
        else { R.n_aligned = step; R.aln_begin = cap - step; }
    }
}

// ------------------------------------------------------------------------------------------------
// k2_post: QC + cleaned pairs from the recorded path.  Path slot j (forward order) corresponds to walk step
// n-1-j; the reference accumulates in WALK order (from the read's end to its start).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lp_match_dev(double x, double mu, const FillConsts &fc) {
    const double d = x - mu;
    const double q = d * fc.rsigma;
    const double rem = fma(-q, fc.sigma, d);
    const float a = (float)fma(rem, fc.rsigma, q);
    float t = -0.5f * a;
    t = t * a;
    return (float)(fc.C + (double)t);
}

__global__ __launch_bounds__(256) void k2_post(BatchDev B, const uint8_t *path_from, float *path_lp, FillConsts fc) {
    __shared__ int redi[256];
    __shared__ unsigned s_base;
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    ReadRes &R = B.res[r];
    if (R.status != 0) {
        if (tid == 0) { R.n_cleaned = 0; R.avg_log_emission = 0.; R.spanned = 0; R.max_gap = 0; }
        return;
    }
    const unsigned n = R.n_aligned;
    const uint64_t a0 = B.aln_off[r];
    const unsigned beg = R.aln_begin;
    const unsigned *ae = B.aln_event + a0 + beg, *ak = B.aln_kmer + a0 + beg;
    const uint8_t *pf = path_from + a0 + beg;
    float *lp = path_lp + a0 + beg;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *means = B.ev_mean + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    const int32_t *q2r = B.query2ref + B.base_off[r] + r;
    const unsigned *rank_r = B.rank_r + B.ref_off[r];
    const unsigned n_kr = R.n_kr;
    double *cl_sig = B.cl_sig + a0; unsigned *cl_rank = B.cl_rank + a0;

    // 1. emission of every aligned pair (:362-363), parallel
    for (unsigned j = tid; j < n; j += 256) lp[j] = lp_match_dev(xs[ae[j]], mus[ak[j]], fc);
    __syncthreads();

    // 2. cleaned pairs (:380-395).  A pair is emitted at every diagonal step whose query position maps to the
    //    reference; its signal is the mean of the event means buffered since the previous diagonal step, summed in
    //    push (walk) order.  Walk step w <-> slot j = n-1-w.  Ordered compaction in walk order by block scan.
    unsigned out_base = 0;
    for (unsigned wb = 0; wb < n; wb += 256) {
        const unsigned w = wb + tid;
        int flag = 0; double sig = 0.; unsigned rk = 0;
        if (w < n) {
            const unsigned j = n - 1 - w;
            if (pf[j] == 0) {
                const int32_t pos = q2r[ak[j]];
                if (pos >= 0 && (unsigned)pos < n_kr) {
                    flag = 1; rk = rank_r[pos];
                    // buffered steps: walk steps w' < w back to (excluding) the previous diagonal; slots j' > j
                    unsigned jj = j + 1;
                    while (jj < n && pf[jj] != 0) jj++;
                    // push order == walk order == descending slot from jj-1 down to j
                    double total = 0.; unsigned cnt = 0;
                    for (unsigned s = jj; s-- > j;) {
                        if (pf[s] != 2) { total += means[ae[s]]; cnt++; }     // FROM_L pushes nothing (:407-411)
                    }
                    sig = total / (double)cnt;                                // vectorMean common.h:185
                }
            }
        }
        // block exclusive scan of flags
        redi[tid] = flag;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            int t = (tid >= d) ? redi[tid - d] : 0;
            __syncthreads();
            redi[tid] += t;
            __syncthreads();
        }
        const unsigned pos = out_base + (unsigned)redi[tid] - (unsigned)flag;
        if (flag) { cl_sig[pos] = sig; cl_rank[pos] = rk; }
        if (tid == 255) s_base = out_base + (unsigned)redi[255];
        __syncthreads();
        out_base = s_base;
        __syncthreads();
    }

    // 3. ordered fp64 sum of the emissions and the gap statistic: one wavefront, walk order, staged through LDS
    if (tid < 64) {
        double sum_em = 0.; int gap = 0, max_gap = 0;
        for (unsigned wb = 0; wb < n; wb += 64) {
            const unsigned w = wb + tid;
            float v = 0.f; unsigned f = 1;
            if (w < n) { v = lp[n - 1 - w]; f = pf[n - 1 - w]; }
            const unsigned lim = min(64u, n - wb);
            for (unsigned i = 0; i < lim; i++) {
                const float vi = bcast_f(v, (int)i);
                const unsigned fi = (unsigned)__builtin_amdgcn_readlane((int)f, (int)i);
                sum_em += (double)vi;                                         // :364
                if (fi == 2) { gap += 1; max_gap = max(max_gap, gap); } else gap = 0;   // :399-410
            }
        }
        if (tid == 0) {
            R.avg_log_emission = sum_em / (double)n;                          // :420 (n_aligned_events is a double count)
            R.spanned = (n > 0 && ak[0] == 0 && ak[n - 1] == (unsigned)(R.n_kq - 1)) ? 1 : 0;   // :421
            R.max_gap = max_gap;
            R.n_cleaned = out_base;
            int fail = 0;
            if (R.avg_log_emission < -2.0 || !R.spanned || max_gap > 5) fail = 1;   // :433, config.h:41
            if (out_base < 1000) fail = 1;                                          // :438
            if (fail) R.status = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
int k2_selftest_run(hipStream_t st) {
    int *d = nullptr; int h = 0;
    if (hipMalloc(&d, sizeof(int)) != hipSuccess) return -1;
    hipMemsetAsync(d, 0, sizeof(int), st);
    hipLaunchKernelGGL(k2_selftest, dim3(1), dim3(64), 0, st, d);
    hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    hipFree(d);
    return h;
}
void k2_launch_fill(const BatchDev &B, const void *bc, const void *fc, bool dpp, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    if (dpp) hipLaunchKernelGGL(k2_fill<true>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
    else hipLaunchKernelGGL(k2_fill<false>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
}
void k2_launch_chase(const BatchDev &B, uint8_t *path_from, hipStream_t st) {
    hipLaunchKernelGGL(k2_chase, dim3(B.n_reads), dim3(64), 0, st, B, path_from);
}
void k2_launch_post(const BatchDev &B, const uint8_t *path_from, float *path_lp, const void *fc, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    hipLaunchKernelGGL(k2_post, dim3(B.n_reads), dim3(256), 0, st, B, path_from, path_lp, f);
}
