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
#include <stdlib.h>

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

// v_writelane_b32 with the value in an SGPR and the lane select in M0 (gfx9 allows one SGPR per VALU instruction; M0 is the
// second scalar source the instruction accepts).  val and lane are wave-uniform.
__device__ __forceinline__ int writelane_(int old, int val, int lane) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(lane) : "m0");
    return old;
}

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
    // primitives of k2_fill6: wave rotate (lane l <- lane l - 1 mod 64), a uniform mask as a lane predicate, writelane
    const int rr = __builtin_amdgcn_update_dpp(0, lane * 5 + 2, 0x13C, 0xf, 0xf, false);
    ok = ok && (rr == ((lane + 63) & 63) * 5 + 2);
    const unsigned long long pat = 0x8000000000000001ull | (0x5ull << 20);
    ok = ok && (__builtin_amdgcn_inverse_ballot_w64(pat) == (((pat >> lane) & 1ull) != 0ull));
    const int wl = writelane_(lane, 777, 37);
    ok = ok && (wl == (lane == 37 ? 777 : lane));
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
    // :300-306  max = d; if (u > max) max = u; from = (max == u) ? U : D; then the same for l: the result is the
    // maximum of the three with ties resolved L over U over D (scores are never NaN), i.e. one v_max3 + two compares.
    float mx;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(sd), "v"(su), "v"(sl));
    unsigned f = (su == mx) ? 1u : 0u;
    f = (sl == mx) ? 2u : f;
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
            const float nq = from_next<DPP>(Q0, NINF, lane);
            up0 = P1; lf0 = P0; up1 = from_next<DPP>(P0, NINF, lane); lf1 = P1;
            dg0 = prev_right ? Q1 : Q0; dg1 = prev_right ? nq : Q1;
        } else {
            ev += 1;
            const double t = from_prev_d<DPP>(X1, lane);
            X1 = X0; X0 = (lane == 0) ? nx : t;
            const float pq = from_prev<DPP>(Q1, NINF, lane);
            up0 = P0; lf0 = from_prev<DPP>(P1, NINF, lane); up1 = P1; lf1 = P0;
            dg0 = prev_right ? Q0 : pq; dg1 = prev_right ? Q1 : Q0;
        }
        // prefetch for the next band (consumed one iteration later)
        nx = xs_c[min(ev + 1, E - 1)];
        nm = mu_c[max(min(km + 100, K - 1), 0)];

        // ---- the two cells of this lane ----
        float S0, S1; unsigned F0, F1;
        cell(dg0, up0, lf0, X0, M0, fc, lp_step, lp_stay, S0, F0);
        cell(dg1, up1, lf1, X1, M1, fc, lp_step, lp_stay, S1, F1);
        if (km >= 0 && km + (DN_W - 1) < K && ev >= DN_W - 1 && ev < E) {
            // interior band (almost all of them): every cell 0..99 is inside the matrix; only the 14 idle lanes are masked
            S0 = inb ? S0 : NINF; S1 = inb ? S1 : NINF;
        } else {
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
// k2_fill2: TWO wavefronts per read, one band cell per lane (100 of 128 lanes).
//
// A single wavefront issues one VALU instruction every ~5 cycles at best (measured, profiles/r01_valu_issue_microbench.txt)
// and a 1 000-read batch gives the chip only 1 000 of them, one per SIMD.  Splitting a read over two wavefronts halves the
// per-band instruction stream of each and puts two wavefronts on every SIMD, which the SIMD interleaves for free.  The
// price is one workgroup barrier per band, so the band state moves to LDS:
//   * the scores of the last three bands live in a 3-slot LDS ring; the neighbours of cell o are plain LDS reads at
//     o-1, o, o+1 (both Suzuki moves are read at once, then selected: one LDS round trip per band, no dependent second read);
//   * x_e and mu_k are served from two 512-entry LDS rings refilled 128 entries at a time, the global load for the next
//     refill being issued one refill (>= 128 bands) ahead;
//   * trace: one byte per lane into the same 128-byte row layout as before.
// ------------------------------------------------------------------------------------------------
#define F2_RING 512
#define F2_PW 136

__global__ __launch_bounds__(128) void k2_fill2(BatchDev B, const BandConsts *bc, FillConsts fc) {
    __shared__ float Pb[3][F2_PW];                        // [band % 3][cell offset + 1]; [0] and [101..] stay -inf
    __shared__ double rx[F2_RING], rm[F2_RING];           // x[e] at e & 511, mu[k] at k & 511
    __shared__ float red_s[128]; __shared__ int red_e[128];
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = (int)R.n_events, K = (int)R.n_kq;
    const int n_bands = E + K + 2;
    const double lp_stay = bc[r].lp_stay, lp_step = bc[r].lp_step;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    uint8_t *rows = B.trace + B.trace_off[r] * DN_TROW;
    const float NINF = neg_inf();
    const int o = tid;                                    // cell offset of this lane
    const bool inb = o < DN_W;
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };

    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    for (int i = tid; i < 3 * F2_PW; i += 128) (&Pb[0][0])[i] = NINF;
    for (int i = tid; i < 384; i += 128) { rx[i] = ldx(i); rm[i] = ldm(i); }
    int xhi = 384, mhi = 384;                             // rings hold [.., xhi) / [.., mhi)
    double pendx = ldx(xhi + tid), pendm = ldm(mhi + tid);
    __syncthreads();
    if (tid == 0) { Pb[0][50 + 1] = 0.0f; Pb[1][50 + 1] = (float)fc.lp_trim; }
    int ev = 50, km = -51;                                // lower-left of band 1; band 0 is (49, -51)
    {
        uint8_t b0 = 0, b1 = (o == 50) ? 1 : 0;           // trace[1][50] = FROM_U
        if (tid >= 104 && tid < 108) { b0 = (uint8_t)((49u >> (8 * (tid - 104))) & 0xff); b1 = (uint8_t)((50u >> (8 * (tid - 104))) & 0xff); }
        rows[tid] = b0; rows[DN_TROW + tid] = b1;
    }
    double x = ldx(ev - o), mu = ldm(km + o);
    float best = NINF; int best_e = 0x7fffffff;
    int prev_right = 0;
    __syncthreads();
    // retire the pre-loop global loads HERE (vmcnt(0), builtin form so the compiler's scoreboard sees it): otherwise
    // the first use of x / mu inside the loop carries a vmcnt(0) on every iteration and each band waits for its own store
    __builtin_amdgcn_s_waitcnt(0x0F70);

    for (int b = 2; b < n_bands; b++) {
        const int sp = (b + 2) % 3, sq = (b + 1) % 3, sc = b % 3;
        const float *Pp = Pb[sp], *Qq = Pb[sq];
        // ---- one LDS round trip: everything either move can need ----
        const float lo = Pp[1], hi = Pp[DN_W];
        const float pm1 = Pp[o], p0 = Pp[o + 1], pp1 = Pp[o + 2];            // P[o-1], P[o], P[o+1]
        const float qa = Qq[o + prev_right], qb = Qq[o + 1 + prev_right];    // diag for a down / right move
        const double xn = rx[(ev + 1 - o) & (F2_RING - 1)];                  // x of this cell after a down move
        const double mn = rm[(km + 1 + o) & (F2_RING - 1)];                  // mu of this cell after a right move
        // ---- Suzuki-Kasahara move (:237-253) ----
        const bool vright = (lo == NINF && hi == NINF) ? ((b & 1) != 0) : (lo < hi);
        const float up = vright ? pp1 : p0, left = vright ? p0 : pm1, diag = vright ? qb : qa;
        x = vright ? x : xn;
        mu = vright ? mn : mu;
        const int right = __builtin_amdgcn_readfirstlane((int)vright);
        ev += 1 - right; km += right;

        float S; unsigned F;
        cell(diag, up, left, x, mu, fc, lp_step, lp_stay, S, F);
        if (km >= 0 && km + (DN_W - 1) < K && ev >= DN_W - 1 && ev < E) {
            S = inb ? S : NINF;                           // interior band: every cell 0..99 is inside the matrix
        } else {
            const unsigned e_u = (unsigned)ev - (unsigned)o;
            const bool ok = inb && ((unsigned)(km + o) < (unsigned)K) && (e_u < (unsigned)E);   // :269-278
            S = ok ? S : NINF; F = ok ? F : 0u;
            if (inb && km + o == -1 && e_u < (unsigned)E) { S = (float)(fc.lp_trim * (double)(e_u + 1u)); F = 1; }   // :256-265
        }
        Pb[sc][o + 1] = S;                                // lanes >= 100 rewrite -inf into the sentinel cells
        // ---- trace row: byte per cell + the band's lower-left event index in bytes 104..107 ----
        uint8_t tb = (uint8_t)F;
        if (tid >= 104 && tid < 108) tb = (uint8_t)(((unsigned)ev >> (8 * (tid - 104))) & 0xffu);
        rows[(size_t)b * DN_TROW + tid] = tb;
        // ---- end cell (:329-340): the lane that holds column K-1 keeps its best; reduced after the loop ----
        const int oe = K - 1 - km;
        if (oe >= 0 && oe < DN_W) {
            const int ee = ev - o;
            if (o == oe && ee >= 0 && ee < E) {
                const float sv = (float)((double)S + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                if (sv > best) { best = sv; best_e = ee; }
            }
        }
        // ---- ring refills (uniform, rare) ----
        if (ev + 130 >= xhi) { rx[(xhi + tid) & (F2_RING - 1)] = pendx; xhi += 128; pendx = ldx(xhi + tid); }
        if (km + 230 >= mhi) { rm[(mhi + tid) & (F2_RING - 1)] = pendm; mhi += 128; pendm = ldm(mhi + tid); }
        prev_right = right;
        // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for this band's trace store to be acknowledged
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // first maximum in increasing event order == maximum score, smallest event on ties
    red_s[tid] = best; red_e[tid] = best_e;
    __syncthreads();
    if (tid == 0) {
        float bs = NINF; int be = 0x7fffffff;
        for (int i = 0; i < DN_W; i++)
            if (red_e[i] != 0x7fffffff && (red_s[i] > bs || (red_s[i] == bs && red_e[i] < be))) { bs = red_s[i]; be = red_e[i]; }
        const int found = (be != 0x7fffffff) && (bs > NINF);
        R.n_bands = (unsigned)n_bands;
        R.end_event = found ? be : 0;
        R.end_score = bs;
        if (!found) R.status = 3;
    }
}

// diagnostic build of k2_fill2 with s_memtime stamps per phase (DN_FILL_VARIANT=22); never used for results or timing
__global__ __launch_bounds__(128) void k2_fill2p(BatchDev B, const BandConsts *bc, FillConsts fc) {
    __shared__ float Pb[3][F2_PW];                        // [band % 3][cell offset + 1]; [0] and [101..] stay -inf
    __shared__ double rx[F2_RING], rm[F2_RING];           // x[e] at e & 511, mu[k] at k & 511
    __shared__ float red_s[128]; __shared__ int red_e[128];
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = (int)R.n_events, K = (int)R.n_kq;
    const int n_bands = E + K + 2;
    const double lp_stay = bc[r].lp_stay, lp_step = bc[r].lp_step;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    uint8_t *rows = B.trace + B.trace_off[r] * DN_TROW;
    const float NINF = neg_inf();
    const int o = tid;                                    // cell offset of this lane
    const bool inb = o < DN_W;
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };

    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    for (int i = tid; i < 3 * F2_PW; i += 128) (&Pb[0][0])[i] = NINF;
    for (int i = tid; i < 384; i += 128) { rx[i] = ldx(i); rm[i] = ldm(i); }
    int xhi = 384, mhi = 384;                             // rings hold [.., xhi) / [.., mhi)
    double pendx = ldx(xhi + tid), pendm = ldm(mhi + tid);
    __syncthreads();
    if (tid == 0) { Pb[0][50 + 1] = 0.0f; Pb[1][50 + 1] = (float)fc.lp_trim; }
    int ev = 50, km = -51;                                // lower-left of band 1; band 0 is (49, -51)
    {
        uint8_t b0 = 0, b1 = (o == 50) ? 1 : 0;           // trace[1][50] = FROM_U
        if (tid >= 104 && tid < 108) { b0 = (uint8_t)((49u >> (8 * (tid - 104))) & 0xff); b1 = (uint8_t)((50u >> (8 * (tid - 104))) & 0xff); }
        rows[tid] = b0; rows[DN_TROW + tid] = b1;
    }
    double x = ldx(ev - o), mu = ldm(km + o);
    float best = NINF; int best_e = 0x7fffffff;
    int prev_right = 0;
    __syncthreads();
    // retire the pre-loop global loads HERE (vmcnt(0), builtin form so the compiler's scoreboard sees it): otherwise
    // the first use of x / mu inside the loop carries a vmcnt(0) on every iteration and each band waits for its own store
    __builtin_amdgcn_s_waitcnt(0x0F70);

    unsigned long long T[6] = {0,0,0,0,0,0};
#define STAMP(i) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); T[i] += t_ - tl; tl = t_; }
    unsigned long long tl; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl) :: "memory");
    for (int b = 2; b < n_bands; b++) {
        const int sp = (b + 2) % 3, sq = (b + 1) % 3, sc = b % 3;
        const float *Pp = Pb[sp], *Qq = Pb[sq];
        // ---- one LDS round trip: everything either move can need ----
        const float lo = Pp[1], hi = Pp[DN_W];
        const float pm1 = Pp[o], p0 = Pp[o + 1], pp1 = Pp[o + 2];            // P[o-1], P[o], P[o+1]
        const float qa = Qq[o + prev_right], qb = Qq[o + 1 + prev_right];    // diag for a down / right move
        const double xn = rx[(ev + 1 - o) & (F2_RING - 1)];                  // x of this cell after a down move
        const double mn = rm[(km + 1 + o) & (F2_RING - 1)];                  // mu of this cell after a right move
        asm volatile("" :: "v"(lo), "v"(hi), "v"(pm1), "v"(p0), "v"(pp1), "v"(qa), "v"(qb), "v"(xn), "v"(mn));
        STAMP(0)
        // ---- Suzuki-Kasahara move (:237-253) ----
        const bool vright = (lo == NINF && hi == NINF) ? ((b & 1) != 0) : (lo < hi);
        const float up = vright ? pp1 : p0, left = vright ? p0 : pm1, diag = vright ? qb : qa;
        x = vright ? x : xn;
        mu = vright ? mn : mu;
        const int right = __builtin_amdgcn_readfirstlane((int)vright);
        ev += 1 - right; km += right;

        asm volatile("" :: "v"(up), "v"(left), "v"(diag), "v"(x), "v"(mu));
        STAMP(1)
        float S; unsigned F;
        cell(diag, up, left, x, mu, fc, lp_step, lp_stay, S, F);
        if (km >= 0 && km + (DN_W - 1) < K && ev >= DN_W - 1 && ev < E) {
            S = inb ? S : NINF;                           // interior band: every cell 0..99 is inside the matrix
        } else {
            const unsigned e_u = (unsigned)ev - (unsigned)o;
            const bool ok = inb && ((unsigned)(km + o) < (unsigned)K) && (e_u < (unsigned)E);   // :269-278
            S = ok ? S : NINF; F = ok ? F : 0u;
            if (inb && km + o == -1 && e_u < (unsigned)E) { S = (float)(fc.lp_trim * (double)(e_u + 1u)); F = 1; }   // :256-265
        }
        asm volatile("" :: "v"(S), "v"(F));
        STAMP(2)
        Pb[sc][o + 1] = S;                                // lanes >= 100 rewrite -inf into the sentinel cells
        // ---- trace row: byte per cell + the band's lower-left event index in bytes 104..107 ----
        uint8_t tb = (uint8_t)F;
        if (tid >= 104 && tid < 108) tb = (uint8_t)(((unsigned)ev >> (8 * (tid - 104))) & 0xffu);
        rows[(size_t)b * DN_TROW + tid] = tb;
        // ---- end cell (:329-340): the lane that holds column K-1 keeps its best; reduced after the loop ----
        const int oe = K - 1 - km;
        if (oe >= 0 && oe < DN_W) {
            const int ee = ev - o;
            if (o == oe && ee >= 0 && ee < E) {
                const float sv = (float)((double)S + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                if (sv > best) { best = sv; best_e = ee; }
            }
        }
        // ---- ring refills (uniform, rare) ----
        if (ev + 130 >= xhi) { rx[(xhi + tid) & (F2_RING - 1)] = pendx; xhi += 128; pendx = ldx(xhi + tid); }
        if (km + 230 >= mhi) { rm[(mhi + tid) & (F2_RING - 1)] = pendm; mhi += 128; pendm = ldm(mhi + tid); }
        prev_right = right;
        STAMP(3)
        // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for this band's trace store to be acknowledged
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(4)
    }
    if (blockIdx.x == 0 && (tid == 0 || tid == 64))
        printf("fill2p wave %d: bands %d  cycles/band: lds-reads %.1f  move+select %.1f  cell+mask %.1f  write+store+endcell+refill %.1f  barrier %.1f  total %.1f\n",
               tid / 64, n_bands, (double)T[0] / n_bands, (double)T[1] / n_bands, (double)T[2] / n_bands, (double)T[3] / n_bands, (double)T[4] / n_bands,
               (double)(T[0] + T[1] + T[2] + T[3] + T[4]) / n_bands);
    // first maximum in increasing event order == maximum score, smallest event on ties
    red_s[tid] = best; red_e[tid] = best_e;
    __syncthreads();
    if (tid == 0) {
        float bs = NINF; int be = 0x7fffffff;
        for (int i = 0; i < DN_W; i++)
            if (red_e[i] != 0x7fffffff && (red_s[i] > bs || (red_s[i] == bs && red_e[i] < be))) { bs = red_s[i]; be = red_e[i]; }
        const int found = (be != 0x7fffffff) && (bs > NINF);
        R.n_bands = (unsigned)n_bands;
        R.end_event = found ? be : 0;
        R.end_score = bs;
        if (!found) R.status = 3;
    }
}

// ------------------------------------------------------------------------------------------------
// k2_fill3<NR>: the two-wavefront LDS design of k2_fill2 with NR reads interleaved per workgroup.
//
// One band cell is ONE long dependency chain (x-mu -> /sigma -> -a*a/2 -> +C -> three candidates -> max), and a
// lone wavefront issues a dependent instruction only every ~8 cycles (4 independent chains: ~5).  Giving every lane
// one cell of each of NR different reads puts NR independent chains into the same instruction stream, and the NR reads
// share the one barrier per band.  The code is written phase by phase (all LDS reads, all selects, all cells, all
// writes) and branch-free inside a phase so the scheduler can interleave the reads.
// ------------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(128) void k2_fill3(BatchDev B, const BandConsts *bc, FillConsts fc) {
    __shared__ float Pb[NR][3][F2_PW];
    __shared__ double rx[NR][F2_RING], rm[NR][F2_RING];
    __shared__ float red_s[NR][128]; __shared__ int red_e[NR][128];
    const int tid = threadIdx.x;
    const int o = tid;
    const bool inb = o < DN_W;
    const float NINF = neg_inf();
    int E[NR], K[NR], nb[NR], ev[NR], km[NR], prev_right[NR], xhi[NR], mhi[NR];
    double lp_stay[NR], lp_step[NR], x[NR], mu[NR], pendx[NR], pendm[NR];
    const double *xs[NR], *mus[NR];
    uint8_t *rows[NR];
    float best[NR]; int best_e[NR];
    int max_nb = 0;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int r = blockIdx.x * NR + q;
        const bool live = r < B.n_reads && B.res[r < B.n_reads ? r : 0].status == 0;
        const int rr = r < B.n_reads ? r : 0;
        E[q] = live ? (int)B.res[rr].n_events : 0; K[q] = live ? (int)B.res[rr].n_kq : 0;
        nb[q] = live ? E[q] + K[q] + 2 : 0;
        max_nb = max(max_nb, nb[q]);
        lp_stay[q] = bc[rr].lp_stay; lp_step[q] = bc[rr].lp_step;
        xs[q] = B.ev_x + B.ev_off[rr]; mus[q] = B.mu_q + B.base_off[rr];
        rows[q] = B.trace + B.trace_off[rr] * DN_TROW;
        best[q] = NINF; best_e[q] = 0x7fffffff; prev_right[q] = 0;
        ev[q] = 50; km[q] = -51; xhi[q] = 384; mhi[q] = 384;
    }
    if (max_nb == 0) return;
#define LDX(q, e) (((e) >= 0 && (e) < E[q]) ? xs[q][e] : 0.0)
#define LDM(q, k) (((k) >= 0 && (k) < K[q]) ? mus[q][k] : 0.0)
    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
#pragma unroll
    for (int q = 0; q < NR; q++) {
        for (int i = tid; i < 3 * F2_PW; i += 128) (&Pb[q][0][0])[i] = NINF;
        for (int i = tid; i < 384; i += 128) { rx[q][i] = LDX(q, i); rm[q][i] = LDM(q, i); }
        pendx[q] = LDX(q, xhi[q] + tid); pendm[q] = LDM(q, mhi[q] + tid);
        x[q] = LDX(q, ev[q] - o); mu[q] = LDM(q, km[q] + o);
        if (nb[q]) {
            uint8_t b0 = 0, b1 = (o == 50) ? 1 : 0;       // trace[1][50] = FROM_U
            if (tid >= 104 && tid < 108) { b0 = (uint8_t)((49u >> (8 * (tid - 104))) & 0xff); b1 = (uint8_t)((50u >> (8 * (tid - 104))) & 0xff); }
            rows[q][tid] = b0; rows[q][DN_TROW + tid] = b1;
        }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < NR; q++) { Pb[q][0][50 + 1] = 0.0f; Pb[q][1][50 + 1] = (float)fc.lp_trim; }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // retire the pre-loop global loads (see k2_fill2)

    for (int b = 2; b < max_nb; b++) {
        const int sp = (b + 2) % 3, sq = (b + 1) % 3, sc = b % 3;
        float lo[NR], hi[NR], pm1[NR], p0[NR], pp1[NR], qa[NR], qb[NR];
        double xn[NR], mn[NR];
        // ---- phase 1: one LDS round trip for everything either move can need ----
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const float *Pp = Pb[q][sp], *Qq = Pb[q][sq];
            lo[q] = Pp[1]; hi[q] = Pp[DN_W];
            pm1[q] = Pp[o]; p0[q] = Pp[o + 1]; pp1[q] = Pp[o + 2];
            qa[q] = Qq[o + prev_right[q]]; qb[q] = Qq[o + 1 + prev_right[q]];
            xn[q] = rx[q][(ev[q] + 1 - o) & (F2_RING - 1)];
            mn[q] = rm[q][(km[q] + 1 + o) & (F2_RING - 1)];
        }
        // ---- phase 2: Suzuki-Kasahara move (:237-253) and neighbour selection, branch-free ----
        float up[NR], left[NR], diag[NR];
        bool all_interior = true, any_end = false;
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const bool both_ob = (lo[q] == NINF) & (hi[q] == NINF);
            const bool vright = both_ob ? ((b & 1) != 0) : (lo[q] < hi[q]);
            up[q] = vright ? pp1[q] : p0[q]; left[q] = vright ? p0[q] : pm1[q]; diag[q] = vright ? qb[q] : qa[q];
            x[q] = vright ? x[q] : xn[q];
            mu[q] = vright ? mn[q] : mu[q];
            const int right = __builtin_amdgcn_readfirstlane((int)vright);
            ev[q] += 1 - right; km[q] += right; prev_right[q] = right;
            all_interior = all_interior && (km[q] >= 0 && km[q] + (DN_W - 1) < K[q] && ev[q] >= DN_W - 1 && ev[q] < E[q] && b < nb[q]);
            any_end = any_end || (K[q] - 1 - km[q] < DN_W);
        }
        // ---- phase 3: the cells ----
        float S[NR]; unsigned F[NR];
#pragma unroll
        for (int q = 0; q < NR; q++) cell(diag[q], up[q], left[q], x[q], mu[q], fc, lp_step[q], lp_stay[q], S[q], F[q]);
        if (all_interior) {
#pragma unroll
            for (int q = 0; q < NR; q++) S[q] = inb ? S[q] : NINF;
        } else {
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const unsigned e_u = (unsigned)ev[q] - (unsigned)o;
                const bool ok = inb && ((unsigned)(km[q] + o) < (unsigned)K[q]) && (e_u < (unsigned)E[q]);   // :269-278
                S[q] = ok ? S[q] : NINF; F[q] = ok ? F[q] : 0u;
                if (inb && km[q] + o == -1 && e_u < (unsigned)E[q]) { S[q] = (float)(fc.lp_trim * (double)(e_u + 1u)); F[q] = 1; }   // :256-265
            }
        }
        // ---- phase 4: scores to LDS, trace row to HBM ----
#pragma unroll
        for (int q = 0; q < NR; q++) {
            Pb[q][sc][o + 1] = S[q];
            uint8_t tb = (uint8_t)F[q];
            if (tid >= 104 && tid < 108) tb = (uint8_t)(((unsigned)ev[q] >> (8 * (tid - 104))) & 0xffu);
            if (b < nb[q]) rows[q][(size_t)b * DN_TROW + tid] = tb;
        }
        // ---- end cell (:329-340) and ring refills: uniform, rare ----
        if (any_end) {
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const int oe = K[q] - 1 - km[q];
                const int ee = ev[q] - o;
                if (b < nb[q] && oe >= 0 && o == oe && ee >= 0 && ee < E[q]) {
                    const float sv = (float)((double)S[q] + (double)(unsigned long long)(E[q] - ee) * fc.lp_trim);
                    if (sv > best[q]) { best[q] = sv; best_e[q] = ee; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NR; q++) {
            if (ev[q] + 130 >= xhi[q]) { rx[q][(xhi[q] + tid) & (F2_RING - 1)] = pendx[q]; xhi[q] += 128; pendx[q] = LDX(q, xhi[q] + tid); }
            if (km[q] + 230 >= mhi[q]) { rm[q][(mhi[q] + tid) & (F2_RING - 1)] = pendm[q]; mhi[q] += 128; pendm[q] = LDM(q, mhi[q] + tid); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier (no vmcnt: stores stay in flight)
    }
#undef LDX
#undef LDM
#pragma unroll
    for (int q = 0; q < NR; q++) { red_s[q][tid] = best[q]; red_e[q][tid] = best_e[q]; }
    __syncthreads();
    if (tid < NR) {
        const int q = tid;
        const int r = blockIdx.x * NR + q;
        bool live = false; int nbq = 0;
#pragma unroll
        for (int z = 0; z < NR; z++) if (z == q) { live = nb[z] > 0; nbq = nb[z]; }
        if (live) {
            float bs = NINF; int be = 0x7fffffff;
            for (int i = 0; i < DN_W; i++) {
                const float si = red_s[q][i]; const int ei = red_e[q][i];
                if (ei != 0x7fffffff && (si > bs || (si == bs && ei < be))) { bs = si; be = ei; }   // first maximum in event order
            }
            const int found = (be != 0x7fffffff) && (bs > NINF);
            ReadRes &R = B.res[r];
            R.n_bands = (unsigned)nbq;
            R.end_event = found ? be : 0;
            R.end_score = bs;
            if (!found) R.status = 3;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k2_fill4: the single-wavefront register/DPP design (k2_fill) rewritten around the measured costs of a lone
// wavefront on a SIMD: ~8 cycles per VALU instruction, and ~40+ cycles for every VALU->SALU hop or VALU-dependent
// branch (profiles/r01_valu_issue_microbench.txt, stamped phases of k2_fill2p):
//   * the Suzuki move is decided on the SCALAR unit with integer arithmetic on the two broadcast edge scores
//     (float order == signed order of sign-magnitude keys), so the loop has one VALU->SALU hop (two v_readlane) and
//     no VALU-dependent branch;
//   * every rare condition (band touching a matrix edge, trim column, end column) is one merged scalar test and one
//     branch to a general slow path; the fast path has no other branch than the move itself;
//   * idle lanes are kept at -inf with a single select (only cell 100 can leak into the band).
// ------------------------------------------------------------------------------------------------
template <bool DPP>
__global__ __launch_bounds__(64) void k2_fill4(BatchDev B, const BandConsts *bc, FillConsts fc) {
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
    const unsigned o0 = inb ? (unsigned)(2 * lane) : 0x40000000u, o1 = inb ? (unsigned)(2 * lane + 1) : 0x40000000u;
    const unsigned meta_shift = (lane == 53) ? 16u : 0u;
    const unsigned meta_mask = (lane == 52 || lane == 53) ? 0xffffu : 0u;

    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    int ev = 50, km = -51;
    float Q0 = (o0 == 50u) ? 0.0f : NINF, Q1 = NINF;
    float P0 = (o0 == 50u) ? (float)fc.lp_trim : NINF, P1 = NINF;
    {
        const unsigned short w1 = (o0 == 50u) ? 1u : 0u;
        reinterpret_cast<unsigned short *>(rows)[lane] = inb ? (unsigned short)0 : (unsigned short)((49u >> meta_shift) & meta_mask);
        reinterpret_cast<unsigned short *>(rows + DN_TROW)[lane] = inb ? w1 : (unsigned short)((50u >> meta_shift) & meta_mask);
    }
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };
    double X0 = ldx(ev - (int)o0), X1 = ldx(ev - (int)o1);
    double M0 = ldm(km + (int)o0), M1 = ldm(km + (int)o1);
    double nx = xs_c[min(ev + 1, E - 1)];
    double nm = mu_c[max(min(km + 100, K - 1), 0)];
    float best = NINF; int best_e = 0; int found = 0;
    int prev_right = 0;
    unsigned short *wrow = reinterpret_cast<unsigned short *>(rows) + lane;      // + 64 shorts per band
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // retire the pre-loop vector loads once (not per band)

    for (int b = 2; b < n_bands; b++) {
        // ---- Suzuki-Kasahara move (:237-253), on the scalar unit ----
        const int lo_i = __builtin_amdgcn_readlane(__float_as_int(P0), 0);
        const int hi_i = __builtin_amdgcn_readlane(__float_as_int(P1), 49);
        const int klo = lo_i ^ ((lo_i >> 31) & 0x7fffffff), khi = hi_i ^ ((hi_i >> 31) & 0x7fffffff);
        int right = klo < khi;                                            // lo < hi for non-NaN floats ...
        if ((((unsigned)lo_i | (unsigned)hi_i) & 0x7fffffffu) == 0u) right = 0;   // ... except -0 vs +0, which compare equal
        if (lo_i == (int)0xff800000 && hi_i == (int)0xff800000) right = b & 1;   // both edges out of band: alternate

        float up0, up1, lf0, lf1, dg0, dg1;
        if (right) {
            km += 1;
            const double t = from_next_d<DPP>(M0, lane);
            M0 = M1; M1 = (lane == 49) ? nm : t;
            const float nq = from_next<DPP>(Q0, NINF, lane);
            up0 = P1; lf0 = P0; up1 = from_next<DPP>(P0, NINF, lane); lf1 = P1;
            dg0 = prev_right ? Q1 : Q0; dg1 = prev_right ? nq : Q1;
        } else {
            ev += 1;
            const double t = from_prev_d<DPP>(X1, lane);
            X1 = X0; X0 = (lane == 0) ? nx : t;
            const float pq = from_prev<DPP>(Q1, NINF, lane);
            up0 = P0; lf0 = from_prev<DPP>(P1, NINF, lane); up1 = P1; lf1 = P0;
            dg0 = prev_right ? Q0 : pq; dg1 = prev_right ? Q1 : Q0;
        }
        nx = xs_c[min(ev + 1, E - 1)];                    // scalar prefetch for the next band
        nm = mu_c[max(min(km + 100, K - 1), 0)];

        float S0, S1; unsigned F0, F1;
        cell(dg0, up0, lf0, X0, M0, fc, lp_step, lp_stay, S0, F0);
        cell(dg1, up1, lf1, X1, M1, fc, lp_step, lp_stay, S1, F1);
        S0 = inb ? S0 : NINF;                             // cell 100 (lane 50) is the only idle cell a band cell can read
        if (__builtin_expect(!(km >= 0 && km + DN_W < K && ev >= DN_W - 1 && ev < E), 0)) {
            // ---- slow path: the band touches a matrix edge, the trim column or the end column ----
            const unsigned e0 = (unsigned)ev - o0, e1 = (unsigned)ev - o1;
            const bool ok0 = ((unsigned)km + o0 < (unsigned)K) && (e0 < (unsigned)E);    // :269-278
            const bool ok1 = ((unsigned)km + o1 < (unsigned)K) && (e1 < (unsigned)E);
            S0 = ok0 ? S0 : NINF; F0 = ok0 ? F0 : 0u;
            S1 = ok1 ? S1 : NINF; F1 = ok1 ? F1 : 0u;
            if (km <= -1) {                               // trim column kmer == -1 (:256-265)
                const bool t0 = ((unsigned)km + o0 == 0xffffffffu) && (e0 < (unsigned)E);
                const bool t1 = ((unsigned)km + o1 == 0xffffffffu) && (e1 < (unsigned)E);
                if (t0) { S0 = (float)(fc.lp_trim * (double)(e0 + 1u)); F0 = 1; }
                if (t1) { S1 = (float)(fc.lp_trim * (double)(e1 + 1u)); F1 = 1; }
            }
            const int oe = K - 1 - km;                    // end column (:329-340)
            if (oe >= 0 && oe < DN_W) {
                const int ee = ev - oe;
                if (ee >= 0 && ee < E) {
                    const float sv = bcast_f((oe & 1) ? S1 : S0, oe >> 1);
                    const float sc = (float)((double)sv + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                    if (sc > best) { best = sc; best_e = ee; found = 1; }
                }
            }
        }
        // ---- one 128-byte row: 100 trace bytes (lanes 0..49) + the band's lower-left event index (lanes 52, 53) ----
        const unsigned meta = ((unsigned)ev >> meta_shift) & meta_mask;
        wrow += DN_TROW / 2;
        *wrow = (unsigned short)(inb ? (F0 | (F1 << 8)) : meta);
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

// diagnostic build of k2_fill4 with s_memtime stamps (DN_FILL_VARIANT=44); never used for results or timing
__global__ __launch_bounds__(64) void k2_fill4p(BatchDev B, const BandConsts *bc, FillConsts fc) {
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
    const unsigned o0 = inb ? (unsigned)(2 * lane) : 0x40000000u, o1 = inb ? (unsigned)(2 * lane + 1) : 0x40000000u;
    const unsigned meta_shift = (lane == 53) ? 16u : 0u;
    const unsigned meta_mask = (lane == 52 || lane == 53) ? 0xffffu : 0u;

    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    int ev = 50, km = -51;
    float Q0 = (o0 == 50u) ? 0.0f : NINF, Q1 = NINF;
    float P0 = (o0 == 50u) ? (float)fc.lp_trim : NINF, P1 = NINF;
    {
        const unsigned short w1 = (o0 == 50u) ? 1u : 0u;
        reinterpret_cast<unsigned short *>(rows)[lane] = inb ? (unsigned short)0 : (unsigned short)((49u >> meta_shift) & meta_mask);
        reinterpret_cast<unsigned short *>(rows + DN_TROW)[lane] = inb ? w1 : (unsigned short)((50u >> meta_shift) & meta_mask);
    }
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };
    double X0 = ldx(ev - (int)o0), X1 = ldx(ev - (int)o1);
    double M0 = ldm(km + (int)o0), M1 = ldm(km + (int)o1);
    double nx = xs_c[min(ev + 1, E - 1)];
    double nm = mu_c[max(min(km + 100, K - 1), 0)];
    float best = NINF; int best_e = 0; int found = 0;
    int prev_right = 0;
    unsigned short *wrow = reinterpret_cast<unsigned short *>(rows) + lane;      // + 64 shorts per band
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // retire the pre-loop vector loads once (not per band)

    unsigned long long T[6] = {0,0,0,0,0,0};
#define STAMP4(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); T[i] += t_ - tl; tl = t_; }
    unsigned long long tl; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl) :: "memory");
    for (int b = 2; b < n_bands; b++) {
        // ---- Suzuki-Kasahara move (:237-253), on the scalar unit ----
        const int lo_i = __builtin_amdgcn_readlane(__float_as_int(P0), 0);
        const int hi_i = __builtin_amdgcn_readlane(__float_as_int(P1), 49);
        const int klo = lo_i ^ ((lo_i >> 31) & 0x7fffffff), khi = hi_i ^ ((hi_i >> 31) & 0x7fffffff);
        int right = klo < khi;                                            // lo < hi for non-NaN floats ...
        if ((((unsigned)lo_i | (unsigned)hi_i) & 0x7fffffffu) == 0u) right = 0;   // ... except -0 vs +0, which compare equal
        if (lo_i == (int)0xff800000 && hi_i == (int)0xff800000) right = b & 1;   // both edges out of band: alternate

        asm volatile("" :: "s"(right));
        STAMP4(0)
        float up0, up1, lf0, lf1, dg0, dg1;
        if (right) {
            km += 1;
            const double t = from_next_d<true>(M0, lane);
            M0 = M1; M1 = (lane == 49) ? nm : t;
            const float nq = from_next<true>(Q0, NINF, lane);
            up0 = P1; lf0 = P0; up1 = from_next<true>(P0, NINF, lane); lf1 = P1;
            dg0 = prev_right ? Q1 : Q0; dg1 = prev_right ? nq : Q1;
        } else {
            ev += 1;
            const double t = from_prev_d<true>(X1, lane);
            X1 = X0; X0 = (lane == 0) ? nx : t;
            const float pq = from_prev<true>(Q1, NINF, lane);
            up0 = P0; lf0 = from_prev<true>(P1, NINF, lane); up1 = P1; lf1 = P0;
            dg0 = prev_right ? Q0 : pq; dg1 = prev_right ? Q1 : Q0;
        }
        asm volatile("" :: "v"(up0), "v"(up1), "v"(lf0), "v"(lf1), "v"(dg0), "v"(dg1), "v"(X0), "v"(X1), "v"(M0), "v"(M1));
        STAMP4(1)
        nx = xs_c[min(ev + 1, E - 1)];                    // scalar prefetch for the next band
        nm = mu_c[max(min(km + 100, K - 1), 0)];

        float S0, S1; unsigned F0, F1;
        cell(dg0, up0, lf0, X0, M0, fc, lp_step, lp_stay, S0, F0);
        cell(dg1, up1, lf1, X1, M1, fc, lp_step, lp_stay, S1, F1);
        asm volatile("" :: "v"(S0), "v"(S1), "v"(F0), "v"(F1));
        STAMP4(2)
        S0 = inb ? S0 : NINF;                             // cell 100 (lane 50) is the only idle cell a band cell can read
        if (__builtin_expect(!(km >= 0 && km + DN_W < K && ev >= DN_W - 1 && ev < E), 0)) {
            // ---- slow path: the band touches a matrix edge, the trim column or the end column ----
            const unsigned e0 = (unsigned)ev - o0, e1 = (unsigned)ev - o1;
            const bool ok0 = ((unsigned)km + o0 < (unsigned)K) && (e0 < (unsigned)E);    // :269-278
            const bool ok1 = ((unsigned)km + o1 < (unsigned)K) && (e1 < (unsigned)E);
            S0 = ok0 ? S0 : NINF; F0 = ok0 ? F0 : 0u;
            S1 = ok1 ? S1 : NINF; F1 = ok1 ? F1 : 0u;
            if (km <= -1) {                               // trim column kmer == -1 (:256-265)
                const bool t0 = ((unsigned)km + o0 == 0xffffffffu) && (e0 < (unsigned)E);
                const bool t1 = ((unsigned)km + o1 == 0xffffffffu) && (e1 < (unsigned)E);
                if (t0) { S0 = (float)(fc.lp_trim * (double)(e0 + 1u)); F0 = 1; }
                if (t1) { S1 = (float)(fc.lp_trim * (double)(e1 + 1u)); F1 = 1; }
            }
            const int oe = K - 1 - km;                    // end column (:329-340)
            if (oe >= 0 && oe < DN_W) {
                const int ee = ev - oe;
                if (ee >= 0 && ee < E) {
                    const float sv = bcast_f((oe & 1) ? S1 : S0, oe >> 1);
                    const float sc = (float)((double)sv + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                    if (sc > best) { best = sc; best_e = ee; found = 1; }
                }
            }
        }
        // ---- one 128-byte row: 100 trace bytes (lanes 0..49) + the band's lower-left event index (lanes 52, 53) ----
        const unsigned meta = ((unsigned)ev >> meta_shift) & meta_mask;
        wrow += DN_TROW / 2;
        *wrow = (unsigned short)(inb ? (F0 | (F1 << 8)) : meta);
        Q0 = P0; Q1 = P1; P0 = S0; P1 = S1;
        prev_right = right;
        STAMP4(3)
    }
    if (blockIdx.x == 0 && lane == 0)
        printf("fill4p: bands %d cycles/band: decide %.1f  shift %.1f  sload+cells %.1f  mask+slow+store %.1f  total %.1f\n", n_bands,
               (double)T[0] / n_bands, (double)T[1] / n_bands, (double)T[2] / n_bands, (double)T[3] / n_bands, (double)(T[0] + T[1] + T[2] + T[3]) / n_bands);
    if (lane == 0) {
        R.n_bands = (unsigned)n_bands;
        R.end_event = best_e;
        R.end_score = best;
        if (!found) R.status = 3;
    }
}

// ------------------------------------------------------------------------------------------------
// k2_fill5: single wavefront per read, 2 cells per lane, written for what the stamped builds measured on a lone
// wavefront: streamed VALU arithmetic costs ~3.5 cycles/instruction, but every scalar dependency chain, VALU->SALU hop
// and taken branch costs tens of cycles (k2_fill4p: 54 cell instructions 187 cycles, the 10-instruction move branch 400).
//   * the Suzuki move is a VALU mask; BOTH moves' neighbours / shifted x / shifted mu are formed with DPP and picked with
//     v_cndmask: no branch on the move at all;
//   * bands are processed in runs: the number of following bands that cannot touch a matrix edge, the trim column or the
//     end column is computed once (each band moves the corner by exactly one), and that run executes a loop body with no
//     edge test; only the first/last ~100 bands of a read take the general body.
// ------------------------------------------------------------------------------------------------
struct F5State {
    float P0, P1, Q0, Q1;
    double X0, X1, M0, M1;
    int km;                  // lower-left kmer index of the last band; event index ev = b - 2 - km
    unsigned long long pr;   // lane mask: previous move was "right"
};

template <bool FAST, int ABL>
__device__ __forceinline__ void f5_band(F5State &st, const int b, const int E, const int K, const int lane, const bool inb,
                                        const unsigned o0, const unsigned o1, const unsigned meta_shift, const unsigned meta_mask,
                                        const cdptr_t xs_c, const cdptr_t mu_c, double &nx, double &nm, const FillConsts &fc,
                                        const double lp_step, const double lp_stay, unsigned short *rows16, float &best, int &best_e,
                                        int &found) {
    const float NINF = neg_inf();
    // ---- Suzuki-Kasahara move (:237-253) as a lane mask ----
    const float lo = bcast_f(st.P0, 0), hi = bcast_f(st.P1, 49);
    const bool ob = fmaxf(lo, hi) == NINF;                 // both edge cells out of band
    const bool vright = (ABL & 2) ? ((b & 1) != 0) : (ob ? ((b & 1) != 0) : (lo < hi));
    const unsigned long long r = __ballot(vright);         // all-ones or zero (uniform)
    const bool R = r != 0ull, PR = st.pr != 0ull;
    // ---- neighbours for both moves, then select ----
    const float nP0 = from_next<true>(st.P0, NINF, lane), pP1 = from_prev<true>(st.P1, NINF, lane);
    const float nQ0 = from_next<true>(st.Q0, NINF, lane), pQ1 = from_prev<true>(st.Q1, NINF, lane);
    const float up0 = vright ? st.P1 : st.P0, lf0 = vright ? st.P0 : pP1;
    const float up1 = vright ? nP0 : st.P1,   lf1 = vright ? st.P1 : st.P0;
    const float dA0 = PR ? st.Q1 : st.Q0, dA1 = PR ? nQ0 : st.Q1;        // diagonal if this move is "right"
    const float dB0 = PR ? st.Q0 : pQ1,   dB1 = PR ? st.Q1 : st.Q0;      // ... if it is "down"
    float dg0 = vright ? dA0 : dB0, dg1 = vright ? dA1 : dB1;
    float up0_ = up0, up1_ = up1, lf0_ = lf0, lf1_ = lf1;
    if (ABL & 8) { dg0 = st.Q0; dg1 = st.Q1; up0_ = st.P0; up1_ = st.P1; lf0_ = st.P1; lf1_ = st.P0; }
    // ---- x moves one cell on "down", mu on "right"; the entering value was prefetched by the scalar unit ----
    double tX = from_prev_d<true>(st.X1, lane); tX = (lane == 0) ? nx : tX;
    double tM = from_next_d<true>(st.M0, lane); tM = (lane == 49) ? nm : tM;
    const double X0 = (ABL & 1) ? st.X0 : (vright ? st.X0 : tX), X1 = (ABL & 1) ? st.X1 : (vright ? st.X1 : st.X0);
    const double M0 = (ABL & 1) ? st.M0 : (vright ? st.M1 : st.M0), M1 = (ABL & 1) ? st.M1 : (vright ? tM : st.M1);
    st.X0 = X0; st.X1 = X1; st.M0 = M0; st.M1 = M1;
    const int km = st.km + (R ? 1 : 0);
    const int ev = b - 2 - km;
    st.km = km; st.pr = r;
    // scalar prefetch of the values entering the NEXT band (FAST: indices are in range by construction)
    nx = xs_c[FAST ? ev + 1 : min(ev + 1, E - 1)];
    nm = mu_c[FAST ? km + 100 : max(min(km + 100, K - 1), 0)];

    float S0, S1; unsigned F0, F1;
    if (ABL & 16) { S0 = up0_ + dg0 + (float)X0; S1 = up1_ + dg1 + (float)M1; F0 = lf0_ > S0; F1 = lf1_ > S1; }
    else {
        cell(dg0, up0_, lf0_, X0, M0, fc, lp_step, lp_stay, S0, F0);
        cell(dg1, up1_, lf1_, X1, M1, fc, lp_step, lp_stay, S1, F1);
    }
    S0 = inb ? S0 : NINF;                                  // cell 100 (lane 50) is the only idle cell a band cell can read
    if (!FAST) {
        const unsigned e0 = (unsigned)ev - o0, e1 = (unsigned)ev - o1;
        const bool ok0 = ((unsigned)km + o0 < (unsigned)K) && (e0 < (unsigned)E);    // :269-278
        const bool ok1 = ((unsigned)km + o1 < (unsigned)K) && (e1 < (unsigned)E);
        S0 = ok0 ? S0 : NINF; F0 = ok0 ? F0 : 0u;
        S1 = ok1 ? S1 : NINF; F1 = ok1 ? F1 : 0u;
        if (km <= -1) {                                    // trim column kmer == -1 (:256-265)
            const bool t0 = ((unsigned)km + o0 == 0xffffffffu) && (e0 < (unsigned)E);
            const bool t1 = ((unsigned)km + o1 == 0xffffffffu) && (e1 < (unsigned)E);
            if (t0) { S0 = (float)(fc.lp_trim * (double)(e0 + 1u)); F0 = 1; }
            if (t1) { S1 = (float)(fc.lp_trim * (double)(e1 + 1u)); F1 = 1; }
        }
        const int oe = K - 1 - km;                         // end column (:329-340)
        if (oe >= 0 && oe < DN_W) {
            const int ee = ev - oe;
            if (ee >= 0 && ee < E) {
                const float sv = bcast_f((oe & 1) ? S1 : S0, oe >> 1);
                const float sc = (float)((double)sv + (double)(unsigned long long)(E - ee) * fc.lp_trim);
                if (sc > best) { best = sc; best_e = ee; found = 1; }
            }
        }
    }
    // ---- one 128-byte row: 100 trace bytes (lanes 0..49) + the band's lower-left event index (lanes 52, 53) ----
    const unsigned meta = ((unsigned)ev >> meta_shift) & meta_mask;
    if (!(ABL & 4)) rows16[(size_t)b * (DN_TROW / 2) + lane] = (unsigned short)(inb ? (F0 | (F1 << 8)) : meta);
    else asm volatile("" :: "v"(F0), "v"(F1), "v"(meta));
    st.Q0 = st.P0; st.Q1 = st.P1; st.P0 = S0; st.P1 = S1;
}

template <int ABL>
__global__ __launch_bounds__(64) void k2_fill5(BatchDev B, const BandConsts *bc, FillConsts fc) {
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
    unsigned short *rows16 = reinterpret_cast<unsigned short *>(rows);
    const float NINF = neg_inf();
    const bool inb = lane < 50;
    const unsigned o0 = inb ? (unsigned)(2 * lane) : 0x40000000u, o1 = inb ? (unsigned)(2 * lane + 1) : 0x40000000u;
    const unsigned meta_shift = (lane == 53) ? 16u : 0u;
    const unsigned meta_mask = (lane == 52 || lane == 53) ? 0xffffu : 0u;
    // ---- bands 0 and 1 (event_handling.cpp:213-228) ----
    F5State st;
    st.Q0 = (o0 == 50u) ? 0.0f : NINF; st.Q1 = NINF;
    st.P0 = (o0 == 50u) ? (float)fc.lp_trim : NINF; st.P1 = NINF;
    {
        const unsigned short w1 = (o0 == 50u) ? 1u : 0u;
        rows16[lane] = inb ? (unsigned short)0 : (unsigned short)((49u >> meta_shift) & meta_mask);
        rows16[DN_TROW / 2 + lane] = inb ? w1 : (unsigned short)((50u >> meta_shift) & meta_mask);
    }
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };
    const int ev1 = 50, km1 = -51;
    st.X0 = ldx(ev1 - (int)o0); st.X1 = ldx(ev1 - (int)o1);
    st.M0 = ldm(km1 + (int)o0); st.M1 = ldm(km1 + (int)o1);
    st.km = km1; st.pr = 0ull;
    double nx = xs_c[min(ev1 + 1, E - 1)];
    double nm = mu_c[max(min(km1 + 100, K - 1), 0)];
    float best = NINF; int best_e = 0; int found = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // retire the pre-loop vector loads once (not per band)

    int b = 2;
    while (b < n_bands) {
        const int km = st.km, ev = b - 3 - km;             // corner of band b-1
        // after j more moves: kmer corner <= km + j, event corner <= ev + j.  A band is "fast" when, after its move,
        // 0 <= km', km' + 100 < K (no end column, mu prefetch in range), 99 <= ev' < E - 1 (x prefetch in range).
        int run = 0;
        if (km >= 0 && ev >= DN_W - 1) run = min(K - 101 - km, E - 2 - ev);
        run = min(run, n_bands - b);
        if (run > 0) {
            const int bend = b + run;
            for (; b < bend; b++)
                f5_band<true, ABL>(st, b, E, K, lane, inb, o0, o1, meta_shift, meta_mask, xs_c, mu_c, nx, nm, fc, lp_step, lp_stay, rows16, best, best_e, found);
        } else {
            f5_band<false, ABL>(st, b, E, K, lane, inb, o0, o1, meta_shift, meta_mask, xs_c, mu_c, nx, nm, fc, lp_step, lp_stay, rows16, best, best_e, found);
            b++;
        }
    }
    if (lane == 0) {
        R.n_bands = (unsigned)n_bands;
        R.end_event = best_e;
        R.end_score = best;
        if (!found) R.status = 3;
    }
}

// ------------------------------------------------------------------------------------------------
// k2_fill6: event-keyed slots.  The 100 cells of a band are the events [ev - 99, ev]; cell (event e, kmer k = b - 2 - e)
// lives in slot e & 127 = lane (e & 127) >> 1, register e & 1 (A: even events, B: odd events) for as long as event e is in
// the band.  Consequences, all of which remove per-band selects that k2_fill5 needs because its cells are keyed by the
// band offset:
//   * the scaled event level x is STATIONARY in its slot; a new event is written once (v_writelane) when it enters;
//   * "left" (e, k - 1) is the SAME slot of the previous band; "up" (e - 1, k) is the previous slot of the previous band:
//     upB = PA, upA = wave_ror(PB), one DPP move, independent of the band move; "diag" is the previous band's up operand;
//   * the k-mer level mu rotates by one slot EVERY band (k grows by one per band for a fixed event), again independent of
//     the move: MB' = MA, MA' = wave_ror(MB); the one level that enters the window is written by v_writelane;
//   * the Suzuki move only changes WHICH slots are in the band: two uniform 64-bit masks (a 50-lane cyclic run each),
//     built on the scalar unit, applied with one v_cndmask per register; the decision itself is integer arithmetic on the
//     two end scores read with v_readlane.
// Trace row b (128 B): byte s = from-code of the cell in slot s, 0xFF for the 28 slots outside the band.  The backtrack
// indexes it with e & 127 and needs no per-band corner; a path that steps out of the band reads 0xFF.
// ------------------------------------------------------------------------------------------------
// every lane has a source under a rotate, so the "old" operand is dead: bound_ctrl lets the compiler drop its initialisation
__device__ __forceinline__ float ror_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, true));
}
__device__ __forceinline__ double ror_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x13C, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x13C, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double writelane_d(double v, double nv, int lane_sel) {      // nv and lane_sel are wave-uniform
    const long long b = __double_as_longlong(v), n = __double_as_longlong(nv);
    const int lo = writelane_((int)(b & 0xffffffffll), (int)(n & 0xffffffffll), lane_sel);
    const int hi = writelane_((int)(b >> 32), (int)(n >> 32), lane_sel);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ unsigned long long rotl64_(unsigned long long m, unsigned n) {   // n in 0..63, uniform
    return (m << n) | ((m >> 1) >> (63u - n));
}
__device__ __forceinline__ int ordered_f32_bits(int x) { return x ^ ((x >> 31) & 0x7fffffff); }   // a < b as floats <=> as these ints

// both registers of one lane at once: A[lane_sel] = va, B[lane_sel] = vb (four v_writelane under one M0)
__device__ __forceinline__ void writelane_pair_d(double &A, double &Bv, double va, double vb, int lane_sel) {
    const long long a = __double_as_longlong(A), b = __double_as_longlong(Bv);
    const long long na = __double_as_longlong(va), nb = __double_as_longlong(vb);
    int alo = (int)(a & 0xffffffffll), ahi = (int)(a >> 32), blo = (int)(b & 0xffffffffll), bhi = (int)(b >> 32);
    asm volatile("s_mov_b32 m0, %8\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\tv_writelane_b32 %3, %7, m0"
                 : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi)
                 : "s"((int)(na & 0xffffffffll)), "s"((int)(na >> 32)), "s"((int)(nb & 0xffffffffll)), "s"((int)(nb >> 32)), "s"(lane_sel)
                 : "m0");
    A = __longlong_as_double(((long long)ahi << 32) | (unsigned)alo);
    Bv = __longlong_as_double(((long long)bhi << 32) | (unsigned)blo);
}

struct F6State {
    float PA, PB, DA, DB;
    double XA, XB, MA, MB;
    int ev, km;              // lower-left corner of the last band
};
struct F6In { double x0, x1, m0, m1; };   // prefetched for the next band: x of an event pair (2j, 2j + 1), mu of kmers (kA - 1, kA)

template <bool FAST>
__device__ __forceinline__ void f6_prefetch(F6In &in, int b_next, int ev, int E, int K, const cdptr_t xs_c, const cdptr_t mu_c) {
    // band b_next writes the event pair that holds event ev + 1 and the kmer levels of the pair that holds event ev - 99
    const int p = (ev + 1) & ~1;
    const int q = (ev - (DN_W - 1)) & ~1;                  // two's complement: rounds down for negative events too
    const int kA = b_next - 2 - q;                         // kmer of the even event of that pair in band b_next; the odd one has kA - 1
    if (FAST) {
        // indices are in range and non-negative here: one 16-byte scalar load per pair, 32-bit byte offsets (s_load ... soffset)
        typedef double d2_t __attribute__((ext_vector_type(2)));
        typedef const d2_t __attribute__((address_space(4), aligned(8))) *cd2ptr_t;
        typedef const char __attribute__((address_space(4))) *ccptr_t;
        const d2_t xv = *(cd2ptr_t)((ccptr_t)xs_c + (unsigned)(p << 3));
        const d2_t mv = *(cd2ptr_t)((ccptr_t)mu_c + (unsigned)((kA - 1) << 3));
        in.x0 = xv[0]; in.x1 = xv[1]; in.m0 = mv[0]; in.m1 = mv[1];
    }
    else {
        in.x0 = xs_c[max(min(p, E - 1), 0)]; in.x1 = xs_c[max(min(p + 1, E - 1), 0)];
        in.m0 = mu_c[max(min(kA - 1, K - 1), 0)]; in.m1 = mu_c[max(min(kA, K - 1), 0)];
    }
}

template <bool FAST>
__device__ __forceinline__ void f6_band(F6State &st, const int b, const int E, const int K, const int lane2, const cdptr_t xs_c,
                                        const cdptr_t mu_c, F6In &in, const FillConsts &fc, const double lp_step,
                                        const double lp_stay, unsigned short *rows16, float &best, int &best_e, int &found) {
    const int NINF_BITS = (int)0xff800000;
    const float NINF = neg_inf();
    // ---- Suzuki-Kasahara move (:237-253) from the end cells of the previous band: events ev (lower left) and ev - 99 ----
    const int ev0 = st.ev, el0 = ev0 - (DN_W - 1);
    const int l_lo = (ev0 & 127) >> 1, l_hi = (el0 & 127) >> 1;
    const int loA = __builtin_amdgcn_readlane(__float_as_int(st.PA), l_lo), loB = __builtin_amdgcn_readlane(__float_as_int(st.PB), l_lo);
    const int hiA = __builtin_amdgcn_readlane(__float_as_int(st.PA), l_hi), hiB = __builtin_amdgcn_readlane(__float_as_int(st.PB), l_hi);
    const int lo = (ev0 & 1) ? loB : loA, hi = (el0 & 1) ? hiB : hiA;
    // integer 0/1 arithmetic (scalar unit, no branch): both end cells out of band -> alternate by parity, else ll < ur
    const int ol = ordered_f32_bits(lo), oh = ordered_f32_bits(hi);
    const int lt = (ol < oh) ? 1 : 0;
    const int right = (max(ol, oh) == ordered_f32_bits(NINF_BITS)) ? (b & 1) : lt;
    // ---- entering values, written pair-wise (both registers of a lane under one M0), independent of the move and
    //      idempotent: the event pair that holds event ev0 + 1; after the rotation, the kmer levels of the pair that
    //      holds event ev0 - 99.  Whatever is not needed yet lies outside the band or already has exactly that value. ----
    writelane_pair_d(st.XA, st.XB, in.x0, in.x1, ((ev0 + 1) & 127) >> 1);
    {
        const double nMA = ror_d(st.MB);                   // event 2l now has the kmer event 2l - 1 had
        st.MB = st.MA;                                     // event 2l + 1 the kmer event 2l had
        st.MA = nMA;
    }
    writelane_pair_d(st.MA, st.MB, in.m1, in.m0, l_hi);
    const int km = st.km + right, ev = ev0 + (right ^ 1);
    st.km = km; st.ev = ev;
    f6_prefetch<FAST>(in, b + 1, ev, E, K, xs_c, mu_c);
    // ---- in-band slots: ((ev - event) & 127) < 100 ----
    const unsigned tA = (unsigned)(ev - lane2) & 127u;
    const bool actA = tA < (unsigned)DN_W, actB = (tA - 1u) < (unsigned)DN_W;
    // ---- operands: left = same slot, up = previous slot, diag = the previous band's up ----
    const float upA = ror_f(st.PB), upB = st.PA;
    float SA, SB; unsigned FA, FB;
    cell(st.DA, upA, st.PA, st.XA, st.MA, fc, lp_step, lp_stay, SA, FA);
    cell(st.DB, upB, st.PB, st.XB, st.MB, fc, lp_step, lp_stay, SB, FB);
    if (FAST) {
        SA = actA ? SA : NINF; SB = actB ? SB : NINF;
    } else {
        const int eA = ev - (int)tA, eB = ev - (int)((tA - 1u) & 127u);
        const int kA = b - 2 - eA, kB = b - 2 - eB;
        const bool okA = actA && (unsigned)kA < (unsigned)K && (unsigned)eA < (unsigned)E;       // :269-278
        const bool okB = actB && (unsigned)kB < (unsigned)K && (unsigned)eB < (unsigned)E;
        SA = okA ? SA : NINF; FA = okA ? FA : 0u;
        SB = okB ? SB : NINF; FB = okB ? FB : 0u;
        if (km <= -1) {                                    // trim column kmer == -1 (:256-265)
            if (actA && kA == -1 && (unsigned)eA < (unsigned)E) { SA = (float)(fc.lp_trim * (double)((unsigned)eA + 1u)); FA = 1; }
            if (actB && kB == -1 && (unsigned)eB < (unsigned)E) { SB = (float)(fc.lp_trim * (double)((unsigned)eB + 1u)); FB = 1; }
        }
        const int ee = b - 2 - (K - 1);                    // end column kmer == K - 1 (:329-340)
        if (ee <= ev && ee > ev - DN_W && ee >= 0 && ee < E) {
            const int svA = __builtin_amdgcn_readlane(__float_as_int(SA), (ee & 127) >> 1), svB = __builtin_amdgcn_readlane(__float_as_int(SB), (ee & 127) >> 1);
            const float sv = __int_as_float((ee & 1) ? svB : svA);
            const float sc = (float)((double)sv + (double)(unsigned long long)(E - ee) * fc.lp_trim);
            if (sc > best) { best = sc; best_e = ee; found = 1; }
        }
    }
    FA = actA ? FA : 0xFFu; FB = actB ? FB : 0xFFu;
    rows16[(size_t)b * (DN_TROW / 2) + (lane2 >> 1)] = (unsigned short)(FA | (FB << 8));
    st.DA = upA; st.DB = upB; st.PA = SA; st.PB = SB;
}

__global__ __launch_bounds__(64) void k2_fill6(BatchDev B, const BandConsts *bc, FillConsts fc) {
    const int r = blockIdx.x;
    const int lane = threadIdx.x, lane2 = 2 * lane;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = __builtin_amdgcn_readfirstlane((int)R.n_events), K = __builtin_amdgcn_readfirstlane((int)R.n_kq);   // wave-uniform: keep them scalar
    const int n_bands = E + K + 2;
    const double lp_stay = bc[r].lp_stay, lp_step = bc[r].lp_step;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    const cdptr_t xs_c = (cdptr_t)(uintptr_t)xs;
    const cdptr_t mu_c = (cdptr_t)(uintptr_t)mus;
    unsigned short *rows16 = reinterpret_cast<unsigned short *>(B.trace + B.trace_off[r] * DN_TROW);
    const float NINF = neg_inf();
    // ---- bands 0 and 1 (event_handling.cpp:213-228): corners (49, -51) and (50, -51) ----
    F6State st;
    st.ev = 50; st.km = -51;
    // slot events as seen from band 1 (events -49 .. 50 are in the band; the other slots belong to the events that enter next)
    const int eA = 50 - (int)(((unsigned)(50 - lane2)) & 127u), eB = 50 - (int)(((unsigned)(50 - lane2 - 1)) & 127u);
    st.PA = (eA == 0) ? (float)fc.lp_trim : NINF; st.PB = NINF;          // band 1: cell (event 0, kmer -1) = lp_trim (:224-228)
    st.DA = (lane == 0) ? 0.0f : NINF; st.DB = NINF;                     // band 0: cell (event -1, kmer -1) = 0 is the up operand of event 0
    auto ldx = [&](int e) -> double { return (e >= 0 && e < E) ? xs[e] : 0.0; };
    auto ldm = [&](int k) -> double { return (k >= 0 && k < K) ? mus[k] : 0.0; };
    st.XA = ldx(eA); st.XB = ldx(eB);
    st.MA = ldm(1 - 2 - eA); st.MB = ldm(1 - 2 - eB);                    // kmer of event e in band 1
    {
        // rows 0 and 1: band 0 holds events -50 .. 49 (all from-codes 0), band 1 events -49 .. 50 (event 0: from U, :226)
        const int e0A = 49 - (int)(((unsigned)(49 - lane2)) & 127u), e0B = 49 - (int)(((unsigned)(49 - lane2 - 1)) & 127u);
        const unsigned a0 = (e0A >= -50) ? 0u : 0xFFu, b0 = (e0B >= -50) ? 0u : 0xFFu;
        const unsigned a1 = (eA >= -49) ? (eA == 0 ? 1u : 0u) : 0xFFu, b1 = (eB >= -49) ? 0u : 0xFFu;
        rows16[lane] = (unsigned short)(a0 | (b0 << 8));
        rows16[DN_TROW / 2 + lane] = (unsigned short)(a1 | (b1 << 8));
    }
    F6In in;
    f6_prefetch<false>(in, 2, st.ev, E, K, xs_c, mu_c);
    float best = NINF; int best_e = 0; int found = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // retire the pre-loop vector loads once (not per band)

    int b = 2;
    while (b < n_bands) {
        const int km = st.km, ev = st.ev;                  // corner of band b-1
        // a band is "fast" when, after its move, 0 <= km', km' + 101 < K and 99 <= ev' < E - 1: all 100 cells inside the matrix,
        // no trim / end column, every prefetch index in range.  Each band moves the corner by exactly one.
        int run = 0;
        if (km >= 0 && ev >= DN_W - 1) run = min(K - 102 - km, E - 3 - ev);
        run = min(run, n_bands - b);
        if (run > 0) {
            const int bend = b + run;
            for (; b + 1 < bend; b += 2) {                 // two bands per trip: the rotating state is renamed instead of moved
                f6_band<true>(st, b, E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows16, best, best_e, found);
                f6_band<true>(st, b + 1, E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows16, best, best_e, found);
            }
            if (b < bend) { f6_band<true>(st, b, E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows16, best, best_e, found); b++; }
        } else {
            f6_band<false>(st, b, E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows16, best, best_e, found);
            b++;
        }
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

template <bool SLOT>
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
    typedef int i32x4 __attribute__((ext_vector_type(4)));        // a native vector type: the HIP int4 struct array went to scratch
    auto load_tile = [&](int tlo, i32x4 (&regs)[8]) {
        // 64 rows * 128 B = 512 pieces of 16 B; lane handles pieces lane, lane+64, ...
        const i32x4 *src = reinterpret_cast<const i32x4 *>(rows + (size_t)tlo * DN_TROW);
#pragma unroll
        for (int i = 0; i < 8; i++) regs[i] = src[lane + 64 * i];
    };
    auto store_tile = [&](int which, const i32x4 (&regs)[8]) {
        i32x4 *dst = reinterpret_cast<i32x4 *>(tile[which]);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[lane + 64 * i] = regs[i];
    };
    i32x4 regs[8];
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
    int evrow = SLOT ? 0 : row_ev(cur);

    unsigned step = 0;
    unsigned rec_e = 0, rec_k = 0;                     // rec_e carries the from-code in its top two bits until the flush
    int bad = 0;
    // one walk step with a known from-code; returns false when the walk is over (matrix edge reached or path invalid)
    auto take = [&](unsigned from) -> bool {
        if (from == 0xFFu || step >= cap) { bad = 1; return false; }        // reference: out-of-bounds read (UB)
        // stash the step in lane (step & 63); flush 64 steps at a time, back to front
        const bool mine = lane == (int)(step & 63u);
        rec_e = mine ? ((unsigned)e | (from << 30)) : rec_e;
        rec_k = mine ? (unsigned)k : rec_k;
        step++;
        if ((step & 63u) == 0u) {
            const unsigned idx = cap - (step - 64u) - 1u - (unsigned)lane;     // step-64+lane -> slot cap-1-(step-64+lane)
            ae[idx] = rec_e & 0x3fffffffu; ak[idx] = rec_k; pf[idx] = (uint8_t)(rec_e >> 30);
        }
        // from 0 (diag): e-1, k-1, b-2; 1 (up): e-1, b-1; 2 (left): k-1, b-1 -- as arithmetic on the scalar unit, no branch
        e -= (int)((from >> 1) ^ 1u);
        k -= (int)((from & 1u) ^ 1u);
        b -= 2 - (int)((from + 1u) >> 1);
        return (k | e) >= 0;
    };
    // four-step lookahead (slot rows): lane L < 40 stands for a prefix of up to three moves (1 + 3 + 9 + 27 nodes of the ternary
    // tree of continuations); ONE LDS read fetches the from-codes of all 40 candidate cells, four dependent v_readlane then walk
    // the tree -- four steps per LDS round trip instead of one.
    int la_db = 0, la_de = 0;                          // band / event offset of this lane's node from the current cell
    {
        const int base[4] = {0, 1, 4, 13};
        int lvl = lane >= 13 ? 3 : (lane >= 4 ? 2 : (lane >= 1 ? 1 : 0));
        int code = lane - base[lvl];
        for (int j = 0; j < lvl; j++) {                // digits, last move first
            const int m = code % 3; code /= 3;
            la_db += 2 - ((m + 1) >> 1); la_de += ((m >> 1) ^ 1);
        }
        if (lane >= 40) { la_db = 0; la_de = 0; }
    }
    while ((k | e) >= 0) {
        if (b < lo) {
            // switch to the prefetched tile
            cur ^= 1;
            store_tile(cur, regs);
            lo = nlo;
            nlo = lo - CH_ROWS; if (nlo < 0) nlo = 0;
            if (lo > 0) load_tile(nlo, regs);
            __syncthreads();
            if (!SLOT) evrow = row_ev(cur);
        }
        const int bi = b - lo;
        if (SLOT) {                                        // k2_fill6 rows: byte = slot of the event, 0xFF outside the band
            if (bi >= 6) {
                const unsigned v = tile[cur][(bi - la_db) * DN_TROW + ((e - la_de) & 127)];
                const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
                if (!take(f0)) break;
                const unsigned f1 = (unsigned)__builtin_amdgcn_readlane((int)v, 1 + (int)f0);
                if (!take(f1)) break;
                const unsigned f2 = (unsigned)__builtin_amdgcn_readlane((int)v, 4 + 3 * (int)f0 + (int)f1);
                if (!take(f2)) break;
                const unsigned f3 = (unsigned)__builtin_amdgcn_readlane((int)v, 13 + 9 * (int)f0 + 3 * (int)f1 + (int)f2);
                if (!take(f3)) break;
            } else {
                if (!take((unsigned)__builtin_amdgcn_readfirstlane((int)tile[cur][bi * DN_TROW + (e & 127)]))) break;
            }
        } else {
            const int ev_b = __builtin_amdgcn_readlane(evrow, bi);
            const int off = ev_b - e;
            if (off < 0 || off >= DN_W) { bad = 1; break; }                   // reference: out-of-bounds read (UB)
            if (!take(tile[cur][bi * DN_TROW + off])) break;
        }
    }
    const unsigned rem = step & 63u;
    if (!bad && rem && (unsigned)lane < rem) {
        const unsigned base = step - rem;
        const unsigned idx = cap - (base + (unsigned)lane) - 1u;
        ae[idx] = rec_e & 0x3fffffffu; ak[idx] = rec_k; pf[idx] = (uint8_t)(rec_e >> 30);
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
int k2_fill_variant() {      // 6 = event-keyed slots (slot-indexed trace rows); anything else writes offset-indexed rows
    static const int variant = getenv("DN_FILL_VARIANT") ? atoi(getenv("DN_FILL_VARIANT")) : 6;
    return variant;
}
void k2_launch_fill(const BatchDev &B, const void *bc, const void *fc, bool dpp, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    const int variant = k2_fill_variant();
    if (variant == 6) { hipLaunchKernelGGL(k2_fill6, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f); return; }
    if (variant == 5) {
        static const int abl = getenv("DN_FILL_ABL") ? atoi(getenv("DN_FILL_ABL")) : 0;   // timing-only ablations (wrong results)
#define L5(A) case A: hipLaunchKernelGGL(k2_fill5<A>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f); return;
        switch (abl) { L5(0) L5(1) L5(2) L5(4) L5(8) L5(16) L5(31) default: break; }
#undef L5
    }
    if (variant == 44) { hipLaunchKernelGGL(k2_fill4p, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f); return; }
    if (variant == 4) {
        if (dpp) hipLaunchKernelGGL(k2_fill4<true>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
        else hipLaunchKernelGGL(k2_fill4<false>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
        return;
    }
    if (variant == 3) { hipLaunchKernelGGL(k2_fill3<2>, dim3((B.n_reads + 1) / 2), dim3(128), 0, st, B, (const BandConsts *)bc, f); return; }
    if (variant == 31) { hipLaunchKernelGGL(k2_fill3<1>, dim3(B.n_reads), dim3(128), 0, st, B, (const BandConsts *)bc, f); return; }
    if (variant == 22) { hipLaunchKernelGGL(k2_fill2p, dim3(B.n_reads), dim3(128), 0, st, B, (const BandConsts *)bc, f); return; }
    if (variant == 2) { hipLaunchKernelGGL(k2_fill2, dim3(B.n_reads), dim3(128), 0, st, B, (const BandConsts *)bc, f); return; }
    if (dpp) hipLaunchKernelGGL(k2_fill<true>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
    else hipLaunchKernelGGL(k2_fill<false>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
}
void k2_launch_chase(const BatchDev &B, uint8_t *path_from, hipStream_t st) {
    if (k2_fill_variant() == 6) hipLaunchKernelGGL(k2_chase<true>, dim3(B.n_reads), dim3(64), 0, st, B, path_from);
    else hipLaunchKernelGGL(k2_chase<false>, dim3(B.n_reads), dim3(64), 0, st, B, path_from);
}
void k2_launch_post(const BatchDev &B, const uint8_t *path_from, float *path_lp, const void *fc, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    hipLaunchKernelGGL(k2_post, dim3(B.n_reads), dim3(256), 0, st, B, path_from, path_lp, f);
}
