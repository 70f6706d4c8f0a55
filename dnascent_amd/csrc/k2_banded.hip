// k2_banded.hip -- K2: adaptive banded event-to-9mer alignment (event_handling.cpp:148-448) on gfx950.
//
//   k2_fill6  (default) one wavefront per read, EVENT-KEYED slots: the cell of event e lives in slot e & 127 for as long as
//             e is in the band, so x is stationary, the neighbours sit at fixed offsets, the k-mer level rotates by one
//             slot every band regardless of the Suzuki move, and the move only changes which slots are active (see the
//             comment above the kernel).  The whole recurrence lives in registers; the value that enters a band is
//             wave-uniform and fetched one band ahead through the scalar unit (s_load), so the loop holds no vector load
//             and never waits on vmcnt.  Per band one 128-byte row is stored: the from-codes by slot, 0xFF outside the band.
//   k2_fill5  (DN_FILL_VARIANT=5) the offset-keyed predecessor, kept for A/B measurements: cells keyed by band offset,
//             both moves' operands formed with DPP and picked with v_cndmask; rows hold 100 from-codes + the band corner.
//             Arithmetic of both is the reference's, cast by cast (float scores, candidates evaluated in fp64 and
//             rounded back; event_handling.cpp:116-137, :296-306); the fp64 division by sigma is done exactly
//             with an FMA-corrected reciprocal (3 ops, brute-force verified against IEEE division).
//   k2_chase  backtrack (event_handling.cpp:356-412), one wavefront per read: trace rows are staged through a
//             double-buffered LDS tile (coalesced 16-B loads); with slot rows one LDS read fetches the from-codes of the
//             40 cells reachable in the next three moves and four dependent v_readlane walk that tree.
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

// the same cell with the diagonal operand already widened: "diag" of band b is "up" of band b - 1, whose (double) conversion
// band b - 1 computed anyway -- k2_fill6 carries it over instead of converting the same float twice
__device__ __forceinline__ void cell_d(double ddiag, float up, float left, double x, double mu, const FillConsts &fc,
                                       double lp_step, double lp_stay, float &score, unsigned &from, double &dup) {
    const double d = x - mu;
    const double q = d * fc.rsigma;
    const double rem = fma(-q, fc.sigma, d);
    const double ad = fma(rem, fc.rsigma, q);
    const float a = (float)ad;                            // :133
    float t = -0.5f * a;                                  // :135
    t = t * a;
    const float em = (float)(fc.C + (double)t);           // :135-136
    const double emd = (double)em;
    dup = (double)up;
    const float sd = (float)((ddiag + lp_step) + emd);          // :296
    const float su = (float)((dup + lp_stay) + emd);            // :297
    const float sl = (float)((double)left + fc.lp_skip);        // :298
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
    float PA, PB;
    double DA, DB;           // the diagonal operands, kept widened (see cell_d)
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
    // Program order is chosen for a lone wavefront (in-order issue): the end scores of the previous band are read first,
    // the cell arithmetic -- which needs neither the move nor the masks -- follows, and only then comes the scalar chain of
    // the Suzuki move, so that its VALU -> SALU hazard and dependent scalar instructions hide behind the cell work.
    const int ev0 = st.ev, el0 = ev0 - (DN_W - 1);
    const int l_lo = (ev0 & 127) >> 1, l_hi = (el0 & 127) >> 1;
    const int loA = __builtin_amdgcn_readlane(__float_as_int(st.PA), l_lo), loB = __builtin_amdgcn_readlane(__float_as_int(st.PB), l_lo);
    const int hiA = __builtin_amdgcn_readlane(__float_as_int(st.PA), l_hi), hiB = __builtin_amdgcn_readlane(__float_as_int(st.PB), l_hi);
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
    // ---- operands: left = same slot, up = previous slot, diag = the previous band's up ----
    const float upA = ror_f(st.PB), upB = st.PA;
    float SA, SB; unsigned FA, FB; double dupA, dupB;
    cell_d(st.DA, upA, st.PA, st.XA, st.MA, fc, lp_step, lp_stay, SA, FA, dupA);
    cell_d(st.DB, upB, st.PB, st.XB, st.MB, fc, lp_step, lp_stay, SB, FB, dupB);
    // ---- Suzuki-Kasahara move (:237-253) from the end cells of the previous band: events ev (lower left) and ev - 99 ----
    const int lo = (ev0 & 1) ? loB : loA, hi = (el0 & 1) ? hiB : hiA;
    // integer 0/1 arithmetic (scalar unit, no branch): both end cells out of band -> alternate by parity, else ll < ur
    const int ol = ordered_f32_bits(lo), oh = ordered_f32_bits(hi);
    const int lt = (ol < oh) ? 1 : 0;
    const int right = (max(ol, oh) == ordered_f32_bits(NINF_BITS)) ? (b & 1) : lt;
    const int km = st.km + right, ev = ev0 + (right ^ 1);
    st.km = km; st.ev = ev;
    f6_prefetch<FAST>(in, b + 1, ev, E, K, xs_c, mu_c);
    // ---- in-band slots: ((ev - event) & 127) < 100, a cyclic run of 50 lanes in each register.  With several batches in
    //      flight the vector unit is the scarce one, so the two masks are built on the scalar unit (rotate a 50-bit run) ----
    const unsigned tA = (unsigned)(ev - lane2) & 127u;     // only the edge path below needs the per-lane distance
    const unsigned p0 = (unsigned)(ev - (DN_W - 1)) & 127u;
    const unsigned long long FIFTY = (1ull << 50) - 1ull;
    const bool actA = __builtin_amdgcn_inverse_ballot_w64(rotl64_(FIFTY, ((p0 + 1u) >> 1) & 63u));
    const bool actB = __builtin_amdgcn_inverse_ballot_w64(rotl64_(FIFTY, p0 >> 1));
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
    st.DA = dupA; st.DB = dupB; st.PA = SA; st.PB = SB;
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
    st.DA = (lane == 0) ? 0.0 : (double)NINF; st.DB = (double)NINF;      // band 0: cell (event -1, kmer -1) = 0 is the up operand of event 0
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
    (void)E;                                              // only the offset-keyed (SLOT == false) walk needs it
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
    int bad = 0;
    if (SLOT) {
        // ---- slot rows: the walk records ONLY the from-codes, 2 bits per step, 32 steps per 64-bit word (written by lane 0
        //      into the still-unused cleaned-signal workspace); k2_expand turns the code stream into (event, kmer) pairs with a
        //      parallel scan.  Four-step lookahead: lane L < 40 stands for a prefix of up to three moves (1 + 3 + 9 + 27 nodes
        //      of the ternary tree of continuations); ONE LDS read fetches the from-codes of all 40 candidate cells and four
        //      dependent v_readlane walk the tree.  Away from the matrix edge a group of four steps needs no per-step test:
        //      the position update is two popcounts of the packed codes (diag: e-1 k-1 b-2, up: e-1 b-1, left: k-1 b-1). ----
        unsigned long long *words = reinterpret_cast<unsigned long long *>(B.cl_sig + a0);
        unsigned long long acc = 0ull; unsigned nacc = 0, nword = 0;
        int la_db = 0, la_de = 0;                          // band / event offset of this lane's node from the current cell
        {
            const int base[4] = {0, 1, 4, 13};
            const int lvl = lane >= 13 ? 3 : (lane >= 4 ? 2 : (lane >= 1 ? 1 : 0));
            int code = lane - base[lvl];
            for (int j = 0; j < lvl; j++) {
                const int m = code % 3; code /= 3;
                la_db += 2 - ((m + 1) >> 1); la_de += ((m >> 1) ^ 1);
            }
            if (lane >= 40) { la_db = 0; la_de = 0; }
        }
        auto push = [&](unsigned codes, unsigned cnt) {     // cnt codes of 2 bits, oldest in the low bits; cnt <= 4
            acc |= (unsigned long long)codes << (2u * nacc);
            const unsigned room = 32u - nacc;
            if (cnt >= room) {
                if (lane == 0) words[nword] = acc;
                nword++;
                acc = (cnt > room) ? ((unsigned long long)codes >> (2u * room)) : 0ull;
                nacc = cnt - room;
            } else nacc += cnt;
        };
        while ((k | e) >= 0) {
            if (b < lo) {
                cur ^= 1;
                store_tile(cur, regs);
                lo = nlo;
                nlo = lo - CH_ROWS; if (nlo < 0) nlo = 0;
                if (lo > 0) load_tile(nlo, regs);
                __syncthreads();
            }
            const int bi = b - lo;
            if (bi >= 6 && e >= 4 && k >= 4 && step + 4u <= cap) {
                const unsigned v = tile[cur][(bi - la_db) * DN_TROW + ((e - la_de) & 127)];
                const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
                const unsigned f1 = (unsigned)__builtin_amdgcn_readlane((int)v, 1 + (int)(f0 & 3u));
                const unsigned f2 = (unsigned)__builtin_amdgcn_readlane((int)v, 4 + 3 * (int)(f0 & 3u) + (int)(f1 & 3u));
                const unsigned f3 = (unsigned)__builtin_amdgcn_readlane((int)v, 13 + 9 * (int)(f0 & 3u) + 3 * (int)(f1 & 3u) + (int)(f2 & 3u));
                if (((f0 | f1 | f2 | f3) & 0xFCu) == 0u && f0 != 3u && f1 != 3u && f2 != 3u && f3 != 3u) {
                    const unsigned p = f0 | (f1 << 2) | (f2 << 4) | (f3 << 6);
                    push(p, 4u);
                    const int de = 4 - __builtin_popcount(p & 0xAAu), dk = 4 - __builtin_popcount(p & 0x55u);
                    e -= de; k -= dk; b -= de + dk;
                    step += 4u;
                    continue;
                }
            }
            // single step (matrix edge, tile seam, or an invalid code ahead)
            const unsigned from = (unsigned)__builtin_amdgcn_readfirstlane((int)tile[cur][bi * DN_TROW + (e & 127)]);
            if (from > 2u || step >= cap) { bad = 1; break; }                  // reference: out-of-bounds read (UB)
            push(from, 1u);
            step++;
            e -= (int)((from >> 1) ^ 1u);
            k -= (int)((from & 1u) ^ 1u);
            b -= 2 - (int)((from + 1u) >> 1);
        }
        if (nacc && lane == 0) words[nword] = acc;
    } else {
    unsigned rec_e = 0, rec_k = 0;                     // rec_e carries the from-code in its top two bits until the flush
    auto take = [&](unsigned from) -> bool {
        if (from == 0xFFu || step >= cap) { bad = 1; return false; }        // reference: out-of-bounds read (UB)
        const bool mine = lane == (int)(step & 63u);
        rec_e = mine ? ((unsigned)e | (from << 30)) : rec_e;
        rec_k = mine ? (unsigned)k : rec_k;
        step++;
        if ((step & 63u) == 0u) {
            const unsigned idx = cap - (step - 64u) - 1u - (unsigned)lane;     // step-64+lane -> slot cap-1-(step-64+lane)
            ae[idx] = rec_e & 0x3fffffffu; ak[idx] = rec_k; pf[idx] = (uint8_t)(rec_e >> 30);
        }
        e -= (int)((from >> 1) ^ 1u);
        k -= (int)((from & 1u) ^ 1u);
        b -= 2 - (int)((from + 1u) >> 1);
        return (k | e) >= 0;
    };
    while ((k | e) >= 0) {
        if (b < lo) {
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
        if (off < 0 || off >= DN_W) { bad = 1; break; }                   // reference: out-of-bounds read (UB)
        if (!take(tile[cur][bi * DN_TROW + off])) break;
    }
    const unsigned rem = step & 63u;
    if (!bad && rem && (unsigned)lane < rem) {
        const unsigned base = step - rem;
        const unsigned idx = cap - (base + (unsigned)lane) - 1u;
        ae[idx] = rec_e & 0x3fffffffu; ak[idx] = rec_k; pf[idx] = (uint8_t)(rec_e >> 30);
    }
    }
    if (lane == 0) {
        if (bad) { R.status = 3; R.n_aligned = 0; R.aln_begin = cap; }
        else { R.n_aligned = step; R.aln_begin = cap - step; }
    }
}

// ------------------------------------------------------------------------------------------------
// k2_expand: the from-code stream of k2_chase<true> (2 bits per walk step, step 0 = the end cell) -> alignment pairs.
// The event / kmer of step i is the end cell minus the number of earlier steps that moved in that dimension: an exclusive
// prefix sum over the codes (per 32-step word: two popcounts), done per read by one block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k2_expand(BatchDev B, uint8_t *path_from) {
    __shared__ unsigned s_e[256], s_k[256];
    __shared__ unsigned base_e, base_k;
    const int r = blockIdx.x, tid = threadIdx.x;
    const ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const unsigned n = R.n_aligned;
    const uint64_t a0 = B.aln_off[r];
    const unsigned cap = (unsigned)(B.aln_off[r + 1] - a0);
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(B.cl_sig + a0);
    unsigned *ae = B.aln_event + a0, *ak = B.aln_kmer + a0;
    uint8_t *pf = path_from + a0;
    const unsigned nw = (n + 31u) / 32u;
    if (tid == 0) { base_e = 0; base_k = 0; }
    __syncthreads();
    const unsigned e_end = (unsigned)R.end_event, k_end = R.n_kq - 1u;
    for (unsigned w0 = 0; w0 < nw; w0 += 256) {
        const unsigned w = w0 + tid;
        unsigned long long word = 0ull; unsigned cnt = 0;
        if (w < nw) { word = words[w]; cnt = min(32u, n - w * 32u); }
        const unsigned long long live = cnt >= 32u ? ~0ull : ((1ull << (2u * cnt)) - 1ull);
        // steps that decrement e have code 0 or 1 (high bit clear), steps that decrement k have code 0 or 2 (low bit clear)
        const unsigned de = cnt - (unsigned)__popcll(word & live & 0xAAAAAAAAAAAAAAAAull);
        const unsigned dk = cnt - (unsigned)__popcll(word & live & 0x5555555555555555ull);
        s_e[tid] = de; s_k[tid] = dk;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned te = tid >= d ? s_e[tid - d] : 0u, tk = tid >= d ? s_k[tid - d] : 0u;
            __syncthreads();
            s_e[tid] += te; s_k[tid] += tk;
            __syncthreads();
        }
        unsigned e = e_end - (base_e + s_e[tid] - de), k = k_end - (base_k + s_k[tid] - dk);
        for (unsigned j = 0; j < cnt; j++) {
            const unsigned from = (unsigned)(word >> (2u * j)) & 3u;
            const unsigned idx = cap - 1u - (w * 32u + j);              // walk step -> slot: the arrays end up in forward order
            ae[idx] = e; ak[idx] = k; pf[idx] = (uint8_t)from;
            e -= (from >> 1) ^ 1u; k -= (from & 1u) ^ 1u;
        }
        __syncthreads();
        if (tid == 255) { base_e += s_e[255]; base_k += s_k[255]; }
        __syncthreads();
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
    __shared__ int redi[4];
    __shared__ double emd[64];                            // the 64 emissions of a chunk, widened, for the ordered sum
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
        // ordered compaction: ballot + popcount inside a wavefront, one LDS hop for the four wavefront totals
        const unsigned long long bm = __ballot(flag != 0);
        const unsigned within = (unsigned)__popcll(bm & ((1ull << (tid & 63)) - 1ull));
        if ((tid & 63) == 0) redi[tid >> 6] = (int)__popcll(bm);
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const unsigned c = (unsigned)redi[q]; before += (q < (tid >> 6)) ? c : 0u; total += c; }
        const unsigned pos = out_base + before + within;
        if (flag) { cl_sig[pos] = sig; cl_rank[pos] = rk; }
        out_base += total;
        __syncthreads();
    }

    // 3. ordered fp64 sum of the emissions (:364, walk order: one dependent add per step, nothing else in the chain) and the
    //    longest run of FROM_L steps (:399-410) from 64-step ballots with scalar bit arithmetic
    if (tid < 64) {
        double sum_em = 0.; int gap = 0, max_gap = 0;
        for (unsigned wb = 0; wb < n; wb += 64) {
            const unsigned w = wb + tid;
            float v = 0.f; unsigned f = 1;
            if (w < n) { v = lp[n - 1 - w]; f = pf[n - 1 - w]; }
            const unsigned lim = min(64u, n - wb);
            unsigned long long m2 = __ballot(f == 2u);                        // bit i: walk step wb + i came from the left
            // the operands come back as broadcast LDS reads (their own issue port): ONE vector instruction per step, the add
            // (a v_readlane + v_cvt_f64_f32 + v_add_f64 chain was three)
            emd[tid] = (double)v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wavefront: LDS operations are in order
            if (lim == 64u) {
#pragma unroll 16
                for (unsigned i = 0; i < 64u; i++) sum_em += emd[i];
            } else {
                for (unsigned i = 0; i < lim; i++) sum_em += emd[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (m2 == 0ull) { gap = 0; continue; }
            const unsigned long long valid = lim == 64u ? ~0ull : ((1ull << lim) - 1ull);
            const unsigned long long inv = (~m2) & valid;                     // the steps of the chunk that are NOT from the left
            const unsigned lead = inv ? (unsigned)__builtin_ctzll(inv) : lim; // run that continues the previous chunk's
            max_gap = max(max_gap, gap + (int)lead);
            unsigned long long x = m2; int longest = 0;
            while (x) { x &= x << 1; longest++; }                             // longest run inside the chunk
            max_gap = max(max_gap, longest);
            // run that reaches the end of the chunk (carried into the next one)
            gap = inv ? (int)(lim - 1u - (63u - (unsigned)__builtin_clzll(inv))) : gap + (int)lim;
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
    (void)dpp;
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    if (k2_fill_variant() == 6) { hipLaunchKernelGGL(k2_fill6, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f); return; }
    hipLaunchKernelGGL(k2_fill5<0>, dim3(B.n_reads), dim3(64), 0, st, B, (const BandConsts *)bc, f);
}
void k2_launch_chase(const BatchDev &B, uint8_t *path_from, hipStream_t st) {
    if (k2_fill_variant() == 6) {
        hipLaunchKernelGGL(k2_chase<true>, dim3(B.n_reads), dim3(64), 0, st, B, path_from);
        hipLaunchKernelGGL(k2_expand, dim3(B.n_reads), dim3(256), 0, st, B, path_from);
    } else hipLaunchKernelGGL(k2_chase<false>, dim3(B.n_reads), dim3(64), 0, st, B, path_from);
}
void k2_launch_post(const BatchDev &B, const uint8_t *path_from, float *path_lp, const void *fc, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    hipLaunchKernelGGL(k2_post, dim3(B.n_reads), dim3(256), 0, st, B, path_from, path_lp, f);
}
