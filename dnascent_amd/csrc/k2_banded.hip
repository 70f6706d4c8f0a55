// k2_banded.hip -- K2: adaptive banded event-to-9mer alignment (event_handling.cpp:148-448) on gfx950.
//
//   k2_fill6  one wavefront per read, EVENT-KEYED slots: the cell of event e lives in slot e & 127 for as long as e is in the
//             band, so x is stationary, the neighbours sit at fixed offsets, the k-mer level rotates by one slot every band
//             regardless of the Suzuki move, and the move only changes which slots are active (see the comment above the
//             kernel).  The whole recurrence lives in registers; the value that enters a band is wave-uniform and fetched
//             one band ahead through the scalar unit (s_load), so the loop holds no vector load and never waits on vmcnt.
//             Per band one 32-byte row is produced: 2-bit from-codes by slot as four 64-bit lane masks (put_row); eight rows
//             leave in one 256-byte store.  Arithmetic is the reference's, cast by cast (float scores, candidates evaluated in
//             fp64 and rounded back; event_handling.cpp:116-137, :296-306); the fp64 division by sigma is done exactly with an
//             FMA-corrected reciprocal (3 ops, brute-force verified against IEEE division).  (The offset-keyed predecessor
//             k2_fill5 and its ablation switches live in tools/k2_fill5_ablation.hip, outside the product.)
//   k2_chase  backtrack (event_handling.cpp:356-412), one wavefront per read: trace rows are staged through a double-buffered
//             LDS tile (coalesced 16-B loads); one LDS read per lane fetches the from-code of one of the 40 cells reachable in
//             the next three moves and four dependent v_readlane walk that tree.  Output: the 2-bit code stream.
//   k2_expand code stream -> (event, kmer, from) arrays in forward order, coalesced.
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


// one cell of the recurrence (event_handling.cpp:280-311 + :116-137).  The diagonal operand arrives already widened: "diag" of
// band b is "up" of band b - 1, whose (double) conversion band b - 1 computed anyway.  The from-code is returned as the two
// compare results it is made of (eu: max == up, el: max == left): the caller turns them into wave-wide masks, which is the
// form the 2-bit trace rows are stored in.
__device__ __forceinline__ void cell_d(double ddiag, float up, float left, double x, double mu, const FillConsts &fc,
                                       double lp_step, double lp_stay, float &score, bool &eu, bool &el, double &dup) {
    const double d = x - mu;
    const double q = d * fc.rsigma;                       // exact (x - mu) / sigma via FMA-corrected reciprocal
    const double rem = fma(-q, fc.sigma, d);
    const double ad = fma(rem, fc.rsigma, q);
    const float a = (float)ad;                            // :133
    float t = -0.5f * a;                                  // :135 (-0.5f * a) * a in float
    t = t * a;
    const float em = (float)(fc.C + (double)t);           // :135-136
    const double emd = (double)em;
    dup = (double)up;
    const float sd = (float)((ddiag + lp_step) + emd);          // :296
    const float su = (float)((dup + lp_stay) + emd);            // :297
    const float sl = (float)((double)left + fc.lp_skip);        // :298
    // :300-306  max = d; if (u > max) max = u; from = (max == u) ? U : D; then the same for l: the result is the maximum of
    // the three with ties resolved L over U over D (scores are never NaN), i.e. one v_max3 + two compares.
    float mx;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(sd), "v"(su), "v"(sl));
    eu = su == mx; el = sl == mx;
    score = mx;
}

// uniform (wave-wide) loads of the one new x / mu value a band needs go through the scalar unit (s_load, counted
// on lgkmcnt): the loop then holds no vector load, so its trace stores are never waited for (loads and stores share
// the in-order vmcnt on gfx950).  The arrays were written by earlier kernels and are read-only here, which is what
// the constant address space promises.
typedef const double __attribute__((address_space(4))) *cdptr_t;


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

// ---- trace rows.  Row b is 32 bytes: the 2-bit from-codes of the 128 slots, PLANAR -- four 64-bit lane masks
//      {A0, A1, B0, B1}: bit l of A0 / A1 = low / high code bit of slot 2l (even events), B0 / B1 of slot 2l + 1.  Codes:
//      0 diag, 1 up, 2 left, 3 = slot outside the band.  This is the form the compare results of the cell arithmetic have
//      anyway (a v_cmp writes a lane mask into a scalar register pair), so the codes are never materialised per lane; the
//      eight dwords of a row are dropped into lanes 8 (b & 7) .. + 7 of an accumulator register (v_writelane) and every
//      eighth band the wavefront stores the accumulator: 256 contiguous bytes = 8 rows.  (The reference stores one byte per
//      cell, event_handling.cpp:192-199; round 1 stored 128-byte rows.) ----
template <int PH>
__device__ __forceinline__ void put_row(int &acc, const int b, const unsigned long long A0, const unsigned long long A1,
                                        const unsigned long long B0, const unsigned long long B1, unsigned *rows32, const int lane) {
    const int d[8] = { (int)(unsigned)A0, (int)(A0 >> 32), (int)(unsigned)A1, (int)(A1 >> 32),
                       (int)(unsigned)B0, (int)(B0 >> 32), (int)(unsigned)B1, (int)(B1 >> 32) };
    if (PH >= 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(d[j]), "i"(8 * (PH < 0 ? 0 : PH) + j));   // lane select: inline constant
        if (PH == 7) rows32[(size_t)(b >> 3) * 64 + lane] = (unsigned)acc;
    } else {
        const int base = 8 * (b & 7);
#pragma unroll
        for (int j = 0; j < 8; j++) acc = writelane_(acc, d[j], base + j);
        if ((b & 7) == 7) rows32[(size_t)(b >> 3) * 64 + lane] = (unsigned)acc;
    }
}

struct F6State {
    float PA, PB;
    double DA, DB;           // the diagonal operands, kept widened (see cell_d)
    double XA, XB, MA, MB;
    int ev, km;              // lower-left corner of the last band
    int acc;                 // trace rows of the current group of eight bands (put_row)
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

// PH: b & 7 when it is known at compile time (the eight-band trips of the fast loop), -1 otherwise
template <bool FAST, int PH>
__device__ __forceinline__ void f6_band(F6State &st, const int b, const int E, const int K, const int lane2, const cdptr_t xs_c,
                                        const cdptr_t mu_c, F6In &in, const FillConsts &fc, const double lp_step,
                                        const double lp_stay, unsigned *rows32, float &best, int &best_e, int &found) {
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
    float SA, SB; bool euA, elA, euB, elB; double dupA, dupB;
    cell_d(st.DA, upA, st.PA, st.XA, st.MA, fc, lp_step, lp_stay, SA, euA, elA, dupA);
    cell_d(st.DB, upB, st.PB, st.XB, st.MB, fc, lp_step, lp_stay, SB, euB, elB, dupB);
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
    const unsigned p0 = (unsigned)(ev - (DN_W - 1)) & 127u;
    const unsigned long long FIFTY = (1ull << 50) - 1ull;
    const unsigned long long mA = rotl64_(FIFTY, ((p0 + 1u) >> 1) & 63u), mB = rotl64_(FIFTY, p0 >> 1);
    const bool actA = __builtin_amdgcn_inverse_ballot_w64(mA);
    const bool actB = __builtin_amdgcn_inverse_ballot_w64(mB);
    unsigned long long A0, A1, B0, B1;
    if (FAST) {
        SA = actA ? SA : NINF; SB = actB ? SB : NINF;
        // from-code = 2 if max == left, else 1 if max == up, else 0 (:300-306); 3 outside the band -- all on the scalar unit
        const unsigned long long uA = __ballot(euA), lA = __ballot(elA), uB = __ballot(euB), lB = __ballot(elB);
        A1 = lA | ~mA; A0 = (uA & ~lA) | ~mA;
        B1 = lB | ~mB; B0 = (uB & ~lB) | ~mB;
    } else {
        const unsigned tA = (unsigned)(ev - lane2) & 127u;     // per-lane distance of the even slot's event from the corner
        const int eA = ev - (int)tA, eB = ev - (int)((tA - 1u) & 127u);
        const int kA = b - 2 - eA, kB = b - 2 - eB;
        unsigned FA = elA ? 2u : (euA ? 1u : 0u), FB = elB ? 2u : (euB ? 1u : 0u);
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
        FA = actA ? FA : 3u; FB = actB ? FB : 3u;
        A0 = __ballot((FA & 1u) != 0u); A1 = __ballot((FA & 2u) != 0u);
        B0 = __ballot((FB & 1u) != 0u); B1 = __ballot((FB & 2u) != 0u);
    }
    put_row<PH>(st.acc, b, A0, A1, B0, B1, rows32, lane2 >> 1);
    st.DA = dupA; st.DB = dupB; st.PA = SA; st.PB = SB;
}

#define F6_ARGS st, E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows32, best, best_e, found
#define F6_BAND(FAST_, PH_, b_) f6_band<FAST_, PH_>(st, (b_), E, K, lane2, xs_c, mu_c, in, fc, lp_step, lp_stay, rows32, best, best_e, found)

// K2_FILL_W reads (= wavefronts: a read is one wavefront, they share nothing) per WORKGROUP.  The dispatcher places a workgroup on ONE CU, so W > 1 packs the
// batch's 500 long-lived wavefronts onto n / W CUs instead of spreading one over every CU: the network's kernels are sized to own a CU's registers (the
// 256-row convolution: two workgroups x 128 registers x 2 wavefronts per SIMD = all 512), and a single foreign wavefront of 48 registers on one SIMD
// keeps the second workgroup off that whole CU for as long as it lives (round 4).
#ifndef K2_FILL_W
#define K2_FILL_W 4                                          /* round 4, one session (gpurun_out/r4m, r4n): W = 1 760-764 Msamples/s, 4 with K2B_W 4: 782; 8 / 6: 775-781; 16 / 4: no gain; 2 / 2: none */
#endif
__global__ __launch_bounds__(64 * K2_FILL_W) void k2_fill6(BatchDev B, const BandConsts *bc, FillConsts fc) {
    const int r = K2_FILL_W > 1 ? __builtin_amdgcn_readfirstlane((int)(blockIdx.x * K2_FILL_W + (threadIdx.x >> 6))) : (int)blockIdx.x;     // wave-uniform, and the compiler must know it
    if (r >= B.n_reads) return;
    const int lane = threadIdx.x & 63, lane2 = 2 * lane;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int E = __builtin_amdgcn_readfirstlane((int)R.n_events), K = __builtin_amdgcn_readfirstlane((int)R.n_kq);   // wave-uniform: keep them scalar
    const int n_bands = E + K + 2;
    const double lp_stay = bc[r].lp_stay, lp_step = bc[r].lp_step;
    const double *xs = B.ev_x + B.ev_off[r];
    const double *mus = B.mu_q + B.base_off[r];
    const cdptr_t xs_c = (cdptr_t)(uintptr_t)xs;
    const cdptr_t mu_c = (cdptr_t)(uintptr_t)mus;
    unsigned *rows32 = reinterpret_cast<unsigned *>(B.trace + B.trace_off[r] * DN_TROW);
    const float NINF = neg_inf();
    // ---- bands 0 and 1 (event_handling.cpp:213-228): corners (49, -51) and (50, -51) ----
    F6State st;
    st.ev = 50; st.km = -51; st.acc = 0;
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
        const unsigned a0 = (e0A >= -50) ? 0u : 3u, b0 = (e0B >= -50) ? 0u : 3u;
        const unsigned a1 = (eA >= -49) ? (eA == 0 ? 1u : 0u) : 3u, b1 = (eB >= -49) ? 0u : 3u;
        put_row<-1>(st.acc, 0, __ballot((a0 & 1u) != 0u), __ballot((a0 & 2u) != 0u), __ballot((b0 & 1u) != 0u), __ballot((b0 & 2u) != 0u), rows32, lane);
        put_row<-1>(st.acc, 1, __ballot((a1 & 1u) != 0u), __ballot((a1 & 2u) != 0u), __ballot((b1 & 1u) != 0u), __ballot((b1 & 2u) != 0u), rows32, lane);
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
            for (; b < bend && (b & 7); b++) F6_BAND(true, -1, b);
            // eight bands per trip: b & 7 is a compile-time constant (row deposit with constant lane selects, one 256-byte store per
            // trip), and the rotating state is renamed instead of moved
            for (; b + 8 <= bend; b += 8) {
                F6_BAND(true, 0, b);     F6_BAND(true, 1, b + 1); F6_BAND(true, 2, b + 2); F6_BAND(true, 3, b + 3);
                F6_BAND(true, 4, b + 4); F6_BAND(true, 5, b + 5); F6_BAND(true, 6, b + 6); F6_BAND(true, 7, b + 7);
            }
            for (; b < bend; b++) F6_BAND(true, -1, b);
        } else {
            F6_BAND(false, -1, b);
            b++;
        }
    }
    if (n_bands & 7) rows32[(size_t)((n_bands - 1) >> 3) * 64 + lane] = (unsigned)st.acc;     // the last, partial group of rows
    if (lane == 0) {
        R.n_bands = (unsigned)n_bands;
        R.end_event = best_e;
        R.end_score = best;
        if (!found) R.status = 3;
    }
}

// ------------------------------------------------------------------------------------------------
// k2_chase: record the backtrack path (event_handling.cpp:356-412) as a stream of from-codes.
// One wavefront per read; trace rows (32 B each) are staged through double-buffered LDS tiles of 256 rows (coalesced 16-byte
// loads; the whole trace is read exactly once: the path visits every band or every other band).  The walk records ONLY the
// from-codes, 2 bits per step, 32 steps per 64-bit word (written by lane 0 into the still-unused cleaned-signal workspace);
// k2_expand turns the code stream into (event, kmer) pairs.  Four-step lookahead: lane L < 40 stands for a prefix of up to three
// moves (1 + 3 + 9 + 27 nodes of the ternary tree of continuations); ONE 16-byte LDS read per lane fetches the two code planes
// of its candidate cell and four dependent v_readlane walk the tree.  Away from the matrix edge a group of four steps needs no
// per-step test: the position update is two popcounts of the packed codes (diag: e-1 k-1 b-2, up: e-1 b-1, left: k-1 b-1).
// ------------------------------------------------------------------------------------------------
#define CH_ROWS 256

__device__ __forceinline__ unsigned trace_code(const uint8_t *tile, int row, int ev) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const unsigned s = (unsigned)ev & 127u;
    const u64x2 w = *reinterpret_cast<const u64x2 *>(tile + row * DN_TROW + (s & 1u) * 16u);    // {X0, X1} of the slot's register
    const unsigned l = s >> 1;
    return ((unsigned)(w[0] >> l) & 1u) | (((unsigned)(w[1] >> l) & 1u) << 1);
}

// K2_CHASE_W reads (= wavefronts) per workgroup, as K2_FILL_W: every wavefront has its own pair of tiles, and the barriers (which only ever ordered one
// wavefront's LDS traffic) are wave-level fences
#ifndef K2_CHASE_W
#define K2_CHASE_W 1
#endif
#if K2_CHASE_W > 1
#define K2C_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define K2C_SYNC() __syncthreads()
#endif
__global__ __launch_bounds__(64 * K2_CHASE_W) void k2_chase(BatchDev B) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles_[K2_CHASE_W][2][CH_ROWS * DN_TROW];
    uint8_t (*tile)[CH_ROWS * DN_TROW] = tiles_[K2_CHASE_W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0];
    const int r = K2_CHASE_W > 1 ? __builtin_amdgcn_readfirstlane((int)(blockIdx.x * K2_CHASE_W + (threadIdx.x >> 6))) : (int)blockIdx.x;
    if (r >= B.n_reads) return;
    const int lane = threadIdx.x & 63;
    ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int K = (int)R.n_kq;
    const uint8_t *rows = B.trace + B.trace_off[r] * DN_TROW;
    const uint64_t a0 = B.aln_off[r];
    const unsigned cap = (unsigned)(B.aln_off[r + 1] - a0);

    int e = R.end_event, k = K - 1;
    int b = e + k + 2;
    // a tile covers bands [lo, lo + CH_ROWS); rows below band 0 do not exist: tiles are clamped at 0 and always hold CH_ROWS rows
    // starting at `lo` (the trace allocation of a read is padded by CH_ROWS rows)
    int lo = b - (CH_ROWS - 1); if (lo < 0) lo = 0;
    typedef int i32x4 __attribute__((ext_vector_type(4)));        // a native vector type: the HIP int4 struct array went to scratch
    auto load_tile = [&](int tlo, i32x4 (&regs)[8]) {
        // 256 rows * 32 B = 512 pieces of 16 B; lane handles pieces lane, lane+64, ...
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
    load_tile(lo, regs);
    store_tile(0, regs);
    int cur = 0;
    int nlo = lo - CH_ROWS; if (nlo < 0) nlo = 0;
    if (lo > 0) load_tile(nlo, regs);
    K2C_SYNC();

    unsigned step = 0;
    int bad = 0;
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
            K2C_SYNC();
        }
        const int bi = b - lo;
        if (bi >= 6 && e >= 4 && k >= 4 && step + 4u <= cap) {
            const unsigned v = trace_code(tile[cur], bi - la_db, e - la_de);
            const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
            const unsigned f1 = (unsigned)__builtin_amdgcn_readlane((int)v, 1 + (int)f0);
            const unsigned f2 = (unsigned)__builtin_amdgcn_readlane((int)v, 4 + 3 * (int)f0 + (int)f1);
            const unsigned f3 = (unsigned)__builtin_amdgcn_readlane((int)v, 13 + 9 * (int)f0 + 3 * (int)f1 + (int)f2);
            // a code 3 (slot outside the band) makes the lane indices above meaningless, but every index stays below 64 and the
            // group is discarded here
            if (f0 != 3u && f1 != 3u && f2 != 3u && f3 != 3u) {
                const unsigned p = f0 | (f1 << 2) | (f2 << 4) | (f3 << 6);
                push(p, 4u);
                const int de = 4 - __builtin_popcount(p & 0xAAu), dk = 4 - __builtin_popcount(p & 0x55u);
                e -= de; k -= dk; b -= de + dk;
                step += 4u;
                continue;
            }
        }
        // single step (matrix edge, tile seam, or an invalid code ahead)
        const unsigned from = (unsigned)__builtin_amdgcn_readfirstlane((int)trace_code(tile[cur], bi, e));
        if (from > 2u || step >= cap) { bad = 1; break; }                  // reference: out-of-bounds read (UB)
        push(from, 1u);
        step++;
        e -= (int)((from >> 1) ^ 1u);
        k -= (int)((from & 1u) ^ 1u);
        b -= 2 - (int)((from + 1u) >> 1);
    }
    if (nacc && lane == 0) words[nword] = acc;
    if (lane == 0) {
        if (bad) { R.status = 3; R.n_aligned = 0; R.aln_begin = cap; }
        else { R.n_aligned = step; R.aln_begin = cap - step; }
    }
}

// ------------------------------------------------------------------------------------------------
// k2_expand: the from-code stream of k2_chase (2 bits per walk step, step 0 = the end cell) -> alignment pairs.
// The event / kmer of step i is the end cell minus the number of earlier steps that moved in that dimension: an exclusive
// prefix sum over the codes.  One block per read works on 8 192 steps at a time: 256 words are scanned (two popcounts per word),
// then thread t takes steps t, t + 256, ... of the group, finds its own position with two more popcounts of the masked word,
// and writes slot cap - 1 - step: consecutive threads write consecutive addresses (the arrays end up in forward order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k2_expand(BatchDev B, uint8_t *path_from) {
    __shared__ unsigned s_e[256], s_k[256];
    __shared__ unsigned long long s_w[256];
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
        word &= live;
        // steps that decrement e have code 0 or 1 (high bit clear), steps that decrement k have code 0 or 2 (low bit clear)
        const unsigned de = cnt - (unsigned)__popcll(word & 0xAAAAAAAAAAAAAAAAull);
        const unsigned dk = cnt - (unsigned)__popcll(word & 0x5555555555555555ull);
        s_e[tid] = de; s_k[tid] = dk; s_w[tid] = word;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned te = tid >= d ? s_e[tid - d] : 0u, tk = tid >= d ? s_k[tid - d] : 0u;
            __syncthreads();
            s_e[tid] += te; s_k[tid] += tk;
            __syncthreads();
        }
        const unsigned be = base_e, bk = base_k;
        const unsigned gsteps = min(8192u, n - w0 * 32u);
        for (unsigned li = tid; li < gsteps; li += 256) {
            const unsigned wl = li >> 5, bit = li & 31u;
            const unsigned long long wd = s_w[wl];
            const unsigned long long below = wd & ((1ull << (2u * bit)) - 1ull);
            const unsigned pe = wl ? s_e[wl - 1] : 0u, pk = wl ? s_k[wl - 1] : 0u;      // steps of the earlier words of the group
            const unsigned e = e_end - (be + pe + bit - (unsigned)__popcll(below & 0xAAAAAAAAAAAAAAAAull));
            const unsigned k = k_end - (bk + pk + bit - (unsigned)__popcll(below & 0x5555555555555555ull));
            const unsigned idx = cap - 1u - (w0 * 32u + li);
            ae[idx] = e; ak[idx] = k; pf[idx] = (uint8_t)((unsigned)(wd >> (2u * bit)) & 3u);
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
void k2_launch_fill(const BatchDev &B, const void *bc, const void *fc, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    hipLaunchKernelGGL(k2_fill6, dim3((B.n_reads + K2_FILL_W - 1) / K2_FILL_W), dim3(64 * K2_FILL_W), 0, st, B, (const BandConsts *)bc, f);
}
void k2_launch_chase(const BatchDev &B, uint8_t *path_from, hipStream_t st) {
    hipLaunchKernelGGL(k2_chase, dim3((B.n_reads + K2_CHASE_W - 1) / K2_CHASE_W), dim3(64 * K2_CHASE_W), 0, st, B);
    hipLaunchKernelGGL(k2_expand, dim3(B.n_reads), dim3(256), 0, st, B, path_from);
}
void k2_launch_post(const BatchDev &B, const uint8_t *path_from, float *path_lp, const void *fc, hipStream_t st) {
    const FillConsts f = *reinterpret_cast<const FillConsts *>(fc);
    hipLaunchKernelGGL(k2_post, dim3(B.n_reads), dim3(256), 0, st, B, path_from, path_lp, f);
}
