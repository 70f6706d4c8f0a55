// k2_fill5_ablation.hip -- NOT part of libdnascent_hip.so.  The offset-keyed predecessor of k2_fill6 (cells keyed by band
// offset, both moves' operands formed with DPP and picked with v_cndmask; 128-byte rows of 100 from-codes + the band corner)
// together with its ABL ablation switches, kept here as the A/B source the round-1 measurements in DESIGN.md s4 refer to
// (k2_fill5 27.9 ms vs k2_fill6 24.3 ms per 1 000 x 20 kb reads).  It is a fragment: it needs the shift helpers, FillConsts,
// BandConsts and cdptr_t of dnascent_amd/csrc/k2_banded.hip around it to compile.
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
