// k_scaling.hip -- k-mer ranks, rough quantile scaling, Theil-Sen refinement (gfx950).
//
//   k_ranks     kmer2index (data_IO.cpp:129-141: A0 T1 G2 C3, big-endian base 4, unknown -> 0) for every
//               query / reference 9-mer (event_handling.cpp:578-592), plus the gathered model mean of each
//               query k-mer (feeds the banded kernel without a dependent table lookup).
//   k_quantile  estimateScaling_quantiles (event_handling.cpp:510-541).  The reference sorts both arrays only to
//               read 10 order statistics each (quantileMedians :451-475); here they are found exactly by an
//               8-pass MSB radix select over order-preserving 64-bit keys, 10 targets at once, histograms in
//               LDS.  The OLS of linear_regression (:478-507) is then one thread.
//   k_prep      x_e = (mean_e - shift) / scale in fp64 (the division of event_handling.cpp:130, hoisted:
//               it only depends on the event).
//   k_theilsen  estimateScaling_theilSen (event_handling.cpp:24-110): <= 1000 points, all pairwise slopes,
//               the upper median slope and the median intercept.  A median is an order statistic, so no sort is
//               needed: slopes are regenerated on the fly in four passes (three 11-bit MSB histograms in LDS,
//               then an in-LDS finish on the few survivors that share 33 leading key bits).  fp64 '/' on gfx950 is IEEE, so each slope has the
//               reference's bits.
#include "dn_dev.h"

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ranks(BatchDev B) {
    const int r = blockIdx.y;
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t b0 = B.base_off[r];
    const unsigned nb = (unsigned)(B.base_off[r + 1] - b0);
    const uint64_t f0 = B.ref_off[r];
    const unsigned nf = (unsigned)(B.ref_off[r + 1] - f0);
    if (nb >= DN_K && i < nb - DN_K + 1) {
        const char *s = B.basecall + b0 + i;
        unsigned v = 0;
#pragma unroll
        for (int j = 0; j < DN_K; j++) v = v * 4u + base_code(s[j]);
        B.rank_q[b0 + i] = v;
        B.mu_q[b0 + i] = B.model_mean[v];
    }
    if (nf >= DN_K && i < nf - DN_K + 1) {
        const char *s = B.refseq + f0 + i;
        unsigned v = 0;
#pragma unroll
        for (int j = 0; j < DN_K; j++) v = v * 4u + base_code(s[j]);
        B.rank_r[f0 + i] = v;
    }
    if (i == 0) {
        B.res[r].n_kq = nb >= DN_K ? nb - DN_K + 1 : 0;
        B.res[r].n_kr = nf >= DN_K ? nf - DN_K + 1 : 0;
    }
}

// wave-aggregated histogram update: lanes of a wavefront that hit the same bin are combined into ONE LDS atomic.
// Scaled signal and slope values share their leading key bits, so plain atomics would serialise on one address.
__device__ __forceinline__ void hist_add_agg(unsigned *hist, unsigned bin, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(active && bin == b);
        if (lane == leader) atomicAdd(&hist[b], (unsigned)__popcll(same));
        todo &= ~same;
    }
}

// ------------------------------------------------------------------------------------------------
// 10-target radix select over 32-bit keys (NBITS significant), 8 bits per pass from the top.  key(i) is supplied by a
// functor; out[t] receives the selected KEY of target t (rank ((t m + (t + 1) m) / 2, quantileMedians :467-470).
// Both inputs of estimateScaling_quantiles reduce to 32-bit keys without changing any order statistic:
//   * event means are floats widened to double (event_handling.cpp:549-575 copies event_t.mean), so the order-preserving
//     map of their FLOAT bits orders them exactly like the doubles;
//   * model levels are looked up in a 4^9 table: their position in the sorted table (computed once at load) is an 18-bit key.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fkey(float x) {                // order-preserving float -> u32 (-0 sorts just below +0)
    const unsigned u = (unsigned)__float_as_int(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) { return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k)); }

template <int NBITS, class F>
__device__ __forceinline__ void select10(F key, unsigned n, unsigned *out10, unsigned (*hist)[256], unsigned *prefix, unsigned *rank_in) {
    const int tid = threadIdx.x;
    const unsigned m = n / 10u;                                   // quantileMedians :467
    if (tid < 10) {
        prefix[tid] = 0u;
        rank_in[tid] = ((unsigned)tid * m + (unsigned)(tid + 1) * m) / 2u;   // :470
    }
    __syncthreads();
    constexpr int NPASS = (NBITS + 7) / 8;
    for (int pass = NPASS - 1; pass >= 0; pass--) {
        for (int j = tid; j < 10 * 256; j += 256) (&hist[0][0])[j] = 0u;
        __syncthreads();
        // targets are in rank order, so equal prefixes are adjacent: one histogram per distinct prefix (group).
        // In the leading passes every target shares one prefix and the element loop does a single compare + atomic.
        const int sh = 8 * (pass + 1);
        unsigned gp[10]; int ng = 0;
#pragma unroll
        for (int t = 0; t < 10; t++) {
            const unsigned ph = (pass == NPASS - 1) ? 0u : (prefix[t] >> sh);
            if (ng == 0 || gp[ng - 1] != ph) gp[ng++] = ph;
        }
        for (unsigned i = tid; i < n; i += 256) {
            const unsigned k = key(i);
            const unsigned digit = (k >> (8 * pass)) & 255u;
            const unsigned hi = (pass == NPASS - 1) ? 0u : (k >> sh);
            for (int g = 0; g < ng; g++)
                if (hi == gp[g]) { atomicAdd(&hist[g][digit], 1u); break; }
        }
        __syncthreads();
        if (tid < 10) {
            const unsigned ph = (pass == NPASS - 1) ? 0u : (prefix[tid] >> sh);
            int g = 0;
            for (int q = 0; q < ng; q++) if (gp[q] == ph) g = q;
            unsigned want = rank_in[tid], cum = 0; unsigned d = 255;
            for (unsigned b = 0; b < 256; b++) {
                const unsigned h = hist[g][b];
                if (want < cum + h) { d = b; break; }
                cum += h;
            }
            rank_in[tid] = want - cum;
            prefix[tid] |= d << (8 * pass);
        }
        __syncthreads();
    }
    if (tid < 10) out10[tid] = prefix[tid];
    __syncthreads();
}

__global__ __launch_bounds__(256, 4) void k_quantile(BatchDev B) {     // 4 waves per SIMD: a 1000-read batch must be resident at once
    __shared__ unsigned hist[10][256];
    __shared__ unsigned prefix[10];
    __shared__ unsigned rank_in[10];
    __shared__ unsigned skey[10], mkey[10];
    __shared__ double sq[10], mq[10];
    const int r = blockIdx.x;
    ReadRes &R = B.res[r];
    const unsigned ne = R.n_events, nr = R.n_kr;
    if (ne == 0 || nr == 0) {
        if (threadIdx.x == 0) { R.q_shift = 0.; R.q_scale = 1.; R.status = 5; }
        return;
    }
    const double *em = B.ev_mean + B.ev_off[r];
    const unsigned *rr = B.rank_r + B.ref_off[r];
    const unsigned *mpos = B.model_pos;
    select10<32>([=](unsigned i) { return fkey((float)em[i]); }, ne, skey, hist, prefix, rank_in);     // signal quantiles :532
    select10<18>([=](unsigned i) { return mpos[rr[i]]; }, nr, mkey, hist, prefix, rank_in);            // model quantiles  :533
    if (threadIdx.x < 10) { sq[threadIdx.x] = (double)fkey_inv(skey[threadIdx.x]); mq[threadIdx.x] = B.model_sorted[mkey[threadIdx.x]]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // linear_regression(x = model quantiles, y = signal quantiles) :478-507, :535
        double sx = 0., sx2 = 0., sy = 0., sxy = 0.;
        const int n = 10;
        for (int i = 0; i < n; i++) {
            sx = sx + mq[i];
            sx2 = sx2 + mq[i] * mq[i];
            sy = sy + sq[i];
            sxy = sxy + mq[i] * sq[i];
        }
        const double slope = ((double)n * sxy - sx * sy) / ((double)n * sx2 - sx * sx);
        const double icpt = (sy - slope * sx) / (double)n;
        R.q_shift = icpt;                                         // :537
        R.q_scale = slope;                                        // :538
        R.status = 0;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(BatchDev B) {
    const int r = blockIdx.y;
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    const ReadRes &R = B.res[r];
    if (e >= R.n_events) return;
    const uint64_t e0 = B.ev_off[r];
    B.ev_x[e0 + e] = (B.ev_mean[e0 + e] - R.q_shift) / R.q_scale; // event_handling.cpp:130
}

// ------------------------------------------------------------------------------------------------
// Theil-Sen
// ------------------------------------------------------------------------------------------------
#define TS_MAXP 1000
#define TS_CAND 2048       // candidate keys kept in LDS (also holds the 2 x 2048 second-level counters of pass 1)

// visit every pair (a < b) of the np points: wave w takes rows a = w, w+4, ...; lanes stride over b
template <class F>
__device__ __forceinline__ void for_each_slope(const double *x, const double *y, unsigned np, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned a = wave; a + 1 < np; a += 4) {
        const double xa = x[a], ya = y[a];
        for (unsigned b0 = a + 1; b0 < np; b0 += 64) {
            const unsigned b = b0 + lane;
            const bool act = b < np;
            const double s = act ? (ya - y[b]) / (xa - x[b]) : 0.0;   // event_handling.cpp:70-73, IEEE fp64 division
            // 0 / 0 (two points with equal signal AND equal level) is NaN, on which std::sort is undefined; the oracle and this
            // kernel define it as sorting LAST.  The sign of that NaN is the hardware's business (the division here returns it with
            // the sign bit set, whose order-preserving key would sort FIRST and shift every rank by one), so the key is forced.
            f(s != s ? ~0ull : dkey(s), act);
        }
    }
}

__global__ __launch_bounds__(256) void k_theilsen(BatchDev B, int force_full_histogram /* diagnostics: take the general path */) {
    __shared__ double x[TS_MAXP], y[TS_MAXP];
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long cand[TS_CAND];
    __shared__ unsigned ncand, sel_digit, sel_digit2, sel_rank;
    __shared__ unsigned long long slope_key, icpt_sel;
    __shared__ unsigned cnt0;
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    ReadRes &R = B.res[r];
    const uint64_t a0 = B.aln_off[r];
    const double *sig = B.cl_sig + a0;
    const unsigned *rk = B.cl_rank + a0;
    const unsigned n = R.n_cleaned;
    const double shift = R.q_shift, scale = R.q_scale;
    const uint64_t b0 = B.base_off[r];
    const unsigned nbase = (unsigned)(B.base_off[r + 1] - b0);
    if (tid == 0) {
        R.events_per_base = (double)R.n_scrappie / (double)(nbase - DN_K);   // :606
        R.ts_slope = __longlong_as_double(0x7ff8000000000000ll);
        R.ts_intercept = __longlong_as_double(0x7ff8000000000000ll);
        R.shift = shift; R.scale = scale;
    }
    if (R.status == 3 || R.status == 5) return;                   // nothing aligned
    if (n < TS_MAXP) return;                                      // :33 (such a read already failed the n_cleaned QC, :438)
    const unsigned eff = n - 100u;                                // trimSize 50 at both ends :35
    unsigned skip = 1, np = eff;
    if (eff > TS_MAXP) { skip = eff / TS_MAXP; np = TS_MAXP; }
    for (unsigned j = tid; j < np; j += 256) {
        const unsigned i = 50u + j * skip;
        x[j] = (sig[i] - shift) / scale;                          // :51
        y[j] = B.model_mean[rk[i]];                               // :58-61
    }
    const unsigned long long ns = (unsigned long long)np * (np - 1) / 2ull;
    unsigned want = (unsigned)(ns / 2ull);                        // :78 slopes[size/2]
    // ---- pass 1: 11-bit MSB histogram (sign + exponent) of every slope, and -- because x is already roughly scaled, so
    //      the median slope is close to 1 -- a second-level histogram of the next 11 bits for the two exponent buckets that
    //      straddle 1.0 ([0.5, 1) and [1, 2)).  When the median falls in one of them (the normal case) 22 leading bits are
    //      known after ONE regeneration of the 499 500 slopes. ----
    unsigned long long pref = 0ull;                               // selected leading bits so far (right aligned)
    int bits_done = 0;
    const unsigned GA = (unsigned)(dkey(0.75) >> 53), GB = (unsigned)(dkey(1.5) >> 53);
    unsigned *hist2 = reinterpret_cast<unsigned *>(cand);         // 2 x 2048 counters, dead before the candidates are collected
    for (int j = tid; j < 2048; j += 256) { hist[j] = 0; hist2[j] = 0; hist2[2048 + j] = 0; }
    __syncthreads();
    // The first level only has to say WHICH exponent bucket holds the median, and it is almost always one of the two around 1:
    // instead of a 2048-bin histogram (a data-dependent loop of wave-aggregated LDS atomics per 64 slopes) the common pass just
    // counts the slopes below / inside the two buckets with ballots.  Only if the median lies outside them is the full
    // first-level histogram built in an extra pass.
    unsigned nBelow = 0, nA = 0, nB = 0;                          // wave-uniform counters
    for_each_slope(x, y, np, [&](unsigned long long k, bool act) {
        const unsigned top = (unsigned)(k >> 53);
        const bool inB = act && top == GB, inA = act && top == GA && !inB;   // (0.75 and 1.5 share their 10 leading exponent bits: GA == GB)
        nBelow += (unsigned)__popcll(__ballot(act && top < GA));
        nA += (unsigned)__popcll(__ballot(inA)); nB += (unsigned)__popcll(__ballot(inB));
        if (inA || inB) atomicAdd(&hist2[(inB ? 2048u : 0u) + ((unsigned)(k >> 42) & 2047u)], 1u);
    });
    if ((tid & 63) == 0) { atomicAdd(&hist[0], nBelow); atomicAdd(&hist[1], nA); atomicAdd(&hist[2], nB); }   // hist[] is free here
    __syncthreads();
    const unsigned tBelow = hist[0], tA = hist[1], tB = hist[2];
    const bool around_one = !force_full_histogram && want >= tBelow && want < tBelow + tA + tB;     // block-uniform
    __syncthreads();
    if (!around_one) {
        for (int j = tid; j < 2048; j += 256) hist[j] = 0;
        __syncthreads();
        for_each_slope(x, y, np, [&](unsigned long long k, bool act) { hist_add_agg(hist, (unsigned)(k >> 53), act); });
        __syncthreads();
    }
    if (tid == 0) {
        unsigned d, rank, d2 = 0xffffffffu;
        if (around_one) {
            const bool inGA = want < tBelow + tA;                   // tA counts top == GA && top != GB only
            d = inGA ? GA : GB;
            rank = want - tBelow - (inGA ? 0u : tA);
            const unsigned *h2 = hist2 + (inGA ? 0u : 2048u);
            unsigned c2 = 0; d2 = 2047;
            for (unsigned b = 0; b < 2048; b++) { if (rank < c2 + h2[b]) { d2 = b; break; } c2 += h2[b]; }
            cnt0 = h2[d2];                                        // slopes sharing the 22 leading bits
            rank -= c2;
        } else {
            unsigned cum = 0; d = 2047;
            for (unsigned b = 0; b < 2048; b++) { if (want < cum + hist[b]) { d = b; break; } cum += hist[b]; }
            rank = want - cum;                                    // d is neither GA nor GB here: no second level
        }
        sel_digit = d; sel_digit2 = d2; sel_rank = rank;
    }
    __syncthreads();
    pref = sel_digit; want = sel_rank; bits_done = 11;
    if (sel_digit2 != 0xffffffffu) { pref = (pref << 11) | sel_digit2; bits_done = 22; }
    const bool collect22 = bits_done == 22 && cnt0 <= TS_CAND;
    __syncthreads();
    // ---- further 11-bit passes until 33 bits are fixed (skipped when the 22-bit bucket already fits the candidate buffer) ----
    while (!collect22 && bits_done < 33) {
        const int lo_bit = 53 - bits_done;                        // digit = key bits [lo_bit+10 : lo_bit]
        for (int j = tid; j < 2048; j += 256) hist[j] = 0;
        __syncthreads();
        {
            const unsigned long long want_hi = pref;
            for_each_slope(x, y, np, [&](unsigned long long k, bool act) {
                if (act && (k >> (lo_bit + 11)) == want_hi) atomicAdd(&hist[(unsigned)(k >> lo_bit) & 2047u], 1u);
            });
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0, d = 2047;
            for (unsigned b = 0; b < 2048; b++) { if (want < cum + hist[b]) { d = b; break; } cum += hist[b]; }
            sel_digit = d; sel_rank = want - cum;
        }
        __syncthreads();
        pref = (pref << 11) | sel_digit;
        want = sel_rank;
        bits_done += 11;
        __syncthreads();
    }
    // ---- collect the survivors (they share bits_done leading bits), finish by rank counting in LDS ----
    const int rest = 64 - bits_done;
    if (tid == 0) ncand = 0;
    __syncthreads();
    {
        // The survivors lie in [lo, hi] (the doubles whose key starts with `pref`), a window 2^-11 wide around the median.  A
        // reciprocal estimate (v_rcp_f64, ~26 bits) is enough to discard the other 99.7 % of the pairs without the ~25-
        // instruction IEEE division; whatever falls inside the window widened by 2^-16 is divided exactly and tested exactly.
        // Pairs with dx == 0 give +-inf / NaN estimates and are discarded, which is right: their quotient is +-inf / NaN and
        // the window is finite (checked: otherwise the plain loop runs).
        const double lo_d = dkey_inv(pref << rest), hi_d = dkey_inv(((pref + 1ull) << rest) - 1ull);
        const double mag = fmax(fabs(lo_d), fabs(hi_d));
        const bool finite_window = mag < 1e300 && lo_d == lo_d && hi_d == hi_d;
        if (finite_window) {
            const double wlo = lo_d - mag * 0x1p-16, whi = hi_d + mag * 0x1p-16;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            for (unsigned a = wave; a + 1 < np; a += 4) {
                const double xa = x[a], ya = y[a];
                for (unsigned b0 = a + 1; b0 < np; b0 += 64) {
                    const unsigned b = b0 + lane;
                    if (b >= np) continue;
                    const double dy = ya - y[b], dx = xa - x[b];
                    const double est = dy * __builtin_amdgcn_rcp(dx);
                    if (est >= wlo && est <= whi) {
                        const unsigned long long k = dkey(dy / dx);                  // event_handling.cpp:70-73, IEEE fp64 division
                        if ((k >> rest) == pref) {
                            const unsigned slot = atomicAdd(&ncand, 1u);
                            if (slot < TS_CAND) cand[slot] = k;
                        }
                    }
                }
            }
        } else {
            for_each_slope(x, y, np, [&](unsigned long long k, bool act) {
                if (act && (k >> rest) == pref) {
                    const unsigned slot = atomicAdd(&ncand, 1u);
                    if (slot < TS_CAND) cand[slot] = k;
                }
            });
        }
    }
    __syncthreads();
    const unsigned nc = ncand;
    if (nc <= TS_CAND) {
        for (unsigned i = tid; i < nc; i += 256) {
            const unsigned long long k = cand[i];
            unsigned less = 0, eq = 0;
            for (unsigned j = 0; j < nc; j++) { const unsigned long long o = cand[j]; less += (o < k); eq += (o == k); }
            if (less <= want && want < less + eq) slope_key = k;   // all writers hold the same key value
        }
        __syncthreads();
    } else {
        // > TS_CAND slopes share 33 leading bits (degenerate data): resolve the remaining bits one per pass
        unsigned long long full = pref << rest;
        for (int bit = rest - 1; bit >= 0; bit--) {
            if (tid == 0) cnt0 = 0;
            __syncthreads();
            unsigned local = 0;
            for_each_slope(x, y, np, [&](unsigned long long k, bool act) {
                if (act && (k >> (bit + 1)) == (full >> (bit + 1)) && !((k >> bit) & 1ull)) local++;
            });
            atomicAdd(&cnt0, local);
            __syncthreads();
            const unsigned c0 = cnt0;
            if (want >= c0) { want -= c0; full |= (1ull << bit); }
            __syncthreads();
        }
        if (tid == 0) slope_key = full;
        __syncthreads();
    }
    const double slope_med = dkey_inv(slope_key);
    unsigned long long *icpt_key = cand;                          // the candidate keys are dead: same LDS (4 blocks per CU fit)
    __syncthreads();
    // ---- intercepts :79-87 : median of y - slope*x over <= 1000 points (rank np/2) ----
    for (unsigned j = tid; j < np; j += 256) {
        const double prod = slope_med * x[j];                     // product rounded, then subtracted (no contraction)
        icpt_key[j] = dkey(y[j] - prod);
    }
    __syncthreads();
    {
        const unsigned wanti = np / 2u;
        for (unsigned i = tid; i < np; i += 256) {
            const unsigned long long k = icpt_key[i];
            unsigned less = 0, eq = 0;
            for (unsigned j = 0; j < np; j++) { const unsigned long long o = icpt_key[j]; less += (o < k); eq += (o == k); }
            if (less <= wanti && wanti < less + eq) icpt_sel = k;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double icpt_med = dkey_inv(icpt_sel);
        R.ts_slope = slope_med; R.ts_intercept = icpt_med;
        if (slope_med == 0.) {                                    // :90-95
            R.shift = -1.; R.scale = -1.;
            if (R.status == 0) R.status = 2;
        } else {
            const double scale_corr = 1. / slope_med;             // :98-101
            const double shift_corr = -icpt_med / slope_med;
            R.shift = shift + (shift_corr * scale);
            R.scale = scale * scale_corr;
        }
    }
}

// ------------------------------------------------------------------------------------------------
void ks_launch_ranks(const BatchDev &B, unsigned max_len, hipStream_t st) {
    hipLaunchKernelGGL(k_ranks, dim3((max_len + 255) / 256, B.n_reads), dim3(256), 0, st, B);
}
void ks_launch_quantile(const BatchDev &B, hipStream_t st) {
    hipLaunchKernelGGL(k_quantile, dim3(B.n_reads), dim3(256), 0, st, B);
}
void ks_launch_prep(const BatchDev &B, unsigned max_events, hipStream_t st) {
    hipLaunchKernelGGL(k_prep, dim3((max_events + 255) / 256, B.n_reads), dim3(256), 0, st, B);
}
void ks_launch_theilsen(const BatchDev &B, hipStream_t st) {
    // DN_TS_FULL=1: always build the full first-level histogram (the path a median slope outside [0.5, 2) takes; same results)
    static const int full = (getenv("DN_TS_FULL") && atoi(getenv("DN_TS_FULL")) != 0) ? 1 : 0;
    hipLaunchKernelGGL(k_theilsen, dim3(B.n_reads), dim3(256), 0, st, B, full);
}
