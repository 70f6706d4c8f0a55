// k_hmm.hip -- the `--HMM` branch of detect (detect.cpp:885): llAcrossRead (detect.cpp:393-574) with its forward
// algorithm sequenceProbability (detect.cpp:235-378).
//
// Per read: every reference T at least 2 W from either end (getPOIs :381) is a position of interest (POI).  For a POI the
// reference gathers the rough-aligned events whose query k-mer lies in [refToQuery[pos - W], refToQuery[pos + W]) and runs
// the forward algorithm twice over 2 W positions x {I, D, M}: once with the analogue emission model substituted for
// T-containing k-mers within +-4 of the centre, once without; the call is the difference of the two log-likelihoods.
//
//   k_hmm_pois     wavefront per read: ordered compaction of the POIs (ascending strand coordinate).
//   k_hmm_forward  four lanes per (POI, pass) chain, six states each, skewed in time (see the comment above the kernel);
//                  a 20 kb read has ~5 k POIs x 2 passes, a batch ten million chains.  One ascending sweep over the states per
//                  observation updates I, M and D in place (the previous column's values at i - 1 are carried along).
//
// The recursion runs in the probability domain with exact binary rescaling (see k_hmm_forward); the grouping of products
// and sums follows the source statement by statement; the bar is the north_star's 1e-3 relative on log-likelihoods (observed
// ~1e-12 against the log-space restatement in oracle/).
#include "dn_dev.h"

#define HMM_W 12                 // detect.cpp:885
#define HMM_N (2 * HMM_W)        // states per kind
#define HMM_SNIP (2 * HMM_W + 9) // readSnippet length (:420)

struct HmmConsts { double D2D, D2M, I2M, M2D, M2I, I2I, ln025, ln05; };     // :245-250 as PROBABILITIES (exp of the log values, host libm)
struct HmmRead { double iM2M, eM2M, endM; };                                 // exp of :253-254 and of lnSum(eM2M, M2D) (:366)
struct HmmDev {
    const double4 *unl, *ana;    // per 9-mer rank: {mu, 2 sigma^2, -1 / (2 sigma^2), 1 / sqrt(2 sigma^2 pi)}
    unsigned *poi;               // at ref_off: POIs of the read, ascending
    unsigned *n_poi;             // [n_reads]
    unsigned *n_ev;              // at ref_off, per POI: eventSnippet.size() (0 = no call made)
    unsigned char *ok;           // at ref_off, per POI: 1 = a call was made
    double *la, *lt;             // at ref_off, per POI: log P(analogue), log P(thymidine)
};

__global__ __launch_bounds__(64) void k_hmm_pois(BatchDev B, HmmDev H) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (B.res[r].status != 0) { if (lane == 0) H.n_poi[r] = 0; return; }
    const uint64_t f0 = B.ref_off[r];
    const long L = (long)(B.ref_off[r + 1] - f0);
    const char *ref = B.refseq + f0;
    unsigned n = 0;
    for (long base = 2 * HMM_W; base < L - 2 * HMM_W; base += 64) {          // :385
        const long i = base + lane;
        const bool is = i < L - 2 * HMM_W && ref[i] == 'T';
        const unsigned long long m = __ballot(is);
        if (is) H.poi[f0 + n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
        n += (unsigned)__popcll(m);
    }
    if (lane == 0) H.n_poi[r] = n;
}

// ------------------------------------------------------------------------------------------------
// k_hmm_forward: the forward algorithm in the PROBABILITY domain with exact power-of-two rescaling.
//
// The reference works in log space: every "+" is lnSum = max + log(1 + exp(min - max)) (probability.cpp:50), i.e. five
// exp + five log per (observation, position).  The same quantity is a sum of products of probabilities; here it is
// accumulated as such -- (x t) e for lnProd(lnProd(x, t), e), left-to-right "+" for the lnSum chain, so the grouping of
// the source is kept -- with ONE exp per cell (the Gaussian emission itself, normalPDF of probability.cpp:145).  The
// range problem that log space solves is handled by carrying a binary exponent per lane and rescaling inputs with
// v_ldexp_f64 at every step: a power-of-two scale changes no mantissa bit, so the only rounding is that of the fused
// multiply-adds (~1e-16 each).  Result: log P = log(sum) + E ln 2, within ~1e-12 of the log-space restatement (the
// north_star bar for log-likelihoods is 1e-3 relative), at one eighth of the instructions.  log 0 (NaN in the reference)
// is probability 0.
//
// Mapping: FOUR lanes per (POI, pass) chain, six positions each, skewed in time (a systolic pipeline): lane g works on
// observation t = step - g, so that when it starts a column its left neighbour has just finished the same column.  What
// crosses a lane boundary per step is the neighbour's last position -- previous-column I, M, D, current-column M, D -- and
// the neighbour's exponent (DPP row_shr:1).  The state of a lane is 18 doubles in registers; no LDS.
// ------------------------------------------------------------------------------------------------
#define HMM_G 4                  // lanes per chain
#define HMM_S (HMM_N / HMM_G)    // positions per lane
#define HMM_EZERO (-1000000)     // "exponent" of an all-zero set of values

__device__ __forceinline__ double shr1_d(double v) {       // lane l <- lane l - 1 within a row of 16 (lane 0 of a row gets 0)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x111, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int shr1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ int exp_of(double m, int e) { return m > 0.0 ? __builtin_amdgcn_frexp_exp(m) + e : HMM_EZERO; }

__global__ __launch_bounds__(64, 3) void k_hmm_forward(BatchDev B, HmmDev H, const HmmRead *hr, HmmConsts hc) {
    const int r = blockIdx.y, lane = threadIdx.x;
    const ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const int g = lane & (HMM_G - 1);                      // position group of this lane: positions [6 g, 6 g + 6)
    const unsigned idx = blockIdx.x * (64 / HMM_G) + (lane >> 2);
    const unsigned k = idx >> 1, pass = idx & 1u;         // pass 0: analogue, 1: thymidine
    const unsigned npoi = H.n_poi[r];
    const bool chain = k < npoi;
    const uint64_t f0 = B.ref_off[r];
    const unsigned pos = chain ? H.poi[f0 + k] : (unsigned)(2 * HMM_W);
    const char *ref = B.refseq + f0;
    // ---- readSnippet fully A/T/G/C (:423-442) + k-mer ranks of this lane's six positions ----
    const char *snip = ref + pos - HMM_W;
    bool acgt = true;
    unsigned ki[HMM_S];
    {
        unsigned rank = 0, tmask = 0;                     // rolling 18-bit rank; tmask: which of the 9 bases are T
#pragma unroll
        for (int z = 0; z < HMM_SNIP; z++) {
            const char c = snip[z];
            acgt = acgt && (c == 'A' || c == 'T' || c == 'G' || c == 'C');
            const unsigned code = base_code(c);
            rank = ((rank << 2) | code) & 0x3ffffu;
            tmask = ((tmask << 1) | (code == 1u ? 1u : 0u)) & 0x1ffu;
            if (z >= 8) {
                const int i = z - 8;
                if (i < HMM_N) {
                    const bool an = pass == 0 && i >= HMM_W - 4 && i <= HMM_W + 4 && tmask != 0u;   // :319, BrdUStart/End :544-545
#pragma unroll
                    for (int q = 0; q < HMM_S; q++) if (i == g * HMM_S + q) ki[q] = rank | (an ? 0x80000000u : 0u);
                }
            }
        }
    }
    // ---- events of the window: rough alignment pairs with lo <= query k-mer < hi (:446-507) ----
    const unsigned *r2q = B.ref2query + f0;
    const unsigned lo = r2q[pos - HMM_W], hi = r2q[pos + HMM_W];
    const uint64_t a0 = B.aln_off[r] + R.aln_begin;
    const unsigned *ae = B.aln_event + a0, *ak = B.aln_kmer + a0;
    const int na = (int)R.n_aligned;
    int j0, j1;
    { int a = 0, b = na; while (a < b) { const int m = (a + b) >> 1; if (ak[m] < lo) a = m + 1; else b = m; } j0 = a; }
    { int a = j0, b = na; while (a < b) { const int m = (a + b) >> 1; if (ak[m] < hi) a = m + 1; else b = m; } j1 = a; }
    const double *evm = B.ev_mean + B.ev_off[r];
    unsigned ns = 0;
    for (int j = j0; j < j1; j++) { const double ev = evm[ae[j]]; ns += (ev > 0. && ev < 250.0) ? 1u : 0u; }
    const bool go = chain && acgt && ns >= 2 * HMM_W - 9;                                       // :442, :510
    if (chain && !go && g == 0 && pass == 0) { H.ok[f0 + k] = 0; H.n_ev[f0 + k] = 0; }
    // the four lanes of a chain agree on `go`; a wavefront leaves only when none of its chains runs
    if (__ballot(go) == 0ull) return;
    // a reverse-strand snippet is only put back in forward order when the scan leaves through the break at :476-480,
    // i.e. when some aligned pair lies below the window; otherwise it stays in descending order
    const bool descending = B.is_rev[r] != 0 && j0 == 0;
    const double shift = R.shift, scale = R.scale;
    const double iM2M = hr[r].iM2M, eM2M = hr[r].eM2M;     // probabilities (the exp of the reference's log values)
    const bool first = g == 0;
    // ---- initialisation (:257-271): D_prev[i] = 0.25 * 0.3^i, built left to right; I, M = 0; true value = stored * 2^E ----
    double SI[HMM_S], SM[HMM_S], SD[HMM_S];
    int E = 0;
    {
        double d = hc.ln025;
        for (int i = 0; i < HMM_N; i++) {
#pragma unroll
            for (int q = 0; q < HMM_S; q++) if (i == g * HMM_S + q) { SI[q] = 0.0; SM[q] = 0.0; SD[q] = d; }
            d = d * hc.D2D;
        }
    }
    double firstI_prev = 0.0, start_prev = 1.0;
    // boundary handed to the next lane: this lane's LAST position, previous column (I, M, D) and current column (M, D), exponent
    double bI = 0.0, bM = 0.0, bD = 0.0, bcM = 0.0, bcD = 0.0; int bE = 0;
    int cur = descending ? j1 - 1 : j0;                    // this lane's cursor over the aligned pairs
    const int dir = descending ? -1 : 1;
    const int T = (int)ns;
    for (int step = 0; step < T + HMM_G - 1; step++) {
        // what the left neighbour left behind in the step before (for g == 0 the values are not used)
        double pI0 = shr1_d(bI), pM0 = shr1_d(bM), pD0 = shr1_d(bD), cM0 = shr1_d(bcM), cD0 = shr1_d(bcD);
        const int inE = shr1_i(bE);
        const int t = step - g;
        const bool act = go && t >= 0 && t < T;
        if (act) {
            double ev = evm[ae[cur]];
            while (!(ev > 0. && ev < 250.0)) { cur += dir; ev = evm[ae[cur]]; }      // events outside (0, 250) are not observations (:466)
            cur += dir;
            const double x = (ev - shift) / scale;
            // ---- bring everything this step reads onto one binary scale (exact: only exponents change) ----
            double own = fmax(firstI_prev, start_prev);
#pragma unroll
            for (int q = 0; q < HMM_S; q++) own = fmax(own, fmax(SI[q], fmax(SM[q], SD[q])));
            const double inc = first ? 0.0 : fmax(fmax(pI0, pM0), fmax(pD0, fmax(cM0, cD0)));
            const int eo = exp_of(own, E), ei = exp_of(inc, inE);
            const int En = max(eo, ei) == HMM_EZERO ? E : max(eo, ei);
            const int so = max(E - En, -2000), si = max(inE - En, -2000);
#pragma unroll
            for (int q = 0; q < HMM_S; q++) { SI[q] = ldexp(SI[q], so); SM[q] = ldexp(SM[q], so); SD[q] = ldexp(SD[q], so); }
            firstI_prev = ldexp(firstI_prev, so); start_prev = ldexp(start_prev, so);
            pI0 = ldexp(pI0, si); pM0 = ldexp(pM0, si); pD0 = ldexp(pD0, si); cM0 = ldexp(cM0, si); cD0 = ldexp(cD0, si);
            E = En;
            const double firstI_curr = start_prev * hc.ln025 + firstI_prev * hc.ln025;               // :297-298 (insProb = 1)
            // incoming edge of this lane's first position: for position 0 the start / first-insertion states take the place of
            // the (i - 1) states (:305-311); everything else is the generic recursion (:336-350)
            double pI = first ? firstI_prev : pI0, pM = first ? 0.0 : pM0, pD = first ? start_prev : pD0;
            double cM = first ? 0.0 : cM0, cD = first ? firstI_curr : cD0;
            double tI = first ? hc.ln05 : hc.I2M, tD = first ? hc.ln05 : hc.D2M, tcD = first ? hc.ln025 : hc.D2D;
            const double oI5 = SI[HMM_S - 1], oM5 = SM[HMM_S - 1], oD5 = SD[HMM_S - 1];
#pragma unroll
            for (int q = 0; q < HMM_S; q++) {
                const unsigned kk = ki[q];
                const double4 prm = (kk & 0x80000000u) ? H.ana[kk & 0x3ffffu] : H.unl[kk & 0x3ffffu];
                const double dd = x - prm.x;
                const double match = prm.w * exp((dd * dd) * prm.z);                                 // normalPDF (probability.cpp:145)
                const double oI = SI[q], oM = SM[q], oD = SD[q];
                const double nI = oI * hc.I2I + oM * hc.M2I;                                         // :301-302 / :336-337
                const double nM = ((((pI * tI) * match + (pM * eM2M) * match) + (oM * iM2M) * match) + (pD * tD) * match);   // :305-307 / :340-343
                const double nD = cM * hc.M2D + cD * tcD;                                            // :310-311 / :349-350
                SI[q] = nI; SM[q] = nM; SD[q] = nD;
                pI = oI; pM = oM; pD = oD; cM = nM; cD = nD;
                tI = hc.I2M; tD = hc.D2M; tcD = hc.D2D;
            }
            bI = oI5; bM = oM5; bD = oD5; bcM = SM[HMM_S - 1]; bcD = SD[HMM_S - 1]; bE = E;
            firstI_prev = firstI_curr;
            start_prev = 0.0;                              // the start state is left for good after the first observation (:261, :357)
        }
    }
    // ---- termination (:362-367): the last position lives in the last lane of the chain ----
    if (go && g == HMM_G - 1) {
        const double tot = (SD[HMM_S - 1] + SM[HMM_S - 1] * hr[r].endM) + SI[HMM_S - 1] * hc.I2M;
        const double fwd = tot > 0.0 ? log(tot) + (double)E * 0.6931471805599453 : neg_inf_d();
        if (pass == 0) { H.la[f0 + k] = fwd; H.ok[f0 + k] = 1; H.n_ev[f0 + k] = ns; }
        else H.lt[f0 + k] = fwd;
    }
}

void k_hmm_launch(const BatchDev &B, const void *hdev, const void *hreads, const void *hconsts, unsigned max_ref, hipStream_t st) {
    const HmmDev &H = *(const HmmDev *)hdev;
    hipLaunchKernelGGL(k_hmm_pois, dim3(B.n_reads), dim3(64), 0, st, B, H);
    // upper bound on POIs per read = reference length; blocks beyond a read's POI count exit at once
    hipLaunchKernelGGL(k_hmm_forward, dim3((2 * max_ref + 15) / 16, B.n_reads), dim3(64), 0, st, B, H, (const HmmRead *)hreads,
                       *(const HmmConsts *)hconsts);
}
