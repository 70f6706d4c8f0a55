// k_hmm.hip -- the `--HMM` branch of detect (detect.cpp:885): llAcrossRead (detect.cpp:393-574) with its forward
// algorithm sequenceProbability (detect.cpp:235-378).
//
// Per read: every reference T at least 2 W from either end (getPOIs :381) is a position of interest (POI).  For a POI the
// reference gathers the rough-aligned events whose query k-mer lies in [refToQuery[pos - W], refToQuery[pos + W]) and runs
// the forward algorithm twice over 2 W positions x {I, D, M}: once with the analogue emission model substituted for
// T-containing k-mers within +-4 of the centre, once without; the call is the difference of the two log-likelihoods.
//
//   k_hmm_pois     wavefront per read: ordered compaction of the POIs (ascending strand coordinate).
//   k_hmm_forward  THREAD per (POI, pass): the recursion is a serial chain in t and the D states chain in i, so the
//                  work-efficient mapping is one lane per chain; a 20 kb read has ~5 k POIs x 2 passes, a batch millions.
//                  One ascending sweep over i per observation updates I, M and D in place (the previous column's values
//                  at i - 1 are carried in registers), state lives in LDS ([array][i][lane]: conflict-free), the loop over i
//                  is rolled so the ~200 transcendental call sites of an unrolled version do not blow the instruction cache.
//
// log(0) is NaN in the reference (probability.cpp:35-77); here it is -inf: lnProd is a plain add (x + -inf = -inf) and
// lnSum(a, b) = max + log(1 + exp(min - max)) with min == -inf short-circuited, which is the reference's case split.
// Arithmetic order (lnProd nesting, lnSum accumulation order) follows the source statement by statement; the bar is the
// north_star's 1e-3 relative on log-likelihoods (device exp/log vs glibc differ in the last ulps).
#include "dn_dev.h"

#define HMM_W 12                 // detect.cpp:885
#define HMM_N (2 * HMM_W)        // states per kind
#define HMM_SNIP (2 * HMM_W + 9) // readSnippet length (:420)

struct HmmConsts { double D2D, D2M, I2M, M2D, M2I, I2I, ln025, ln05; };     // :245-250 via host libm
struct HmmRead { double iM2M, eM2M, endM; };                                 // :253-254, lnSum(eM2M, M2D) of :366
struct HmmDev {
    const double4 *unl, *ana;    // per 9-mer rank: {mu, 2 sigma^2, log(1 / sqrt(2 sigma^2 pi)), 1 / sqrt(2 sigma^2 pi)}
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

__device__ __forceinline__ double lnsum_(double a, double b) {                // probability.cpp:50-77
    const double hi = fmax(a, b), lo = fmin(a, b);
    if (lo == neg_inf_d()) return hi;
    return hi + log(1.0 + exp(lo - hi));
}

__device__ __forceinline__ double emission_(double x, const double4 p) {      // eln(normalPDF(mu, sigma, x)) (:293, probability.cpp:145)
    const double d = x - p.x;
    const double arg = -(d * d) / p.y;
    double e = p.z + arg;
    if (arg < -708.0) {                                   // exp() subnormal or zero in the reference: evaluate the literal chain
        const double q = p.w * exp(arg);
        e = (q == 0.0) ? neg_inf_d() : log(q);
    }
    return e;
}

__global__ __launch_bounds__(64) void k_hmm_forward(BatchDev B, HmmDev H, const HmmRead *hr, HmmConsts hc) {
    __shared__ double S[3][HMM_N][64];                    // I, M, D of the previous / current column (updated in place)
    __shared__ unsigned KI[HMM_N][64];                    // k-mer rank of state i; bit 31: analogue model applies in pass 0
    const int r = blockIdx.y, lane = threadIdx.x;
    const ReadRes &R = B.res[r];
    if (R.status != 0) return;
    const unsigned idx = blockIdx.x * 64 + lane;
    const unsigned k = idx >> 1, pass = idx & 1u;         // pass 0: analogue, 1: thymidine
    const unsigned npoi = H.n_poi[r];
    if (k >= npoi) return;
    const uint64_t f0 = B.ref_off[r];
    const unsigned pos = H.poi[f0 + k];
    const char *ref = B.refseq + f0;
    // ---- readSnippet fully A/T/G/C (:423-442) + k-mer ranks of the 2 W states ----
    const char *snip = ref + pos - HMM_W;
    bool acgt = true;
    unsigned code[HMM_SNIP];
#pragma unroll
    for (int z = 0; z < HMM_SNIP; z++) {
        const char c = snip[z];
        acgt = acgt && (c == 'A' || c == 'T' || c == 'G' || c == 'C');
        code[z] = base_code(c);
    }
    if (!acgt) { if (pass == 0) { H.ok[f0 + k] = 0; H.n_ev[f0 + k] = 0; } return; }
    {
        unsigned rank = 0, tmask = 0;                     // rolling 18-bit rank; tmask: which of the 9 bases are T
#pragma unroll
        for (int z = 0; z < HMM_SNIP; z++) {
            rank = ((rank << 2) | code[z]) & 0x3ffffu;
            tmask = ((tmask << 1) | (code[z] == 1u ? 1u : 0u)) & 0x1ffu;
            if (z >= 8) {
                const int i = z - 8;
                if (i < HMM_N) {
                    const bool an = i >= HMM_W - 4 && i <= HMM_W + 4 && tmask != 0u;       // :319, BrdUStart/End :544-545 (i >= 1 holds)
                    KI[i][lane] = rank | (an ? 0x80000000u : 0u);
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
    if (ns < 2 * HMM_W - 9) { if (pass == 0) { H.ok[f0 + k] = 0; H.n_ev[f0 + k] = 0; } return; }   // :510
    // a reverse-strand snippet is only put back in forward order when the scan leaves through the break at :476-480,
    // i.e. when some aligned pair lies below the window; otherwise it stays in descending order
    const bool descending = B.is_rev[r] != 0 && j0 == 0;
    const double shift = R.shift, scale = R.scale;
    const double iM2M = hr[r].iM2M, eM2M = hr[r].eM2M;
    const double NI = neg_inf_d();
    // ---- initialisation (:257-271) ----
    {
        double d = hc.ln025;                               // lnProd(start_prev = 0, eln(0.25))
        for (int i = 0; i < HMM_N; i++) { S[0][i][lane] = NI; S[1][i][lane] = NI; S[2][i][lane] = d; d = d + hc.D2D; }
    }
    double firstI_prev = NI, start_prev = 0.0;
    const bool use_an = pass == 0;
    // ---- recursion (:277-358) ----
    for (int jj = 0; jj < j1 - j0; jj++) {
        const int j = descending ? (j1 - 1 - jj) : (j0 + jj);
        const double ev = evm[ae[j]];
        if (!(ev > 0. && ev < 250.0)) continue;
        const double x = (ev - shift) / scale;
        // position 0 (:286-311): always the unlabelled model
        unsigned ki = KI[0][lane];
        double match = emission_(x, H.unl[ki & 0x3ffffu]);
        const double firstI_curr = lnsum_(start_prev + hc.ln025, firstI_prev + hc.ln025);            // :297-298 (insProb = 0)
        double oI = S[0][0][lane], oM = S[1][0][lane], oD = S[2][0][lane];
        double nI = lnsum_(oI + hc.I2I, oM + hc.M2I);                                                // :301-302
        double nM = lnsum_(lnsum_((firstI_prev + hc.ln05) + match, (oM + iM2M) + match), (start_prev + hc.ln05) + match);   // :305-307
        double nD = firstI_curr + hc.ln025;                                                          // :310-311 (start -> D is log 0)
        S[0][0][lane] = nI; S[1][0][lane] = nM; S[2][0][lane] = nD;
        double pI = oI, pM = oM, pD = oD, cM = nM, cD = nD;
        for (int i = 1; i < HMM_N; i++) {
            ki = KI[i][lane];
            const double4 prm = (use_an && (ki & 0x80000000u)) ? H.ana[ki & 0x3ffffu] : H.unl[ki & 0x3ffffu];
            match = emission_(x, prm);
            oI = S[0][i][lane]; oM = S[1][i][lane]; oD = S[2][i][lane];
            nI = lnsum_(oI + hc.I2I, oM + hc.M2I);                                                   // :336-337
            nM = lnsum_(lnsum_(lnsum_((pI + hc.I2M) + match, (pM + eM2M) + match), (oM + iM2M) + match), (pD + hc.D2M) + match);   // :340-343
            nD = lnsum_(cM + hc.M2D, cD + hc.D2D);                                                   // :349-350
            S[0][i][lane] = nI; S[1][i][lane] = nM; S[2][i][lane] = nD;
            pI = oI; pM = oM; pD = oD; cM = nM; cD = nD;
        }
        firstI_prev = firstI_curr;
        start_prev = NI;                                   // start_curr is log 0 (:261, :357)
    }
    // ---- termination (:362-367) ----
    const double fwd = lnsum_(lnsum_(S[2][HMM_N - 1][lane] + 0.0, S[1][HMM_N - 1][lane] + hr[r].endM), S[0][HMM_N - 1][lane] + hc.I2M);
    if (pass == 0) { H.la[f0 + k] = fwd; H.ok[f0 + k] = 1; H.n_ev[f0 + k] = ns; }
    else H.lt[f0 + k] = fwd;
}

void k_hmm_launch(const BatchDev &B, const void *hdev, const void *hreads, const void *hconsts, unsigned max_ref, hipStream_t st) {
    const HmmDev &H = *(const HmmDev *)hdev;
    hipLaunchKernelGGL(k_hmm_pois, dim3(B.n_reads), dim3(64), 0, st, B, H);
    // upper bound on POIs per read = reference length; blocks beyond a read's POI count exit at once
    hipLaunchKernelGGL(k_hmm_forward, dim3((2 * max_ref + 63) / 64, B.n_reads), dim3(64), 0, st, B, H, (const HmmRead *)hreads,
                       *(const HmmConsts *)hconsts);
}
