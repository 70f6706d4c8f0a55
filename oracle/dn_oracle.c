/*
 * dn_oracle.c -- CPU restatement of DNAscent `detect`'s per-read numerical path.
 *
 * TEST INFRASTRUCTURE ONLY (see dn_oracle.h for the pinning status of every function).
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared  (no -ffast-math, no -march=native:
 * the reference is built with plain -O2, Makefile:6-7, so no FMA contraction may happen here).
 *
 * The arithmetic types of every expression follow the reference exactly (float vs double,
 * evaluation order); where the reference relies on C/C++ implicit conversions the cast is
 * written out and the source line cited.  All citations are relative to /root/reference/src/.
 */
#define _GNU_SOURCE
#include "dn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * probability.cpp
 * ---------------------------------------------------------------------------------------- */
double dno_eexp(double x) { return isnan(x) ? 0.0 : exp(x); }            /* :23-31 */

double dno_eln(double x, int *neg) {                                     /* :35-47 */
    if (x == 0.0) return NAN;
    if (x > 0.0) return log(x);
    if (neg) *neg = 1;                                                   /* reference: throw NegativeLog() */
    return NAN;
}

double dno_lnSum(double a, double b) {                                   /* :50-76 */
    int na = isnan(a), nb = isnan(b);
    if (na || nb) {
        if (na && nb) return NAN;
        return na ? b : a;
    }
    if (a > b) return a + dno_eln(1.0 + dno_eexp(b - a), NULL);
    return b + dno_eln(1.0 + dno_eexp(a - b), NULL);
}

double dno_lnProd(double a, double b) {                                  /* :79-88 */
    if (isnan(a) || isnan(b)) return NAN;
    return a + b;
}

int dno_lnGreaterThan(double a, double b) {                              /* :107-131 */
    int na = isnan(a), nb = isnan(b);
    if (na || nb) {
        if (na || !nb) return 0;      /* :112 */
        if (!na || nb) return 1;      /* :115 */
        return 0;
    }
    return a > b;
}

double dno_normalPDF(double mu, double sigma, double x) {                /* :145-148 */
    return (1.0 / sqrt(2.0 * pow(sigma, 2.0) * M_PI)) * exp(-pow(x - mu, 2.0) / (2.0 * pow(sigma, 2.0)));
}

/* ANALYSIS ONLY (tools/gpu_sequence_fuzz.py, tests/test_gpu_fuzz.py): the Viterbi's emission eln(normalPDF(mu, sigma, x)) evaluated the way the DEVICE lattice does
 * (csrc/k2b_viterbi.hip emission(): log c + arg, arg = -(x - mu)^2 / (2 s^2) through an FMA-corrected reciprocal; the literal exp -> log chain only where exp() would be
 * subnormal) instead of the reference's log(c * exp(arg)).  The two agree to a few ulps; inside low-complexity sequence (identical or periodically repeating k-mers) the
 * Viterbi has exact ties that those last bits decide.  With this switch on, the oracle's labels must equal the device's bit for bit -- which is how the claim "the only
 * difference is the emission's last bits" is TESTED rather than believed.  Off (the default) the oracle is the reference's arithmetic. */
static int g_device_emission = 0;
void dno_set_device_emission(int on) { g_device_emission = on; }
static double viterbi_emission(double mu, double sigma, double x) {
    if (!g_device_emission) return dno_eln(dno_normalPDF(mu, sigma, x), NULL);
    const double s2 = sigma * sigma, d2 = s2 + s2, rd2 = 1.0 / d2, c = 1.0 / sqrt(M_PI * d2), logc = log(c);     /* dn_capi.hip: the host computes them with its libm */
    const double d = x - mu, sq = d * d, n = -sq, q = n * rd2, rem = fma(-q, d2, n), arg = fma(rem, rd2, q);
    if (arg < -708.0) { const double p_ = c * exp(arg); return p_ == 0.0 ? NAN : log(p_); }
    return logc + arg;
}

/* ------------------------------------------------------------------------------------------
 * data_IO.cpp:129-141  kmer2index: A0 T1 G2 C3, big-endian base 4; anything else -> 0
 * (std::map::operator[] default-inserts 0 for an unknown one-letter key).
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t base_code(char c) {
    switch (c) { case 'A': return 0; case 'T': return 1; case 'G': return 2; case 'C': return 3; default: return 0; }
}
uint32_t dno_kmer2index(const char *kmer, unsigned k) {
    uint32_t r = 0;
    for (unsigned i = 0; i < k; i++) r = r * 4u + base_code(kmer[i]);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * pod5.cpp:57-61   pA = ((float)adc + (float)offset) * (float)scale, widened to double
 * ---------------------------------------------------------------------------------------- */
void dno_adc_to_pa(const int16_t *adc, size_t n, float offset, float scale, double *out) {
    for (size_t i = 0; i < n; i++) {
        float v = ((float)adc[i] + offset) * scale;
        out[i] = (double)v;
    }
}

/* ------------------------------------------------------------------------------------------
 * scrappie/event_detection.c
 * ---------------------------------------------------------------------------------------- */
static void tstat_window(const double *sum, const double *sumsq, size_t n, size_t w, float *t) {
    /* compute_tstat :60-115 */
    const float eta = FLT_MIN;
    const float wf = (float)w;
    for (size_t i = 0; i < n; i++) t[i] = 0.0f;                          /* calloc :67 + :76-86 */
    if (n < 2 * w || w < 2) return;
    for (size_t i = w; i <= n - w; i++) {                                /* :89 inclusive upper bound */
        double sum1 = sum[i], sumsq1 = sumsq[i];
        if (i > w) { sum1 -= sum[i - w]; sumsq1 -= sumsq[i - w]; }       /* :92-95 */
        float sum2 = (float)(sum[i + w] - sum[i]);                       /* :96 */
        float sumsq2 = (float)(sumsq[i + w] - sumsq[i]);                 /* :97 */
        float mean1 = (float)(sum1 / (double)wf);                        /* :98 double/float -> double -> float */
        float mean2 = sum2 / wf;                                         /* :99 float/float */
        float m1sq = mean1 * mean1, m2sq = mean2 * mean2;                /* float products */
        float s2w = sumsq2 / wf;
        /* :100-101  ((sumsq1/wf - m1sq) + s2w) - m2sq evaluated in double, then stored to float */
        float var = (float)(((sumsq1 / (double)wf - (double)m1sq) + (double)s2w) - (double)m2sq);
        var = fmaxf(var, eta);                                           /* :104 */
        const float dm = mean2 - mean1;                                  /* :110 */
        float vw = var / wf;
        t[i] = (float)(fabs((double)dm) / sqrt((double)vw));             /* :111 */
    }
}

typedef struct {
    const float *sig; float thr; size_t win; size_t masked_to; long peak_pos; float peak_val; int valid;
} det_t;

size_t dno_detect_events(const double *raw, size_t n, dno_sevent **out,
                         float *tstat1, float *tstat2, uint64_t *peaks_out, size_t *npeaks_out) {
    *out = NULL;
    if (!raw || n == 0) return 0;
    double *sum = (double *)calloc(n + 1, sizeof(double));
    double *sumsq = (double *)calloc(n + 1, sizeof(double));
    float *t1 = (float *)malloc(n * sizeof(float));
    float *t2 = (float *)malloc(n * sizeof(float));
    uint64_t *peaks = (uint64_t *)calloc(n, sizeof(uint64_t));
    /* compute_sum_sumsq :42-47, strictly left to right */
    for (size_t i = 0; i < n; i++) {
        sum[i + 1] = sum[i] + raw[i];
        sumsq[i + 1] = sumsq[i] + raw[i] * raw[i];
    }
    tstat_window(sum, sumsq, n, 3, t1);                                  /* event_detection.h:19-25 defaults */
    tstat_window(sum, sumsq, n, 6, t2);

    /* short_long_peak_detector :122-198 */
    det_t d[2] = {
        { t1, 1.4f, 3, 0, -1, FLT_MAX, 0 },
        { t2, 9.0f, 6, 0, -1, FLT_MAX, 0 },
    };
    const float peak_height = 0.2f;
    size_t npk = 0;
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 2; k++) {
            det_t *q = &d[k];
            if (q->masked_to >= i) continue;                             /* :140 */
            float v = q->sig[i];
            if (q->peak_pos == -1) {                                     /* :146 */
                if (v < q->peak_val) q->peak_val = v;
                else if (v - q->peak_val > peak_height) { q->peak_val = v; q->peak_pos = (long)i; }
            } else {
                if (v > q->peak_val) { q->peak_val = v; q->peak_pos = (long)i; }
                if (k == 0 && q->peak_val > q->thr) {                    /* :166-176 short dominates long */
                    d[1].masked_to = (size_t)q->peak_pos + q->win;
                    d[1].peak_pos = -1; d[1].peak_val = FLT_MAX; d[1].valid = 0;
                }
                if (q->peak_val - v > peak_height && q->peak_val > q->thr) q->valid = 1;   /* :178 */
                if (q->valid && (i - (size_t)q->peak_pos) > q->win / 2) {                  /* :183 */
                    peaks[npk++] = (uint64_t)q->peak_pos;
                    q->peak_pos = -1; q->peak_val = v; q->valid = 0;
                }
            }
        }
    }
    /* create_events :234-266 */
    size_t ne = 1;
    for (size_t i = 0; i < n; i++) if (peaks[i] > 0 && peaks[i] < n) ne++;
    dno_sevent *ev = (dno_sevent *)calloc(ne, sizeof(dno_sevent));
    for (size_t e = 0; e < ne; e++) {
        uint64_t s = (e == 0) ? 0 : peaks[e - 1];
        uint64_t en = (e == ne - 1) ? (uint64_t)n : peaks[e];
        /* create_event :224-229 */
        ev[e].start = s;
        ev[e].length = (float)(en - s);
        ev[e].mean = (float)(sum[en] - sum[s]) / ev[e].length;
        const float dsq = (float)(sumsq[en] - sumsq[s]);
        const float var = dsq / ev[e].length - ev[e].mean * ev[e].mean;
        ev[e].stdv = sqrtf(fmaxf(var, 0.0f));
    }
    if (tstat1) memcpy(tstat1, t1, n * sizeof(float));
    if (tstat2) memcpy(tstat2, t2, n * sizeof(float));
    if (peaks_out) memcpy(peaks_out, peaks, n * sizeof(uint64_t));
    if (npeaks_out) *npeaks_out = npk;
    free(sum); free(sumsq); free(t1); free(t2); free(peaks);
    *out = ev;
    return ne;
}

/* ------------------------------------------------------------------------------------------
 * event_handling.cpp
 * ---------------------------------------------------------------------------------------- */
static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    /* NaN (only reachable through 0/0 slopes, UB under std::sort in the reference) sorts last */
    if (isnan(x)) return isnan(y) ? 0 : 1;
    if (isnan(y)) return -1;
    return (x < y) ? -1 : (x > y);
}

static void quantile_medians(const double *data, size_t n, double *q10) {
    /* quantileMedians :451-475 */
    double *s = (double *)malloc((n ? n : 1) * sizeof(double));
    memcpy(s, data, n * sizeof(double));
    qsort(s, n, sizeof(double), cmp_double);
    unsigned int m = (unsigned int)(n / 10);
    for (int i = 0; i < 10; i++) q10[i] = s[((unsigned)i * m + (unsigned)(i + 1) * m) / 2];
    free(s);
}

void dno_quantile_scaling(const dno_model *m, const double *event_means, size_t ne,
                          const uint32_t *rank_r, size_t nr, double *shift, double *scale) {
    /* estimateScaling_quantiles :510-541 + linear_regression :478-507 */
    double *mm = (double *)malloc((nr ? nr : 1) * sizeof(double));
    for (size_t i = 0; i < nr; i++) mm[i] = m->mean[rank_r[i]];
    double sq[10], mq[10];
    quantile_medians(event_means, ne, sq);
    quantile_medians(mm, nr, mq);
    free(mm);
    double sx = 0., sx2 = 0., sy = 0., sxy = 0.;
    const int n = 10;
    for (int i = 0; i < n; i++) {            /* x = model quantiles, y = signal quantiles (:535) */
        sx = sx + mq[i];
        sx2 = sx2 + mq[i] * mq[i];
        sy = sy + sq[i];
        sxy = sxy + mq[i] * sq[i];
    }
    double slope = (n * sxy - sx * sy) / (n * sx2 - sx * sx);
    double icpt = (sy - slope * sx) / n;
    *shift = icpt;   /* :537 */
    *scale = slope;  /* :538 */
}

int dno_theil_sen(const dno_model *m, const double *sig, const uint32_t *rank, size_t n,
                  double in_shift, double in_scale, double *out_shift, double *out_scale,
                  double *slope_med, double *icpt_med) {
    /* estimateScaling_theilSen :24-110.  returns 1 if refinement ran, 0 if skipped (:33) */
    const size_t maxPoints = 1000, trimSize = 50;
    if (slope_med) *slope_med = NAN;
    if (icpt_med) *icpt_med = NAN;
    *out_shift = in_shift; *out_scale = in_scale;
    if (n < maxPoints) return 0;
    size_t eff = n - 2 * trimSize;
    size_t skip = 1, npts = eff;
    if (eff > maxPoints) { skip = eff / maxPoints; npts = maxPoints; }
    double *x = (double *)malloc(npts * sizeof(double)), *y = (double *)malloc(npts * sizeof(double));
    size_t i = trimSize;
    for (size_t j = 0; j < npts; j++) {
        x[j] = (sig[i] - in_shift) / in_scale;
        y[j] = m->mean[rank[i]];
        i += skip;
    }
    size_t ns = npts * (npts - 1) / 2;
    double *sl = (double *)malloc((ns ? ns : 1) * sizeof(double));
    size_t c = 0;
    for (size_t a = 0; a < npts; a++)
        for (size_t b = a + 1; b < npts; b++) {
            double dy = y[a] - y[b], dx = x[a] - x[b];
            sl[c++] = dy / dx;
        }
    qsort(sl, ns, sizeof(double), cmp_double);
    double smed = sl[ns / 2];
    double *ic = (double *)malloc(npts * sizeof(double));
    for (size_t a = 0; a < npts; a++) {
        double prod = smed * x[a];          /* :83  y - slope*x, product rounded before the subtraction */
        ic[a] = y[a] - prod;
    }
    qsort(ic, npts, sizeof(double), cmp_double);
    double imed = ic[npts / 2];
    free(x); free(y); free(sl); free(ic);
    if (slope_med) *slope_med = smed;
    if (icpt_med) *icpt_med = imed;
    if (smed == 0.) { *out_shift = -1.; *out_scale = -1.; return 1; }   /* :90-95 */
    double scale_corr = 1. / smed;                                      /* :98-101 */
    double shift_corr = -imed / smed;
    *out_shift = in_shift + (shift_corr * in_scale);
    *out_scale = in_scale * scale_corr;
    return 1;
}

typedef struct { double C; double inv_unused; float lisp; } lpm_const;

static inline float lp_match(const dno_model *m, uint32_t rank, double ev_mean, double shift, double scale) {
    /* logProbabilityMatch :116-137 (static model: sigma == 0.14 for every k-mer) */
    double mu = m->mean[rank];
    double sigma = m->sigma;
    double x = (ev_mean - shift) / scale;
    float a = (float)((x - mu) / sigma);                                 /* :133 */
    static float lisp = 0.0f; static int init = 0;
    if (!init) { lisp = (float)log(0.3989422804014327); init = 1; }      /* :134 static const float */
    float t = -0.5f * a;                                                 /* :135  (-0.5f * a) * a in float */
    t = t * a;
    double p = ((double)lisp - log(sigma)) + (double)t;
    return (float)p;                                                     /* :136 return type float */
}

typedef struct { int ev, km; } ekp;

int dno_normalise(const dno_model *m, const dno_read *r, dno_norm *o) {
    memset(o, 0, sizeof(*o));
    o->ts_slope = NAN; o->ts_intercept = NAN;
    const size_t k = DNO_K;
    if (r->n_base < k + 1 || r->n_ref < k || r->n_raw < 16) { o->status = DNO_FAIL_TOO_SHORT; return o->status; }

    /* ---- normaliseEvents :544-575 : segmentation + event build ---- */
    dno_sevent *et = NULL;
    size_t etn = dno_detect_events(r->raw, r->n_raw, &et, NULL, NULL, NULL, NULL);
    o->n_scrappie = etn;
    o->events = (dno_event *)malloc((etn ? etn : 1) * sizeof(dno_event));
    double *event_means = (double *)malloc((etn ? etn : 1) * sizeof(double));
    size_t ne = 0;
    unsigned int rawStart = 0;
    double mean = 0.;
    for (unsigned int i = 0; i < etn; i++) {
        if (et[i].mean > 0.) {
            if (i > 0) {
                uint64_t last = et[i].start - 1;                         /* :563 */
                if (last > (uint64_t)r->n_raw - 1) last = (uint64_t)r->n_raw - 1;
                o->events[ne].mean = mean;
                o->events[ne].raw_start = rawStart;
                o->events[ne].raw_len = (last >= rawStart) ? (uint32_t)(last - rawStart + 1) : 0;
                event_means[ne] = mean;
                ne++;
                mean = (double)et[i].mean;                               /* :570 float -> double */
                rawStart = (unsigned int)et[i].start;
            }
        }
    }
    free(et);
    o->n_events = ne;

    /* ---- k-mer ranks :578-592 ---- */
    o->n_kq = r->n_base - k + 1;
    o->rank_q = (uint32_t *)malloc(o->n_kq * sizeof(uint32_t));
    for (size_t i = 0; i < o->n_kq; i++) o->rank_q[i] = dno_kmer2index(r->basecall + i, k);
    o->n_kr = r->n_ref - k + 1;
    o->rank_r = (uint32_t *)malloc(o->n_kr * sizeof(uint32_t));
    for (size_t i = 0; i < o->n_kr; i++) o->rank_r[i] = dno_kmer2index(r->refseq + i, k);

    /* ---- rough scaling :595 ---- */
    dno_quantile_scaling(m, event_means, ne, o->rank_r, o->n_kr, &o->q_shift, &o->q_scale);
    free(event_means);
    const double shift = o->q_shift, scale = o->q_scale;

    /* ---- adaptive_banded_simple_event_align :148-448 ---- */
    const size_t n_events = ne, n_kmers = o->n_kq;
    const int W = DNO_BANDWIDTH, half = W / 2;
    if (n_events < 1) { o->status = DNO_FAIL_TOO_SHORT; return o->status; }
    const double epk = (double)n_events / (double)n_kmers;               /* :174 */
    const double p_stay = 1 - (1 / (epk + 1));
    const double lp_skip = log(1e-30);
    const double lp_stay = log(p_stay);
    const double lp_step = log(1.0 - exp(lp_skip) - exp(lp_stay));
    const double lp_trim = log(0.01);
    const size_t n_bands = (n_events + 1) + (n_kmers + 1);
    o->n_bands = n_bands;
    float *sc = (float *)malloc(n_bands * W * sizeof(float));
    uint8_t *tr = (uint8_t *)calloc(n_bands * W, 1);
    ekp *ll = (ekp *)malloc(n_bands * sizeof(ekp));
    for (size_t i = 0; i < n_bands * (size_t)W; i++) sc[i] = -INFINITY;
#define SC(b, off) sc[(size_t)(b) * W + (off)]
#define TR(b, off) tr[(size_t)(b) * W + (off)]
    ll[0].ev = half - 1; ll[0].km = -1 - half;                           /* :213-215 */
    ll[1].ev = ll[0].ev + 1; ll[1].km = ll[0].km;
    SC(0, (-1) - ll[0].km) = 0.0f;                                       /* :218-221 */
    { int fo = ll[1].ev - 0; SC(1, fo) = (float)lp_trim; TR(1, fo) = 1; }/* :224-228 */
    uint64_t fills = 0;
    for (size_t b = 2; b < n_bands; b++) {
        float lo = SC(b - 1, 0), hi = SC(b - 1, W - 1);                  /* :237-247 */
        int lo_ob = (lo == -INFINITY), hi_ob = (hi == -INFINITY);
        int right = (lo_ob && hi_ob) ? ((b % 2) == 1) : (lo < hi);
        if (right) { ll[b].ev = ll[b - 1].ev; ll[b].km = ll[b - 1].km + 1; }
        else { ll[b].ev = ll[b - 1].ev + 1; ll[b].km = ll[b - 1].km; }
        int trim_off = (-1) - ll[b].km;                                  /* :256-265 */
        if (trim_off >= 0 && trim_off < W) {
            unsigned int e = (unsigned int)(ll[b].ev - trim_off);
            if (e < n_events) { SC(b, trim_off) = (float)(lp_trim * (double)(e + 1u)); TR(b, trim_off) = 1; }
            else SC(b, trim_off) = -INFINITY;
        }
        int kmin = 0 - ll[b].km, kmax = (int)n_kmers - ll[b].km;         /* :269-278 */
        int emin = ll[b].ev - ((int)n_events - 1), emax = ll[b].ev + 1;
        int omin = kmin > emin ? kmin : emin; if (omin < 0) omin = 0;
        int omax = kmax < emax ? kmax : emax; if (omax > W) omax = W;
        for (int off = omin; off < omax; off++) {
            int e = ll[b].ev - off, km = ll[b].km + off;
            int ou = ll[b - 1].ev - (e - 1);
            int ol = (km - 1) - ll[b - 1].km;
            int od = (km - 1) - ll[b - 2].km;
            float up = (ou >= 0 && ou < W) ? SC(b - 1, ou) : -INFINITY;
            float left = (ol >= 0 && ol < W) ? SC(b - 1, ol) : -INFINITY;
            float diag = (od >= 0 && od < W) ? SC(b - 2, od) : -INFINITY;
            float em = lp_match(m, o->rank_q[km], o->events[e].mean, shift, scale);
            float s_d = (float)(((double)diag + lp_step) + (double)em);  /* :296-298 */
            float s_u = (float)(((double)up + lp_stay) + (double)em);
            float s_l = (float)((double)left + lp_skip);
            float mx = s_d; uint8_t from = 0;                            /* :300-306 */
            mx = s_u > mx ? s_u : mx;
            from = (mx == s_u) ? 1 : from;
            mx = s_l > mx ? s_l : mx;
            from = (mx == s_l) ? 2 : from;
            SC(b, off) = mx; TR(b, off) = from;
            fills++;
        }
    }
    o->fills = fills;

    /* end cell :324-340 */
    float best = -INFINITY; int cur_e = 0; int cur_k = (int)n_kmers - 1; int found = 0;
    for (unsigned int e = 0; e < n_events; e++) {
        size_t b = (size_t)(e + 1) + (size_t)(cur_k + 1);
        int off = ll[b].ev - (int)e;
        if (off >= 0 && off < W) {
            float s = (float)((double)SC(b, off) + (double)(n_events - e) * lp_trim);
            if (s > best) { best = s; cur_e = (int)e; found = 1; }
        }
    }
    o->end_event = cur_e;
    if (!found) {   /* reference would read trace[][] out of bounds here (UB); we fail the read */
        free(sc); free(tr); free(ll);
        o->status = DNO_FAIL_NO_END_CELL; return o->status;
    }

    /* backtrack :356-412 */
    size_t cap = n_events + n_kmers + 2;
    o->aln_event = (uint32_t *)malloc(cap * sizeof(uint32_t));
    o->aln_kmer = (uint32_t *)malloc(cap * sizeof(uint32_t));
    o->cleaned_sig = (double *)malloc(cap * sizeof(double));
    o->cleaned_rank = (uint32_t *)malloc(cap * sizeof(uint32_t));
    double sum_em = 0., n_al = 0.;
    int gap = 0, max_gap = 0;
    double buf_total = 0.; size_t buf_n = 0;     /* signalBuffer + vectorMean (common.h:185) */
    size_t na = 0, nc = 0;
    int bad = 0;
    while (cur_k >= 0 && cur_e >= 0) {
        o->aln_event[na] = (uint32_t)cur_e; o->aln_kmer[na] = (uint32_t)cur_k; na++;
        float lp = lp_match(m, o->rank_q[cur_k], o->events[cur_e].mean, shift, scale);
        sum_em += (double)lp;
        n_al += 1;
        size_t b = (size_t)(cur_e + 1) + (size_t)(cur_k + 1);
        int off = ll[b].ev - cur_e;
        if (off < 0 || off >= W) { bad = 1; break; }   /* reference: UB (path left the band) */
        uint8_t from = TR(b, off);
        if (from == 0) {
            buf_total += o->events[cur_e].mean; buf_n++;
            int32_t q2r = r->query2ref[cur_k];
            if (q2r >= 0) {                                              /* :386 queryToRef.count() */
                unsigned int posOnRef = (unsigned int)q2r;
                if (posOnRef < o->n_kr) {
                    o->cleaned_rank[nc] = o->rank_r[posOnRef];
                    o->cleaned_sig[nc] = buf_total / (double)buf_n;            /* == dno_vector_mean of the buffer, pinned in tests */
                    nc++;
                }
            }
            buf_total = 0.; buf_n = 0;
            cur_k -= 1; cur_e -= 1; gap = 0;
        } else if (from == 1) {
            buf_total += o->events[cur_e].mean; buf_n++;
            cur_e -= 1; gap = 0;
        } else {
            cur_k -= 1; gap += 1; if (gap > max_gap) max_gap = gap;
        }
    }
    free(sc); free(tr); free(ll);
#undef SC
#undef TR
    if (bad) { o->status = DNO_FAIL_NO_END_CELL; return o->status; }
    /* std::reverse :413 */
    for (size_t i = 0; i < na / 2; i++) {
        uint32_t t = o->aln_event[i]; o->aln_event[i] = o->aln_event[na - 1 - i]; o->aln_event[na - 1 - i] = t;
        t = o->aln_kmer[i]; o->aln_kmer[i] = o->aln_kmer[na - 1 - i]; o->aln_kmer[na - 1 - i] = t;
    }
    o->n_aln = na; o->n_cleaned = nc;
    o->avg_log_emission = sum_em / n_al;                                 /* :420 */
    o->spanned = (o->aln_kmer[0] == 0 && o->aln_kmer[na - 1] == n_kmers - 1);
    o->max_gap = max_gap;

    int qc_fail = 0;
    if (o->avg_log_emission < -2.0 || !o->spanned || max_gap > 5) qc_fail = 1;   /* :433 config.h:41 */
    if (nc < 1000) qc_fail = 1;                                                  /* :438 */

    /* ---- Theil-Sen :601-604 (runs even when QC failed; result only matters if it passed) ---- */
    dno_theil_sen(m, o->cleaned_sig, o->cleaned_rank, nc, shift, scale, &o->shift, &o->scale,
                  &o->ts_slope, &o->ts_intercept);
    o->events_per_base = (double)etn / (double)(r->n_base - k);          /* :606 */
    if (qc_fail) o->status = DNO_FAIL_BANDED_QC;
    else if (o->shift == -1.) o->status = DNO_FAIL_SCALING;
    else o->status = DNO_OK;
    return o->status;
}

void dno_norm_free(dno_norm *n) {
    free(n->events); free(n->rank_q); free(n->rank_r); free(n->aln_event); free(n->aln_kmer);
    free(n->cleaned_sig); free(n->cleaned_rank);
    memset(n, 0, sizeof(*n));
}

/* ------------------------------------------------------------------------------------------
 * alignment.cpp:166-516  builtinViterbi  (NaN == log 0)
 * ---------------------------------------------------------------------------------------- */
static inline double vmax(const double *v, int n, int *arg) {
    /* lnVecMax :166-175 / lnArgMax :178-190: first wins ties, NaN never wins */
    double mv = v[0]; int a = 0;
    for (int i = 1; i < n; i++) if (dno_lnGreaterThan(v[i], mv)) { mv = v[i]; a = i; }
    *arg = a; return mv;
}

size_t dno_viterbi(const dno_model *m, const double *obs, size_t T, const char *seq, size_t seqlen,
                   double shift, double scale, double events_per_base,
                   double *score, uint8_t *state, uint32_t *pos, int *err) {
    int neg = 0;
    const double D2D = dno_eln(0.3, &neg), D2M = dno_eln(0.7, &neg), I2M = dno_eln(0.999, &neg);   /* config.h:42 */
    const double M2D = dno_eln(0.0025, &neg), M2I = dno_eln(0.001, &neg), I2I = dno_eln(0.001, &neg);
    const double iM2M = dno_eln(1. - (1. / events_per_base), &neg);      /* :207 */
    const double eM2M = dno_eln(1.0 - M2D - M2I - iM2M, &neg);           /* :208 sic: log values subtracted */
    const double eM2MorD = dno_lnSum(eM2M, M2D);
    const double eOrI = dno_lnSum(eM2M, iM2M);
    if (neg) { if (err) *err = DNO_FAIL_NEGATIVE_LOG; *score = NAN; return 0; }
    if (err) *err = 0;
    const size_t N = seqlen - DNO_K + 1;
    double *mu = (double *)malloc(N * sizeof(double));
    for (size_t i = 0; i < N; i++) mu[i] = m->mean[dno_kmer2index(seq + i, DNO_K)];
    /* predecessor codes per (column, position) */
    uint8_t *bI = (uint8_t *)calloc((T + 1) * N, 1), *bM = (uint8_t *)calloc((T + 1) * N, 1), *bD = (uint8_t *)calloc((T + 1) * N, 1);
    double *buf = (double *)malloc(6 * N * sizeof(double));
    double *Ic = buf, *Dc = buf + N, *Mc = buf + 2 * N, *Ip = buf + 3 * N, *Dp = buf + 4 * N, *Mp = buf + 5 * N;
    for (size_t i = 0; i < 6 * N; i++) buf[i] = NAN;
    double start_prev = 0.0;
    Dp[0] = dno_lnProd(start_prev, M2D);                                 /* :241 */
    for (size_t i = 1; i < N; i++) Dp[i] = Dp[i - 1] + D2D;              /* :246-251 */
    /* bD column 0: i==0 -> START (code 2), else D_{i-1} (code 1) */
    bD[0] = 2; for (size_t i = 1; i < N; i++) bD[i] = 1;
    for (size_t t = 0; t < T; t++) {
        for (size_t i = 0; i < N; i++) { Ic[i] = NAN; Mc[i] = NAN; Dc[i] = NAN; }
        const double xs = (obs[t] - shift) / scale;
        uint8_t *cI = bI + (t + 1) * N, *cM = bM + (t + 1) * N, *cD = bD + (t + 1) * N;
        int a; double v[4];
        double e0 = viterbi_emission(mu[0], m->sigma, xs);               /* :273 */
        v[0] = Ip[0] + I2I + 0.0; v[1] = Mp[0] + M2I + 0.0; v[2] = start_prev + M2I + 0.0;   /* :278-285 */
        Ic[0] = vmax(v, 3, &a); cI[0] = (uint8_t)a;                      /* 0:I0 1:M0 2:START */
        v[0] = Mp[0] + iM2M + e0; v[1] = start_prev + eOrI + e0;         /* :305-310 */
        Mc[0] = vmax(v, 2, &a); cM[0] = (uint8_t)(a == 0 ? 2 : 4);       /* 2:M_i 4:START */
        Dc[0] = dno_lnProd(NAN, M2D); cD[0] = 2;                         /* :326-328 */
        for (size_t i = 1; i < N; i++) {
            double e = viterbi_emission(mu[i], m->sigma, xs);            /* :347 */
            v[0] = Ip[i] + I2I + 0.0; v[1] = Mp[i] + M2I + 0.0;          /* :351-356 */
            Ic[i] = vmax(v, 2, &a); cI[i] = (uint8_t)a;
            v[0] = Ip[i - 1] + I2M + e; v[1] = Mp[i - 1] + eM2M + e;     /* :372-381 */
            v[2] = Mp[i] + iM2M + e;    v[3] = Dp[i - 1] + D2M + e;
            Mc[i] = vmax(v, 4, &a); cM[i] = (uint8_t)a;                  /* 0:I_{i-1} 1:M_{i-1} 2:M_i 3:D_{i-1} */
        }
        for (size_t i = 1; i < N; i++) {                                 /* :405-427 silent deletions */
            v[0] = Mc[i - 1] + M2D; v[1] = Dc[i - 1] + D2D;
            Dc[i] = vmax(v, 2, &a); cD[i] = (uint8_t)a;                  /* 0:M_{i-1} 1:D_{i-1} (same column) */
        }
        double *tp;
        tp = Ip; Ip = Ic; Ic = tp; tp = Mp; Mp = Mc; Mc = tp; tp = Dp; Dp = Dc; Dc = tp;
        start_prev = NAN;                                                /* :432 start_curr stays NaN */
    }
    /* after the loop *_prev hold the last column (== *_curr in the reference; for T==0 the reference reads the
     * never-written *_curr = NaN, reproduced here) */
    double vend[3]; int a;
    if (T == 0) { vend[0] = NAN; vend[1] = NAN; vend[2] = NAN; }
    else { vend[0] = Dp[N - 1]; vend[1] = Mp[N - 1] + eM2MorD; vend[2] = Ip[N - 1] + I2M; }   /* :447-458 */
    *score = vmax(vend, 3, &a);
    /* traceback :460-509 */
    size_t n = 0;
    int st = a;                 /* 0 D, 1 M, 2 I */
    size_t i = N - 1, col = T;
    int done = 0;
    size_t guard = 3 * N * (T + 1) + 8;
    while (!done && guard--) {
        state[n] = (uint8_t)st; pos[n] = (uint32_t)i; n++;
        if (st == 0) {
            uint8_t c = bD[col * N + i];
            if (c == 2) done = 1;
            else if (c == 0) { st = 1; i = i - 1; }
            else { st = 0; i = i - 1; }
            /* column unchanged (backtraceT = t+1) */
        } else if (st == 1) {
            uint8_t c = bM[col * N + i];
            if (c == 4) done = 1;
            else if (c == 0) { st = 2; i = i - 1; }
            else if (c == 1) { st = 1; i = i - 1; }
            else if (c == 2) { st = 1; }
            else { st = 0; i = i - 1; }
            col = col - 1;
        } else {
            uint8_t c = bI[col * N + i];
            if (c == 2) done = 1;
            else if (c == 0) st = 2;
            else st = 1;
            col = col - 1;
        }
    }
    for (size_t q = 0; q < n / 2; q++) {
        uint8_t ts = state[q]; state[q] = state[n - 1 - q]; state[n - 1 - q] = ts;
        uint32_t tp = pos[q]; pos[q] = pos[n - 1 - q]; pos[n - 1 - q] = tp;
    }
    free(mu); free(bI); free(bM); free(bD); free(buf);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * alignment.cpp:519-744  eventalign  (+ reads.h:292-372 feature/tensor packing)
 * ---------------------------------------------------------------------------------------- */
static int ref_defined(const char *s, size_t n) {                        /* referenceDefined :519-544 */
    for (size_t i = 0; i < n; i++) if (s[i] != 'A' && s[i] != 'T' && s[i] != 'G' && s[i] != 'C') return 0;
    return 1;
}

static uint32_t sub_index(const char *p, int n) {                        /* reads.h:112-138 without the +1 */
    uint32_t r = 0; for (int i = 0; i < n; i++) r = r * 4u + base_code(p[i]); return r;
}

int dno_eventalign(const dno_model *m, const dno_read *r, const dno_norm *nm, dno_align *o) {
    memset(o, 0, sizeof(*o));
    const unsigned k = DNO_K, totalW = 50;
    const size_t n_ref = r->n_ref;
    size_t cap = n_ref + 16;
    o->coord = (uint32_t *)malloc(cap * 4); o->query_idx = (uint32_t *)malloc(cap * 4); o->ref_idx = (uint32_t *)malloc(cap * 4);
    o->indel_score = (int32_t *)malloc(cap * 4); o->kmer = (char *)malloc(cap * 9); o->n_signal = (uint32_t *)calloc(cap, 4);
    o->signal = (float *)calloc(cap * DNO_RAWDEPTH, sizeof(float));
    o->core = (float *)malloc(cap * 4); o->residual = (float *)malloc(cap * 4);
    size_t wcap = n_ref / 8 + 16;
    o->win_ref = (uint32_t *)malloc(wcap * 4); o->win_len = (uint32_t *)malloc(wcap * 4); o->win_T = (uint32_t *)malloc(wcap * 4);
    o->win_score = (double *)malloc(wcap * 8);
    /* an event appears once per rough-alignment pair it is part of, so the table can be longer than the signal: grow on demand */
    size_t rcap = r->n_raw + 1024;
    o->row_coord = (uint32_t *)malloc(rcap * 4); o->row_rpos = (uint32_t *)malloc(rcap * 4);
    o->row_val = (double *)malloc(rcap * 8); o->row_kind = (uint8_t *)malloc(rcap);
#define DNO_ROW_ROOM() do { if (o->n_rows == rcap) { rcap *= 2; o->row_coord = (uint32_t *)realloc(o->row_coord, rcap * 4); \
        o->row_rpos = (uint32_t *)realloc(o->row_rpos, rcap * 4); o->row_val = (double *)realloc(o->row_val, rcap * 8); \
        o->row_kind = (uint8_t *)realloc(o->row_kind, rcap); } } while (0)
    size_t tcap = 4096;
    uint32_t *taken = (uint32_t *)malloc(tcap * 4);
    double *tmeans = (double *)malloc(tcap * 8);
    uint8_t *lst = NULL; uint32_t *lpos = NULL; size_t lcap = 0;
    int readHead = 0;
    unsigned int ri = 0;
    int rc = DNO_OK;
    while (ri < n_ref - k + 1) {                                         /* :556 */
        unsigned int toEnd = (unsigned int)(n_ref - ri);
        unsigned int W = toEnd < totalW ? toEnd : totalW;
        if ((double)toEnd > 1.5 * totalW) {                              /* :564 */
            const char *snip = r->refseq + ri;                           /* substr(ri, 1.5*W) = 75 */
            size_t sl = (size_t)(1.5 * W);
            if (!ref_defined(snip, sl)) { ri += W; continue; }
            const double lim = 1.5 * W - k - 1;                          /* :574 evaluated before W changes */
            for (unsigned int i = W; (double)i < lim; i++) {
                double mu = m->mean[dno_kmer2index(snip + i, k)];
                double mb = m->mean[dno_kmer2index(snip + i - 1, k)];
                double mf = m->mean[dno_kmer2index(snip + i + 1, k)];
                double g1 = fabs(mu - mf), g2 = fabs(mu - mb);
                if (g1 > 0.75 && g2 > 0.75) { W = i + k; break; }
            }
        }
        const char *seq = r->refseq + ri;
        if (!ref_defined(seq, W)) { ri += W; continue; }                 /* :599-604 */
        const uint32_t qlo = r->ref2query[ri], qhi = r->ref2query[ri + W - k + 1];
        size_t nt = 0; int first = 1;
        for (unsigned int j = (unsigned int)readHead; j < nm->n_aln; j++) {      /* :611-632 */
            uint32_t q = nm->aln_kmer[j];
            if (qlo <= q && q < qhi) {
                if (first) { readHead = (int)j; first = 0; }
                double em = nm->events[nm->aln_event[j]].mean;
                if (0. < em && em < 250.) {
                    if (nt == tcap) { tcap *= 2; taken = (uint32_t *)realloc(taken, tcap * 4); tmeans = (double *)realloc(tmeans, tcap * 8); }
                    taken[nt] = nm->aln_event[j]; tmeans[nt] = em; nt++;
                }
            }
            if (q >= qhi) break;
        }
        int querySpan = (int)(qhi - qlo);
        int indelScore = querySpan - (int)(W - k + 1);                   /* :635-638 */
        if (nt < 2) { ri += W; continue; }                               /* :641 */
        int coord0 = r->is_reverse ? (r->ref_end - (int)ri - (int)(k / 2)) : (r->ref_start + (int)ri + (int)(k / 2));
        const size_t N = W - k + 1;
        if (nt + N + 2 > lcap) { lcap = 2 * (nt + N + 2); lst = (uint8_t *)realloc(lst, lcap); lpos = (uint32_t *)realloc(lpos, lcap * 4); }
        double vs; int verr = 0;
        size_t nl = dno_viterbi(m, tmeans, nt, seq, W, nm->shift, nm->scale, nm->events_per_base, &vs, lst, lpos, &verr);
        if (verr) { rc = verr; break; }
        if (o->n_windows == wcap) {
            wcap *= 2;
            o->win_ref = (uint32_t *)realloc(o->win_ref, wcap * 4); o->win_len = (uint32_t *)realloc(o->win_len, wcap * 4);
            o->win_T = (uint32_t *)realloc(o->win_T, wcap * 4); o->win_score = (double *)realloc(o->win_score, wcap * 8);
        }
        o->win_ref[o->n_windows] = ri; o->win_len[o->n_windows] = W; o->win_T[o->n_windows] = (uint32_t)nt;
        o->win_score[o->n_windows] = vs;
        o->n_windows++; o->sum_TN += (uint64_t)nt * N; if (!isnan(vs)) o->score_sum += vs;
        size_t lastM_ev = 0, lastM_ref = 0, evIdx = 0;                   /* :655-672 */
        for (size_t i = 0; i < nl; i++) {
            if (lst[i] == 1) { lastM_ev = evIdx; lastM_ref = lpos[i]; }
            if (lst[i] != 0) evIdx++;
        }
        evIdx = 0;
        for (size_t i = 0; i < nl; i++) {                                /* :676-736 */
            if (lst[i] == 0) continue;
            uint32_t p = lpos[i];
            const char *kmerStrand = r->refseq + ri + p;
            unsigned int coord = r->is_reverse ? (unsigned int)(coord0 - (int)p - 1) : (unsigned int)(coord0 + (int)p);
            unsigned int idxRef = ri + p + k / 2;
            unsigned int idxQ = r->ref2query[idxRef];
            if (lst[i] == 1) {
                const dno_event *ev = &nm->events[taken[evIdx]];
                for (uint32_t s = 0; s < ev->raw_len; s++) {
                    double scaled = (r->raw[ev->raw_start + s] - nm->shift) / nm->scale;   /* :705 */
                    /* addSignal reads.h:292-304 (std::map keyed by coord; first touch fixes the metadata) */
                    size_t idx = (size_t)-1;
                    if (o->n_pos > 0) {
                        unsigned int lastc = o->coord[o->n_pos - 1];
                        if (lastc == coord) idx = o->n_pos - 1;
                        else if (r->is_reverse ? (coord > lastc) : (coord < lastc)) {
                            /* non-monotonic coordinate: emulate the map lookup */
                            for (size_t z = o->n_pos; z-- > 0;) if (o->coord[z] == coord) { idx = z; break; }
                        }
                    }
                    if (idx == (size_t)-1) {
                        if (o->n_pos == cap) { rc = DNO_FAIL_TOO_SHORT; goto done; }
                        idx = o->n_pos++;
                        o->coord[idx] = coord; o->query_idx[idx] = idxQ; o->ref_idx[idx] = idxRef;
                        o->indel_score[idx] = indelScore; memcpy(o->kmer + idx * 9, kmerStrand, 9);
                        o->n_signal[idx] = 0;
                    }
                    if (o->n_signal[idx] < DNO_RAWDEPTH) o->signal[idx * DNO_RAWDEPTH + o->n_signal[idx]] = (float)scaled;  /* reads.h:156 */
                    o->n_signal[idx]++;
                    DNO_ROW_ROOM();                                      /* :717-722 (refCoordToCalls is empty in `align`) */
                    o->row_coord[o->n_rows] = coord; o->row_rpos[o->n_rows] = ri + p; o->row_val[o->n_rows] = scaled;
                    o->row_kind[o->n_rows] = 0; o->n_rows++;
                }
            }
            else if (lst[i] == 2 && evIdx < lastM_ev) {                  /* :728-733 insertions before the last match */
                const dno_event *ev = &nm->events[taken[evIdx]];
                for (uint32_t s = 0; s < ev->raw_len; s++) {
                    double scaled = (r->raw[ev->raw_start + s] - nm->shift) / nm->scale;
                    DNO_ROW_ROOM();
                    o->row_coord[o->n_rows] = coord; o->row_rpos[o->n_rows] = ri + p; o->row_val[o->n_rows] = scaled;
                    o->row_kind[o->n_rows] = 1; o->n_rows++;
                }
            }
            evIdx++;
        }
        readHead += (int)lastM_ev + 1;                                   /* :739-740 */
        ri += (unsigned int)lastM_ref + 1;
    }
done:
    for (size_t i = 0; i < o->n_pos; i++) {
        const char *km = o->kmer + i * 9;
        o->core[i] = (float)(sub_index(km + 2, 5) + 1u);                 /* reads.h:112-124 */
        char rs[4] = { km[0], km[1], km[7], km[8] };
        o->residual[i] = (float)(sub_index(rs, 4) + 1u);                 /* reads.h:125-138 */
    }
#undef DNO_ROW_ROOM
    free(taken); free(tmeans); free(lst); free(lpos);
    return rc;
}

/* TEST / ANALYSIS ONLY (tools/k2b_resync_sim.py): the window chain of eventalign (alignment.cpp:556-740) started at an ARBITRARY state (ri0, readHead0), walked for at most
 * max_w windows that reach the Viterbi; records every such window's state at the moment its events are gathered: the reference index and readHead AFTER the scan's
 * reset-on-first-hit (:616-618) -- two chains that agree on that pair at one window agree on everything after it.  Returns the number of windows recorded.
 * seq_only: no Viterbi -- every window's last match is assumed at its last position (ri += N), readHead left to the scan: what the sequence alone predicts. */
size_t dno_eventalign_chain(const dno_model *m, const dno_read *r, const dno_norm *nm, unsigned int ri0, int readHead0, size_t max_w, uint32_t *out_ri, int32_t *out_rh, int seq_only) {
    const unsigned k = DNO_K, totalW = 50;
    const size_t n_ref = r->n_ref;
    size_t tcap = 4096, nw = 0;
    double *tmeans = (double *)malloc(tcap * 8);
    uint8_t *lst = NULL; uint32_t *lpos = NULL; size_t lcap = 0;
    int readHead = readHead0;
    unsigned int ri = ri0;
    while (ri < n_ref - k + 1 && nw < max_w) {
        unsigned int toEnd = (unsigned int)(n_ref - ri);
        unsigned int W = toEnd < totalW ? toEnd : totalW;
        if ((double)toEnd > 1.5 * totalW) {
            const char *snip = r->refseq + ri;
            size_t sl = (size_t)(1.5 * W);
            if (!ref_defined(snip, sl)) { ri += W; continue; }
            const double lim = 1.5 * W - k - 1;
            for (unsigned int i = W; (double)i < lim; i++) {
                double mu = m->mean[dno_kmer2index(snip + i, k)], mb = m->mean[dno_kmer2index(snip + i - 1, k)], mf = m->mean[dno_kmer2index(snip + i + 1, k)];
                if (fabs(mu - mf) > 0.75 && fabs(mu - mb) > 0.75) { W = i + k; break; }
            }
        }
        const char *seq = r->refseq + ri;
        if (!ref_defined(seq, W)) { ri += W; continue; }
        const uint32_t qlo = r->ref2query[ri], qhi = r->ref2query[ri + W - k + 1];
        size_t nt = 0; int first = 1;
        for (unsigned int j = (unsigned int)readHead; j < nm->n_aln; j++) {
            uint32_t q = nm->aln_kmer[j];
            if (qlo <= q && q < qhi) {
                if (first) { readHead = (int)j; first = 0; }
                double em = nm->events[nm->aln_event[j]].mean;
                if (0. < em && em < 250.) {
                    if (nt == tcap) { tcap *= 2; tmeans = (double *)realloc(tmeans, tcap * 8); }
                    tmeans[nt++] = em;
                }
            }
            if (q >= qhi) break;
        }
        if (nt < 2) { ri += W; continue; }
        const size_t N = W - k + 1;
        if (seq_only) {                                                  /* the chain the SEQUENCE alone predicts: every window's last match at its last position */
            out_ri[nw] = ri; out_rh[nw] = readHead; nw++;
            ri += (unsigned int)N;
            continue;
        }
        if (nt + N + 2 > lcap) { lcap = 2 * (nt + N + 2); lst = (uint8_t *)realloc(lst, lcap); lpos = (uint32_t *)realloc(lpos, lcap * 4); }
        double vs; int verr = 0;
        size_t nl = dno_viterbi(m, tmeans, nt, seq, W, nm->shift, nm->scale, nm->events_per_base, &vs, lst, lpos, &verr);
        if (verr) break;
        out_ri[nw] = ri; out_rh[nw] = readHead; nw++;
        size_t lastM_ev = 0, lastM_ref = 0, evIdx = 0;
        for (size_t i = 0; i < nl; i++) {
            if (lst[i] == 1) { lastM_ev = evIdx; lastM_ref = lpos[i]; }
            if (lst[i] != 0) evIdx++;
        }
        readHead += (int)lastM_ev + 1;
        ri += (unsigned int)lastM_ref + 1;
    }
    free(tmeans); free(lst); free(lpos);
    return nw;
}

void dno_align_free(dno_align *a) {
    free(a->coord); free(a->query_idx); free(a->ref_idx); free(a->indel_score); free(a->kmer); free(a->n_signal);
    free(a->signal); free(a->core); free(a->residual); free(a->win_ref); free(a->win_len); free(a->win_T); free(a->win_score);
    free(a->row_coord); free(a->row_rpos); free(a->row_val); free(a->row_kind);
    memset(a, 0, sizeof(*a));
}

/* alignment.cpp:553 + :697-733.  kmerRef is the reverse complement of kmerStrand for reverse reads (:688-690). */
size_t dno_format_align(const dno_model *m, const char *read_id, const char *contig, const dno_read *r, const dno_align *a,
                        char *buf, size_t cap) {
    size_t len = 0;
    char line[160];
    int n = snprintf(line, sizeof line, ">%s %s %d %d %s\n", read_id, contig, r->ref_start, r->ref_end, r->is_reverse ? "rev" : "fwd");
    if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
    len += (size_t)n;
    for (size_t i = 0; i < a->n_rows; i++) {
        char ks[10], kr[10];
        memcpy(ks, r->refseq + a->row_rpos[i], 9); ks[9] = 0;
        if (r->is_reverse) dno_reverse_complement(ks, 9, kr); else memcpy(kr, ks, 9);
        kr[9] = 0;
        if (a->row_kind[i] == 0)
            n = snprintf(line, sizeof line, "%u\t%s\t%f\t%s\t%f\n", a->row_coord[i], kr, a->row_val[i], ks, m->mean[dno_kmer2index(ks, DNO_K)]);
        else
            n = snprintf(line, sizeof line, "%u\t%s\t%f\tNNNNNNNNN\t0\n", a->row_coord[i], kr, a->row_val[i]);
        if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
        len += (size_t)n;
    }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * detect.cpp:235-378  sequenceProbability: forward algorithm over 2*window positions x {I, D, M}
 * (NaN == log 0).  Statement order, lnProd nesting and lnSum accumulation order follow the source.
 * ---------------------------------------------------------------------------------------- */
double dno_sequence_probability(const dno_fit_models *fm, const double *obs, size_t T, const char *seq, size_t window,
                                int use_analogue, double shift, double scale, double events_per_base,
                                size_t brdu_start, size_t brdu_end, int *neg) {
    const unsigned k = DNO_K;
    int ng = 0;
    const double externalD2D = dno_eln(0.3, &ng), externalD2M1 = dno_eln(0.7, &ng), externalI2M1 = dno_eln(0.999, &ng);   /* :245-250, config.h:42 */
    const double externalM12D = dno_eln(0.0025, &ng), internalM12I = dno_eln(0.001, &ng), internalI2I = dno_eln(0.001, &ng);
    const double internalM12M1 = dno_eln(1. - (1. / events_per_base), &ng);                   /* :253 */
    const double externalM12M1 = dno_eln(1.0 - externalM12D - internalM12I - internalM12M1, &ng);   /* :254 (sic: log values) */
    const double ln025 = dno_eln(0.25, &ng), ln05 = dno_eln(0.5, &ng);
    const size_t N = 2 * window;
    double *buf = (double *)malloc(6 * N * sizeof(double));
    double *Ic = buf, *Dc = buf + N, *Mc = buf + 2 * N, *Ip = buf + 3 * N, *Dp = buf + 4 * N, *Mp = buf + 5 * N;
    for (size_t i = 0; i < 6 * N; i++) buf[i] = NAN;
    double firstI_curr = NAN, firstI_prev = NAN;
    const double start_curr = NAN; double start_prev = 0.0;
    Dp[0] = dno_lnProd(start_prev, ln025);                                                    /* :265 */
    for (size_t i = 1; i < N; i++) Dp[i] = dno_lnProd(Dp[i - 1], externalD2D);                /* :268-271 */
    for (size_t t = 0; t < T; t++) {
        for (size_t i = 0; i < N; i++) { Ic[i] = NAN; Mc[i] = NAN; Dc[i] = NAN; }
        firstI_curr = NAN;
        const double x = (obs[t] - shift) / scale;
        uint32_t ki = dno_kmer2index(seq, k);
        double matchProb = dno_eln(dno_normalPDF(fm->unl_mean[ki], fm->unl_std[ki], x), &ng);  /* :286-293 */
        double insProb = 0.0;
        firstI_curr = dno_lnSum(firstI_curr, dno_lnProd(dno_lnProd(start_prev, ln025), insProb));     /* :297 */
        firstI_curr = dno_lnSum(firstI_curr, dno_lnProd(dno_lnProd(firstI_prev, ln025), insProb));    /* :298 */
        Ic[0] = dno_lnSum(Ic[0], dno_lnProd(dno_lnProd(Ip[0], internalI2I), insProb));                /* :301 */
        Ic[0] = dno_lnSum(Ic[0], dno_lnProd(dno_lnProd(Mp[0], internalM12I), insProb));               /* :302 */
        Mc[0] = dno_lnSum(Mc[0], dno_lnProd(dno_lnProd(firstI_prev, ln05), matchProb));               /* :305 */
        Mc[0] = dno_lnSum(Mc[0], dno_lnProd(dno_lnProd(Mp[0], internalM12M1), matchProb));            /* :306 */
        Mc[0] = dno_lnSum(Mc[0], dno_lnProd(dno_lnProd(start_prev, ln05), matchProb));                /* :307 */
        Dc[0] = dno_lnSum(Dc[0], dno_lnProd(NAN, ln025));                                             /* :310 */
        Dc[0] = dno_lnSum(Dc[0], dno_lnProd(firstI_curr, ln025));                                     /* :311 */
        for (size_t i = 1; i < N; i++) {
            const char *km = seq + i;
            ki = dno_kmer2index(km, k);
            insProb = 0.0;
            int hasT = 0;
            for (unsigned z = 0; z < k; z++) if (km[z] == 'T') hasT = 1;
            if (use_analogue && brdu_start <= i && i <= brdu_end && hasT)                              /* :319 */
                matchProb = dno_eln(dno_normalPDF(fm->ana_mean[ki], fm->ana_std[ki], x), &ng);
            else
                matchProb = dno_eln(dno_normalPDF(fm->unl_mean[ki], fm->unl_std[ki], x), &ng);
            Ic[i] = dno_lnSum(Ic[i], dno_lnProd(dno_lnProd(Ip[i], internalI2I), insProb));            /* :336 */
            Ic[i] = dno_lnSum(Ic[i], dno_lnProd(dno_lnProd(Mp[i], internalM12I), insProb));           /* :337 */
            Mc[i] = dno_lnSum(Mc[i], dno_lnProd(dno_lnProd(Ip[i - 1], externalI2M1), matchProb));     /* :340 */
            Mc[i] = dno_lnSum(Mc[i], dno_lnProd(dno_lnProd(Mp[i - 1], externalM12M1), matchProb));    /* :341 */
            Mc[i] = dno_lnSum(Mc[i], dno_lnProd(dno_lnProd(Mp[i], internalM12M1), matchProb));        /* :342 */
            Mc[i] = dno_lnSum(Mc[i], dno_lnProd(dno_lnProd(Dp[i - 1], externalD2M1), matchProb));     /* :343 */
        }
        for (size_t i = 1; i < N; i++) {
            Dc[i] = dno_lnSum(Dc[i], dno_lnProd(Mc[i - 1], externalM12D));                            /* :349 */
            Dc[i] = dno_lnSum(Dc[i], dno_lnProd(Dc[i - 1], externalD2D));                             /* :350 */
        }
        memcpy(Ip, Ic, N * sizeof(double)); memcpy(Mp, Mc, N * sizeof(double)); memcpy(Dp, Dc, N * sizeof(double));
        firstI_prev = firstI_curr;
        start_prev = start_curr;
    }
    double fwd = NAN;
    fwd = dno_lnSum(fwd, dno_lnProd(Dc[N - 1], dno_eln(1.0, &ng)));                                   /* :365 */
    fwd = dno_lnSum(fwd, dno_lnProd(Mc[N - 1], dno_lnSum(externalM12M1, externalM12D)));              /* :366 */
    fwd = dno_lnSum(fwd, dno_lnProd(Ic[N - 1], externalI2M1));                                        /* :367 */
    free(buf);
    if (neg) *neg = ng;
    return fwd;
}

/* ------------------------------------------------------------------------------------------
 * detect.cpp:381-574  getPOIs + llAcrossRead.  The readHead scan over r.eventAlignment is restated
 * as written (including: a reverse-strand snippet is only put back in forward order when the scan
 * leaves the window through the break at :476-480).
 * ---------------------------------------------------------------------------------------- */
int dno_ll_across_read(const dno_fit_models *fm, const dno_read *r, const dno_norm *nm, unsigned W, dno_hmm *out) {
    const unsigned k = DNO_K;
    memset(out, 0, sizeof(*out));
    const size_t L = r->n_ref;
    if (L < 4 * (size_t)W + 1) return 0;
    size_t npoi = 0;
    uint32_t *poi = (uint32_t *)malloc((L + 1) * sizeof(uint32_t));
    for (size_t i = 2 * W; i < L - 2 * W; i++) if (r->refseq[i] == 'T') poi[npoi++] = (uint32_t)i;     /* :385-388 */
    if (r->is_reverse) for (size_t a = 0, b = npoi; a + 1 < b; a++, b--) { const uint32_t t = poi[a]; poi[a] = poi[b - 1]; poi[b - 1] = t; }   /* :405 */
    out->pos_on_ref = (uint32_t *)malloc((npoi + 1) * sizeof(uint32_t)); out->pos_on_query = (uint32_t *)malloc((npoi + 1) * sizeof(uint32_t));
    out->global_pos = (int32_t *)malloc((npoi + 1) * sizeof(int32_t)); out->n_events = (uint32_t *)malloc((npoi + 1) * sizeof(uint32_t));
    out->log_analogue = (double *)malloc((npoi + 1) * sizeof(double)); out->log_thymidine = (double *)malloc((npoi + 1) * sizeof(double));
    out->llr = (double *)malloc((npoi + 1) * sizeof(double));
    const long n_aln = (long)nm->n_aln;
    long readHead = r->is_reverse ? n_aln - 1 : 0;                                                    /* :402-411 */
    double *snip = (double *)malloc(((size_t)n_aln + 1) * sizeof(double));
    char seq[96];
    for (size_t pi = 0; pi < npoi; pi++) {
        const unsigned posOnRef = poi[pi];
        const unsigned posOnQuery = r->ref2query[posOnRef];
        const size_t slen = 2 * W + k;
        if (posOnRef - W + slen > L) continue;                       /* substr would be short -> caught by the ACGT count below */
        memcpy(seq, r->refseq + posOnRef - W, slen); seq[slen] = 0;
        size_t acgt = 0;
        for (size_t z = 0; z < slen; z++) if (seq[z] == 'A' || seq[z] == 'T' || seq[z] == 'G' || seq[z] == 'C') acgt++;
        if (acgt != slen) continue;                                                                   /* :442 */
        const unsigned lo = r->ref2query[posOnRef - W], hi = r->ref2query[posOnRef + W];
        size_t ns = 0; int first = 1;
        if (r->is_reverse) {
            for (long j = readHead; j >= 0; j--) {                                                    /* :453-482 */
                const unsigned q = nm->aln_kmer[j];
                if (lo <= q && q < hi) {
                    if (first) { readHead = j; first = 0; }
                    const double ev = nm->events[nm->aln_event[j]].mean;
                    if (ev > 0. && ev < 250.0) snip[ns++] = ev;
                }
                if (q < lo) {                                                                         /* :476-480 */
                    for (size_t a = 0, b = ns; a + 1 < b; a++, b--) { const double t = snip[a]; snip[a] = snip[b - 1]; snip[b - 1] = t; }
                    break;
                }
            }
        } else {
            for (long j = readHead; j < n_aln; j++) {                                                 /* :484-507 */
                const unsigned q = nm->aln_kmer[j];
                if (lo <= q && q < hi) {
                    if (first) { readHead = j; first = 0; }
                    const double ev = nm->events[nm->aln_event[j]].mean;
                    if (ev > 0. && ev < 250.0) snip[ns++] = ev;
                }
                if (q >= hi) break;
            }
        }
        if (ns < 2 * W - k) continue;                                                                 /* :510 */
        int gpos = r->is_reverse ? (r->ref_end - (int)posOnRef - 1) : (r->ref_start + (int)posOnRef); /* :531-541 */
        const size_t bs = W - k / 2, be = W + k / 2;                                                  /* :544-545 */
        int ng = 0;
        const double la = dno_sequence_probability(fm, snip, ns, seq, W, 1, nm->shift, nm->scale, nm->events_per_base, bs, be, &ng);
        const double lt = dno_sequence_probability(fm, snip, ns, seq, W, 0, nm->shift, nm->scale, nm->events_per_base, 0, 0, &ng);
        const size_t o = out->n++;
        out->pos_on_ref[o] = posOnRef; out->pos_on_query[o] = posOnQuery; out->global_pos[o] = gpos; out->n_events[o] = (uint32_t)ns;
        out->log_analogue[o] = la; out->log_thymidine[o] = lt; out->llr[o] = la - lt;                 /* :548 */
    }
    free(snip); free(poi);
    return 0;
}

void dno_hmm_free(dno_hmm *h) {
    free(h->pos_on_ref); free(h->pos_on_query); free(h->global_pos); free(h->n_events); free(h->log_analogue); free(h->log_thymidine); free(h->llr);
    memset(h, 0, sizeof(*h));
}

/* common.h:91-150 */
static char comp_iupac(char c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; case 'U': return 'A';
        case 'Y': return 'R'; case 'R': return 'Y'; case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V';
        case 'D': return 'H'; case 'H': return 'D'; case 'V': return 'B'; default: return c;      /* N, W, S map to themselves */
    }
}
void dno_reverse_complement(const char *in, size_t n, char *out) {
    for (size_t i = 0; i < n; i++) out[i] = comp_iupac(in[n - 1 - i]);
}
/* common.h:185-195: left-to-right fp64 sum, then one division; 0 for an empty vector */
double dno_vector_mean(const double *v, size_t n) {
    double total = 0.0;
    for (size_t i = 0; i < n; i++) total += v[i];
    return n == 0 ? 0.0 : total / (double)n;
}

static char comp(char c);
size_t dno_format_hmm(const char *read_id, const char *contig, const dno_read *r, const dno_hmm *h, char *buf, size_t cap) {
    size_t len = 0; char line[256];
    int n = snprintf(line, sizeof line, ">%s %s %d %d %s\n", read_id, contig, r->ref_start, r->ref_end, r->is_reverse ? "rev" : "fwd");   /* :414 */
    if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
    len += (size_t)n;
    for (size_t i = 0; i < h->n; i++) {
        char kq[10], kr[10];
        for (int z = 0; z < 9; z++) {                                  /* substr(pos - k/2, k) (:534-536), reverse complement for rev (:540-541) */
            const long iq = (long)h->pos_on_query[i] - 4 + z, ir = (long)h->pos_on_ref[i] - 4 + z;
            const char cq = (iq >= 0 && (size_t)iq < r->n_base) ? r->basecall[iq] : 'N', cr = (ir >= 0 && (size_t)ir < r->n_ref) ? r->refseq[ir] : 'N';
            if (r->is_reverse) { kq[8 - z] = comp(cq); kr[8 - z] = comp(cr); } else { kq[z] = cq; kr[z] = cr; }
        }
        kq[9] = kr[9] = 0;
        n = snprintf(line, sizeof line, "%d\t%f\t%s\t%s\n", h->global_pos[i], h->llr[i], kr, kq);                                           /* :571 */
        if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
        len += (size_t)n;
    }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * detect.cpp:684-727 human-readable record
 * ---------------------------------------------------------------------------------------- */
static char comp(char c) {                                               /* common.h:91 (ACGT subset) */
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return c; }
}

size_t dno_format_detect(const char *read_id, const char *contig, const dno_read *r,
                         const dno_align *a, const float *probs, char *buf, size_t cap) {
    size_t len = 0;
    char line[256];
    int n = snprintf(line, sizeof line, ">%s %s %d %d %s\n", read_id, contig, r->ref_start, r->ref_end, r->is_reverse ? "rev" : "fwd");
    if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
    len += (size_t)n;
    /* tensors are in creation order; fwd emits in that order, rev reverses the lines (detect.cpp:722) so both ascend */
    for (size_t q = 0; q < a->n_pos; q++) {
        size_t i = r->is_reverse ? (a->n_pos - 1 - q) : q;
        const char *km = a->kmer + i * 9;
        if (km[4] != 'T') continue;                                      /* :690 */
        char ks[10];
        if (r->is_reverse) { for (int z = 0; z < 9; z++) ks[z] = comp(km[8 - z]); }
        else memcpy(ks, km, 9);
        ks[9] = 0;
        /* std::to_string(float) == "%f" of the value promoted to double (:698) */
        n = snprintf(line, sizeof line, "%u\t%f\t%f\t%s\n", a->coord[i], (double)probs[i * 3 + 2], (double)probs[i * 3 + 1], ks);
        if (len + (size_t)n <= cap) memcpy(buf + len, line, (size_t)n);
        len += (size_t)n;
    }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * detect.cpp:704-707 (queryIndexToCalls) + reads.h:453-512 (writeModBamTag, no existing MM/ML)
 * The std::map is emulated by a dense table over query indices: value = index of the LAST position
 * that wrote the key (operator[] assignment), walk ascending.
 * ---------------------------------------------------------------------------------------- */
size_t dno_modbam_tags(const dno_read *r, const dno_align *a, const float *probs, char *mm, size_t mm_cap, uint8_t *ml, size_t ml_cap) {
    size_t nq = r->n_base + 1, calls = 0;
    for (size_t i = 0; i < a->n_pos; i++) if ((size_t)a->query_idx[i] + 1 > nq) nq = (size_t)a->query_idx[i] + 1;
    int64_t *slot = (int64_t *)malloc(nq * sizeof(int64_t));
    for (size_t q = 0; q < nq; q++) slot[q] = -1;
    for (size_t i = 0; i < a->n_pos; i++) {
        if (a->kmer[i * 9 + 4] != 'T') continue;                         /* detect.cpp:690 */
        if (r->ref2del[a->ref_idx[i]]) continue;                         /* :704 */
        slot[a->query_idx[i]] = (int64_t)i;                              /* :706 */
    }
    for (size_t q = 0; q < nq; q++) if (slot[q] >= 0) calls++;
    /* two passes over the map: the BrdU field then the EdU field carry the same deltas (reads.h:472-476) */
    size_t len = 0;
    char num[32];
    for (int field = 0; field < 2; field++) {
        const char *head = field == 0 ? "N+b?" : "N+e?";
        for (size_t z = 0; z < 4; z++) { if (len + 1 < mm_cap) mm[len] = head[z]; len++; }
        unsigned prev = 0;
        for (size_t q = 0; q < nq; q++) {
            if (slot[q] < 0) continue;
            int n = snprintf(num, sizeof num, ",%u", (unsigned)q - prev);
            for (int z = 0; z < n; z++) { if (len + 1 < mm_cap) mm[len] = num[z]; len++; }
            prev = (unsigned)q + 1;
        }
        if (len + 1 < mm_cap) mm[len] = ';';
        len++;
    }
    if (mm_cap) mm[len < mm_cap ? len : mm_cap - 1] = 0;
    size_t w = 0;
    for (int field = 0; field < 2; field++)                                /* ML = BrdU calls, then EdU calls (:503-504) */
        for (size_t q = 0; q < nq; q++) {
            if (slot[q] < 0) continue;
            const float p = probs[(size_t)slot[q] * 3 + (field == 0 ? 1 : 2)];
            if (w < ml_cap) ml[w] = (uint8_t)((double)p * 255.0);          /* static_cast<uint8_t>(float * 255.0) (:482-483) */
            w++;
        }
    free(slot);
    return calls;
}

/* ------------------------------------------------------------------------------------------
 * htsInterface.cpp:59-157 parseCigar (std::map semantics flattened; later writes win)
 * ---------------------------------------------------------------------------------------- */
int dno_parse_cigar(const uint32_t *ops, const uint32_t *lens, size_t n_ops, int is_reverse,
                    uint32_t *ref2query, uint8_t *ref2del, size_t n_ref_cap,
                    int32_t *query2ref, size_t n_q_cap) {
    for (size_t i = 0; i < n_q_cap; i++) query2ref[i] = -1;
    int qp = 0, rp = 0;
    for (size_t c = 0; c < n_ops; c++) {
        size_t i = is_reverse ? (n_ops - 1 - c) : c;                     /* :69 reversed walk for rev strand */
        const int op = (int)ops[i], ol = (int)lens[i];
        if (op == 0 || op == 7 || op == 8) {                             /* M,=,X */
            for (int j = rp; j < rp + ol; j++) {
                if ((size_t)j < n_ref_cap) { ref2query[j] = (uint32_t)qp; ref2del[j] = 0; }
                if ((size_t)qp < n_q_cap) query2ref[qp] = j;
                qp++;
            }
            rp += ol;
        } else if (op == 2 || op == 3) {                                 /* D,N */
            for (int j = rp; j < rp + ol; j++) {
                if ((size_t)j < n_ref_cap) { ref2query[j] = (uint32_t)qp; ref2del[j] = 1; }
                if ((size_t)qp < n_q_cap) query2ref[qp] = j;
            }
            rp += ol;
        } else if (op == 4 || op == 1) {                                 /* S,I : sic, writes ref slots ahead */
            for (int j = rp; j < rp + ol; j++) {
                if ((size_t)j < n_ref_cap) { ref2query[j] = (uint32_t)qp; ref2del[j] = 0; }
                if ((size_t)qp < n_q_cap) query2ref[qp] = j;
                qp++;
            }
        }
    }
    return rp;
}


/* ---- bench.py cpu_baseline: the reference's per-read loop (detect.cpp:852: #pragma omp parallel for schedule(dynamic)) over an
 *      array of reads, one read per thread: normaliseEvents and, if do_align, eventalign.  Returns the wall time; per-read
 *      statuses and position counts come back for the caller's bookkeeping.  Test / measurement infrastructure only. ---- */
#include <omp.h>
double dno_bench_reads(const dno_model *m, const dno_read *reads, size_t n, int do_align, int n_threads, int *status, uint64_t *n_pos) {
    if (n_threads > 0) omp_set_num_threads(n_threads);
    const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic)
    for (long i = 0; i < (long)n; i++) {
        dno_norm nm; memset(&nm, 0, sizeof nm);
        int st = dno_normalise(m, &reads[i], &nm);
        uint64_t np = 0;
        if (st == 0 && do_align) {
            dno_align al; memset(&al, 0, sizeof al);
            st = dno_eventalign(m, &reads[i], &nm, &al);
            np = st == 0 ? (uint64_t)al.n_pos : 0;
            dno_align_free(&al);
        }
        dno_norm_free(&nm);
        if (status) status[i] = st;
        if (n_pos) n_pos[i] = np;
    }
    return omp_get_wtime() - t0;
}
