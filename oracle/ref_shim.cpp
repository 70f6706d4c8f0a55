// ref_shim.cpp -- C-linkage driver around the REFERENCE's own probability.cpp and
// scrappie/event_detection.c, which are compiled in place from /root/reference (never copied).
// TEST INFRASTRUCTURE ONLY: used to pin oracle/dn_oracle.c (tests/test_oracle_vs_ref.py) and to
// generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).
//
// This file declares nothing the reference lacks: it only includes the reference's headers
// (common.h, probability.h, scrappie/event_detection.h) and forwards calls.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "common.h"                      // /root/reference/src/common.h (reverseComplement, vectorMean; common.cpp links in)
#include "probability.h"                 // /root/reference/src/probability.h
#include "scrappie/event_detection.h"    // /root/reference/src/scrappie/event_detection.h

extern "C" {

double ref_eexp(double x) { return eexp(x); }
// returns NaN and sets *neg when the reference throws NegativeLog (probability.cpp:45)
double ref_eln(double x, int *neg) {
    try { return eln(x); } catch (NegativeLog &) { if (neg) *neg = 1; return NAN; }
}
double ref_lnSum(double a, double b) { return lnSum(a, b); }
double ref_lnProd(double a, double b) { return lnProd(a, b); }
int ref_lnGreaterThan(double a, double b) { return lnGreaterThan(a, b) ? 1 : 0; }
double ref_normalPDF(double mu, double sigma, double x) { return normalPDF(mu, sigma, x); }

// detect_events with the reference's own default parameters (event_detection.h:19-25).
// Copies start/length/mean/stdv of each event into caller arrays of capacity cap; returns et.n.
size_t ref_detect_events(double *raw, size_t n, uint64_t *start, float *length, float *mean, float *stdv, size_t cap) {
    event_table et = detect_events(raw, n, event_detection_defaults);
    size_t m = et.n < cap ? et.n : cap;
    for (size_t i = 0; i < m; i++) {
        start[i] = et.event[i].start;
        length[i] = et.event[i].length;
        mean[i] = et.event[i].mean;
        stdv[i] = et.event[i].stdv;
    }
    size_t total = et.n;
    free(et.event);
    return total;
}

// common.h:91 reverseComplement (IUPAC input only: anything else makes the reference exit); returns the length written
size_t ref_reverseComplement(const char *in, size_t n, char *out) {
    const std::string r = reverseComplement(std::string(in, n));
    memcpy(out, r.data(), r.size());
    return r.size();
}
// common.h:185 vectorMean<double>
double ref_vectorMean(const double *v, size_t n) {
    std::vector<double> x(v, v + n);
    return vectorMean(x);
}

}  // extern "C"
