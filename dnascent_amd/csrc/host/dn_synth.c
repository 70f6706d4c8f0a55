/*
 * dn_synth.c -- deterministic synthetic R10.4.1-like data.
 *
 * Own PRNG (splitmix64 + Box-Muller) so the same seed gives the same bytes on any platform with
 * the same libm.  Produces: a synthetic static pore-model table in kmer2index order
 * (data_IO.cpp:129: A0 T1 G2 C3), strand-direction reference / basecall sequences, a CIGAR in BAM
 * (reference-forward) order and an int16 ADC trace with POD5-style calibration (pod5.cpp:60).
 */
#include "dn_synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; int have; double spare; } rng_t;

static inline uint64_t sm64(rng_t *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(rng_t *r) { return (double)(sm64(r) >> 11) * (1.0 / 9007199254740992.0); }
static double gauss(rng_t *r) {
    if (r->have) { r->have = 0; return r->spare; }
    double u1 = u01(r), u2 = u01(r);
    if (u1 < 1e-300) u1 = 1e-300;
    double m = sqrt(-2.0 * log(u1));
    r->spare = m * sin(6.283185307179586 * u2);
    r->have = 1;
    return m * cos(6.283185307179586 * u2);
}

static const char BASES[4] = { 'A', 'T', 'G', 'C' };   /* kmer2index digit order */

void dns_pore_model(uint64_t seed, double *mean) {
    rng_t r = { seed, 0, 0.0 };
    for (uint32_t k = 0; k < 262144u; k++) {
        /* the reference reads a text table through atof (data_IO.cpp:160-175): keep 6 decimals */
        double v = gauss(&r);
        mean[k] = round(v * 1e6) / 1e6;
    }
}

void dns_fit_models(uint64_t seed, const double *static_mean, double *unl_std, double *ana_mean, double *ana_std) {
    rng_t r = { seed ^ 0xF17F17F17ull, 0, 0.0 };
    for (uint32_t k = 0; k < 262144u; k++) {
        unl_std[k] = round((0.12 + 0.04 * fabs(gauss(&r))) * 1e6) / 1e6;
        ana_mean[k] = round((static_mean[k] + 0.3 * gauss(&r)) * 1e6) / 1e6;
        ana_std[k] = 0.15;
    }
}

void dns_index_to_kmer(uint32_t idx, char *out9) {
    for (int i = 8; i >= 0; i--) { out9[i] = BASES[idx & 3u]; idx >>= 2; }
}

static inline uint32_t code_of(char c) {
    switch (c) { case 'A': return 0; case 'T': return 1; case 'G': return 2; case 'C': return 3; default: return 0; }
}

size_t dns_max_samples(uint32_t n_bases) { return (size_t)n_bases * 64 + 4096; }

int dns_make_read(const double *model_mean, const dns_read_spec *sp, dns_read_out *o) {
    rng_t r = { sp->seed * 0x2545F4914F6CDD1Dull + 0x1234567ull, 0, 0.0 };
    const uint32_t L = sp->n_bases;
    if (L < 32) return -1;
    /* strand-direction truth (== referenceSeqMappedTo after the reverse-complement step, reads.h:283) */
    for (uint32_t i = 0; i < L; i++) o->refseq[i] = BASES[sm64(&r) & 3u];
    for (uint32_t i = 0; i < sp->n_unknown && L > 200; i++) {
        uint32_t p = 100 + (uint32_t)(u01(&r) * (L - 200));
        o->refseq[p] = 'N';
    }
    o->n_ref = L;

    /* basecall + CIGAR in strand direction */
    uint32_t nq = 0, nops = 0;
    uint32_t cur_op = 99, cur_len = 0;
#define PUSH_OP(OP) do { if (cur_op == (uint32_t)(OP)) cur_len++; else { if (cur_len) { o->cigar_op[nops] = cur_op; o->cigar_len[nops] = cur_len; nops++; } cur_op = (OP); cur_len = 1; } } while (0)
    for (uint32_t i = 0; i < sp->soft_clip_head; i++) { o->basecall[nq++] = BASES[sm64(&r) & 3u]; PUSH_OP(4); }
    for (uint32_t i = 0; i < L; i++) {
        double u = u01(&r);
        int edge = (i < 20 || i + 20 >= L);
        if (!edge && u < sp->del_rate) { PUSH_OP(2); continue; }
        if (!edge && u < sp->del_rate + sp->ins_rate) { o->basecall[nq++] = BASES[sm64(&r) & 3u]; PUSH_OP(1); }
        char b = o->refseq[i];
        if (b == 'N') b = BASES[sm64(&r) & 3u];
        if (!edge && u01(&r) < sp->sub_rate) b = BASES[(code_of(b) + 1 + (sm64(&r) % 3)) & 3u];
        o->basecall[nq++] = b; PUSH_OP(0);
    }
    for (uint32_t i = 0; i < sp->soft_clip_tail; i++) { o->basecall[nq++] = BASES[sm64(&r) & 3u]; PUSH_OP(4); }
    if (cur_len) { o->cigar_op[nops] = cur_op; o->cigar_len[nops] = cur_len; nops++; }
#undef PUSH_OP
    o->n_base = nq; o->n_cigar = nops;
    if (sp->is_reverse) {   /* BAM stores the CIGAR in reference-forward order (htsInterface.cpp:69 walks it backwards) */
        for (uint32_t i = 0; i < nops / 2; i++) {
            uint32_t t = o->cigar_op[i]; o->cigar_op[i] = o->cigar_op[nops - 1 - i]; o->cigar_op[nops - 1 - i] = t;
            t = o->cigar_len[i]; o->cigar_len[i] = o->cigar_len[nops - 1 - i]; o->cigar_len[nops - 1 - i] = t;
        }
    }
    o->is_reverse = sp->is_reverse;
    o->ref_start = (int32_t)sp->ref_start;
    o->ref_end = (int32_t)(sp->ref_start + L);

    /* signal from the molecule that was sequenced (the basecalled strand incl. clips): pA = mean*14 + 95 + noise */
    const float cal_off = -240.0f, cal_scale = 0.1755f;
    o->cal_offset = cal_off; o->cal_scale = cal_scale;
    size_t ns = 0; const size_t cap = dns_max_samples(L);
    const double p_geo = 1.0 / sp->mean_dwell;   /* dwell = 1 + Geometric(mean mean_dwell) */
    const char *mol = o->refseq; uint32_t mol_len = L;
    uint32_t idx = 0;
    for (uint32_t i = 0; i + 9 <= mol_len; i++) {
        if (i == 0) { for (int z = 0; z < 9; z++) idx = idx * 4u + code_of(mol[z]); }
        else idx = ((idx << 2) & 0x3FFFFu) | code_of(mol[i + 8]);
        double level = model_mean[idx] * 14.0 + 95.0;
        double u = u01(&r); if (u < 1e-300) u = 1e-300;
        uint32_t dwell = 1u + (uint32_t)floor(log(u) / log(1.0 - p_geo));
        if (dwell > 400) dwell = 400;
        for (uint32_t d = 0; d < dwell && ns < cap; d++) {
            double pa = level + sp->noise_pa * gauss(&r);
            long adc = lround(pa / (double)cal_scale - (double)cal_off);
            if (adc > 32767) adc = 32767;
            if (adc < -32768) adc = -32768;
            o->adc[ns++] = (int16_t)adc;
        }
    }
    o->n_samples = ns;
    return 0;
}
