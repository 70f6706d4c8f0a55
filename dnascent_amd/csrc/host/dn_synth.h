/* dn_synth.h -- deterministic synthetic reads for tests and bench. */
#ifndef DN_SYNTH_H
#define DN_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t seed;
    uint32_t n_bases;        /* reference span of the read */
    uint32_t ref_start;      /* coordinate on the contig */
    int32_t is_reverse;
    double noise_pa;         /* 1.6 nominal; ~6 makes the banded QC fail */
    double mean_dwell;       /* 11.5 -> ~12.5 samples/base */
    double sub_rate, ins_rate, del_rate;
    uint32_t soft_clip_head, soft_clip_tail;
    uint32_t n_unknown;      /* number of 'N' placed in the reference slice */
} dns_read_spec;

typedef struct {
    char *refseq;  uint32_t n_ref;        /* strand direction; capacity n_bases */
    char *basecall; uint32_t n_base;      /* strand direction; capacity 2*n_bases + clips */
    uint32_t *cigar_op, *cigar_len; uint32_t n_cigar;   /* BAM order; capacity 2*n_bases + 4 */
    int16_t *adc; size_t n_samples;       /* capacity dns_max_samples(n_bases) */
    float cal_offset, cal_scale;
    int32_t ref_start, ref_end, is_reverse;
} dns_read_out;

void dns_pore_model(uint64_t seed, double *mean /* [262144] */);
/* fit tables of the --HMM path (config.h:53-54) derived from the static table: unlabelled = (mean, 0.12 + 0.04 |N|),
 * analogue = (mean + 0.3 N, 0.15), all rounded to 6 decimals like a text model file (data_IO.cpp:209-226) */
void dns_fit_models(uint64_t seed, const double *static_mean, double *unl_std, double *ana_mean, double *ana_std);
void dns_index_to_kmer(uint32_t idx, char *out9);
size_t dns_max_samples(uint32_t n_bases);
int dns_make_read(const double *model_mean, const dns_read_spec *sp, dns_read_out *out);

#ifdef __cplusplus
}
#endif
#endif
