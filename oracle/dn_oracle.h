/*
 * dn_oracle.h -- CPU restatement of DNAscent `detect`'s per-read numerical path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the reported CPU baseline.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   PINNED    dno_detect_events          vs  oracle/_ref  (reference scrappie/event_detection.c, compiled in place)
 *   PINNED    dno_eexp/eln/lnSum/...     vs  oracle/_ref  (reference probability.cpp, compiled in place)
 *   PINNED    dno_reverse_complement, dno_vector_mean  vs  oracle/_ref  (reference common.h / common.cpp)
 *   UNPINNED  banded alignment, scaling, Viterbi, eventalign, tensor packing, the --HMM forward path,
 *             output formatting (.detect / modbam):
 *             the reference has no tests / golden vectors for them and
 *             event_handling.cpp / alignment.cpp cannot be compiled here without
 *             stand-in htslib / TensorFlow / generated headers, which is not allowed.
 *             -> "parity unpinned": restated line by line from the cited sources.
 *
 * All file:line citations are relative to /root/reference/src/.
 */
#ifndef DN_ORACLE_H
#define DN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNO_K 9                 /* config.h:45 kmer_len */
#define DNO_NKMER 262144        /* 4^9 */
#define DNO_BANDWIDTH 100       /* config.h:41 */
#define DNO_RAWDEPTH 20         /* reads.h:12 */

enum {
    DNO_OK = 0,
    DNO_FAIL_BANDED_QC = 1,     /* event_handling.cpp:433-441 */
    DNO_FAIL_SCALING = 2,       /* event_handling.cpp:90-95,604 */
    DNO_FAIL_NO_END_CELL = 3,   /* reference would index out of bounds (UB) -- we fail the read */
    DNO_FAIL_NEGATIVE_LOG = 4,  /* probability.cpp:45 throws NegativeLog (uncaught in the reference) */
    DNO_FAIL_TOO_SHORT = 5      /* inputs the reference would assert/UB on */
};

/* ---- probability.cpp ---- */
double dno_eexp(double x);                         /* :23 */
double dno_eln(double x, int *neg);                /* :35  (*neg set instead of throwing) */
double dno_lnSum(double a, double b);              /* :50 */
double dno_lnProd(double a, double b);             /* :79 */
int    dno_lnGreaterThan(double a, double b);      /* :107 */
double dno_normalPDF(double mu, double sigma, double x); /* :145 */

/* ---- data_IO.cpp:129 ---- */
uint32_t dno_kmer2index(const char *kmer, unsigned k);

/* ---- scrappie/event_detection.c ---- */
typedef struct {
    uint64_t start;
    float length;
    float mean;
    float stdv;
} dno_sevent;
/* returns number of events (et.n); *out is malloc'd (caller frees). tstat1/tstat2/peaks may be NULL
 * or caller buffers of n floats / n uint64 to receive the intermediates. */
size_t dno_detect_events(const double *raw, size_t n, dno_sevent **out,
                         float *tstat1, float *tstat2, uint64_t *peaks, size_t *npeaks);

/* ---- pod5.cpp:57-61 ---- */
void dno_adc_to_pa(const int16_t *adc, size_t n, float offset, float scale, double *out);

/* pore model (static sigma, data_IO.cpp:144-190) */
typedef struct {
    const double *mean;  /* [DNO_NKMER], indexed by kmer2index */
    double sigma;        /* 0.14 */
} dno_model;

typedef struct {
    const double *raw;  size_t n_raw;          /* pA, already trimmed */
    const char *basecall; size_t n_base;       /* sequencing direction */
    const char *refseq;   size_t n_ref;        /* sequencing direction */
    const uint32_t *ref2query;                 /* [n_ref]            (htsInterface.cpp:59 map, flattened) */
    const int32_t *query2ref;                  /* [n_base+1], -1 = key absent */
    const uint8_t *ref2del;                    /* [n_ref] */
    int32_t ref_start, ref_end;
    int is_reverse;
} dno_read;

typedef struct {
    double mean;
    uint32_t raw_start, raw_len;
} dno_event;

typedef struct {
    int status;
    size_t n_scrappie;                 /* et.n */
    dno_event *events; size_t n_events;
    uint32_t *rank_q; size_t n_kq;     /* event_handling.cpp:578-584 */
    uint32_t *rank_r; size_t n_kr;     /* :586-592 */
    double q_shift, q_scale;           /* estimateScaling_quantiles :510 */
    /* adaptive banded alignment :148 */
    size_t n_bands; uint64_t fills;
    int end_event;                     /* curr_event_idx chosen at :329-340 */
    uint32_t *aln_event, *aln_kmer; size_t n_aln;   /* eventAlignment BEFORE any QC clear */
    double avg_log_emission; int spanned; int max_gap;
    double *cleaned_sig; uint32_t *cleaned_rank; size_t n_cleaned;
    double ts_slope, ts_intercept;     /* Theil-Sen medians :78,:87 (NaN if not run) */
    double shift, scale, events_per_base;  /* final r.scalings */
} dno_norm;

int  dno_normalise(const dno_model *m, const dno_read *r, dno_norm *out);   /* event_handling.cpp:544 */
void dno_norm_free(dno_norm *n);

/* sub-steps, exposed for unit parity */
void dno_quantile_scaling(const dno_model *m, const double *event_means, size_t ne,
                          const uint32_t *rank_r, size_t nr, double *shift, double *scale);
int  dno_theil_sen(const dno_model *m, const double *sig, const uint32_t *rank, size_t n,
                   double in_shift, double in_scale, double *out_shift, double *out_scale,
                   double *slope_med, double *icpt_med);

/* ---- alignment.cpp:193 builtinViterbi ---- */
/* labels: state[i] 0=D 1=M 2=I, pos[i]; returns count (forward order). caller buffers sized >= T+N+1 */
size_t dno_viterbi(const dno_model *m, const double *obs, size_t T, const char *seq, size_t seqlen,
                   double shift, double scale, double events_per_base,
                   double *score, uint8_t *state, uint32_t *pos, int *err);

/* ---- alignment.cpp:547 eventalign + reads.h:292-372 tensors ---- */
typedef struct {
    size_t n_pos;
    uint32_t *coord;       /* reference coordinate (creation order = sequencing direction) */
    uint32_t *query_idx, *ref_idx;
    int32_t *indel_score;
    char *kmer;            /* n_pos*9, strand orientation */
    uint32_t *n_signal;    /* samples pushed at that position */
    float *signal;         /* n_pos*20, zero padded (reads.h:147) */
    float *core, *residual;/* reads.h:112,125 (+1) */
    size_t n_windows;      /* Viterbi calls made */
    uint64_t sum_TN;       /* sum of T*N over windows */
    double score_sum;      /* sum of finite window Viterbi scores (diagnostic) */
    /* per-window log for unit parity: ref_index, window_len, T */
    uint32_t *win_ref, *win_len, *win_T; double *win_score;
    /* `DNAscent align` table (alignment.cpp:697-733): one row per raw sample of every event labelled M, and of every event
     * labelled I before the window's last match, in emission order.  ref_pos = reference_index + pos (start of kmerStrand). */
    size_t n_rows;
    uint32_t *row_coord, *row_rpos; double *row_val; uint8_t *row_kind;   /* kind 0 = match, 1 = insertion */
} dno_align;

int  dno_eventalign(const dno_model *m, const dno_read *r, const dno_norm *n, dno_align *out);
/* analysis only: the Viterbi's emission as the device evaluates it (see dn_oracle.c); 0 = the reference's arithmetic (default) */
void dno_set_device_emission(int on);
/* analysis only (tools/k2b_resync_sim.py): the window chain from an arbitrary start state; see dn_oracle.c */
size_t dno_eventalign_chain(const dno_model *m, const dno_read *r, const dno_norm *nm, unsigned int ri0, int readHead0, size_t max_w, uint32_t *out_ri, int32_t *out_rh, int seq_only);
void dno_align_free(dno_align *a);
/* text `DNAscent align` writes for the read (alignment.cpp:553 header + :697-733 rows, std::to_string = "%f") */
size_t dno_format_align(const dno_model *m, const char *read_id, const char *contig, const dno_read *r, const dno_align *a,
                        char *buf, size_t cap);

/* ---- detect.cpp:235-378 sequenceProbability (forward algorithm, --HMM) ----
 * fit models: (mean, std) per 9-mer in kmer2index order (config.h:53-54, data_IO.cpp:192-240).
 * seq: 2*window + 9 bases; obs: raw event means.  Returns the forward log-probability (NaN = log 0); *neg is set when an
 * eln of a negative number would have thrown (probability.cpp:45). */
typedef struct {
    const double *unl_mean, *unl_std;      /* Pore_Substrate_Config.unlabelled_model */
    const double *ana_mean, *ana_std;      /* Pore_Substrate_Config.analogue_model */
} dno_fit_models;
double dno_sequence_probability(const dno_fit_models *fm, const double *obs, size_t T, const char *seq, size_t window,
                                int use_analogue, double shift, double scale, double events_per_base,
                                size_t brdu_start, size_t brdu_end, int *neg);

/* ---- detect.cpp:381-574 getPOIs + llAcrossRead (window 12, detect.cpp:885) ----
 * One entry per emitted line, in emission order. */
typedef struct {
    size_t n;
    uint32_t *pos_on_ref, *pos_on_query;   /* strand coordinates of the call */
    int32_t *global_pos;                   /* :531-541 */
    uint32_t *n_events;                    /* eventSnippet.size() */
    double *log_analogue, *log_thymidine, *llr;
} dno_hmm;
int  dno_ll_across_read(const dno_fit_models *fm, const dno_read *r, const dno_norm *n, unsigned window, dno_hmm *out);
void dno_hmm_free(dno_hmm *h);
/* text of the read's record in --HMM mode (:414, :571): ">id contig start end strand" + "pos\tllr\tkmerRef\tkmerQuery" lines */
size_t dno_format_hmm(const char *read_id, const char *contig, const dno_read *r, const dno_hmm *h, char *buf, size_t cap);

/* ---- common.h:91 reverseComplement (A/C/G/T + the IUPAC codes the reference maps), common.h:185 vectorMean ---- */
void   dno_reverse_complement(const char *in, size_t n, char *out);
double dno_vector_mean(const double *v, size_t n);

/* ---- detect.cpp:684-731 human-readable record from CNN outputs ---- */
/* probs: n_pos*3 (class0 thymidine, class1 BrdU, class2 EdU). Writes text to buf, returns length
 * (or required length if > cap). */
size_t dno_format_detect(const char *read_id, const char *contig, const dno_read *r,
                         const dno_align *a, const float *probs, char *buf, size_t cap);

/* ---- detect.cpp:704-707 + reads.h:453-512: modbam MM / ML fields of a read with no pre-existing tags ----
 * queryIndexToCalls is a std::map keyed by query index (ascending walk, a later position with the same key overwrites);
 * positions whose reference base is deleted in the read (refToDel) are skipped.  mm receives
 * "N+b?,d..;N+e?,d..;" (NUL-terminated when it fits), ml the BrdU bytes followed by the EdU bytes; returns the number of calls. */
size_t dno_modbam_tags(const dno_read *r, const dno_align *a, const float *probs, char *mm, size_t mm_cap, uint8_t *ml, size_t ml_cap);

/* htsInterface.cpp:59 parseCigar, flattened to arrays.  ops: BAM op codes, lens: lengths.
 * ref2query/ref2del sized n_ref_cap, query2ref sized n_q_cap (filled with -1 first). returns ref length. */
int dno_parse_cigar(const uint32_t *ops, const uint32_t *lens, size_t n_ops, int is_reverse,
                    uint32_t *ref2query, uint8_t *ref2del, size_t n_ref_cap,
                    int32_t *query2ref, size_t n_q_cap);

#ifdef __cplusplus
}
#endif
/* OpenMP driver for bench.py's cpu_baseline: the reference's per-read loop (detect.cpp:852) over an array of reads */
double dno_bench_reads(const dno_model *m, const dno_read *reads, size_t n, int do_align, int n_threads, int *status, uint64_t *n_pos);

#endif
