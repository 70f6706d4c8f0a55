/*
 * dnascent_hip.h -- C-ABI of the MI355X-native `DNAscent detect` hot path (libdnascent_hip.so).
 *
 * The reference has no plugin / FFI layer: its seams are ordinary C++ calls inside one OpenMP loop
 * (detect.cpp:852-907).  Each entry point below replaces one of those seams for a whole BATCH of reads
 * (the reference processes one read per OpenMP thread); the citation names the reference interface it
 * stands in for.  All paths are relative to /root/reference/src/.
 *
 * Conventions: int return codes (0 = DN_OK, <0 = error, text via dn_last_error), caller-owned host
 * buffers, library-owned device memory, no exceptions and no torch / HIP types across the boundary.
 * A context holds ONE batch and is single-producer; the stages of the batch are stream-ordered on the context's HIP
 * stream and every dn_run_* only ENQUEUES (no host synchronisation: the per-read constants the reference takes from libm
 * are computed by stream-ordered host functions on page-locked mirrors).  dn_collect / dn_sync / the dn_get_* taps wait.
 * Throughput comes from several contexts in flight, driven by one host thread:
 *     for each slot:  dn_collect(slot) -> write records;  dn_batch_upload(slot, next batch);  dn_run_detect(slot)
 * There is NO CPU fallback: every dn_run_* fails with DN_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef DNASCENT_HIP_H
#define DNASCENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_ABI_VERSION 7
#define DN_KMER 9            /* config.h:45 */
#define DN_NKMER 262144      /* 4^9, data_IO.cpp:177 */
#define DN_BANDWIDTH 100     /* config.h:41 AdaptiveBanded_Params.bandwidth */
#define DN_RAWDEPTH 20       /* reads.h:12 */

enum {
    DN_OK = 0,
    DN_ERR_NO_DEVICE = -1,
    DN_ERR_HIP = -2,
    DN_ERR_ARG = -3,
    DN_ERR_STATE = -4,       /* stage called out of order */
    DN_ERR_OVERFLOW = -5     /* a device workspace bound was exceeded (reported, never silently truncated) */
};

/* per-read status after normalise / eventalign (detect.cpp:879-894: such reads are counted as failed and skipped) */
enum {
    DN_READ_OK = 0,
    DN_READ_FAIL_BANDED_QC = 1,   /* event_handling.cpp:433-441 */
    DN_READ_FAIL_SCALING = 2,     /* event_handling.cpp:90-95,604 */
    DN_READ_FAIL_NO_END_CELL = 3, /* path left the band / no end cell: undefined behaviour in the reference */
    DN_READ_FAIL_NEGATIVE_LOG = 4,/* probability.cpp:45 NegativeLog */
    DN_READ_FAIL_TOO_SHORT = 5,
    DN_READ_FAIL_WINDOW_EVENTS = 6/* more than 8 192 events inside one eventalign window (alignment.cpp:611-632 has no limit; a 50-base window holds ~120, a
                                     stalled pore thousands): up to 224 / 512 the lattice lives in LDS, up to 8 192 in global memory (round 6; at most 32 such
                                     reads per batch), beyond that the read is reported failed */
};

typedef struct dn_ctx dn_ctx;

/* ---- lifecycle (replaces model_load_gpu_twoInputs' device selection, tensor.cpp:66-106) ---- */
int dn_abi_version(void);
int dn_device_count(void);
/* hip_stream: NULL = the context creates its own stream; otherwise a hipStream_t owned by the caller */
int dn_ctx_create(int device, void *hip_stream, dn_ctx **out);
void dn_ctx_destroy(dn_ctx *ctx);
const char *dn_last_error(const dn_ctx *ctx);
int dn_sync(dn_ctx *ctx);

/* ---- model (replaces Global_Config::configure_DNA_R10 -> import_poreModel_staticStdv, config.h:44-52, data_IO.cpp:144) ---- */
int dn_load_pore_model(dn_ctx *ctx, const double *mean /* [DN_NKMER] kmer2index order */, double sigma /* 0.14 */);

/* ---- batch upload: the fields of DNAscent::read that normaliseEvents / eventalign consume (reads.h:178-207) ----
 * All sequences are in sequencing direction (after reads.h:280-286).  Arrays are concatenated over reads;
 * *_off have n_reads+1 entries.  ref2query / ref2del share refseq_off; query2ref holds n_base+1 entries per
 * read (-1 = key absent in the reference's std::map) starting at basecall_off[r] + r. */
typedef struct {
    uint32_t n_reads;
    const int16_t *adc;            const uint64_t *adc_off;       /* pod5.cpp:55-56 raw samples, already trimmed (pod5.cpp:75-93) */
    const float *cal_offset;       const float *cal_scale;        /* pod5.cpp:60 */
    const char *basecall;          const uint64_t *basecall_off;  /* r.basecall */
    const char *refseq;            const uint64_t *refseq_off;    /* r.referenceSeqMappedTo */
    const uint32_t *ref2query;     const int32_t *query2ref;      /* r.refToQuery / r.queryToRef (htsInterface.cpp:59) */
    const uint8_t *ref2del;                                       /* r.refToDel */
    const int32_t *ref_start;      const int32_t *ref_end;        /* r.refStart / r.refEnd */
    const uint8_t *is_reverse;                                    /* r.isReverse */
} dn_batch_desc;

/* H2D; sizes every workspace (bounds only: nothing waits for a result).  Waits for the previous batch of the context first.
 * With page-locked input arrays (dn_host_alloc / dn_host_register) the copies themselves are asynchronous too and the arrays
 * must stay untouched until the next dn_sync / dn_collect of this context; with pageable arrays the call returns after the copies. */
int dn_batch_upload(dn_ctx *ctx, const dn_batch_desc *batch);
/* page-locked host memory for the input arrays: hipHostMalloc / hipHostRegister behind a C signature */
int dn_host_alloc(size_t bytes, void **p);
void dn_host_free(void *p);
int dn_host_register(void *p, size_t bytes);
int dn_host_unregister(void *p);

/* ---- stages (stream-ordered; each works on the uploaded batch) ---- */
int dn_run_segment(dn_ctx *ctx);        /* detect_events (scrappie/event_detection.c:268) + event build + k-mer ranks (event_handling.cpp:546-592) */
int dn_run_rough_scaling(dn_ctx *ctx);  /* estimateScaling_quantiles (event_handling.cpp:510) */
int dn_run_banded(dn_ctx *ctx);         /* adaptive_banded_simple_event_align (event_handling.cpp:148): fill, backtrack, QC */
int dn_run_theilsen(dn_ctx *ctx);       /* estimateScaling_theilSen (event_handling.cpp:24) + eventsPerBase (:606) */
int dn_run_normalise(dn_ctx *ctx);      /* normaliseEvents (event_handling.h:13) == the four stages above */
int dn_run_eventalign(dn_ctx *ctx);     /* eventalign (alignment.h:22): windowed Viterbi + feature fill + tensor packing (reads.h:305-372) */
int dn_run_detect(dn_ctx *ctx);         /* the body of the reference's per-read loop (detect.cpp:876-896): dn_run_normalise + dn_run_eventalign + dn_run_cnn */

/* ---- CNN (replaces model_load_*_twoInputs + TF_SessionRun, tensor.cpp:12-106, detect.cpp:577-675) ----
 * The network is DATA: an ordered op list over `n_buffers` activation buffers + one fp32 weight blob
 * (dnascent_amd/cnn_model.py; the reference's SavedModel graph is absent from its checkout, only the weight shapes
 * survive).  Inputs are the three tensors runCNN feeds (core / residual sequence, 20 raw samples per position,
 * reads.h:305-372), already on the device after dn_run_eventalign; output = [n_positions, 3] class probabilities
 * (0 thymidine, 1 BrdU, 2 EdU; detect.cpp:695). */
enum { DN_CNN_ENCODE_GRU = 0, DN_CNN_CONV = 1, DN_CNN_DWCONV = 2, DN_CNN_ADD_RELU = 3, DN_CNN_DENSE_SOFTMAX = 4,
       DN_CNN_CONV_ADD = 5 /* CONV whose epilogue adds buffer `a` before the activation (residual join) */ };
typedef struct {
    int32_t op;                 /* DN_CNN_* */
    int32_t src, dst, a, b;     /* activation buffer indices */
    int32_t k, cin, cout, relu; /* kernel width, channels, fused ReLU */
    int32_t reserved;
    int64_t w, scale, shift;    /* offsets (in floats) into the weight blob: kernel, folded BatchNorm scale / shift(+bias) */
    int64_t aux[6];             /* ENCODE_GRU: kernel/recurrent/bias offsets of the two GRUs */
} dn_cnn_op;
int dn_load_cnn(dn_ctx *ctx, const dn_cnn_op *ops, uint32_t n_ops, const float *weights, uint64_t n_weights, uint32_t n_buffers);
/* how the convolutions multiply:
 *   FP32    exact fp32 MFMA;
 *   BF16X6  fp32 operands split exactly into three bf16 pieces, the six significant products accumulated in fp32 on the
 *           bf16 matrix cores (fp32-equivalent: dropped terms < 2^-24);
 *   F16X3   (default) two fp16 pieces, three products (dropped terms < 2^-22; half the matrix work of BF16X6).  fp16 has a
 *           narrow exponent range: a pass in which some activation exceeds 65504 -- or in which a whole layer's activations are
 *           below 2^-6, where the low pieces are subnormal and the split keeps an absolute 2^-25 instead of a relative 2^-22 --
 *           is detected on the device and repeated in BF16X6 automatically, and the context then stays on BF16X6 until the next
 *           dn_load_cnn / dn_cnn_set_math; dn_cnn_range_escalations counts those repeats.
 *           ABI 7, the canary: the maximum cannot see a layer that is mostly tiny beside a few large values, so the first sequences of every batch
 *           (>= 4 096 positions) also run with bf16 pieces and the two sets of probabilities are compared on the device; a difference above 1e-4
 *           (DN_CNN_CANARY_TOL; DN_CNN_CANARY=0: off) repeats the batch in BF16X6 like a range report.  dn_cnn_canaries: how many ran. */
enum { DN_CNN_MATH_FP32 = 0, DN_CNN_MATH_BF16X6 = 1, DN_CNN_MATH_F16X3 = 2 };
int dn_cnn_set_math(dn_ctx *ctx, int mode);
uint64_t dn_cnn_range_escalations(dn_ctx *ctx);
uint64_t dn_cnn_canaries(dn_ctx *ctx);
int dn_run_cnn(dn_ctx *ctx);            /* runCNN for every read that passed eventalign */
int dn_get_probabilities(dn_ctx *ctx, uint32_t read, uint64_t cap /* positions `probs` holds */, float *probs /* [n_positions * 3] */);
/* the TF_SessionRun seam itself (detect.cpp:653): n_seq sequences given as the three host tensors runCNN builds --
 * core [sum len], residual [sum len], signal [sum len][20] -- -> probs [sum len][3].  Needs only dn_load_cnn; it lets a
 * maintainer keep the reference's CPU eventalign and replace just the TensorFlow call. */
int dn_cnn_infer(dn_ctx *ctx, uint32_t n_seq, const uint32_t *len, const float *core, const float *residual, const float *signal,
                 float *probs);

/* ---- `DNAscent align` (alignment.cpp:747-898): the per-sample event table eventalign prints (alignment.cpp:697-733) ----
 * dn_set_align_table(ctx, 1) before dn_run_eventalign makes that call also materialise, for every read that passes, one row per
 * raw sample of every event labelled M and of every event labelled I before its window's last match, in the reference's
 * emission order: reference coordinate of the event, start of kmerStrand in referenceSeqMappedTo (reference_index + pos),
 * scaled sample (raw - shift) / scale, kind (0 match, 1 insertion).  The text of a record is host work (k-mers, "%f"). */
int dn_set_align_table(dn_ctx *ctx, int on);
int dn_get_align_rows(dn_ctx *ctx, uint32_t *n_rows /* [n_reads] */);
/* cap: rows EACH of the caller's arrays holds; the read's own row count (dn_get_align_rows) is what is written, DN_ERR_ARG if it exceeds cap */
int dn_get_align_table(dn_ctx *ctx, uint32_t read, uint32_t cap /* rows */, uint32_t *coord, uint32_t *ref_pos, double *value, uint8_t *kind);

/* ---- `detect --HMM` (detect.cpp:885): llAcrossRead (detect.cpp:393-574) + sequenceProbability (:235-378) ----
 * Fit models (config.h:53-54, import_poreModel_fitStdv data_IO.cpp:192): (mean, std) per 9-mer in kmer2index order.
 * dn_run_hmm needs dn_run_normalise only (the reference does not call eventalign in this mode).  Calls come back in the
 * order the reference emits them: ascending strand position for forward reads, descending for reverse reads (:405). */
int dn_load_fit_models(dn_ctx *ctx, const double *unlabelled_mean, const double *unlabelled_std, const double *analogue_mean,
                       const double *analogue_std /* each [DN_NKMER] */);
int dn_run_hmm(dn_ctx *ctx);
int dn_get_hmm_calls(dn_ctx *ctx, uint32_t read, uint64_t cap /* calls every array holds */, uint32_t *pos_on_ref, uint32_t *pos_on_query,
                     int32_t *global_pos /* :531-541 */, uint32_t *n_events, double *log_analogue, double *log_thymidine,
                     double *llr /* each [n_hmm_calls] */);

/* ---- per-read results ---- */
typedef struct {
    int32_t status;                 /* DN_READ_* */
    uint32_t n_samples;
    uint32_t n_scrappie;            /* et.n */
    uint32_t n_events;              /* r.events.size() */
    uint32_t n_kmers_query, n_kmers_ref;
    uint32_t n_bands;
    uint64_t band_cells;            /* n_bands * DN_BANDWIDTH */
    double rough_shift, rough_scale;/* after estimateScaling_quantiles */
    int32_t end_event;              /* event chosen at event_handling.cpp:329-340 */
    uint32_t n_aligned;             /* r.eventAlignment.size() before the QC clear */
    double avg_log_emission; int32_t spanned; int32_t max_gap; uint32_t n_cleaned;   /* :420-441 */
    double ts_slope, ts_intercept;  /* Theil-Sen medians (NaN when refinement was skipped, :33) */
    double shift, scale, events_per_base;   /* r.scalings */
    uint32_t n_positions;           /* r.refCoordToAP.size() after eventalign */
    uint32_t n_windows;             /* builtinViterbi calls */
    uint32_t detector_rechecks;     /* speculative segmentation chunks that had to be recomputed (diagnostic) */
    uint32_t n_hmm_calls;           /* lines llAcrossRead emits (after dn_run_hmm) */
} dn_read_summary;

int dn_get_summaries(dn_ctx *ctx, dn_read_summary *out /* [n_reads] */);

/* ---- bulk result of a batch: what runCNN leaves on every read (detect.cpp:677-731), one transfer per array ----
 * A "call" is a position whose strand 9-mer has 'T' in the middle -- the only positions runCNN reports, in the .detect text
 * (detect.cpp:690) and in the modbam tags (:704-707) alike.  Calls of read r are [call_off[r], call_off[r + 1]) in eventalign's
 * creation order (sequencing direction); failed reads have none.  All pointers are page-locked memory owned by the context,
 * valid until its next dn_batch_upload.  dn_collect waits for the batch (after dn_run_cnn / dn_run_detect). */
typedef struct {
    uint32_t n_reads;
    const dn_read_summary *summary;   /* [n_reads]: status, shift, scale, events_per_base, n_positions, ... */
    const uint64_t *call_off;         /* [n_reads + 1] */
    uint64_t n_calls;                 /* == call_off[n_reads] */
    const uint32_t *ref_coord;        /* reference coordinate of the position (alignment.cpp:648-650) */
    const uint32_t *query_idx;        /* refToQuery of its reference index (modbam key, detect.cpp:705) */
    const uint32_t *ref_idx;          /* index of the 9-mer's middle base in referenceSeqMappedTo (refToDel lookup, detect.cpp:704) */
    const float *p_edu, *p_brdu;      /* class 2 / class 1 of the network output (detect.cpp:695) */
    const char *kmer9;                /* [n_calls * 9] strand 9-mer, not NUL-terminated */
} dn_result_batch;
int dn_collect(dn_ctx *ctx, dn_result_batch *out);

/* ---- sizing a context ahead of its batches (hosts whose batches differ in shape: mixed read lengths) ----
 * A context's workspace is one grow-only slab sized by its largest batch so far; growing it frees the old slab, and hipFree waits for
 * the WHOLE device -- with several contexts in flight every regrowth drains the pipeline.  dn_batch_workspace_bytes runs the sizing pass
 * of dn_batch_upload alone (nothing is allocated or copied; the context holds no batch afterwards); dn_ctx_reserve makes the slab hold at
 * least workspace_bytes now and lets the first dn_collect allocate at least collect_bytes of page-locked result space.  Call both between
 * batches (the context's stream is waited for).  The reference has no counterpart: its per-read buffers are malloc'ed per read. */
int dn_batch_workspace_bytes(dn_ctx *ctx, const dn_batch_desc *batch, uint64_t *bytes);
int dn_ctx_reserve(dn_ctx *ctx, uint64_t workspace_bytes, uint64_t collect_bytes);
/* ABI 6.  dn_cnn_reserve: give the CNN lane this context runs on its activation buffers NOW, for passes of up to `rows` rows (0 = the pass cap, DN_CNN_ROWS),
 * instead of at the first pass that needs them: 16-32 GiB of hipMalloc per lane, which a host can take on a helper thread while its stream already runs on
 * the lanes that have theirs (run_detect: 1 s per lane right behind another process's exit, 4 lanes, inside the stream's enqueue path before this call existed).
 * Needs dn_load_cnn.  No reference counterpart (tensor.cpp's session allocates per call). */
int dn_cnn_reserve(dn_ctx *ctx, uint64_t rows);
/* ABI 5.  How many events a read's workspace holds: samples / samples_per_event + 64 (default 2: scrappie's detector cannot place more than one peak per
 * two samples -- its shortest window is 3, event_detection.h:19-25 -- so that bound never overflows).  The event, alignment and trace arrays are sized from it
 * (13 of the 21 GB of a 500 x 50 kb batch at the default); R10.4.1 reads carry one event per 5-8 samples, so a host that RETRIES may ask for 3 .. 16: a read
 * with more events makes dn_collect return DN_ERR_OVERFLOW for its batch (nothing is truncated silently), and the host submits that batch again after
 * dn_ctx_set_event_bound(ctx, 2) -- DNAscent::DetectStream::collect does exactly that.  Takes effect at the next dn_batch_upload.  The reference has no
 * counterpart: event_detection.c:268-319 callocs per read. */
int dn_ctx_set_event_bound(dn_ctx *ctx, uint32_t samples_per_event);
uint32_t dn_ctx_get_event_bound(const dn_ctx *ctx);

/* ---- intermediate taps (parity tests; NULL pointers are skipped) ----
 * Every tap takes `cap`: how many records (samples / events / k-mers / pairs / bands / positions / windows) EACH of the caller's arrays
 * holds.  The library knows the true count (dn_read_summary reports it); a tap whose count exceeds `cap` writes nothing and returns
 * DN_ERR_ARG with the two numbers in dn_last_error -- a stale summary can therefore never make a tap write past a buffer. */
/* prefix sums and t-statistics live only in registers / LDS of the segmentation kernels; dn_debug_keep_k1(ctx, 1) BEFORE
 * dn_batch_upload makes the next batches also write them to HBM (24 bytes per sample) so that the two taps below work */
int dn_debug_keep_k1(dn_ctx *ctx, int on);
/* ABI 7.  The peak detector (event_detection.c:122-198) is a serial state machine; the device runs it speculatively per 1 024-sample chunk from the default
 * state `samples` (default and maximum 192) before the chunk, verifies every hand-off exactly and re-walks a chunk from the true state when the
 * speculation missed (dn_read_summary.detector_rechecks counts those).  A shorter warm-up changes no result -- it only makes the exact redo run more
 * often: tests call dn_debug_seg_warm(ctx, 0) BEFORE dn_batch_upload to put every chunk with a peak pending at its start through that path. */
int dn_debug_seg_warm(dn_ctx *ctx, uint32_t samples);
int dn_get_prefix_sums(dn_ctx *ctx, uint32_t read, uint64_t cap /* samples */, double *sum /* [cap+1] */, double *sumsq /* [cap+1] */);
int dn_get_tstats(dn_ctx *ctx, uint32_t read, uint64_t cap /* samples */, float *t_short, float *t_long /* [n_samples] */);
int dn_get_scrappie_events(dn_ctx *ctx, uint32_t read, uint64_t cap, uint32_t *start, float *length, float *mean /* [n_scrappie] */);
int dn_get_events(dn_ctx *ctx, uint32_t read, uint64_t cap, double *mean, uint32_t *raw_start, uint32_t *raw_len /* [n_events] */);
int dn_get_kmer_ranks(dn_ctx *ctx, uint32_t read, uint64_t cap_query, uint64_t cap_ref, uint32_t *rank_query /* [n_kmers_query] */,
                      uint32_t *rank_ref /* [n_kmers_ref] */);
int dn_get_alignment(dn_ctx *ctx, uint32_t read, uint64_t cap, uint32_t *event_idx, uint32_t *kmer_idx /* [n_aligned] */);
int dn_get_cleaned(dn_ctx *ctx, uint32_t read, uint64_t cap, double *signal, uint32_t *rank /* [n_cleaned], backtrack order */);
int dn_get_trace(dn_ctx *ctx, uint32_t read, uint64_t cap /* bands */, uint8_t *trace /* [n_bands*100] */,
                 int32_t *band_event /* [n_bands] ll.event_idx */, int32_t *band_kmer /* [n_bands] */);
/* eventalign outputs, creation order == sequencing direction (reads.h:305-372) */
int dn_get_positions(dn_ctx *ctx, uint32_t read, uint64_t cap /* positions */, uint32_t *coord, uint32_t *query_idx, uint32_t *ref_idx,
                     int32_t *indel_score, char *kmer9, uint32_t *n_signal, float *signal20, float *core, float *residual);
int dn_get_windows(dn_ctx *ctx, uint32_t read, uint64_t cap /* windows */, uint32_t *ref_index, uint32_t *window_len, uint32_t *n_obs,
                   double *score);

/* the emission term of builtinViterbi as the device lattice evaluates it -- eln(normalPDF(mu, sigma, x)), alignment.cpp:273,347 with
 * probability.cpp:35-47,145-148; sigma of dn_load_pore_model -- for n (observation, level) pairs; log 0 comes back as NaN.
 * Parity tap: tests compare it with values the reference's own probability.cpp produced (tests/golden/ref_emission.npz). */
int dn_debug_emission(dn_ctx *ctx, uint32_t n, const double *x, const double *mu, double *out);

/* ---- measurement ---- */
enum { DN_K_SCAN = 0, DN_K_TSTAT, DN_K_DETECT, DN_K_EVENTS, DN_K_RANKS, DN_K_QUANTILE, DN_K_PREP, DN_K_BAND_FILL,
       DN_K_BAND_TRACE, DN_K_THEILSEN, DN_K_VITERBI, DN_K_CNN /* the whole network of a batch */, DN_K_HMM, DN_K_COUNT };
int dn_profile_enable(dn_ctx *ctx, int on);     /* HIP events around every kernel launch on the context's stream */
int dn_profile_get(dn_ctx *ctx, int kernel, double *total_ms, uint32_t *launches);
int dn_profile_reset(dn_ctx *ctx);
/* the network layer by layer: summed time and launches of op `op` of the loaded description (a fused separable layer is reported under
 * its depthwise op, its pointwise op reads 0), and the name of the kernel the op takes as rocprofv3 prints it ("" for a covered op) */
int dn_profile_get_layer(dn_ctx *ctx, uint32_t op, double *total_ms, uint32_t *launches, char *kernel, size_t kernel_cap);
const char *dn_kernel_name(int kernel);
size_t dn_device_bytes(const dn_ctx *ctx);      /* HBM currently held by the context + the CNN lanes of its device (shared by its contexts) */
/* Frees the process-wide CNN lanes (streams + activation buffers) of every device.  dn_ctx_destroy of the LAST context of a device does
 * it for that device by itself; dn_shutdown is for hosts that keep contexts alive but want the lanes' HBM back between runs.  The
 * lanes come back on the next dn_run_cnn.  A lane on which another host thread is enqueueing a pass at that moment (dn_run_cnn,
 * dn_cnn_infer, dn_collect's repeat) is left alone: the call then returns DN_ERR_STATE after freeing the others -- call it again. */
int dn_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
