// dn_dev.h -- device-side batch layout (SoA in HBM with per-read offsets) and shared device helpers.
// gfx950 only.  Kernels are built with -ffp-contract=off: every fused multiply-add is written as fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DN_W 100            // band width (config.h:41)
#define DN_TROW 32          // bytes per stored trace row: 128 slots x 2-bit from-code, planar (k2_banded.hip put_row)
#define DN_TPAD 264         // rows of padding per read: k2_chase loads whole 256-row tiles, k2_fill6 stores whole 8-row groups
#define DN_K 9
#define DN_RAWDEPTH_DEV 20 // reads.h:12 RAWDEPTH
#define DN_SEG_CHUNK 1024   // samples per speculative detector chunk
#define DN_SEG_WARM 192     // warm-up samples run before a chunk to reach the detector's true state
#define DN_SEG_PEAKCAP 512  // peaks per chunk (peaks are >= 2 samples apart)

struct DetState {           // one scrappie Detector (event_detection.c:10-21)
    int masked_to;
    int peak_pos;           // -1 == DEF_PEAK_POS
    float peak_val;
    int valid;
};
#define DN_SEG_CARRY 256    // k1_carry keeps the exact running sums {sum, sumsq} at every 256th sample (and at the read's end)
struct SegState { DetState s, l; };

struct ReadRes {            // per-read scalars produced on device (mirrors dn_read_summary)
    int status;
    unsigned n_samples, n_scrappie, n_events, n_kq, n_kr, n_bands;
    double q_shift, q_scale;
    int end_event;
    unsigned n_aligned;
    double avg_log_emission; int spanned; int max_gap; unsigned n_cleaned;
    double ts_slope, ts_intercept;
    double shift, scale, events_per_base;
    unsigned n_positions, n_windows, rechecks, aln_begin;   // aln_begin: first valid slot of the (back-filled) alignment array
    unsigned n_hmm_calls;    // --HMM: calls made by llAcrossRead
    float end_score;
    int seg_overflow;
};

struct BatchDev {
    int n_reads;
    // ---- inputs (uploaded) ----
    const int16_t *adc;      const uint64_t *samp_off;     // [n+1]
    const float *cal_off;    const float *cal_scale;
    const char *basecall;    const uint64_t *base_off;     // [n+1]
    const char *refseq;      const uint64_t *ref_off;      // [n+1]
    const uint32_t *ref2query;                              // at ref_off
    const int32_t *query2ref;                               // at base_off[r] + r
    const uint8_t *ref2del;
    const int32_t *ref_start, *ref_end;
    const uint8_t *is_rev;
    const double *model_mean; double sigma;
    const unsigned *model_pos;     // [4^9] position of each 9-mer's level in the sorted table (ties: any order, equal values)
    const double *model_sorted;    // [4^9] the levels in ascending order
    // ---- K1 workspace ----
    double2 *carry;          // [4 * chunks + n_reads] exact {sum[256 j], sumsq[256 j]} of read r at 4 * chunk_off[r] + r + j; the last one is {sum[n], sumsq[n]}
    double2 *psum;           // TAPS ONLY (dn_debug_keep_k1; else null): [samples] psum[i] = {sum[i+1], sumsq[i+1]}
    float *t1, *t2;          // TAPS ONLY: [samples] the two t-statistics
    const uint64_t *chunk_off;   // [n+1] detector chunk offsets
    unsigned *chunk_npk;     // [chunks]
    unsigned *chunk_peaks;   // [chunks * DN_SEG_PEAKCAP] peak positions ...
    double *chunk_psum;      // [chunks * DN_SEG_PEAKCAP] ... and the exact prefix sum sum[peak] that goes with each (event means need nothing else)
    SegState *chunk_in, *chunk_out;   // [chunks]
    double2 *chunk_sums;     // [chunks] {sum[short.peak_pos], sum[long.peak_pos]} of the peaks pending in chunk_out (k1_events' exact redo starts from them)
    int seg_warm;            // detector warm-up before a chunk: DN_SEG_WARM; dn_debug_seg_warm shortens it so that tests reach the redo path
    // scrappie events + DNAscent events; capacity per read = ev_off[r+1]-ev_off[r]
    const uint64_t *ev_off;  // [n+1]
    unsigned *et_start; float *et_mean;     // TAP ONLY (null unless keep_k1): scrappie event_t (start, mean); length = start[i+1]-start[i]
    double *ev_mean; unsigned *ev_start, *ev_len;   // r.events
    double *ev_x;            // (mean - shift)/scale with the rough scaling
    // ---- k-mer ranks ----
    unsigned *rank_q;        // at base_off
    unsigned *rank_r;        // at ref_off
    double *mu_q;            // model mean of rank_q (gathered once)
    // ---- banded alignment ----
    const uint64_t *trace_off;   // [n+1] in rows of DN_TROW bytes
    uint8_t *trace;
    unsigned *aln_event, *aln_kmer;   // capacity per read: (ev cap + n_kq + 2), at aln_off
    const uint64_t *aln_off;
    double *cl_sig; unsigned *cl_rank;    // cleaned signals / ranks (at aln_off)
    ReadRes *res;
};

// ----------------------------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned base_code(char c) {      // data_IO.cpp:131  A0 T1 G2 C3, others 0
    return c == 'T' ? 1u : (c == 'G' ? 2u : (c == 'C' ? 3u : 0u));
}

// order-preserving map double -> uint64 (total order == '<' on non-NaN doubles; -0 sorts just below +0)
__device__ __forceinline__ unsigned long long dkey(double x) {
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
    unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ float neg_inf() { return __int_as_float(0xff800000); }
__device__ __forceinline__ double neg_inf_d() { return __longlong_as_double(0xfff0000000000000ll); }

// uniform (scalar) broadcast of one lane's value
__device__ __forceinline__ float bcast_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double bcast_d(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
