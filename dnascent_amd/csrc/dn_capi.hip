// dn_capi.hip -- implementation of the C-ABI declared in include/dnascent_hip.h.
// Owns the device memory of one batch (SoA in HBM, per-read offsets), orders the stage kernels on one HIP
// stream and exposes the intermediate taps the parity tests read.  No CPU fallback anywhere.
#include "dnascent_hip.h"
#include "dn_dev.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

// kernel launchers (k1_segment.hip, k_scaling.hip, k2_banded.hip, k2b_viterbi.hip)
void k1_launch_scan(const BatchDev &, hipStream_t);
void k1_launch_tstat(const BatchDev &, unsigned, hipStream_t);
void k1_launch_detect(const BatchDev &, unsigned, hipStream_t);
void k1_launch_events(const BatchDev &, hipStream_t);
void ks_launch_ranks(const BatchDev &, unsigned, hipStream_t);
void ks_launch_quantile(const BatchDev &, hipStream_t);
void ks_launch_prep(const BatchDev &, unsigned, hipStream_t);
void ks_launch_theilsen(const BatchDev &, hipStream_t);
int k2_selftest_run(hipStream_t);
void k2_launch_fill(const BatchDev &, const void *, const void *, hipStream_t);
void k2_launch_chase(const BatchDev &, uint8_t *, hipStream_t);
void k2_launch_post(const BatchDev &, const uint8_t *, float *, const void *, hipStream_t);
hipError_t k2b_launch(const BatchDev &, const void *, const void *, const void *, unsigned, void *, hipStream_t);
size_t k2b_huge_scratch_bytes();
struct CnnRows { const unsigned *row_off; const uint8_t *valid; unsigned rows, r0, r1; const unsigned *n_pos; const uint64_t *io_off; };
struct CnnRun { const dn_cnn_op *ops; int n_ops; const float *wts; float *buf[8]; int n_buf; CnnRows rows; uint8_t *valid;
                 const float *core, *resid, *sig; float *probs; unsigned max_pos; const uint16_t *wts_split; const int64_t *wb_off;
                 int pieces; const float *post; unsigned *range_flag;
                 unsigned n_pass_pos; uint8_t *enc_len; unsigned *enc_hist; uint64_t *perm_src; unsigned *perm_row;
                 void (*mark)(void *who, int begin, int kind, hipStream_t st); void *mark_who;
                 unsigned *row_off_w; int *live; };
int k3_run(const CnnRun &, hipStream_t);
void k3_launch_canary_compare(const float *, const float *, const CnnRows &, unsigned, float, unsigned *, hipStream_t);
int k3_describe(const CnnRun &, int, char *, size_t);
struct HmmConstsH { double D2D, D2M, I2M, M2D, M2I, I2I, ln025, ln05; };
struct HmmReadH { double iM2M, eM2M, endM; };
struct HmmDevH { const double4 *unl, *ana; unsigned *poi, *n_poi, *n_ev; unsigned char *ok; double *la, *lt; };
void k_hmm_launch(const BatchDev &, const void *, const void *, const void *, unsigned, hipStream_t);

struct BandConstsH { double lp_stay, lp_step; };
struct FillConstsH { double lp_skip, lp_trim, C, sigma, rsigma; };
struct VitConstsH { double D2D, D2M, I2M, M2D, M2I, I2I; double c, d2, rd2, logc; double initD[66]; };
struct VitReadH { double iM2M, eM2M, eM2MorD, eOrI; int fail, pad; };   // fail: eln() of a negative number (the reference throws NegativeLog)
struct EaDevH { unsigned *coord, *qidx, *ridx; int *indel; unsigned *nsig; float *sig, *core, *resid;
                unsigned *win_ref, *win_len, *win_T; double *win_score;
                unsigned *al_coord, *al_rpos; double *al_val; unsigned char *al_kind; const unsigned long long *al_off; unsigned *al_n;
                unsigned char *redo; unsigned *resume; };
void k2b_rowcap_launch(const BatchDev &, unsigned long long *, hipStream_t);
void k2b_emission_tap_launch(const double *, const double *, double *, unsigned, const void *, hipStream_t);
// k_collect.hip: per-read call counts (centre base T), their exclusive scan, ordered compaction of the per-call outputs
struct CollectDev { const unsigned *coord, *qidx, *ridx; const float *probs; unsigned *cnt; unsigned long long *off;
                    unsigned *o_coord, *o_qidx, *o_ridx; float *o_edu, *o_brdu; char *o_kmer; };
void kc_launch_count(const BatchDev &, const void *, unsigned, hipStream_t);
void kc_launch_scan(const BatchDev &, const void *, hipStream_t);
void kc_launch_pack(const BatchDev &, const void *, unsigned, hipStream_t);
void kc_launch_npos(const BatchDev &, unsigned *, hipStream_t);
void kc_launch_nop(hipStream_t);

namespace {

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
};

struct ProfRec { int k; hipEvent_t a, b; int layer; };      // k == -1: op `layer` of the CNN description

}  // namespace

// The CNN LANE of a device.  The network's kernels fill the whole GPU on their own; two networks running side by side (two
// contexts that reached dn_run_cnn at the same time) only evict each other's activations from L2 / MALL and double the activation
// memory.  All contexts of a process on one device therefore run their CNN passes on ONE extra stream, in submission order, with
// ONE set of activation buffers: a context hands its batch over with an event, the lane's last kernel hands it back.  Meanwhile the
// latency-bound stages of the other contexts (one wavefront per read) run beside the network on their own streams.
struct CnnLane {
    hipStream_t stream = nullptr;
    std::mutex mu;                                       // enqueue order == execution order
    DevBuf buf[8], valid, enclen, enchist, permsrc, permrow, live;
    size_t bytes = 0;
    int users = 0;                                       // guarded by g_lane_mu: passes that hold `mu` or are about to take it (lane_get .. lane_put)
};
#define DN_MAX_LANES 16
static std::mutex g_lane_mu;
static CnnLane *g_lane[64][DN_MAX_LANES] = { { nullptr } };
static unsigned g_ctx_seq = 0;
static unsigned g_dev_ctx[64] = { 0 };                  // live contexts per device: the last one to go takes the device's lanes with it
// caller holds g_lane_mu.  A lane in USE -- a CNN pass of another host thread has looked it up (lane_get counted it in, under g_lane_mu) and is
// enqueueing on it or about to lock it -- is left alone and counted: freeing it would pull the stream, the mutex and the activation buffers from
// under that pass (round-3 advisor).  The count closes the window the mutex alone left open (round-4 advisor): between lane_get returning the
// pointer and the pass locking `mu`, a try_lock here succeeded and the pass went on to lock a deleted mutex.  Returns the number of busy lanes.
static int lanes_free_device(int dev) {
    if (dev < 0 || dev >= 64) return 0;
    int busy = 0;
    for (unsigned l = 0; l < DN_MAX_LANES; l++) {
        CnnLane *L = g_lane[dev][l];
        if (!L) continue;
        if (L->users > 0 || !L->mu.try_lock()) { busy++; continue; }
        (void)hipSetDevice(dev);
        if (L->stream) { (void)hipStreamSynchronize(L->stream); (void)hipStreamDestroy(L->stream); }
        for (DevBuf &b : L->buf) if (b.p) (void)hipFree(b.p);
        for (DevBuf *b : { &L->valid, &L->enclen, &L->enchist, &L->permsrc, &L->permrow, &L->live }) if (b->p) (void)hipFree(b->p);
        g_lane[dev][l] = nullptr;                        // nobody can find the lane any more (lookups go through g_lane under g_lane_mu) ...
        L->mu.unlock();
        delete L;                                        // ... so nobody can be waiting on its mutex
    }
    return busy;
}
static size_t lanes_bytes_device(int dev) {
    size_t n = 0;
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_lane_mu);
    for (unsigned l = 0; l < DN_MAX_LANES; l++) if (g_lane[dev][l]) n += g_lane[dev][l]->bytes;
    return n;
}
// how many lanes a device has (DN_CNN_LANES, default 4): contexts are dealt to them round-robin.  One lane serialises every
// network of the process (least memory: one set of activation buffers); measured with 4 x 500 x 50 kb reads in flight on one box:
// 1 lane 466, 2 lanes 478, 4 lanes 493 Msamples/s -- a second network's kernels fill the gaps a lone one leaves while the CUs' LDS
// is held by the one-wavefront-per-read stages of the other batches
static unsigned lane_count() {
    const char *e = getenv("DN_CNN_LANES");
    const unsigned v = e ? (unsigned)strtoul(e, nullptr, 10) : 4u;
    return std::min<unsigned>(std::max<unsigned>(v, 1u), DN_MAX_LANES);
}

struct dn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool use_dpp = true;
    std::string err;
    size_t dev_bytes = 0;
    // per-batch workspaces come out of grow-only slabs: after the first batches no hipMalloc / hipFree happens per upload
    struct Slab { char *p; size_t cap, used; };
    std::vector<Slab> slabs; size_t slab_cur = 0;
    bool measuring = false; size_t measured = 0;        // dn_batch_upload's first pass: sizes only (see fit_slab)
    // model
    double *d_model = nullptr; double sigma = 0.14; bool have_model = false;
    unsigned *d_model_pos = nullptr; double *d_model_sorted = nullptr;
    // batch
    BatchDev B{};
    bool have_batch = false;
    int stage = 0;   // 0 none, 1 uploaded, 2 segmented, 3 rough scaled, 4 banded, 5 theil-sen, 6 eventaligned
    std::vector<uint64_t> h_samp_off, h_base_off, h_ref_off, h_chunk_off, h_ev_off, h_aln_off, h_trace_off;
    unsigned max_samples = 0, max_chunks = 0, max_len = 0, max_evcap = 0;
    unsigned ev_div = 2;                                  // the event workspace of a read holds samples / ev_div + 64 events (dn_ctx_set_event_bound)
    std::vector<ReadRes> h_res;
    uint8_t *d_path_from = nullptr; float *d_path_lp = nullptr;
    uint64_t *d_trace_off = nullptr;
    void *d_bandc = nullptr;
    // stream-ordered host work (hipLaunchHostFunc): page-locked mirrors of the per-read scalars and of the per-read constants
    // the reference obtains from libm; grow-only, sized for the largest batch seen
    ReadRes *p_res = nullptr; BandConstsH *p_bandc = nullptr; VitReadH *p_vit = nullptr; size_t p_cap = 0;
    int async_err = 0; uint32_t async_err_read = 0;     // written by host functions on the runtime's callback thread, read after a sync
    uint32_t n_batch = 0;                               // reads of the resident batch (what the host functions loop over)
    // dn_collect: device-side compaction + page-locked result block
    unsigned *d_call_cnt = nullptr; unsigned long long *d_call_off = nullptr;
    DevBuf col_dev; void *col_host = nullptr; size_t col_host_cap = 0, col_reserve = 0;
    size_t last_need = 0; bool measure_only = false;      // dn_batch_workspace_bytes: the sizing pass of dn_batch_upload alone
    unsigned long long *p_call_off = nullptr; dn_read_summary *p_summary = nullptr;
    bool upload_pinned = false;
    bool keep_k1 = false;                               // dn_debug_keep_k1
    int seg_warm = DN_SEG_WARM;                         // dn_debug_seg_warm
    size_t n_ref_T = 0; char *d_col = nullptr;
    unsigned *p_cnn_rowoff = nullptr; uint64_t *p_cnn_iooff = nullptr; size_t cnn_meta_cap = 0; unsigned *p_cnn_flag = nullptr; bool cnn_pending = false;
    FillConstsH fc{};
    std::vector<int64_t> cnn_wb_off, cnn_wh_off; uint16_t *d_cnn_wb = nullptr, *d_cnn_wh = nullptr; size_t cnn_nwb = 0, cnn_nwh = 0; int cnn_math = DN_CNN_MATH_F16X3;
    std::vector<float> cnn_post, cnn_one; unsigned *d_cnn_flag = nullptr; uint64_t cnn_escalations = 0; bool cnn_f16_off = false;
    unsigned cnn_underflow_streak = 0; bool cnn_bf16_once = false;        // see cnn_note_escalation
    DevBuf cnn_canary; uint64_t cnn_canaries = 0;                         // the canary's probabilities (cnn_execute); how many canaries ran
    std::vector<dn_cnn_op> cnn_ops; float *d_cnn_w = nullptr; size_t cnn_nw = 0; int cnn_nbuf = 0; DevBuf cnn_rowoff, cnn_npos, cnn_iooff, cnn_in[3], cnn_out;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;    // hand-over to / from the device's CNN lane
    unsigned lane_id = 0;
    float *d_probs = nullptr;
    double4 *d_fit[2] = { nullptr, nullptr }; bool have_fit = false, hmm_done = false;
    DevBuf hmm_poi, hmm_npoi, hmm_nev, hmm_ok, hmm_la, hmm_lt, hmm_reads;
    bool want_align = false, have_align = false; DevBuf al_coord, al_rpos, al_val, al_kind, al_off, al_n; std::vector<unsigned long long> h_al_off; std::vector<unsigned> h_al_n;
    std::vector<unsigned> h_npoi, h_nhmm;
    std::vector<int32_t> h_ref_start, h_ref_end; std::vector<uint8_t> h_is_rev;
    VitConstsH vc{}; EaDevH ea{}; VitReadH *d_vitread = nullptr; unsigned max_ref = 0; void *d_k2b_huge = nullptr;
    // profiling
    bool prof = false;
    std::vector<ProfRec> pending;
    double prof_ms[DN_K_COUNT] = {0};
    uint32_t prof_n[DN_K_COUNT] = {0};
    std::vector<double> layer_ms; std::vector<uint32_t> layer_n;     // per op of the CNN description (dn_profile_get_layer)
};

// Contexts are meant to be used several at a time (one per in-flight batch); their streams only run concurrently when the
// runtime has that many hardware queues (ROCm default: 4).  Effective when this library is loaded before the HIP runtime
// initialises; a host that initialises HIP first sets GPU_MAX_HW_QUEUES itself.
__attribute__((constructor)) static void dn_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static int fail(dn_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail((c), DN_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// smallest new slab; DN_SLAB_MIN_MB exists for the test that forces the exact-size retry of dalloc
static size_t slab_min_bytes() {
    const char *e = getenv("DN_SLAB_MIN_MB");
    return (size_t)(e ? strtoull(e, nullptr, 10) : 256ull) << 20;
}
template <class T>
static int dalloc(dn_ctx *c, T **p, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (c->measuring) { c->measured += bytes; *p = nullptr; return DN_OK; }
    for (; c->slab_cur < c->slabs.size(); c->slab_cur++) {
        dn_ctx::Slab &s = c->slabs[c->slab_cur];
        if (s.used + bytes <= s.cap) { *p = (T *)(s.p + s.used); s.used += bytes; return DN_OK; }
    }
    // a new slab: at least 256 MiB or half of what is already held, so a handful of slabs cover any steady-state batch size
    size_t held = 0;
    for (const dn_ctx::Slab &s : c->slabs) held += s.cap;
    const size_t cap = std::max(bytes, std::max<size_t>(slab_min_bytes(), held / 2));
    void *q = nullptr;
    size_t got = cap;
    hipError_t e = hipMalloc(&q, cap);
    if (e != hipSuccess && cap > bytes) {                                  // memory is tight: exactly what is needed
        (void)hipGetLastError();                                           // the failed attempt must not surface after the next launch
        got = bytes; q = nullptr;
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, DN_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    c->slabs.push_back({ (char *)q, got, bytes }); c->dev_bytes += got;
    c->slab_cur = c->slabs.size() - 1;
    *p = (T *)q;
    return DN_OK;
}
static void dfree_all(dn_ctx *c) {                       // start of a new batch: every slab is reusable from its beginning
    for (dn_ctx::Slab &s : c->slabs) s.used = 0;
    c->slab_cur = 0;
}
static void dfree_slabs(dn_ctx *c) {
    for (dn_ctx::Slab &s : c->slabs) hipFree(s.p);
    c->slabs.clear(); c->slab_cur = 0;
}
// The workspace of a batch is ONE slab sized from the batch itself: dn_batch_upload first runs its placement with dalloc only adding
// up (every size follows from the batch's offsets), then makes sure a single slab of that size (+ 1/16 so that the next, slightly
// larger batch still fits) exists, then places for real.  Growing slab by slab as the placements came (round 1) left a context
// holding ~35 GB for a 500 x 50 kb batch that needs ~14: skipped tails of earlier slabs and geometric growth.
static int fit_slab(dn_ctx *c, size_t need) {
    if (c->slabs.size() == 1 && c->slabs[0].cap >= need) return DN_OK;
    for (dn_ctx::Slab &s : c->slabs) { hipFree(s.p); c->dev_bytes -= s.cap; }
    c->slabs.clear(); c->slab_cur = 0;
    size_t cap = std::max(need + need / 16 + (1u << 20), slab_min_bytes());
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, cap);
    if (e != hipSuccess) { (void)hipGetLastError(); cap = need; q = nullptr; e = hipMalloc(&q, cap); }   // memory is tight: exactly what is needed, and THAT capacity recorded
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, DN_ERR_HIP, "hipMalloc(%zu) for the batch workspace: %s", need, hipGetErrorString(e)); }
    c->slabs.push_back({ (char *)q, cap, 0 }); c->dev_bytes += cap;
    return DN_OK;
}
// grow-only side buffer.  Growth is GEOMETRIC with a floor: hipFree waits for the whole device, so with several batches in flight every regrowth
// drains the pipeline -- on mixed read lengths (28 .. 1 900 reads per batch) a context used to regrow its per-read tables batch after batch
// (round 4: 5.5 s of a 9.7 s run inside dn_batch_upload).  The per-read tables are a few bytes per read: 64 KiB covers any batch at once.
static int dgrow(dn_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return DN_OK;
    bytes = std::max<size_t>(std::max<size_t>(bytes + bytes / 2, 2 * b.cap), 64u << 10);
    if (b.p) { hipFree(b.p); c->dev_bytes -= b.cap; }
    b.p = nullptr; b.cap = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(c, DN_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    b.cap = bytes; c->dev_bytes += bytes;
    return DN_OK;
}

static const char *KNAMES[DN_K_COUNT] = { "k1_scan", "k1_tstat", "k1_detect", "k1_events", "k_ranks", "k_quantile", "k_prep",
                                          "k2_fill", "k2_chase+k2_post", "k_theilsen", "k2b_viterbi", "k3_cnn", "k_hmm" };

struct Timed {
    dn_ctx *c; int k; hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    Timed(dn_ctx *c_, int k_, hipStream_t st_ = nullptr) : c(c_), k(k_), st(st_ ? st_ : c_->stream) {
        if (c->prof) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st); }
    }
    ~Timed() {
        if (c->prof) { hipEventRecord(b, st); c->pending.push_back({k, a, b, -1}); }
    }
};

static int lane_grow(dn_ctx *c, CnnLane *L, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return DN_OK;
    // hipFree waits for the device, so a buffer another context's pass is still using is never pulled from under it; growth only
    // happens while the first batches of a run establish the sizes
    if (b.p) { hipFree(b.p); L->bytes -= b.cap; }
    b.p = nullptr; b.cap = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, DN_ERR_HIP, "hipMalloc(%zu) for the CNN lane: %s", bytes, hipGetErrorString(e)); }
    b.cap = bytes; L->bytes += bytes;
    return DN_OK;
}
// CU partition (DN_FRONT_CUS = n, 0 = off).  The per-batch stages before the CNN are latency-bound chains with one wavefront (or one
// small workgroup) per read: they need few CUs but hold LDS for a long time (k2b_eventalign: 50 KB per wavefront for ~90 ms), which
// keeps the network's large workgroups (k3_sep_ws: 155 KB) off every CU they sit on.  With a CU mask the contexts' streams are
// confined to n CUs (spread over the XCDs) and the CNN lanes to the others.
static unsigned front_cus() {
    const char *e = getenv("DN_FRONT_CUS");
    const unsigned v = e ? (unsigned)strtoul(e, nullptr, 10) : 0u;
    return std::min(v, 192u);
}
static void front_mask(uint32_t (&m)[8], bool complement) {
    for (int i = 0; i < 8; i++) m[i] = 0u;
    const unsigned n = front_cus();
    for (unsigned k = 0; k < n; k++) { const unsigned bit = (k % 8u) * 32u + (k / 8u); m[bit >> 5] |= 1u << (bit & 31u); }
    if (complement) for (int i = 0; i < 8; i++) m[i] = ~m[i];
}

static CnnLane *lane_get(dn_ctx *c) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    if (c->device < 0 || c->device >= 64) return nullptr;
    if (!g_lane[c->device][c->lane_id]) {
        CnnLane *L = new CnnLane();
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = numerically largest = lowest priority
        bool made = false;
        if (front_cus()) {
            uint32_t m[8]; front_mask(m, true);
            made = hipExtStreamCreateWithCUMask(&L->stream, 8, m) == hipSuccess;
            if (!made) (void)hipGetLastError();
        }
        const char *lp = getenv("DN_LANE_PRIO");                   // experiment switch: 0 lowest (default), 1 the middle of the range, 2 highest
        const int lane_prio = !lp || lp[0] == '0' ? lo : lp[0] == '2' ? hi : (lo + hi) / 2;
        if (!made && hipStreamCreateWithPriority(&L->stream, hipStreamNonBlocking, lane_prio) != hipSuccess) { delete L; return nullptr; }
        g_lane[c->device][c->lane_id] = L;
    }
    g_lane[c->device][c->lane_id]->users++;               // counted in while g_lane_mu is held: from here on lanes_free_device leaves the lane alone
    return g_lane[c->device][c->lane_id];
}
static void lane_put(CnnLane *L) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    L->users--;
}
struct LaneUse {                                           // lane_get .. lane_put around a pass's enqueue
    CnnLane *L;
    explicit LaneUse(CnnLane *l) : L(l) {}
    ~LaneUse() { if (L) lane_put(L); }
    LaneUse(const LaneUse &) = delete; LaneUse &operator=(const LaneUse &) = delete;
};

// HIP event pairs around every op of the network (layer = index into the description's op list), profiling only
static void cnn_mark(void *who, int begin, int layer, hipStream_t st) {
    dn_ctx *c = (dn_ctx *)who;
    if (begin) { hipEvent_t a; hipEventCreate(&a); hipEventRecord(a, st); c->pending.push_back({-1, a, nullptr, layer}); }
    else if (!c->pending.empty() && !c->pending.back().b) { hipEvent_t b; hipEventCreate(&b); hipEventRecord(b, st); c->pending.back().b = b; }
}

static void prof_collect(dn_ctx *c) {
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto &p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.k >= 0) { c->prof_ms[p.k] += ms; c->prof_n[p.k]++; }
            else if (p.layer >= 0 && (size_t)p.layer < c->layer_ms.size()) { c->layer_ms[(size_t)p.layer] += ms; c->layer_n[(size_t)p.layer]++; }
        }
        hipEventDestroy(p.a); hipEventDestroy(p.b);
    }
    c->pending.clear();
}

template <class T>
static int upload(dn_ctx *c, const T **dst, const T *src, size_t n) {
    T *d = nullptr;
    int rc = dalloc(c, &d, n);
    if (rc) return rc;
    if (n && !c->measuring) HIPCHK(c, hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dst = d;
    return DN_OK;
}

template <class T>
static int d2h(dn_ctx *c, T *dst, const T *src, size_t n) {
    if (!dst || n == 0) return DN_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DN_OK;
}

extern "C" {

int dn_abi_version(void) { return DN_ABI_VERSION; }

int dn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *dn_kernel_name(int k) { return (k >= 0 && k < DN_K_COUNT) ? KNAMES[k] : "?"; }

// one context less on its device; the last one frees the device's CNN lanes (round-2 advisor: lanes used to outlive every context)
static void ctx_unregister(dn_ctx *c) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    if (c->device < 0 || c->device >= 64 || g_dev_ctx[c->device] == 0) return;
    if (--g_dev_ctx[c->device] == 0) (void)lanes_free_device(c->device);
}

int dn_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    int busy = 0;
    for (int d = 0; d < 64; d++) busy += lanes_free_device(d);
    return busy ? DN_ERR_STATE : DN_OK;                  // a dn_run_cnn / dn_cnn_infer of another host thread was in progress: its lane stays, call again
}

int dn_ctx_create(int device, void *hip_stream, dn_ctx **out) {
    if (!out) return DN_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DN_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return DN_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return DN_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DN_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "dnascent_hip: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
        return DN_ERR_NO_DEVICE;
    }
    dn_ctx *c = new dn_ctx();
    c->device = device;
    { std::lock_guard<std::mutex> lk(g_lane_mu); c->lane_id = (g_ctx_seq++) % lane_count(); if (device < 64) g_dev_ctx[device]++; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        // the per-batch stages are latency-bound chains of small launches: they get the highest stream priority so that their
        // workgroups are placed as soon as they are ready, beside the CNN lane's (lowest priority) big grids
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char *pe = getenv("DN_STREAM_PRIO");
        const bool prio = !(pe && pe[0] == '0');
        bool made = false;
        if (front_cus()) {
            uint32_t m[8]; front_mask(m, false);
            made = hipExtStreamCreateWithCUMask(&c->stream, 8, m) == hipSuccess;
            if (!made) { (void)hipGetLastError(); fprintf(stderr, "dnascent_hip: CU-masked stream unavailable, using a plain one\n"); }
        }
        if (!made && (prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pe && pe[0] == '2' ? lo : hi) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { ctx_unregister(c); delete c; return DN_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
        if (c->own_stream) hipStreamDestroy(c->stream);
        ctx_unregister(c); delete c; return DN_ERR_HIP;
    }
    const int st = k2_selftest_run(c->stream);
    if (st != 1) {
        // wave-shift DPP did not behave as specified: refuse rather than compute wrong bands
        const char *force = getenv("DN_FORCE_SHFL");
        if (!(force && force[0] == '1')) {
            fprintf(stderr, "dnascent_hip: wave-shift DPP self-test failed (%d); set DN_FORCE_SHFL=1 to use ds_bpermute shifts\n", st);
            if (c->own_stream) hipStreamDestroy(c->stream);
            ctx_unregister(c); delete c;
            return DN_ERR_HIP;
        }
    }
    {
        const char *force = getenv("DN_FORCE_SHFL");
        c->use_dpp = !(force && force[0] == '1');
    }
    *out = c;
    return DN_OK;
}

void dn_ctx_destroy(dn_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    dfree_slabs(c);
    if (c->p_res) hipHostFree(c->p_res);
    if (c->p_bandc) hipHostFree(c->p_bandc);
    if (c->p_vit) hipHostFree(c->p_vit);
    if (c->p_call_off) hipHostFree(c->p_call_off);
    if (c->p_summary) hipHostFree(c->p_summary);
    if (c->col_host) hipHostFree(c->col_host);
    if (c->p_cnn_rowoff) { hipHostFree(c->p_cnn_rowoff); hipHostFree(c->p_cnn_iooff); }
    if (c->p_cnn_flag) hipHostFree(c->p_cnn_flag);
    if (c->col_dev.p) hipFree(c->col_dev.p);
    if (c->d_model) hipFree(c->d_model);
    if (c->d_model_pos) hipFree(c->d_model_pos);
    if (c->d_model_sorted) hipFree(c->d_model_sorted);
    if (c->d_cnn_w) hipFree(c->d_cnn_w);
    if (c->d_cnn_wb) hipFree(c->d_cnn_wb);
    if (c->d_cnn_wh) hipFree(c->d_cnn_wh);
    if (c->d_cnn_flag) hipFree(c->d_cnn_flag);
    if (c->d_k2b_huge) hipFree(c->d_k2b_huge);
    for (auto *p : c->d_fit) if (p) hipFree(p);
    for (DevBuf *b : { &c->hmm_poi, &c->hmm_npoi, &c->hmm_nev, &c->hmm_ok, &c->hmm_la, &c->hmm_lt, &c->hmm_reads, &c->al_coord, &c->al_rpos,
                       &c->al_val, &c->al_kind, &c->al_off, &c->al_n }) if (b->p) hipFree(b->p);
    for (DevBuf *b : { &c->cnn_rowoff, &c->cnn_npos, &c->cnn_iooff, &c->cnn_in[0], &c->cnn_in[1], &c->cnn_in[2], &c->cnn_out, &c->cnn_canary }) if (b->p) hipFree(b->p);
    if (c->ev_ready) hipEventDestroy(c->ev_ready);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->own_stream) hipStreamDestroy(c->stream);
    ctx_unregister(c);
    delete c;
}

const char *dn_last_error(const dn_ctx *c) { return c ? c->err.c_str() : "null context"; }

static int async_status(dn_ctx *c);
int dn_sync(dn_ctx *c) {
    if (!c) return DN_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return async_status(c);
}

size_t dn_device_bytes(const dn_ctx *c) { return c ? c->dev_bytes + lanes_bytes_device(c->device) : 0; }

int dn_load_pore_model(dn_ctx *c, const double *mean, double sigma) {
    if (!c || !mean || !(sigma > 0.)) return DN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_model) {
        HIPCHK(c, hipMalloc((void **)&c->d_model, DN_NKMER * sizeof(double)));
        c->dev_bytes += DN_NKMER * sizeof(double);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_model, mean, DN_NKMER * sizeof(double), hipMemcpyHostToDevice, c->stream));
    {
        // the sorted table + each k-mer's position in it: order statistics of model levels (estimateScaling_quantiles,
        // event_handling.cpp:533) become order statistics of 18-bit integers on the device
        std::vector<unsigned> order(DN_NKMER), pos(DN_NKMER);
        for (unsigned k = 0; k < DN_NKMER; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return mean[a] < mean[b]; });
        std::vector<double> sorted(DN_NKMER);
        for (unsigned i = 0; i < DN_NKMER; i++) { pos[order[i]] = i; sorted[i] = mean[order[i]]; }
        if (!c->d_model_pos) {
            HIPCHK(c, hipMalloc((void **)&c->d_model_pos, DN_NKMER * sizeof(unsigned)));
            HIPCHK(c, hipMalloc((void **)&c->d_model_sorted, DN_NKMER * sizeof(double)));
            c->dev_bytes += DN_NKMER * (sizeof(unsigned) + sizeof(double));
        }
        HIPCHK(c, hipMemcpyAsync(c->d_model_pos, pos.data(), DN_NKMER * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_model_sorted, sorted.data(), DN_NKMER * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));          // the vectors are locals
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->sigma = sigma;
    c->have_model = true;
    // constants of logProbabilityMatch (event_handling.cpp:134-135) and the band penalties (:179-183), computed with the
    // host libm exactly as the reference computes them
    const float lisp = (float)log(0.3989422804014327);
    c->fc.C = (double)lisp - log(sigma);
    c->fc.sigma = sigma;
    c->fc.rsigma = 1.0 / sigma;
    c->fc.lp_skip = log(1e-30);
    c->fc.lp_trim = log(0.01);
    // builtinViterbi's fixed transitions (alignment.cpp:199-204, config.h:42) and normalPDF's constants as the reference
    // build evaluates them (probability.cpp:145-148 with pow(v, 2.0) folded to v*v), all with the host libm
    VitConstsH &v = c->vc;
    v.D2D = log(0.3); v.D2M = log(0.7); v.I2M = log(0.999); v.M2D = log(0.0025); v.M2I = log(0.001); v.I2I = log(0.001);
    const double s2 = sigma * sigma;
    v.d2 = s2 + s2;
    v.rd2 = 1.0 / v.d2;
    v.c = 1.0 / sqrt(M_PI * v.d2);
    v.logc = log(v.c);
    v.initD[0] = 0.0 + v.M2D;                                   // alignment.cpp:241
    for (int i = 1; i < 66; i++) v.initD[i] = v.initD[i - 1] + v.D2D;   // :246-251
    return DN_OK;
}

static void cnn_note_escalation(dn_ctx *c, unsigned flag);
int dn_batch_upload(dn_ctx *c, const dn_batch_desc *d) {
    if (!c || !d) return DN_ERR_ARG;
    if (!c->have_model) return fail(c, DN_ERR_STATE, "dn_load_pore_model must be called first");
    HIPCHK(c, hipSetDevice(c->device));
    static const bool up_trace = [] { const char *e = getenv("DN_TRACE_SUBMIT"); return e && e[0] == '1'; }();
    auto up_now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    auto up_cpu = [] { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    const double up_t0 = up_trace ? up_now() : 0.0, up_c0 = up_trace ? up_cpu() : 0.0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double up_t1 = up_trace ? up_now() : 0.0;
    prof_collect(c);
    const double up_t2 = up_trace ? up_now() : 0.0;
    dfree_all(c);
    c->have_batch = false; c->stage = 0;
    // a batch that raised the fp16 range flag and was never collected must not leave it to the next one (round-2 advisor)
    if (c->cnn_pending) {
        c->cnn_pending = false;
        if (c->p_cnn_flag && *c->p_cnn_flag) cnn_note_escalation(c, *c->p_cnn_flag);      // the pass did not fit fp16: remembered, the dropped batch is not repeated
    }
    if (c->p_cnn_flag) *c->p_cnn_flag = 0;
    if (c->d_cnn_flag) HIPCHK(c, hipMemsetAsync(c->d_cnn_flag, 0, sizeof(unsigned), c->stream));
    const uint32_t n = d->n_reads;
    if (n == 0) {                                         // an empty buffer of reads is legal: every stage is a no-op
        memset(&c->B, 0, sizeof(c->B));
        c->h_samp_off.assign(1, 0); c->h_base_off.assign(1, 0); c->h_ref_off.assign(1, 0);
        c->h_res.clear(); c->hmm_done = false;
        c->have_batch = true; c->stage = 1;
        return DN_OK;
    }
    if (n > 65535) return fail(c, DN_ERR_ARG, "at most 65535 reads per batch (grid.y)");
    BatchDev &B = c->B;
    memset(&B, 0, sizeof(B));
    B.n_reads = (int)n;
    B.seg_warm = c->seg_warm;
    B.model_mean = c->d_model; B.sigma = c->sigma;
    B.model_pos = c->d_model_pos; B.model_sorted = c->d_model_sorted;
    c->h_samp_off.assign(d->adc_off, d->adc_off + n + 1);
    c->h_base_off.assign(d->basecall_off, d->basecall_off + n + 1);
    c->h_ref_off.assign(d->refseq_off, d->refseq_off + n + 1);
    c->h_ref_start.assign(d->ref_start, d->ref_start + n); c->h_ref_end.assign(d->ref_end, d->ref_end + n);
    c->h_is_rev.assign(d->is_reverse, d->is_reverse + n);
    c->hmm_done = false;
    const uint64_t S = c->h_samp_off[n], NB = c->h_base_off[n], NR = c->h_ref_off[n];
    c->h_chunk_off.assign(n + 1, 0); c->h_ev_off.assign(n + 1, 0); c->h_aln_off.assign(n + 1, 0); c->h_trace_off.assign(n + 1, 0);
    c->max_samples = c->max_chunks = c->max_len = c->max_evcap = 0;
    for (uint32_t r = 0; r < n; r++) {
        const uint64_t ns = c->h_samp_off[r + 1] - c->h_samp_off[r];
        const uint64_t nb = c->h_base_off[r + 1] - c->h_base_off[r];
        const uint64_t nr = c->h_ref_off[r + 1] - c->h_ref_off[r];
        if (ns >= (1ull << 31) || nb >= (1ull << 31)) return fail(c, DN_ERR_ARG, "read %u too long", r);
        if (ns < 16 || nb < DN_K + 1 || nr < DN_K) return fail(c, DN_ERR_ARG, "read %u too short (samples %llu, bases %llu, ref %llu)", r,
                                                                  (unsigned long long)ns, (unsigned long long)nb, (unsigned long long)nr);
        const uint64_t nch = (ns + DN_SEG_CHUNK - 1) / DN_SEG_CHUNK;
        // events of a read: the detector cannot place more than one peak per two samples (its shortest window is 3: event_detection.h:19-25), which is the
        // default bound; a driver that retries a batch on DN_ERR_OVERFLOW (DNAscent::DetectStream) may ask for a tighter one (dn_ctx_set_event_bound: R10.4.1
        // reads carry one event per 5-8 samples) -- event, alignment and trace arrays are sized from it: 13 of the 21 GB of a 500 x 50 kb batch
        const uint64_t evcap = ns / c->ev_div + (c->ev_div > 2 ? 64 : 8);
        c->h_chunk_off[r + 1] = c->h_chunk_off[r] + nch;
        c->h_ev_off[r + 1] = c->h_ev_off[r] + evcap;
        c->h_aln_off[r + 1] = c->h_aln_off[r] + evcap + nb + 8;
        // trace rows (32 B each): n_bands = events + k-mers + 2 is only known on the device, so the allocation takes the bound
        // events <= evcap -- nothing on the host has to wait for the segmentation.  Padded for whole-tile loads / 8-row stores.
        c->h_trace_off[r + 1] = c->h_trace_off[r] + ((evcap + nb + 2 + DN_TPAD + 7) & ~7ull);
        c->max_samples = std::max<unsigned>(c->max_samples, (unsigned)ns);
        c->max_chunks = std::max<unsigned>(c->max_chunks, (unsigned)nch);
        c->max_len = std::max<unsigned>(c->max_len, (unsigned)std::max(nb, nr));
        c->max_evcap = std::max<unsigned>(c->max_evcap, (unsigned)evcap);
    }
    const uint64_t NCH = c->h_chunk_off[n], NEV = c->h_ev_off[n], NAL = c->h_aln_off[n];
    int rc;
    {   // dn_collect's packed arrays: a call is a thymidine of referenceSeqMappedTo (detect.cpp:690), so their count is the bound
        size_t nt = 0;
        for (uint64_t i = 0; i < NR; i++) nt += d->refseq[i] == 'T';
        c->n_ref_T = (nt + 3) & ~(size_t)3;
    }
    if (n > c->p_cap) {                                   // page-locked mirrors for the stream-ordered host functions
        if (c->p_res) { hipHostFree(c->p_res); hipHostFree(c->p_bandc); hipHostFree(c->p_vit); hipHostFree(c->p_call_off); hipHostFree(c->p_summary); }
        c->p_res = nullptr; c->p_bandc = nullptr; c->p_vit = nullptr; c->p_call_off = nullptr; c->p_summary = nullptr; c->p_cap = 0;
        const size_t cap = std::max<size_t>((size_t)n + n / 2 + 16, 4096);     // hipHostFree waits for the device too: sized once for any batch up to 4 096 reads
        HIPCHK(c, hipHostMalloc((void **)&c->p_res, cap * sizeof(ReadRes), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&c->p_bandc, cap * sizeof(BandConstsH), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&c->p_vit, cap * sizeof(VitReadH), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&c->p_call_off, (cap + 1) * sizeof(unsigned long long), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&c->p_summary, cap * sizeof(dn_read_summary), hipHostMallocDefault));
        c->p_cap = cap;
    }
    auto place = [&]() -> int {
#define UP(field, src, cnt) if ((rc = upload(c, &B.field, src, (size_t)(cnt)))) return rc
    UP(adc, d->adc, S);                  UP(samp_off, c->h_samp_off.data(), n + 1);
    UP(cal_off, d->cal_offset, n);       UP(cal_scale, d->cal_scale, n);
    UP(basecall, d->basecall, NB);       UP(base_off, c->h_base_off.data(), n + 1);
    UP(refseq, d->refseq, NR);           UP(ref_off, c->h_ref_off.data(), n + 1);
    UP(ref2query, d->ref2query, NR);     UP(query2ref, d->query2ref, NB + n);
    UP(ref2del, d->ref2del, NR);
    UP(ref_start, d->ref_start, n);      UP(ref_end, d->ref_end, n);
    UP(is_rev, d->is_reverse, n);
    UP(chunk_off, c->h_chunk_off.data(), n + 1);
    UP(ev_off, c->h_ev_off.data(), n + 1);
    UP(aln_off, c->h_aln_off.data(), n + 1);
    UP(trace_off, c->h_trace_off.data(), n + 1);
#undef UP
#define AL(field, cnt) if ((rc = dalloc(c, &B.field, (size_t)(cnt)))) return rc
    AL(carry, 4 * NCH + n);
    if (c->keep_k1) { AL(psum, S); AL(t1, S); AL(t2, S); AL(et_start, NEV); AL(et_mean, NEV); }      // parity taps only: every prefix sum and both t-statistics in HBM
    AL(chunk_npk, NCH); AL(chunk_peaks, NCH * DN_SEG_PEAKCAP); AL(chunk_psum, NCH * DN_SEG_PEAKCAP); AL(chunk_in, NCH); AL(chunk_out, NCH); AL(chunk_sums, NCH);
    AL(ev_mean, NEV); AL(ev_start, NEV); AL(ev_len, NEV); AL(ev_x, NEV);
    AL(rank_q, NB); AL(rank_r, NR); AL(mu_q, NB);
    AL(aln_event, NAL); AL(aln_kmer, NAL); AL(cl_sig, NAL); AL(cl_rank, NAL);
    AL(res, n);
#undef AL
    if ((rc = dalloc(c, &c->d_path_from, (size_t)NAL))) return rc;
    if ((rc = dalloc(c, &c->d_path_lp, (size_t)NAL))) return rc;
    if ((rc = dalloc(c, &B.trace, (size_t)c->h_trace_off[n] * DN_TROW))) return rc;
    { BandConstsH *bcp = nullptr; if ((rc = dalloc(c, &bcp, (size_t)n))) return rc; c->d_bandc = bcp; }
    if ((rc = dalloc(c, &c->d_call_cnt, (size_t)n)) || (rc = dalloc(c, &c->d_call_off, (size_t)n + 1))) return rc;
    if ((rc = dalloc(c, &c->d_col, std::max<size_t>(c->n_ref_T, 4) * (5 * 4 + DN_KMER)))) return rc;
    if ((rc = dalloc(c, &c->ea.coord, (size_t)NR)) || (rc = dalloc(c, &c->ea.qidx, (size_t)NR)) || (rc = dalloc(c, &c->ea.ridx, (size_t)NR)) ||
        (rc = dalloc(c, &c->ea.indel, (size_t)NR)) || (rc = dalloc(c, &c->ea.nsig, (size_t)NR)) || (rc = dalloc(c, &c->ea.sig, (size_t)NR * DN_RAWDEPTH)) ||
        (rc = dalloc(c, &c->ea.core, (size_t)NR)) || (rc = dalloc(c, &c->ea.resid, (size_t)NR)) || (rc = dalloc(c, &c->ea.win_ref, (size_t)NR)) ||
        (rc = dalloc(c, &c->ea.win_len, (size_t)NR)) || (rc = dalloc(c, &c->ea.win_T, (size_t)NR)) || (rc = dalloc(c, &c->ea.win_score, (size_t)NR)) ||
        (rc = dalloc(c, &c->d_vitread, (size_t)n)) || (rc = dalloc(c, &c->d_probs, (size_t)NR * 3)) || (rc = dalloc(c, &c->ea.redo, (size_t)n)) ||
        (rc = dalloc(c, &c->ea.resume, (size_t)n * 8))) return rc;
        return DN_OK;
    };
    const double up_t3 = up_trace ? up_now() : 0.0;
    c->measuring = true; c->measured = 0;
    rc = place();
    c->measuring = false;
    if (rc) return rc;
    c->last_need = c->measured;
    if (c->measure_only) return DN_OK;                     // dn_batch_workspace_bytes: nothing allocated, nothing copied; the context holds no batch
    if ((rc = fit_slab(c, c->measured))) return rc;
    dfree_all(c);
    if ((rc = place())) return rc;
    c->max_ref = 0;
    for (uint32_t r = 0; r < n; r++) c->max_ref = std::max<unsigned>(c->max_ref, (unsigned)(c->h_ref_off[r + 1] - c->h_ref_off[r]));
    HIPCHK(c, hipMemsetAsync(B.res, 0, n * sizeof(ReadRes), c->stream));
    // Page-locked input arrays (dn_host_alloc / dn_host_register) make the whole upload asynchronous; with pageable memory the
    // runtime may still be reading the caller's arrays when hipMemcpyAsync returns, so the call waits for the copies.
    {
        // EVERY array of the caller must be page-locked for the call to return before the copies are done (round-2 advisor: deciding
        // from adc alone let the runtime read freed pageable arrays); the offset tables were copied into the context above
        const void *arrs[] = { d->adc, d->cal_offset, d->cal_scale, d->basecall, d->refseq, d->ref2query, d->query2ref, d->ref2del, d->ref_start,
                               d->ref_end, d->is_reverse };
        bool pinned = true;
        for (const void *a : arrs) {
            hipPointerAttribute_t at;
            if (!(hipPointerGetAttributes(&at, a) == hipSuccess && at.type == hipMemoryTypeHost)) { pinned = false; break; }
        }
        (void)hipGetLastError();
        c->upload_pinned = pinned;
        const double up_t4 = up_trace ? up_now() : 0.0;
        if (!c->upload_pinned) HIPCHK(c, hipStreamSynchronize(c->stream));
        if (up_trace) fprintf(stderr, "dn_batch_upload: first sync %.1f ms, profile events %.1f ms, host tables %.1f ms, copies issued %.1f ms, final sync %.1f ms; CPU time of this thread %.1f ms\n",
                              (up_t1 - up_t0) * 1e3, (up_t2 - up_t1) * 1e3, (up_t3 - up_t2) * 1e3, (up_t4 - up_t3) * 1e3, (up_now() - up_t4) * 1e3, (up_cpu() - up_c0) * 1e3);
    }
    c->h_res.assign(n, ReadRes{});
    c->n_batch = n; c->async_err = 0;
    c->have_batch = true; c->stage = 1;
    return DN_OK;
}

int dn_batch_workspace_bytes(dn_ctx *c, const dn_batch_desc *d, uint64_t *bytes) {
    if (!c || !d || !bytes) return DN_ERR_ARG;
    c->measure_only = true;
    const int rc = dn_batch_upload(c, d);
    c->measure_only = false;
    c->have_batch = false; c->stage = 0;
    *bytes = rc ? 0 : (uint64_t)c->last_need;
    return rc;
}

int dn_ctx_set_event_bound(dn_ctx *c, uint32_t samples_per_event) {
    if (!c || samples_per_event < 2 || samples_per_event > 16) return DN_ERR_ARG;
    c->ev_div = samples_per_event;
    return DN_OK;
}
uint32_t dn_ctx_get_event_bound(const dn_ctx *c) { return c ? c->ev_div : 0; }

int dn_ctx_reserve(dn_ctx *c, uint64_t workspace_bytes, uint64_t collect_bytes) {
    if (!c) return DN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_batch = false; c->stage = 0;
    dfree_all(c);
    if (workspace_bytes) { const int rc = fit_slab(c, (size_t)workspace_bytes); if (rc) return rc; }
    c->col_reserve = std::max(c->col_reserve, (size_t)collect_bytes);
    return DN_OK;
}

static int need(dn_ctx *c, int stage, const char *what) {
    if (!c) return DN_ERR_ARG;
    if (!c->have_batch || c->stage < stage) return fail(c, DN_ERR_STATE, "%s called before its input stage ran", what);
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, DN_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    return DN_OK;
}

int dn_run_segment(dn_ctx *c) {
    int rc = need(c, 1, "dn_run_segment"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 2); return DN_OK; }
    { Timed t(c, DN_K_SCAN);   k1_launch_scan(c->B, c->stream); }
    if (c->B.psum) { Timed t(c, DN_K_TSTAT); k1_launch_tstat(c->B, c->max_samples, c->stream); }      // taps only
    { Timed t(c, DN_K_DETECT); k1_launch_detect(c->B, c->max_chunks, c->stream); }
    { Timed t(c, DN_K_EVENTS); k1_launch_events(c->B, c->stream); }
    { Timed t(c, DN_K_RANKS);  ks_launch_ranks(c->B, c->max_len, c->stream); }
    HIPCHK(c, hipGetLastError());
    c->stage = 2;
    return DN_OK;
}

int dn_run_rough_scaling(dn_ctx *c) {
    int rc = need(c, 2, "dn_run_rough_scaling"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 3); return DN_OK; }
    { Timed t(c, DN_K_QUANTILE); ks_launch_quantile(c->B, c->stream); }
    HIPCHK(c, hipGetLastError());
    c->stage = 3;
    return DN_OK;
}

// Host functions (hipLaunchHostFunc): they run on the runtime's callback thread between two stream operations, touch only the
// context's page-locked mirrors and call no HIP API.  They exist because three per-read constants of the reference come out of
// libm (log / exp of values that depend on the event count found on the device): computing them with the HOST's libm keeps them
// bit-identical to the reference's, and doing it stream-ordered keeps the host thread out of the pipeline.
static void hf_band_consts(void *p) {
    dn_ctx *c = (dn_ctx *)p;
    for (uint32_t r = 0; r < c->n_batch; r++) {
        const ReadRes &R = c->p_res[r];
        if (R.seg_overflow && !c->async_err) { c->async_err = DN_ERR_OVERFLOW; c->async_err_read = r; }
        const uint64_t E = R.n_events, K = R.n_kq;
        // event_handling.cpp:174-182, with the host libm (the reference's own calls)
        const double epk = (double)E / (double)K;
        const double p_stay = 1 - (1 / (epk + 1));
        const double lp_skip = log(1e-30);
        const double lp_stay = log(p_stay);
        c->p_bandc[r].lp_stay = lp_stay;
        c->p_bandc[r].lp_step = log(1.0 - exp(lp_skip) - exp(lp_stay));
    }
}

int dn_run_banded(dn_ctx *c) {
    int rc = need(c, 3, "dn_run_banded"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 4); return DN_OK; }
    const uint32_t n = (uint32_t)c->B.n_reads;
    // per-read band penalties depend on the number of events found on the device: D2H of the per-read scalars into page-locked
    // memory, host libm in a stream-ordered host function, H2D of the constants -- the calling thread never waits
    HIPCHK(c, hipMemcpyAsync(c->p_res, c->B.res, n * sizeof(ReadRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipLaunchHostFunc(c->stream, hf_band_consts, c));
    HIPCHK(c, hipMemcpyAsync(c->d_bandc, c->p_bandc, n * sizeof(BandConstsH), hipMemcpyHostToDevice, c->stream));
    { Timed t(c, DN_K_PREP);       ks_launch_prep(c->B, c->max_evcap, c->stream); }
    { Timed t(c, DN_K_BAND_FILL);  k2_launch_fill(c->B, c->d_bandc, &c->fc, c->stream); }
    { Timed t(c, DN_K_BAND_TRACE); k2_launch_chase(c->B, c->d_path_from, c->stream);
                                   k2_launch_post(c->B, c->d_path_from, c->d_path_lp, &c->fc, c->stream); }
    HIPCHK(c, hipGetLastError());
    c->stage = 4;
    return DN_OK;
}

int dn_run_theilsen(dn_ctx *c) {
    int rc = need(c, 4, "dn_run_theilsen"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 5); return DN_OK; }
    { Timed t(c, DN_K_THEILSEN); ks_launch_theilsen(c->B, c->stream); }
    HIPCHK(c, hipGetLastError());
    c->stage = 5;
    return DN_OK;
}

int dn_run_normalise(dn_ctx *c) {
    int rc;
    if ((rc = dn_run_segment(c))) return rc;
    if ((rc = dn_run_rough_scaling(c))) return rc;
    if ((rc = dn_run_banded(c))) return rc;
    return dn_run_theilsen(c);
}

static double h_eln(double x, int *neg) {            // probability.cpp:35-47
    if (x == 0.0) return NAN;
    if (x > 0.0) return log(x);
    *neg = 1; return NAN;
}
static double h_lnSum(double a, double b) {           // probability.cpp:50-76
    const bool na = std::isnan(a), nb = std::isnan(b);
    if (na || nb) { if (na && nb) return NAN; return na ? b : a; }
    int neg = 0;
    if (a > b) return a + h_eln(1.0 + (std::isnan(b - a) ? 0.0 : exp(b - a)), &neg);
    return b + h_eln(1.0 + (std::isnan(a - b) ? 0.0 : exp(a - b)), &neg);
}

static void hf_viterbi_consts(void *p) {
    dn_ctx *c = (dn_ctx *)p;
    for (uint32_t r = 0; r < c->n_batch; r++) {
        const ReadRes &R = c->p_res[r];
        VitReadH &v = c->p_vit[r];
        int neg = 0;
        const double iM2M = h_eln(1. - (1. / R.events_per_base), &neg);                 // alignment.cpp:207
        const double eM2M = h_eln(1.0 - c->vc.M2D - c->vc.M2I - iM2M, &neg);            // :208 (sic: log values)
        v.iM2M = iM2M; v.eM2M = eM2M;
        v.eM2MorD = h_lnSum(eM2M, c->vc.M2D);                                            // :209
        v.eOrI = h_lnSum(eM2M, iM2M);                                                    // :210
        v.fail = (R.status == 0 && neg) ? 1 : 0;                                         // the reference throws NegativeLog
        v.pad = 0;
    }
}

int dn_run_eventalign(dn_ctx *c) {
    int rc = need(c, 5, "dn_run_eventalign"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 6); return DN_OK; }
    const uint32_t n = (uint32_t)c->B.n_reads;
    // per-read transitions depend on eventsPerBase (alignment.cpp:207-210): small D2H, host libm in a host function, small H2D
    HIPCHK(c, hipMemcpyAsync(c->p_res, c->B.res, n * sizeof(ReadRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipLaunchHostFunc(c->stream, hf_viterbi_consts, c));
    HIPCHK(c, hipMemcpyAsync(c->d_vitread, c->p_vit, n * sizeof(VitReadH), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->ea.sig, 0, (size_t)c->h_ref_off[n] * DN_RAWDEPTH * sizeof(float), c->stream));
    c->ea.al_coord = nullptr; c->ea.al_rpos = nullptr; c->ea.al_val = nullptr; c->ea.al_kind = nullptr; c->ea.al_off = nullptr; c->ea.al_n = nullptr;
    c->have_align = false;
    if (c->want_align) {
        // rows per read are bounded by sum over its rough-alignment pairs of the event length: size the table exactly for that
        if ((rc = dgrow(c, c->al_off, (n + 1) * sizeof(unsigned long long))) || (rc = dgrow(c, c->al_n, n * sizeof(unsigned)))) return rc;
        k2b_rowcap_launch(c->B, (unsigned long long *)c->al_off.p, c->stream);
        c->h_al_off.assign(n + 1, 0ull);
        HIPCHK(c, hipMemcpyAsync(c->h_al_off.data() + 1, c->al_off.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (uint32_t r = 0; r < n; r++) c->h_al_off[r + 1] += c->h_al_off[r];
        const size_t rows = std::max<size_t>((size_t)c->h_al_off[n], 1);
        if ((rc = dgrow(c, c->al_coord, rows * 4)) || (rc = dgrow(c, c->al_rpos, rows * 4)) || (rc = dgrow(c, c->al_val, rows * 8)) ||
            (rc = dgrow(c, c->al_kind, rows))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->al_off.p, c->h_al_off.data(), (n + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->al_n.p, 0, n * sizeof(unsigned), c->stream));
        c->ea.al_coord = (unsigned *)c->al_coord.p; c->ea.al_rpos = (unsigned *)c->al_rpos.p; c->ea.al_val = (double *)c->al_val.p;
        c->ea.al_kind = (unsigned char *)c->al_kind.p; c->ea.al_off = (const unsigned long long *)c->al_off.p; c->ea.al_n = (unsigned *)c->al_n.p;
        c->have_align = true;
    }
    if (!c->d_k2b_huge) {                                 // once per context: the global-memory lattices of the reads a > 512-observation window stops (6 MB)
        HIPCHK(c, hipMalloc(&c->d_k2b_huge, k2b_huge_scratch_bytes()));
        c->dev_bytes += k2b_huge_scratch_bytes();
    }
    { Timed t(c, DN_K_VITERBI); HIPCHK(c, k2b_launch(c->B, &c->ea, c->d_vitread, &c->vc, c->max_ref, c->d_k2b_huge, c->stream)); }
    HIPCHK(c, hipGetLastError());
    c->stage = 6;
    return DN_OK;
}

int dn_debug_emission(dn_ctx *c, uint32_t n, const double *x, const double *mu, double *out) {
    if (!c || !x || !mu || !out) return DN_ERR_ARG;
    if (!c->have_model) return fail(c, DN_ERR_STATE, "dn_load_pore_model must be called first");
    if (n == 0) return DN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    double *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, (size_t)n * 3 * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, mu, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { k2b_emission_tap_launch(d, d + n, d + 2 * (size_t)n, n, &c->vc, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + 2 * (size_t)n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    return e == hipSuccess ? DN_OK : fail(c, DN_ERR_HIP, "dn_debug_emission: %s", hipGetErrorString(e));
}

int dn_set_align_table(dn_ctx *c, int on) {
    if (!c) return DN_ERR_ARG;
    c->want_align = on != 0;
    return DN_OK;
}

int dn_get_align_rows(dn_ctx *c, uint32_t *n_rows) {
    int rc = need(c, 6, "dn_get_align_rows"); if (rc) return rc;
    if (!n_rows) return DN_ERR_ARG;
    if (c->B.n_reads == 0) return DN_OK;
    if (!c->have_align) return fail(c, DN_ERR_STATE, "dn_set_align_table(ctx, 1) must precede dn_run_eventalign");
    const uint32_t n = (uint32_t)c->B.n_reads;
    c->h_al_n.resize(n);
    HIPCHK(c, hipMemcpyAsync(c->h_al_n.data(), c->al_n.p, n * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(n_rows, c->h_al_n.data(), n * sizeof(unsigned));
    return DN_OK;
}

int dn_get_align_table(dn_ctx *c, uint32_t read, uint32_t cap, uint32_t *coord, uint32_t *ref_pos, double *value, uint8_t *kind) {
    int rc = need(c, 6, "dn_get_align_table"); if (rc) return rc;
    if (read >= (uint32_t)c->B.n_reads) return DN_ERR_ARG;
    if (!c->have_align) return fail(c, DN_ERR_STATE, "dn_set_align_table(ctx, 1) must precede dn_run_eventalign");
    const unsigned long long a0 = c->h_al_off[read];
    unsigned cnt = 0;                                                      // the library's own count (what dn_get_align_rows reports), like every other tap
    HIPCHK(c, hipMemcpyAsync(&cnt, (unsigned *)c->al_n.p + read, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const unsigned long long n_rows = cnt;
    if (a0 + n_rows > c->h_al_off[read + 1]) return fail(c, DN_ERR_STATE, "dn_get_align_table: read %u reports %llu rows, its slice holds %llu", read, n_rows, c->h_al_off[read + 1] - a0);
    if (n_rows > cap) return fail(c, DN_ERR_ARG, "dn_get_align_table: read %u has %llu rows, the caller's arrays hold %u", read, n_rows, cap);
    if (n_rows == 0) return DN_OK;
    if (coord) HIPCHK(c, hipMemcpyAsync(coord, (unsigned *)c->al_coord.p + a0, n_rows * 4ull, hipMemcpyDeviceToHost, c->stream));
    if (ref_pos) HIPCHK(c, hipMemcpyAsync(ref_pos, (unsigned *)c->al_rpos.p + a0, n_rows * 4ull, hipMemcpyDeviceToHost, c->stream));
    if (value) HIPCHK(c, hipMemcpyAsync(value, (double *)c->al_val.p + a0, n_rows * 8ull, hipMemcpyDeviceToHost, c->stream));
    if (kind) HIPCHK(c, hipMemcpyAsync(kind, (unsigned char *)c->al_kind.p + a0, n_rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DN_OK;
}

static int fetch_res(dn_ctx *c) {
    const uint32_t n = (uint32_t)c->B.n_reads;
    if (n == 0) return DN_OK;
    HIPCHK(c, hipMemcpyAsync(c->h_res.data(), c->B.res, n * sizeof(ReadRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DN_OK;
}

static int cnn_settle(dn_ctx *c);
static void fill_summary(const dn_ctx *c, int r, const ReadRes &R, dn_read_summary &s) {
    memset(&s, 0, sizeof(s));
    s.status = R.status;
    s.n_samples = (uint32_t)(c->h_samp_off[r + 1] - c->h_samp_off[r]);
    s.n_scrappie = R.n_scrappie; s.n_events = R.n_events;
    s.n_kmers_query = R.n_kq; s.n_kmers_ref = R.n_kr;
    s.n_bands = R.n_bands; s.band_cells = (uint64_t)R.n_bands * DN_BANDWIDTH;
    s.rough_shift = R.q_shift; s.rough_scale = R.q_scale;
    s.end_event = R.end_event; s.n_aligned = R.n_aligned;
    s.avg_log_emission = R.avg_log_emission; s.spanned = R.spanned; s.max_gap = R.max_gap; s.n_cleaned = R.n_cleaned;
    s.ts_slope = R.ts_slope; s.ts_intercept = R.ts_intercept;
    s.shift = R.shift; s.scale = R.scale; s.events_per_base = R.events_per_base;
    s.n_positions = R.n_positions; s.n_windows = R.n_windows; s.detector_rechecks = R.rechecks;
    s.n_hmm_calls = c->hmm_done && r < (int)c->h_nhmm.size() ? c->h_nhmm[r] : 0;
}

// errors raised by stream-ordered host functions surface at the next call that synchronises
static int async_status(dn_ctx *c) {
    if (c->async_err == DN_ERR_OVERFLOW) return fail(c, DN_ERR_OVERFLOW, "segmentation workspace overflow in read %u", c->async_err_read);
    return c->async_err ? fail(c, c->async_err, "asynchronous stage failed") : DN_OK;
}

int dn_get_summaries(dn_ctx *c, dn_read_summary *out) {
    int rc = need(c, 1, "dn_get_summaries"); if (rc) return rc;
    if (c->B.n_reads == 0) return DN_OK;
    if (!out) return DN_ERR_ARG;
    if ((rc = fetch_res(c))) return rc;
    if ((rc = async_status(c))) return rc;
    for (int r = 0; r < c->B.n_reads; r++) fill_summary(c, r, c->h_res[r], out[r]);
    return DN_OK;
}

int dn_collect(dn_ctx *c, dn_result_batch *out) {
    int rc = need(c, 7, "dn_collect"); if (rc) return rc;
    if (!out) return DN_ERR_ARG;
    memset(out, 0, sizeof(*out));
    const uint32_t n = (uint32_t)c->B.n_reads;
    if (n == 0) return DN_OK;
    if ((rc = cnn_settle(c))) return rc;                  // the stream is idle from here on
    if ((rc = async_status(c))) return rc;
    CollectDev cd{};
    cd.coord = c->ea.coord; cd.qidx = c->ea.qidx; cd.ridx = c->ea.ridx; cd.probs = c->d_probs;
    cd.cnt = c->d_call_cnt; cd.off = c->d_call_off;
    // packed device arrays: every call is a 'T' of referenceSeqMappedTo, so the batch's T count bounds them (slab space set aside
    // at upload); they are compacted on the device before the exact number is known on the host
    const size_t cap = c->n_ref_T;
    char *blk = (char *)c->d_col;
    cd.o_coord = (unsigned *)blk; cd.o_qidx = cd.o_coord + cap; cd.o_ridx = cd.o_qidx + cap;
    cd.o_edu = (float *)(cd.o_ridx + cap); cd.o_brdu = cd.o_edu + cap; cd.o_kmer = (char *)(cd.o_brdu + cap);
    kc_launch_count(c->B, &cd, c->max_ref, c->stream);
    kc_launch_scan(c->B, &cd, c->stream);
    kc_launch_pack(c->B, &cd, c->max_ref, c->stream);
    HIPCHK(c, hipMemcpyAsync(c->p_res, c->B.res, n * sizeof(ReadRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->p_call_off, c->d_call_off, (n + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const size_t total = (size_t)c->p_call_off[n];
    if (total > cap) return fail(c, DN_ERR_OVERFLOW, "dn_collect: %zu calls exceed the reference's %zu thymidines", total, cap);
    // page-locked result block, grow-only: [coord][query_idx][ref_idx][p_edu][p_brdu][kmer9], each `total` long
    const size_t tot4 = (total + 3) & ~(size_t)3;
    const size_t bytes = tot4 * (5 * 4 + DN_KMER) + 64;
    if (bytes > c->col_host_cap) {
        if (c->col_host) hipHostFree(c->col_host);
        c->col_host = nullptr; c->col_host_cap = 0;
        const size_t want = std::max(bytes + bytes / 2, c->col_reserve);
        HIPCHK(c, hipHostMalloc(&c->col_host, want, hipHostMallocDefault));
        c->col_host_cap = want;
    }
    uint32_t *h_coord = (uint32_t *)c->col_host, *h_qidx = h_coord + tot4, *h_ridx = h_qidx + tot4;
    float *h_edu = (float *)(h_ridx + tot4), *h_brdu = h_edu + tot4; char *h_kmer = (char *)(h_brdu + tot4);
    if (total) {
        HIPCHK(c, hipMemcpyAsync(h_coord, cd.o_coord, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_qidx, cd.o_qidx, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_ridx, cd.o_ridx, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_edu, cd.o_edu, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_brdu, cd.o_brdu, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_kmer, cd.o_kmer, total * DN_KMER, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    memcpy(c->h_res.data(), c->p_res, n * sizeof(ReadRes));
    for (uint32_t r = 0; r < n; r++) fill_summary(c, (int)r, c->p_res[r], c->p_summary[r]);
    out->n_reads = n; out->summary = c->p_summary;
    out->call_off = (const uint64_t *)c->p_call_off; out->n_calls = total;
    out->ref_coord = h_coord; out->query_idx = h_qidx; out->ref_idx = h_ridx; out->p_edu = h_edu; out->p_brdu = h_brdu; out->kmer9 = h_kmer;
    return DN_OK;
}

int dn_host_alloc(size_t bytes, void **p) {
    if (!p) return DN_ERR_ARG;
    *p = nullptr;
    return hipHostMalloc(p, std::max<size_t>(bytes, 1), hipHostMallocDefault) == hipSuccess ? DN_OK : DN_ERR_HIP;
}
void dn_host_free(void *p) { if (p) (void)hipHostFree(p); }
int dn_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return DN_ERR_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? DN_OK : DN_ERR_HIP;
}
int dn_host_unregister(void *p) { return (p && hipHostUnregister(p) == hipSuccess) ? DN_OK : DN_ERR_HIP; }

#define CHECK_READ(stage_, name_)                                     \
    int rc = need(c, stage_, name_); if (rc) return rc;              \
    if (read >= (uint32_t)c->B.n_reads) return DN_ERR_ARG;
// every tap states how many records the caller's arrays hold; the true count is the library's
#define CHECK_CAP(name_, need_, cap_)                                 \
    if ((uint64_t)(need_) > (uint64_t)(cap_))                         \
        return fail(c, DN_ERR_ARG, "%s: read %u has %llu records, the caller's arrays hold %llu", name_, read, (unsigned long long)(need_), (unsigned long long)(cap_));

int dn_debug_keep_k1(dn_ctx *c, int on) {
    if (!c) return DN_ERR_ARG;
    c->keep_k1 = on != 0;
    return DN_OK;
}

int dn_debug_seg_warm(dn_ctx *c, uint32_t samples) {
    if (!c || samples > DN_SEG_WARM) return DN_ERR_ARG;
    c->seg_warm = (int)samples;
    return DN_OK;
}

int dn_get_prefix_sums(dn_ctx *c, uint32_t read, uint64_t cap, double *sum, double *sumsq) {
    CHECK_READ(2, "dn_get_prefix_sums");
    CHECK_CAP("dn_get_prefix_sums", c->h_samp_off[read + 1] - c->h_samp_off[read], cap);
    if (!c->B.psum) return fail(c, DN_ERR_STATE, "dn_debug_keep_k1(ctx, 1) must precede dn_batch_upload: prefix sums do not leave the kernels otherwise");
    const uint64_t s0 = c->h_samp_off[read]; const size_t n = (size_t)(c->h_samp_off[read + 1] - s0);
    std::vector<double2> tmp(n);
    if ((rc = d2h(c, tmp.data(), c->B.psum + s0, n))) return rc;
    if (sum) { sum[0] = 0.; for (size_t i = 0; i < n; i++) sum[i + 1] = tmp[i].x; }
    if (sumsq) { sumsq[0] = 0.; for (size_t i = 0; i < n; i++) sumsq[i + 1] = tmp[i].y; }
    return DN_OK;
}

int dn_get_tstats(dn_ctx *c, uint32_t read, uint64_t cap, float *a, float *b) {
    CHECK_READ(2, "dn_get_tstats");
    CHECK_CAP("dn_get_tstats", c->h_samp_off[read + 1] - c->h_samp_off[read], cap);
    if (!c->B.t1) return fail(c, DN_ERR_STATE, "dn_debug_keep_k1(ctx, 1) must precede dn_batch_upload: t-statistics do not leave the kernels otherwise");
    const uint64_t s0 = c->h_samp_off[read]; const size_t n = (size_t)(c->h_samp_off[read + 1] - s0);
    if ((rc = d2h(c, a, c->B.t1 + s0, n))) return rc;
    return d2h(c, b, c->B.t2 + s0, n);
}

int dn_get_scrappie_events(dn_ctx *c, uint32_t read, uint64_t cap, uint32_t *start, float *length, float *mean) {
    CHECK_READ(2, "dn_get_scrappie_events");
    if (!c->B.et_start) return fail(c, DN_ERR_STATE, "dn_debug_keep_k1(ctx, 1) must precede dn_batch_upload: the scrappie event table does not leave the kernels otherwise");
    if ((rc = fetch_res(c))) return rc;
    const size_t n = c->h_res[read].n_scrappie; const uint64_t e0 = c->h_ev_off[read];
    CHECK_CAP("dn_get_scrappie_events", n, cap);
    std::vector<uint32_t> st(n);
    if ((rc = d2h(c, st.data(), c->B.et_start + e0, n))) return rc;
    if (start) memcpy(start, st.data(), n * sizeof(uint32_t));
    if (length) {
        const uint32_t ns = (uint32_t)(c->h_samp_off[read + 1] - c->h_samp_off[read]);
        for (size_t i = 0; i < n; i++) {
            const uint64_t en = (i + 1 < n) ? st[i + 1] : ns;
            length[i] = (float)(uint64_t)(en - (uint64_t)st[i]);
        }
    }
    return d2h(c, mean, c->B.et_mean + e0, n);
}

int dn_get_events(dn_ctx *c, uint32_t read, uint64_t cap, double *mean, uint32_t *raw_start, uint32_t *raw_len) {
    CHECK_READ(2, "dn_get_events");
    if ((rc = fetch_res(c))) return rc;
    const size_t n = c->h_res[read].n_events; const uint64_t e0 = c->h_ev_off[read];
    CHECK_CAP("dn_get_events", n, cap);
    if ((rc = d2h(c, mean, c->B.ev_mean + e0, n))) return rc;
    if ((rc = d2h(c, raw_start, c->B.ev_start + e0, n))) return rc;
    return d2h(c, raw_len, c->B.ev_len + e0, n);
}

int dn_get_kmer_ranks(dn_ctx *c, uint32_t read, uint64_t cap_q, uint64_t cap_r, uint32_t *rq, uint32_t *rr) {
    CHECK_READ(2, "dn_get_kmer_ranks");
    const uint64_t b0 = c->h_base_off[read], f0 = c->h_ref_off[read];
    const size_t nq = (size_t)(c->h_base_off[read + 1] - b0) - DN_K + 1, nr = (size_t)(c->h_ref_off[read + 1] - f0) - DN_K + 1;
    if (rq) { CHECK_CAP("dn_get_kmer_ranks (query)", nq, cap_q); }
    if (rr) { CHECK_CAP("dn_get_kmer_ranks (reference)", nr, cap_r); }
    if ((rc = d2h(c, rq, c->B.rank_q + b0, nq))) return rc;
    return d2h(c, rr, c->B.rank_r + f0, nr);
}

int dn_get_alignment(dn_ctx *c, uint32_t read, uint64_t cap, uint32_t *ev, uint32_t *km) {
    CHECK_READ(4, "dn_get_alignment");
    if ((rc = fetch_res(c))) return rc;
    const ReadRes &R = c->h_res[read];
    CHECK_CAP("dn_get_alignment", R.n_aligned, cap);
    const uint64_t a0 = c->h_aln_off[read] + R.aln_begin;
    if ((rc = d2h(c, ev, c->B.aln_event + a0, R.n_aligned))) return rc;
    return d2h(c, km, c->B.aln_kmer + a0, R.n_aligned);
}

int dn_get_cleaned(dn_ctx *c, uint32_t read, uint64_t cap, double *sig, uint32_t *rank) {
    CHECK_READ(4, "dn_get_cleaned");
    if ((rc = fetch_res(c))) return rc;
    const ReadRes &R = c->h_res[read];
    CHECK_CAP("dn_get_cleaned", R.n_cleaned, cap);
    const uint64_t a0 = c->h_aln_off[read];
    if ((rc = d2h(c, sig, c->B.cl_sig + a0, R.n_cleaned))) return rc;
    return d2h(c, rank, c->B.cl_rank + a0, R.n_cleaned);
}

int dn_get_trace(dn_ctx *c, uint32_t read, uint64_t cap, uint8_t *trace, int32_t *band_event, int32_t *band_kmer) {
    CHECK_READ(4, "dn_get_trace");
    if ((rc = fetch_res(c))) return rc;
    const size_t nb = c->h_res[read].n_bands;
    CHECK_CAP("dn_get_trace", nb, cap);
    std::vector<uint64_t> rows(nb * (DN_TROW / 8));
    if ((rc = d2h(c, (uint8_t *)rows.data(), c->B.trace + c->h_trace_off[read] * DN_TROW, nb * DN_TROW))) return rc;
    // rows are planar 2-bit codes by slot (k2_banded.hip put_row): {A0, A1, B0, B1}; slot s = event & 127, register s & 1, lane s >> 1
    auto code = [&](size_t b, int ev) -> unsigned {
        const unsigned sl = (unsigned)ev & 127u, l = sl >> 1;
        const uint64_t *w = rows.data() + b * 4 + (sl & 1u) * 2;
        return (unsigned)((w[0] >> l) & 1ull) | ((unsigned)((w[1] >> l) & 1ull) << 1);
    };
    int32_t ev_prev = 48;
    for (size_t b = 0; b < nb; b++) {
        // code 3 marks the 28 slots outside the band; the corner moves by exactly one per band: the band moved down iff the slot of
        // event ev_prev + 1 is in the band
        const int32_t ev = (b == 0) ? 49 : ((code(b, ev_prev + 1) != 3u) ? ev_prev + 1 : ev_prev);
        if (trace) for (int o = 0; o < DN_BANDWIDTH; o++) trace[b * DN_BANDWIDTH + o] = (uint8_t)code(b, ev - o);
        ev_prev = ev;
        if (band_event) band_event[b] = ev;
        if (band_kmer) band_kmer[b] = (int32_t)b - 2 - ev;
    }
    return DN_OK;
}

int dn_get_positions(dn_ctx *c, uint32_t read, uint64_t cap, uint32_t *coord, uint32_t *query_idx, uint32_t *ref_idx, int32_t *indel_score, char *kmer9,
                     uint32_t *n_signal, float *signal20, float *core, float *residual) {
    CHECK_READ(6, "dn_get_positions");
    if ((rc = fetch_res(c))) return rc;
    const size_t np = c->h_res[read].n_positions; const uint64_t f0 = c->h_ref_off[read];
    CHECK_CAP("dn_get_positions", np, cap);
    if ((rc = d2h(c, coord, c->ea.coord + f0, np)) || (rc = d2h(c, query_idx, c->ea.qidx + f0, np)) ||
        (rc = d2h(c, indel_score, c->ea.indel + f0, np)) || (rc = d2h(c, n_signal, c->ea.nsig + f0, np)) ||
        (rc = d2h(c, signal20, c->ea.sig + f0 * DN_RAWDEPTH, np * DN_RAWDEPTH)) || (rc = d2h(c, core, c->ea.core + f0, np)) ||
        (rc = d2h(c, residual, c->ea.resid + f0, np))) return rc;
    std::vector<uint32_t> ri(np);
    if ((rc = d2h(c, ri.data(), c->ea.ridx + f0, np))) return rc;
    if (ref_idx) memcpy(ref_idx, ri.data(), np * sizeof(uint32_t));
    if (kmer9) {   // the 9-mer of a position is the reference slice around its index (alignment.cpp:683)
        const size_t nr = (size_t)(c->h_ref_off[read + 1] - f0);
        std::vector<char> ref(nr);
        if ((rc = d2h(c, ref.data(), c->B.refseq + f0, nr))) return rc;
        for (size_t i = 0; i < np; i++) memcpy(kmer9 + 9 * i, ref.data() + ri[i] - DN_KMER / 2, 9);
    }
    return DN_OK;
}

int dn_get_windows(dn_ctx *c, uint32_t read, uint64_t cap, uint32_t *ref_index, uint32_t *window_len, uint32_t *n_obs, double *score) {
    CHECK_READ(6, "dn_get_windows");
    if ((rc = fetch_res(c))) return rc;
    const size_t nw = c->h_res[read].n_windows; const uint64_t f0 = c->h_ref_off[read];
    CHECK_CAP("dn_get_windows", nw, cap);
    if ((rc = d2h(c, ref_index, c->ea.win_ref + f0, nw)) || (rc = d2h(c, window_len, c->ea.win_len + f0, nw)) ||
        (rc = d2h(c, n_obs, c->ea.win_T + f0, nw))) return rc;
    return d2h(c, score, c->ea.win_score + f0, nw);
}

int dn_load_cnn(dn_ctx *c, const dn_cnn_op *ops, uint32_t n_ops, const float *weights, uint64_t n_weights, uint32_t n_buffers) {
    if (!c || !ops || !weights || n_ops == 0 || n_ops > 1024 || n_buffers == 0 || n_buffers > 8) return DN_ERR_ARG;     // 1024: the range report block has two words per op
    HIPCHK(c, hipSetDevice(c->device));
    // the encoder is what marks a pass's live rows (validity bytes): every later epilogue masks by them
    if (ops[0].op != DN_CNN_ENCODE_GRU) return fail(c, DN_ERR_ARG, "cnn op 0 must be ENCODE_GRU (it writes the row validity mask every later op reads)");
    for (uint32_t i = 0; i < n_ops; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op < DN_CNN_ENCODE_GRU || o.op > DN_CNN_CONV_ADD) return fail(c, DN_ERR_ARG, "cnn op %u: unknown type %d", i, o.op);
        if (o.cin > 256 || o.cout > 256) return fail(c, DN_ERR_ARG, "cnn op %u: more than 256 channels", i);
        const uint32_t nb = n_buffers;
        if ((uint32_t)o.src >= nb || (uint32_t)o.dst >= nb || (uint32_t)o.a >= nb || (uint32_t)o.b >= nb) return fail(c, DN_ERR_ARG, "cnn op %u: buffer index out of range", i);
        // this is the trust boundary for model files: every offset + extent must lie inside the blob
        auto inside = [&](int64_t off, uint64_t len) { return off >= 0 && (uint64_t)off <= n_weights && len <= n_weights - (uint64_t)off; };
        const bool conv = o.op == DN_CNN_CONV || o.op == DN_CNN_CONV_ADD;
        if ((conv || o.op == DN_CNN_DWCONV || o.op == DN_CNN_DENSE_SOFTMAX || o.op == DN_CNN_ADD_RELU) && (o.cin <= 0 || o.cout <= 0))
            return fail(c, DN_ERR_ARG, "cnn op %u: channel counts must be positive", i);
        if ((conv || o.op == DN_CNN_DWCONV) && o.k <= 0) return fail(c, DN_ERR_ARG, "cnn op %u: kernel width must be positive", i);
        if (conv && (o.cin % 32 || o.cout % 64 || !(o.k & 1) || o.k > 17)) return fail(c, DN_ERR_ARG, "cnn op %u: conv shape not supported", i);
        if (o.op == DN_CNN_DWCONV && (o.cin % 4 || (o.k != 3 && o.k != 5 && o.k != 7 && o.k != 9 && o.k != 17)))
            return fail(c, DN_ERR_ARG, "cnn op %u: depthwise shape not supported", i);
        // a convolution reads rows (halo) and channels that other workgroups of the same launch write: never in place
        if ((conv || o.op == DN_CNN_DWCONV) && o.src == o.dst) return fail(c, DN_ERR_ARG, "cnn op %u: convolution in place (src == dst)", i);
        if (conv && (!inside(o.w, (uint64_t)o.k * o.cin * o.cout) || !inside(o.scale, (uint64_t)o.cout) || !inside(o.shift, (uint64_t)o.cout)))
            return fail(c, DN_ERR_ARG, "cnn op %u: weights out of range", i);
        if (o.op == DN_CNN_DWCONV && !inside(o.w, (uint64_t)o.k * o.cin)) return fail(c, DN_ERR_ARG, "cnn op %u: depthwise weights out of range", i);
        if (o.op == DN_CNN_DENSE_SOFTMAX && (!inside(o.w, (uint64_t)o.cin * o.cout) || !inside(o.shift, (uint64_t)o.cout)))
            return fail(c, DN_ERR_ARG, "cnn op %u: dense weights out of range", i);
        if (o.op == DN_CNN_ENCODE_GRU) {
            if (o.cout != 64) return fail(c, DN_ERR_ARG, "cnn op %u: the encoder writes 64 channels", i);
            const uint64_t ext[6] = { 48, 16 * 48, 96, 16 * 48, 16 * 48, 96 };     // kernel / recurrent / bias of the two GRUs
            for (int j = 0; j < 6; j++) if (!inside(o.aux[j], ext[j])) return fail(c, DN_ERR_ARG, "cnn op %u: GRU weights out of range", i);
        }
    }
    // conv kernels [k][cin][cout] (Keras order) -> [k][cin / 32][cout][32]: the B tile of k3_conv is then a straight copy
    std::vector<float> wl(weights, weights + n_weights);
    for (uint32_t i = 0; i < n_ops; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op != DN_CNN_CONV && o.op != DN_CNN_CONV_ADD) continue;
        const float *src = weights + o.w; float *dst = wl.data() + o.w;
        for (int t = 0; t < o.k; t++)
            for (int ci = 0; ci < o.cin; ci++)
                for (int n = 0; n < o.cout; n++)
                    dst[(((size_t)t * (o.cin / 32) + ci / 32) * o.cout + n) * 32 + (ci % 32)] = src[((size_t)t * o.cin + ci) * o.cout + n];
    }
    // the bf16 path's weights: three exact pieces per value (round to nearest even), [step][piece][cout][32]
    auto bf16_rne = [](float f) -> uint16_t { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
    auto bf16_f32 = [](uint16_t b) -> float { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    std::vector<uint16_t> wb; std::vector<int64_t> wb_off(n_ops, 0);
    for (uint32_t i = 0; i < n_ops; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op != DN_CNN_CONV && o.op != DN_CNN_CONV_ADD) continue;
        wb_off[i] = (int64_t)wb.size();
        const size_t steps = (size_t)o.k * (o.cin / 32), blk = (size_t)o.cout * 32;
        wb.resize(wb.size() + steps * 3 * blk);
        uint16_t *dst = wb.data() + wb_off[i];
        const float *src = wl.data() + o.w;                // [tap][channel block][cout][32]; the bf16 kernel walks [channel block][tap]
        const size_t cbn = (size_t)(o.cin / 32);
        for (size_t st = 0; st < steps; st++)
            for (size_t e = 0; e < blk; e++) {
                const size_t cb = st / (size_t)o.k, tp = st % (size_t)o.k;
                const float x = src[(tp * cbn + cb) * blk + e];
                const uint16_t h = bf16_rne(x); const float r1 = x - bf16_f32(h);
                const uint16_t m = bf16_rne(r1); const float r2 = r1 - bf16_f32(m);
                dst[(st * 3 + 0) * blk + e] = h; dst[(st * 3 + 1) * blk + e] = m; dst[(st * 3 + 2) * blk + e] = bf16_rne(r2);
            }
    }
    if (c->d_cnn_wb) { hipFree(c->d_cnn_wb); c->dev_bytes -= c->cnn_nwb * 2; c->d_cnn_wb = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_cnn_wb, std::max<size_t>(wb.size(), 8) * 2));
    c->cnn_nwb = wb.size(); c->dev_bytes += wb.size() * 2;
    HIPCHK(c, hipMemcpyAsync(c->d_cnn_wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cnn_wb_off = wb_off;
    // the fp16 path's weights: scaled per layer by a power of two into [2^13, 2^14) (so that the low pieces of all but the
    // tiniest weights are normal fp16 numbers), then two pieces per value (round to nearest even), same layout
    auto f16_bits = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
    auto f16_f32 = [](uint16_t b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
    std::vector<uint16_t> wh; std::vector<int64_t> wh_off(n_ops, 0); std::vector<float> post(n_ops, 1.0f);
    for (uint32_t i = 0; i < n_ops; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op != DN_CNN_CONV && o.op != DN_CNN_CONV_ADD) continue;
        wh_off[i] = (int64_t)wh.size();
        const size_t steps = (size_t)o.k * (o.cin / 32), blk = (size_t)o.cout * 32;
        wh.resize(wh.size() + steps * 2 * blk);
        uint16_t *dst = wh.data() + wh_off[i];
        const float *src = wl.data() + o.w;
        float wmax = 0.0f;
        for (size_t e = 0; e < steps * blk; e++) wmax = std::max(wmax, fabsf(src[e]));
        const int up = (wmax > 0.0f && std::isfinite(wmax)) ? 13 - ilogbf(wmax) : 0;
        const float mul = ldexpf(1.0f, up);
        post[i] = ldexpf(1.0f, -up);
        const size_t cbn = (size_t)(o.cin / 32);
        for (size_t st = 0; st < steps; st++)
            for (size_t e = 0; e < blk; e++) {
                const size_t cb = st / (size_t)o.k, tp = st % (size_t)o.k;
                const float x = src[(tp * cbn + cb) * blk + e] * mul;      // exact
                const uint16_t h = f16_bits(x);
                dst[(st * 2 + 0) * blk + e] = h; dst[(st * 2 + 1) * blk + e] = f16_bits(x - f16_f32(h));
            }
    }
    if (c->d_cnn_wh) { hipFree(c->d_cnn_wh); c->dev_bytes -= c->cnn_nwh * 2; c->d_cnn_wh = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_cnn_wh, std::max<size_t>(wh.size(), 8) * 2));
    c->cnn_nwh = wh.size(); c->dev_bytes += wh.size() * 2;
    HIPCHK(c, hipMemcpyAsync(c->d_cnn_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice, c->stream));
    if (!c->d_cnn_flag) {                                 // the range report block of a pass: word 0 for the host, two words per op (k3_cnn.hip range_report)
        HIPCHK(c, hipMalloc((void **)&c->d_cnn_flag, (2 + 2 * 1024) * sizeof(unsigned)));
        HIPCHK(c, hipMemsetAsync(c->d_cnn_flag, 0, (2 + 2 * 1024) * sizeof(unsigned), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->d_cnn_flag, 0, sizeof(unsigned), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cnn_wh_off = wh_off; c->cnn_post = post; c->cnn_one.assign(n_ops, 1.0f);
    { const char *e = getenv("DN_CNN_MATH");
      if (e) c->cnn_math = strcmp(e, "fp32") == 0 ? DN_CNN_MATH_FP32 : strcmp(e, "bf16x6") == 0 ? DN_CNN_MATH_BF16X6 : DN_CNN_MATH_F16X3; }
    c->cnn_f16_off = false; c->cnn_underflow_streak = 0;
    weights = wl.data();
    if (c->d_cnn_w) { hipFree(c->d_cnn_w); c->dev_bytes -= c->cnn_nw * sizeof(float); c->d_cnn_w = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_cnn_w, n_weights * sizeof(float)));
    c->cnn_nw = n_weights; c->dev_bytes += n_weights * sizeof(float);
    HIPCHK(c, hipMemcpyAsync(c->d_cnn_w, weights, n_weights * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cnn_ops.assign(ops, ops + n_ops);
    c->cnn_nbuf = (int)n_buffers;
    c->layer_ms.assign(n_ops, 0.0); c->layer_n.assign(n_ops, 0u);
    return DN_OK;
}

// activation rows resident per pass: 4 buffers x 256 channels x 4 B = 4 KiB per row -> 16 GiB of HBM at the default cap
static uint64_t cnn_row_cap() {
    const char *e = getenv("DN_CNN_ROWS");
    const uint64_t v = e ? strtoull(e, nullptr, 10) : (4ull << 20);
    return std::max<uint64_t>(v, 1024);
}

int dn_cnn_set_math(dn_ctx *c, int mode) {
    if (!c || (mode != DN_CNN_MATH_FP32 && mode != DN_CNN_MATH_BF16X6 && mode != DN_CNN_MATH_F16X3)) return DN_ERR_ARG;
    c->cnn_math = mode; c->cnn_f16_off = false; c->cnn_underflow_streak = 0;
    return DN_OK;
}

// A pass left fp16's range (word 0 of the range report: bit 0 = some split value > 65 504, bit 1 = a layer whose largest split value is < 2^-6) and is
// repeated with bf16 pieces.  An OVERFLOW is a property of the model: the context stays on bf16 pieces.  An UNDERFLOW may be one odd batch (a tiny
// batch, a layer that is nearly all zero after its ReLU): it is repeated on its own, and only a second one in a row switches the context over
// (round-4 advisor: one such pass used to double the matrix work of every later batch).  Said once on stderr, with the reason.
static void cnn_note_escalation(dn_ctx *c, unsigned flag) {
    c->cnn_escalations++;
    if (flag & 1u) c->cnn_f16_off = true;
    else if (++c->cnn_underflow_streak >= 2) c->cnn_f16_off = true;
    static bool said = false;
    if (!said) {
        said = true;
        fprintf(stderr, "dnascent_hip: a CNN pass was repeated with bf16 pieces (%s); dn_cnn_range_escalations counts the repeats\n",
                (flag & 1u) ? "an activation beyond fp16's range: the context stays on bf16 pieces"
                : (flag & 2u) ? "a whole layer below 2^-6: fp16's low pieces would be subnormal"
                : "the canary sequences' probabilities differ between fp16 and bf16 pieces by more than the tolerance");
    }
}

// THE CANARY (round 6).  The range report above sees a layer's MAXIMUM only: a layer whose values are mostly ~1e-5 beside a few of O(1) keeps the absolute
// 2^-25 error of subnormal low pieces on the small ones and nothing says so.  Instead of guessing from statistics which distributions hurt, the first
// sequences of every batch (>= 4 096 positions, <= 8 sequences) go through the network a second time with bf16 pieces (fp32's exponent range) and the two
// sets of probabilities are compared ON THE DEVICE: a difference above the tolerance (default 1e-4, the contract's bar; DN_CNN_CANARY_TOL) raises bit 2 of
// the report word and the batch is repeated with bf16 pieces exactly as for an overflow.  It measures the contract's own quantity -- whatever the cause:
// range, a cancellation the 22-bit products cannot carry, a kernel bug -- on a sample of ~0.2 % of a 500-read batch (DN_CNN_CANARY=0 switches it off).
// (read per batch, not cached: a test switches the canary off and on inside one process)
static bool cnn_canary_enabled() { const char *e = getenv("DN_CNN_CANARY"); return !(e && atoi(e) == 0); }
static float cnn_canary_tol() { const char *e = getenv("DN_CNN_CANARY_TOL"); return e ? (float)atof(e) : 1e-4f; }
uint64_t dn_cnn_canaries(dn_ctx *c) { return c ? c->cnn_canaries : 0; }

// the lane's activation buffers for passes of up to lane_rows rows (caller holds L->mu)
static int lane_size(dn_ctx *c, CnnLane *L, uint64_t lane_rows) {
    int rc;
    for (int b = 0; b < c->cnn_nbuf; b++)
        if ((rc = lane_grow(c, L, L->buf[b], (size_t)lane_rows * 256 * sizeof(float)))) return rc;
    if ((rc = lane_grow(c, L, L->enclen, (size_t)lane_rows)) || (rc = lane_grow(c, L, L->enchist, 64 * sizeof(unsigned))) ||
        (rc = lane_grow(c, L, L->permsrc, (size_t)lane_rows * sizeof(uint64_t))) || (rc = lane_grow(c, L, L->permrow, (size_t)lane_rows * sizeof(unsigned))) ||
        (rc = lane_grow(c, L, L->valid, (size_t)lane_rows + 256)) || (rc = lane_grow(c, L, L->live, 256))) return rc;    // + 256: k3_sep_pair's 120-row tiles look up to 127 rows past the pass
    return DN_OK;
}

int dn_cnn_reserve(dn_ctx *c, uint64_t rows) {
    if (!c) return DN_ERR_ARG;
    if (c->cnn_ops.empty()) return fail(c, DN_ERR_STATE, "dn_load_cnn must be called first");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, DN_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    const uint64_t cap_rows = (cnn_row_cap() + 255) / 256 * 256;
    const uint64_t want = rows ? std::min<uint64_t>((rows + 255) / 256 * 256, cap_rows) : cap_rows;
    CnnLane *L = lane_get(c);
    if (!L) return fail(c, DN_ERR_HIP, "cannot create the CNN lane of device %d", c->device);
    LaneUse lane_use(L);
    std::lock_guard<std::mutex> lane_lock(L->mu);
    return lane_size(c, L, want);
}

// the CNN over n sequences whose input tensors (core, residual, signal) are already on the device.
// ub[r] (host) bounds the positions of sequence r; the actual counts are on the device (d_npos: 0 = nothing to do).  Activation
// rows are laid out from the BOUNDS -- sequence r owns ub[r] rows + CNN_PAD zero rows, of which the first d_npos[r] are live and
// the rest stay zero / invalid -- so the pass partition, every row offset and every grid size are known without reading anything
// back: the whole network is enqueued without a host synchronisation.  (After eventalign ub = reference length - 8 and a passing
// read fills ~95 % of it.)  check_now: examine the fp16 range flag after every pass (dn_cnn_infer); otherwise the flag is copied
// to page-locked memory at the end and examined by whoever synchronises next (cnn_settle).
static int cnn_execute(dn_ctx *c, uint32_t n, const unsigned *ub, const unsigned *d_npos, const uint64_t *io_off, const float *d_core,
                       const float *d_resid, const float *d_sig, float *d_probs, bool check_now) {
    if (c->cnn_ops.empty()) return fail(c, DN_ERR_STATE, "dn_load_cnn must be called first");
    const uint64_t cap = cnn_row_cap();
    int rc;
    // page-locked staging of the per-sequence tables (the H2D copies below must not read pageable memory after this call returns)
    if (n > c->cnn_meta_cap) {
        if (c->p_cnn_rowoff) { hipHostFree(c->p_cnn_rowoff); hipHostFree(c->p_cnn_iooff); }
        c->p_cnn_rowoff = nullptr; c->p_cnn_iooff = nullptr; c->cnn_meta_cap = 0;
        const size_t m = std::max<size_t>((size_t)n + n / 2 + 16, 4096);
        HIPCHK(c, hipHostMalloc((void **)&c->p_cnn_rowoff, m * sizeof(unsigned), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&c->p_cnn_iooff, m * sizeof(uint64_t), hipHostMallocDefault));
        c->cnn_meta_cap = m;
    }
    unsigned *row_off = c->p_cnn_rowoff;
    memcpy(c->p_cnn_iooff, io_off, n * sizeof(uint64_t));
    struct Pass { uint32_t r0, r1; unsigned rows, max_pos, n_pos; };
    std::vector<Pass> passes;
    // BALANCED passes (round 5): the fewest passes the cap allows, of about equal size -- cut where the running total crosses k / P of the batch, not where the cap is
    // hit.  Filled to the cap, 500 reads of 50 kb at 4 Mi rows were six passes of 83 reads and a seventh of TWO (48 launches over 100 k rows: ~1 % of the step).
    // Sequences are indivisible, so P passes may not hold the batch although P x cap rows would: then P + 1 balanced ones are cut (a few tries at most).
    uint64_t total = 0, max_rows = 0;
    for (uint32_t r = 0; r < n; r++) {
        total += (uint64_t)ub[r] + 8;
        if ((uint64_t)ub[r] + 16 >= (1ull << 31)) return fail(c, DN_ERR_OVERFLOW, "sequence %u has too many positions for one CNN pass", r);
    }
    auto cut = [&](uint64_t n_pass) {
        passes.clear(); max_rows = 0;
        uint64_t rows = 8, done_rows = 0; unsigned max_pos = 1, pass_pos = 0; uint32_t r0 = 0;
        for (uint32_t r = 0; r < n; r++) {
            const unsigned np = ub[r];
            const bool quota = passes.size() + 1 < n_pass && (done_rows + rows - 8) * n_pass >= total * (passes.size() + 1);
            if ((rows + np + 8 > cap || quota) && r > r0) {
                done_rows += rows - 8;
                const uint64_t rr = (rows + 255) / 256 * 256;
                passes.push_back({ r0, r, (unsigned)rr, max_pos, pass_pos }); max_rows = std::max(max_rows, rr);
                r0 = r; rows = 8; max_pos = 1; pass_pos = 0;
            }
            row_off[r] = (unsigned)rows;
            rows += np + 8; pass_pos += np;
            max_pos = std::max(max_pos, np);
        }
        const uint64_t rr = (rows + 255) / 256 * 256; passes.push_back({ r0, n, (unsigned)rr, max_pos, pass_pos }); max_rows = std::max(max_rows, rr);
    };
    {
        static const bool greedy = getenv("DN_CNN_GREEDY_PASSES") && atoi(getenv("DN_CNN_GREEDY_PASSES")) != 0;     // the old partition (A / B)
        uint64_t n_pass = greedy ? 1 : std::max<uint64_t>(1, (total + 8 + cap - 1) / cap);
        for (int tries = 0; tries < 4; tries++, n_pass++) { cut(n_pass); if (greedy || passes.size() <= n_pass) break; }
        if (max_rows >= (1ull << 31)) return fail(c, DN_ERR_OVERFLOW, "a CNN pass of %llu rows", (unsigned long long)max_rows);
    }
    CnnLane *L = lane_get(c);
    if (!L) return fail(c, DN_ERR_HIP, "cannot create the CNN lane of device %d", c->device);
    LaneUse lane_use(L);                                   // declared before the lock: released after it (a concurrent dn_shutdown sees the lane busy until then)
    std::lock_guard<std::mutex> lane_lock(L->mu);
    // A pass of a streamed batch holds between (cap - the longest read) and cap rows: sized by what THIS batch needs, the lane's buffers were freed and
    // reallocated every time a batch came a few rows closer to cap than any before it -- a hipFree (it waits for the whole device: every batch in flight
    // drained) and a 16 GiB hipMalloc, 1.3-2.1 s each, nine times in the first seconds of bench.py --scope mixed (round 4, DN_TRACE_ENQUEUE).  A batch that
    // fills a quarter of a pass gets the full pass allocated once; small calls (tests, dn_cnn_infer of a few reads) keep small buffers.
    const uint64_t cap_rows = (cap + 255) / 256 * 256;
    const uint64_t lane_rows = max_rows >= cap_rows / 4 ? std::max(max_rows, cap_rows) : max_rows;
    if ((rc = lane_size(c, L, lane_rows))) return rc;
    if ((rc = dgrow(c, c->cnn_rowoff, n * sizeof(unsigned))) || (rc = dgrow(c, c->cnn_iooff, n * sizeof(uint64_t)))) return rc;
    if (!c->p_cnn_flag) HIPCHK(c, hipHostMalloc((void **)&c->p_cnn_flag, sizeof(unsigned), hipHostMallocDefault));
    // hand the batch over to the lane: everything the context's stream has enqueued so far (eventalign, the position counts)
    hipStream_t st = L->stream;
    HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_ready, 0));
    HIPCHK(c, hipMemcpyAsync(c->cnn_iooff.p, c->p_cnn_iooff, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));     // (row offsets: k3_layout, on the device)
    if (c->prof) kc_launch_nop(st);     // the profiling event below must be stamped AFTER the hand-over wait: behind a kernel it is
    {
    Timed t(c, DN_K_CNN, st);
    bool canary_due = cnn_canary_enabled() && c->d_cnn_wb != nullptr;     // once per call: on its first pass (if that pass runs on fp16 pieces at all)
    for (const Pass &ps : passes) {
        HIPCHK(c, hipMemsetAsync(L->valid.p, 0, (size_t)ps.rows, st));
        CnnRun run{};
        run.ops = c->cnn_ops.data(); run.n_ops = (int)c->cnn_ops.size(); run.wts = c->d_cnn_w;
        for (int b = 0; b < c->cnn_nbuf; b++) run.buf[b] = (float *)L->buf[b].p;
        run.n_buf = c->cnn_nbuf;
        run.rows.row_off = (const unsigned *)c->cnn_rowoff.p; run.rows.valid = (const uint8_t *)L->valid.p; run.rows.rows = ps.rows;
        run.rows.r0 = ps.r0; run.rows.r1 = ps.r1;
        run.rows.n_pos = d_npos; run.rows.io_off = (const uint64_t *)c->cnn_iooff.p;
        run.valid = (uint8_t *)L->valid.p;
        run.n_pass_pos = ps.n_pos; run.enc_len = (uint8_t *)L->enclen.p; run.enc_hist = (unsigned *)L->enchist.p;
        run.perm_src = (uint64_t *)L->permsrc.p; run.perm_row = (unsigned *)L->permrow.p;
        run.core = d_core; run.resid = d_resid; run.sig = d_sig; run.probs = d_probs; run.max_pos = ps.max_pos;
        run.mark = c->prof ? cnn_mark : nullptr; run.mark_who = c;
        run.row_off_w = (unsigned *)c->cnn_rowoff.p; run.live = (int *)L->live.p;
        // fp16 pieces are only valid while every activation fits fp16: the kernels raise range_flag otherwise and the pass is
        // repeated with bf16 pieces (same result contract, 2x the matrix work) -- never a silently wrong answer
        for (int math = (c->cnn_math == DN_CNN_MATH_F16X3 && (c->cnn_f16_off || c->cnn_bf16_once)) ? DN_CNN_MATH_BF16X6 : c->cnn_math;;) {
            run.wts_split = math == DN_CNN_MATH_BF16X6 ? c->d_cnn_wb : math == DN_CNN_MATH_F16X3 ? c->d_cnn_wh : nullptr;
            run.wb_off = math == DN_CNN_MATH_F16X3 ? c->cnn_wh_off.data() : c->cnn_wb_off.data();
            run.pieces = math == DN_CNN_MATH_F16X3 ? 2 : 3;
            run.post = math == DN_CNN_MATH_F16X3 ? c->cnn_post.data() : c->cnn_one.data();
            run.range_flag = c->d_cnn_flag;
            if (k3_run(run, st)) return fail(c, DN_ERR_ARG, "unsupported op in the CNN description");
            if (math == DN_CNN_MATH_F16X3 && canary_due) {
                // the pass's first sequences once more, with bf16 pieces, into a buffer of their own; then compared with what the fp16 pass wrote
                canary_due = false;
                uint32_t k1 = ps.r0; uint64_t crows = 8, cpos = 0; unsigned cmax = 1;
                while (k1 < ps.r1 && k1 - ps.r0 < 8 && cpos < 4096) { crows += ub[k1] + 8; cpos += ub[k1]; cmax = std::max(cmax, ub[k1]); k1++; }
                const uint64_t span = io_off[k1 - 1] + ub[k1 - 1] - io_off[ps.r0];                 // positions from the first canary sequence's first to the last one's bound
                // (sized once for any read up to 512 k positions: regrowing frees, and hipFree waits for the whole device -- every batch in flight drained)
                if ((rc = dgrow(c, c->cnn_canary, (size_t)std::max<uint64_t>(span, 512u << 10) * 3 * sizeof(float)))) return rc;
                CnnRun cr = run;
                cr.rows.r1 = k1; cr.rows.rows = (unsigned)((crows + 255) / 256 * 256); cr.max_pos = cmax; cr.n_pass_pos = (unsigned)cpos;
                cr.wts_split = c->d_cnn_wb; cr.wb_off = c->cnn_wb_off.data(); cr.pieces = 3; cr.post = c->cnn_one.data();
                cr.probs = (float *)c->cnn_canary.p - 3 * io_off[ps.r0];                            // k3_dense_softmax writes at probs + 3 (io_off[r] + p)
                cr.mark = nullptr;
                HIPCHK(c, hipMemsetAsync(L->valid.p, 0, (size_t)ps.rows, st));
                if (k3_run(cr, st)) return fail(c, DN_ERR_ARG, "unsupported op in the CNN description");
                k3_launch_canary_compare(d_probs, cr.probs, cr.rows, cmax, cnn_canary_tol(), c->d_cnn_flag, st);
                c->cnn_canaries++;
            }
            if (math != DN_CNN_MATH_F16X3 || !check_now) break;
            HIPCHK(c, hipMemcpyAsync(c->p_cnn_flag, c->d_cnn_flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (!*c->p_cnn_flag) { c->cnn_underflow_streak = 0; break; }
            HIPCHK(c, hipMemsetAsync(c->d_cnn_flag, 0, sizeof(unsigned), st));
            cnn_note_escalation(c, *c->p_cnn_flag);
            math = DN_CNN_MATH_BF16X6;
        }
    }
    }
    if (!check_now) HIPCHK(c, hipMemcpyAsync(c->p_cnn_flag, c->d_cnn_flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    // ... and back: the context's stream continues (dn_collect's compaction, the taps) when the lane's last operation is done
    HIPCHK(c, hipEventRecord(c->ev_done, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
    HIPCHK(c, hipGetLastError());
    return DN_OK;
}

uint64_t dn_cnn_range_escalations(dn_ctx *c) { return c ? c->cnn_escalations : 0; }

static int cnn_enqueue_batch(dn_ctx *c) {
    const uint32_t n = (uint32_t)c->B.n_reads;
    // bound of a read's positions: one per 9-mer of referenceSeqMappedTo (alignment.cpp:547: reference index < length - 8)
    std::vector<unsigned> ub(n);
    for (uint32_t r = 0; r < n; r++) ub[r] = (unsigned)(c->h_ref_off[r + 1] - c->h_ref_off[r]) - (DN_KMER - 1);
    int rc;
    if ((rc = dgrow(c, c->cnn_npos, n * sizeof(unsigned)))) return rc;
    kc_launch_npos(c->B, (unsigned *)c->cnn_npos.p, c->stream);      // positions eventalign found; 0 for reads that failed
    return cnn_execute(c, n, ub.data(), (const unsigned *)c->cnn_npos.p, c->h_ref_off.data(), c->ea.core, c->ea.resid, c->ea.sig, c->d_probs, false);
}

// after dn_run_cnn: wait for the stream, and if some activation left fp16's range repeat the batch with bf16 pieces
static int cnn_settle(dn_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->cnn_pending) return DN_OK;
    c->cnn_pending = false;
    if (c->p_cnn_flag && *c->p_cnn_flag) {
        HIPCHK(c, hipMemsetAsync(c->d_cnn_flag, 0, sizeof(unsigned), c->stream));
        cnn_note_escalation(c, *c->p_cnn_flag);
        *c->p_cnn_flag = 0;
        c->cnn_bf16_once = true;                               // this batch again with bf16 pieces, whatever the context does afterwards
        int rc = cnn_enqueue_batch(c);
        c->cnn_bf16_once = false;
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else if (c->p_cnn_flag) c->cnn_underflow_streak = 0;
    return DN_OK;
}

int dn_run_cnn(dn_ctx *c) {
    int rc = need(c, 6, "dn_run_cnn"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->stage = std::max(c->stage, 7); return DN_OK; }
    if ((rc = cnn_enqueue_batch(c))) return rc;
    c->cnn_pending = true;
    c->stage = 7;
    return DN_OK;
}

int dn_run_detect(dn_ctx *c) {
    int rc;
    static const bool trace = [] { const char *e = getenv("DN_TRACE_ENQUEUE"); return e && e[0] == '1'; }();   // which stage's ENQUEUE made the host wait (they should cost launches only)
    if (!trace) {
        if ((rc = dn_run_normalise(c))) return rc;
        if ((rc = dn_run_eventalign(c))) return rc;
        return dn_run_cnn(c);
    }
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    const double t0 = now();
    if ((rc = dn_run_normalise(c))) return rc;
    const double t1 = now();
    if ((rc = dn_run_eventalign(c))) return rc;
    const double t2 = now();
    rc = dn_run_cnn(c);
    const double t3 = now();
    if (t3 - t0 > 0.010) fprintf(stderr, "dn_run_detect: %d reads, max length %u: enqueue normalise %.1f ms, eventalign %.1f ms, cnn %.1f ms\n", c->B.n_reads, c->max_len,
                                 (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    return rc;
}

int dn_cnn_infer(dn_ctx *c, uint32_t n_seq, const uint32_t *len, const float *core, const float *residual, const float *signal, float *probs) {
    if (!c || !len || !core || !residual || !signal || !probs) return DN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> off(n_seq + 1, 0);
    for (uint32_t i = 0; i < n_seq; i++) off[i + 1] = off[i] + len[i];
    const size_t L = off[n_seq];
    if (L == 0) return DN_OK;
    int rc;
    if ((rc = dgrow(c, c->cnn_in[0], L * sizeof(float))) || (rc = dgrow(c, c->cnn_in[1], L * sizeof(float))) ||
        (rc = dgrow(c, c->cnn_in[2], L * DN_RAWDEPTH * sizeof(float))) || (rc = dgrow(c, c->cnn_out, L * 3 * sizeof(float))) ||
        (rc = dgrow(c, c->cnn_npos, n_seq * sizeof(unsigned)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->cnn_in[0].p, core, L * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->cnn_in[1].p, residual, L * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->cnn_in[2].p, signal, L * DN_RAWDEPTH * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->cnn_npos.p, len, n_seq * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // the sources are the caller's pageable arrays
    if ((rc = cnn_execute(c, n_seq, len, (const unsigned *)c->cnn_npos.p, off.data(), (const float *)c->cnn_in[0].p, (const float *)c->cnn_in[1].p,
                          (const float *)c->cnn_in[2].p, (float *)c->cnn_out.p, true))) return rc;
    return d2h(c, probs, (const float *)c->cnn_out.p, L * 3);
}

int dn_get_probabilities(dn_ctx *c, uint32_t read, uint64_t cap, float *probs) {
    CHECK_READ(7, "dn_get_probabilities");
    if ((rc = cnn_settle(c))) return rc;
    if ((rc = fetch_res(c))) return rc;
    CHECK_CAP("dn_get_probabilities", c->h_res[read].n_positions, cap);
    return d2h(c, probs, c->d_probs + c->h_ref_off[read] * 3, (size_t)c->h_res[read].n_positions * 3);
}

int dn_load_fit_models(dn_ctx *c, const double *um, const double *us, const double *am, const double *as) {
    if (!c || !um || !us || !am || !as) return DN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    // normalPDF's per-k-mer constants with the host libm (probability.cpp:145-148): {mu, 2 s^2, log(1/sqrt(2 s^2 pi)), 1/sqrt(2 s^2 pi)}
    std::vector<double4> t(DN_NKMER);
    for (int m = 0; m < 2; m++) {
        const double *mu = m ? am : um, *sd = m ? as : us;
        for (size_t k = 0; k < DN_NKMER; k++) {
            if (!(sd[k] > 0.)) return fail(c, DN_ERR_ARG, "fit model %d: std of k-mer %zu is not positive", m, k);
            const double s2 = sd[k] * sd[k], d2 = 2.0 * s2;
            const double cc = 1.0 / sqrt(d2 * M_PI);
            t[k] = make_double4(mu[k], d2, -1.0 / d2, cc);
        }
        if (!c->d_fit[m]) { HIPCHK(c, hipMalloc((void **)&c->d_fit[m], DN_NKMER * sizeof(double4))); c->dev_bytes += DN_NKMER * sizeof(double4); }
        HIPCHK(c, hipMemcpyAsync(c->d_fit[m], t.data(), DN_NKMER * sizeof(double4), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->have_fit = true;
    return DN_OK;
}

int dn_run_hmm(dn_ctx *c) {
    int rc = need(c, 5, "dn_run_hmm"); if (rc) return rc;
    if (c->B.n_reads == 0) { c->hmm_done = true; c->h_npoi.clear(); c->h_nhmm.clear(); return DN_OK; }
    if (!c->have_fit) return fail(c, DN_ERR_STATE, "dn_load_fit_models must be called first");
    const uint32_t n = (uint32_t)c->B.n_reads;
    const size_t NR = (size_t)c->h_ref_off[n];
    if ((rc = dgrow(c, c->hmm_poi, NR * 4)) || (rc = dgrow(c, c->hmm_npoi, n * 4)) || (rc = dgrow(c, c->hmm_nev, NR * 4)) ||
        (rc = dgrow(c, c->hmm_ok, NR)) || (rc = dgrow(c, c->hmm_la, NR * 8)) || (rc = dgrow(c, c->hmm_lt, NR * 8)) ||
        (rc = dgrow(c, c->hmm_reads, n * sizeof(HmmReadH)))) return rc;
    if ((rc = fetch_res(c))) return rc;
    HmmConstsH hc;
    // the reference's transitions are eln() of the config.h:42 probabilities; the kernel multiplies probabilities, so it gets
    // exp(eln(p)) -- the value the log-space arithmetic effectively uses -- computed with the host libm
    const double lD2D = log(0.3), lD2M = log(0.7), lI2M = log(0.999), lM2D = log(0.0025), lM2I = log(0.001), lI2I = log(0.001);   // :245-250
    hc.D2D = exp(lD2D); hc.D2M = exp(lD2M); hc.I2M = exp(lI2M); hc.M2D = exp(lM2D); hc.M2I = exp(lM2I); hc.I2I = exp(lI2I);
    hc.ln025 = exp(log(0.25)); hc.ln05 = exp(log(0.5));
    std::vector<HmmReadH> hr(n);
    std::vector<int> newstat(n, -1);
    for (uint32_t r = 0; r < n; r++) {
        const ReadRes &R = c->h_res[r];
        int neg = 0;
        const double liM2M = h_eln(1. - (1. / R.events_per_base), &neg);                  // :253
        const double leM2M = h_eln(1.0 - lM2D - lM2I - liM2M, &neg);                      // :254 (sic: log values inside)
        hr[r].iM2M = exp(liM2M); hr[r].eM2M = exp(leM2M);
        hr[r].endM = exp(h_lnSum(leM2M, lM2D));                                           // :366
        if (R.status == 0 && neg) newstat[r] = DN_READ_FAIL_NEGATIVE_LOG;                 // the reference throws NegativeLog
    }
    for (uint32_t r = 0; r < n; r++)
        if (newstat[r] >= 0) HIPCHK(c, hipMemcpyAsync(&c->B.res[r].status, &newstat[r], sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->hmm_reads.p, hr.data(), n * sizeof(HmmReadH), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->hmm_ok.p, 0, NR, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // hr / newstat are locals
    HmmDevH H{ c->d_fit[0], c->d_fit[1], (unsigned *)c->hmm_poi.p, (unsigned *)c->hmm_npoi.p, (unsigned *)c->hmm_nev.p,
               (unsigned char *)c->hmm_ok.p, (double *)c->hmm_la.p, (double *)c->hmm_lt.p };
    { Timed t(c, DN_K_HMM); k_hmm_launch(c->B, &H, c->hmm_reads.p, &hc, c->max_ref, c->stream); }
    HIPCHK(c, hipGetLastError());
    // per-read call counts for the summaries (one flag byte per reference base back: the calls themselves stay on the device)
    c->h_npoi.resize(n); c->h_nhmm.assign(n, 0);
    if ((rc = d2h(c, c->h_npoi.data(), (const unsigned *)c->hmm_npoi.p, n))) return rc;
    std::vector<unsigned char> ok(NR);
    if ((rc = d2h(c, ok.data(), (const unsigned char *)c->hmm_ok.p, NR))) return rc;
    for (uint32_t r = 0; r < n; r++) {
        const unsigned char *o = ok.data() + c->h_ref_off[r];
        unsigned k = 0;
        for (unsigned i = 0; i < c->h_npoi[r]; i++) k += o[i];
        c->h_nhmm[r] = k;
    }
    c->hmm_done = true;
    return DN_OK;
}

int dn_get_hmm_calls(dn_ctx *c, uint32_t read, uint64_t cap, uint32_t *pos_on_ref, uint32_t *pos_on_query, int32_t *global_pos, uint32_t *n_events,
                     double *log_analogue, double *log_thymidine, double *llr) {
    CHECK_READ(5, "dn_get_hmm_calls");
    if (!c->hmm_done) return fail(c, DN_ERR_STATE, "dn_get_hmm_calls before dn_run_hmm");
    CHECK_CAP("dn_get_hmm_calls", read < c->h_nhmm.size() ? c->h_nhmm[read] : 0u, cap);
    const uint64_t f0 = c->h_ref_off[read];
    const unsigned np = c->h_npoi[read];
    if (np == 0) return DN_OK;
    std::vector<unsigned> poi(np), nev(np); std::vector<unsigned char> ok(np); std::vector<double> la(np), lt(np);
    if ((rc = d2h(c, poi.data(), (const unsigned *)c->hmm_poi.p + f0, np)) || (rc = d2h(c, nev.data(), (const unsigned *)c->hmm_nev.p + f0, np)) ||
        (rc = d2h(c, ok.data(), (const unsigned char *)c->hmm_ok.p + f0, np)) || (rc = d2h(c, la.data(), (const double *)c->hmm_la.p + f0, np)) ||
        (rc = d2h(c, lt.data(), (const double *)c->hmm_lt.p + f0, np))) return rc;
    const size_t nref = (size_t)(c->h_ref_off[read + 1] - f0);
    std::vector<uint32_t> r2q;
    if (pos_on_query) { r2q.resize(nref); if ((rc = d2h(c, r2q.data(), c->B.ref2query + f0, nref))) return rc; }
    const bool rev = c->h_is_rev[read] != 0;
    size_t o = 0;
    for (unsigned q = 0; q < np; q++) {
        const unsigned i = rev ? np - 1 - q : q;                     // std::reverse of the POIs for reverse reads (:405)
        if (!ok[i]) continue;
        if (pos_on_ref) pos_on_ref[o] = poi[i];
        if (pos_on_query) pos_on_query[o] = r2q[poi[i]];
        if (global_pos) global_pos[o] = rev ? (c->h_ref_end[read] - (int)poi[i] - 1) : (c->h_ref_start[read] + (int)poi[i]);   // :531-541
        if (n_events) n_events[o] = nev[i];
        // log 0 is NaN in the reference (-inf on the device); NaN - x = NaN
        const double a = std::isinf(la[i]) ? NAN : la[i], t = std::isinf(lt[i]) ? NAN : lt[i];
        if (log_analogue) log_analogue[o] = a;
        if (log_thymidine) log_thymidine[o] = t;
        if (llr) llr[o] = a - t;
        o++;
    }
    return DN_OK;
}

int dn_profile_enable(dn_ctx *c, int on) { if (!c) return DN_ERR_ARG; c->prof = on != 0; return DN_OK; }
int dn_profile_reset(dn_ctx *c) {
    if (!c) return DN_ERR_ARG;
    prof_collect(c);
    for (int i = 0; i < DN_K_COUNT; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    std::fill(c->layer_ms.begin(), c->layer_ms.end(), 0.0); std::fill(c->layer_n.begin(), c->layer_n.end(), 0u);
    return DN_OK;
}
int dn_profile_get(dn_ctx *c, int k, double *ms, uint32_t *launches) {
    if (!c || k < 0 || k >= DN_K_COUNT) return DN_ERR_ARG;
    prof_collect(c);
    if (ms) *ms = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return DN_OK;
}

int dn_profile_get_layer(dn_ctx *c, uint32_t op, double *ms, uint32_t *launches, char *kernel, size_t kernel_cap) {
    if (!c || op >= c->cnn_ops.size()) return DN_ERR_ARG;
    prof_collect(c);
    if (ms) *ms = c->layer_ms[op];
    if (launches) *launches = c->layer_n[op];
    if (kernel && kernel_cap) {
        CnnRun run{};
        run.ops = c->cnn_ops.data(); run.n_ops = (int)c->cnn_ops.size();
        const int math = (c->cnn_math == DN_CNN_MATH_F16X3 && c->cnn_f16_off) ? DN_CNN_MATH_BF16X6 : c->cnn_math;
        run.wts_split = math == DN_CNN_MATH_FP32 ? nullptr : (const uint16_t *)(uintptr_t)1;       // only its nullness is looked at
        run.pieces = math == DN_CNN_MATH_F16X3 ? 2 : 3;
        run.rows.rows = 256;                                   // pass row counts are multiples of 256 (cnn_execute)
        if (k3_describe(run, (int)op, kernel, kernel_cap) < 0) return DN_ERR_ARG;
    }
    return DN_OK;
}

}  // extern "C"
