// dn_host.cpp -- host side above the C-ABI: read model, CIGAR flattening, batch packing, stage drivers.
#include "dn_host.h"
#include <emmintrin.h>
#include <mutex>
#include "dn_synth.h"

#include <math.h>
#include <stdio.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <omp.h>
#include <stdlib.h>

namespace DNAscent {

// Threads of the host-side parallel loops (record formatting, container reads).  NOT every hardware thread: min(64, the CPUs the process may
// actually USE).  The hosts of the pool show 256 hardware threads, but their containers run under a cgroup CPU quota (cpu.max = 16 CPUs):
// a 64-wide team burns the 100 ms period's quota in 25 ms and the kernel then freezes EVERY thread of the process for the rest of the
// period -- the one that drives the GPU included (round 4: dn_run_detect, 2 ms of launches, took 80-130 ms per batch inside run_detect, and
// cpu.stat counted 307 throttled periods of 681; round 3 had seen the same thing as "a 256-wide team formats slower than a 64-wide one").
// DN_HOST_THREADS overrides; the Python drivers also export OMP_WAIT_POLICY=passive before the library loads.
static int cgroupCpus() {                                    // CPUs' worth of quota: cgroup v2 cpu.max "quota period" | v1 cfs_quota_us / cfs_period_us; 0 = unlimited / unknown
    long q = -1, per = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        if (fscanf(f, "%63s %ld", a, &per) == 2 && strcmp(a, "max") != 0) q = atol(a);
        fclose(f);
    } else {
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%ld", &q) == 1 && fscanf(fp, "%ld", &per) == 1) {} else q = -1;
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    return (q > 0 && per > 0) ? (int)((q + per - 1) / per) : 0;
}
int hostThreads() {
    static const int n = [] {
        const char *e = getenv("DN_HOST_THREADS");
        int want = e ? atoi(e) : 0;
        if (want > 0) return std::max(1, std::min(want, omp_get_thread_limit()));      // said explicitly: taken as it is (the loops carry num_threads())
        want = std::min(64, omp_get_num_procs());
        const int quota = cgroupCpus();
        if (quota > 0) want = std::min(want, quota);
        // OMP_NUM_THREADS is honoured when it asks for a team; "1" is what torch.distributed.run exports to every worker by default and would
        // make each rank of a multi-GPU run load, pack and format on ONE thread -- the Python drivers set DN_HOST_THREADS to the rank's share instead
        const int omp = omp_get_max_threads();
        if (omp > 1) want = std::min(want, omp);
        return std::max(1, want);
    }();
    return n;
}

std::string reverseComplement(const std::string &s) {
    // common.h:91-150: reverse, then complement (IUPAC codes the reference maps are kept; anything else is dropped there,
    // which cannot happen for A/C/G/T/N input)
    std::string out(s.rbegin(), s.rend());
    for (char &c : out) {
        switch (c) {
            case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break; case 'C': c = 'G'; break;
            case 'U': c = 'A'; break; case 'Y': c = 'R'; break; case 'R': c = 'Y'; break; case 'K': c = 'M'; break;
            case 'M': c = 'K'; break; case 'B': c = 'V'; break; case 'D': c = 'H'; break; case 'H': c = 'D'; break;
            case 'V': c = 'B'; break; default: break;   // N, S, W stay
        }
    }
    return out;
}

int parseCigar(const std::vector<uint32_t> &ops, const std::vector<uint32_t> &lens, bool isReverse, size_t queryLen,
               std::vector<uint32_t> &ref2query, std::vector<int32_t> &query2ref, std::vector<uint8_t> &ref2del) {
    // htsInterface.cpp:59-157.  The reference fills three std::maps; insertions / soft clips write map slots AHEAD of
    // the current reference position (:99-107) that later ops overwrite, and leave query2ref entries pointing past the
    // insertion point.  Replaying the same writes on arrays reproduces every lookup the hot path performs.
    size_t refLen = 0, maxAhead = 0;
    for (size_t i = 0; i < ops.size(); i++) {
        if (ops[i] == 0 || ops[i] == 7 || ops[i] == 8 || ops[i] == 2 || ops[i] == 3) refLen += lens[i];
        else if (ops[i] == 1 || ops[i] == 4) maxAhead = std::max<size_t>(maxAhead, lens[i]);
    }
    std::vector<uint32_t> r2q(refLen + maxAhead + 1, 0);
    std::vector<uint8_t> r2d(refLen + maxAhead + 1, 0);
    query2ref.assign(queryLen + 1, -1);
    int qp = 0, rp = 0;
    const size_t n = ops.size();
    for (size_t c = 0; c < n; c++) {
        const size_t i = isReverse ? (n - 1 - c) : c;                       // :69
        const int op = (int)ops[i], ol = (int)lens[i];
        if (op == 0 || op == 7 || op == 8) {                                 // BAM_CMATCH / CEQUAL / CDIFF
            for (int j = rp; j < rp + ol; j++) {
                r2q[j] = (uint32_t)qp; r2d[j] = 0;
                if ((size_t)qp <= queryLen) query2ref[qp] = j;
                qp++;
            }
            rp += ol;
        } else if (op == 2 || op == 3) {                                     // BAM_CDEL / CREF_SKIP
            for (int j = rp; j < rp + ol; j++) {
                r2q[j] = (uint32_t)qp; r2d[j] = 1;
                if ((size_t)qp <= queryLen) query2ref[qp] = j;
            }
            rp += ol;
        } else if (op == 4 || op == 1) {                                     // BAM_CSOFT_CLIP / CINS
            for (int j = rp; j < rp + ol; j++) {
                r2q[j] = (uint32_t)qp; r2d[j] = 0;
                if ((size_t)qp <= queryLen) query2ref[qp] = j;
                qp++;
            }
        }                                                                    // hard clips / padding: ignored
    }
    ref2query.assign(r2q.begin(), r2q.begin() + refLen);
    ref2del.assign(r2d.begin(), r2d.begin() + refLen);
    return (int)refLen;
}

void ReadBatch::clear() {                                   // keeps the vectors' capacity: a streamed host reuses its batch objects
    unpin();
    readID.clear(); contig.clear(); adc.clear(); adc_off.assign(1, 0); cal_offset.clear(); cal_scale.clear();
    basecall.clear(); basecall_off.assign(1, 0); refseq.clear(); refseq_off.assign(1, 0);
    ref2query.clear(); query2ref.clear(); ref2del.clear(); ref_start.clear(); ref_end.clear(); is_reverse.clear(); summary.clear();
}

int ReadBatch::pin() {
    unpin();
    auto reg = [&](const void *p, size_t bytes) {
        if (!bytes) return true;
        if (dn_host_register(const_cast<void *>(p), bytes) != DN_OK) return false;
        pinned.push_back(const_cast<void *>(p));
        return true;
    };
    const bool ok = reg(adc.data(), adc.size() * 2) && reg(cal_offset.data(), cal_offset.size() * 4) && reg(cal_scale.data(), cal_scale.size() * 4) &&
                    reg(basecall.data(), basecall.size()) && reg(refseq.data(), refseq.size()) && reg(ref2query.data(), ref2query.size() * 4) &&
                    reg(query2ref.data(), query2ref.size() * 4) && reg(ref2del.data(), ref2del.size()) && reg(ref_start.data(), ref_start.size() * 4) &&
                    reg(ref_end.data(), ref_end.size() * 4) && reg(is_reverse.data(), is_reverse.size());
    if (!ok) { unpin(); return DN_ERR_HIP; }
    return DN_OK;
}
void ReadBatch::unpin() {
    for (void *p : pinned) dn_host_unregister(p);
    pinned.clear();
}

// what add() derives from one read before anything is appended: the signal slice, the flattened CIGAR maps, the strand-direction sequences
namespace {
struct Prepared {
    bool ok = false; size_t lo = 0, hi = 0; int refLen = 0;
    std::vector<uint32_t> r2q; std::vector<int32_t> q2r; std::vector<uint8_t> r2d; std::string bc, rs;
};
void prepareRead(const ReadInput &in, Prepared &P) {
    // ---- signal slice (pod5.cpp:75-93) ----
    size_t lo = 0, hi = in.n_adc;
    if (in.signalLength > 0) {
        if (in.isSplit) { lo = (size_t)(in.signalStartCoord + in.signalTrim); hi = (size_t)(in.signalStartCoord + in.signalLength); }
        else { lo = (size_t)in.signalTrim; hi = (size_t)in.signalLength; }
        hi = std::min(hi, in.n_adc); lo = std::min(lo, hi);
    }
    P.ok = false; P.lo = lo; P.hi = hi;
    if (hi - lo < 16 || in.querySeq.size() < DN_KMER + 1) return;
    P.refLen = parseCigar(in.cigarOp, in.cigarLen, in.isReverse, in.querySeq.size(), P.r2q, P.q2r, P.r2d);
    if (P.refLen < DN_KMER || (size_t)P.refLen != in.refSlice.size()) return;
    // ---- sequencing direction (reads.h:280-286) ----
    P.bc = in.isReverse ? reverseComplement(in.querySeq) : in.querySeq;
    P.rs = in.isReverse ? reverseComplement(in.refSlice) : in.refSlice;
    P.ok = true;
}
}  // namespace

int ReadBatch::add(const ReadInput &in) {
    const ReadInput *one = &in;
    uint8_t took = 0;
    addMany(&one, 1, &took);
    return took ? (int)readID.size() - 1 : -1;
}

// n reads at once: the per-read work of add() -- CIGAR flattening (htsInterface.cpp:59-157: three arrays of the read's reference length),
// reverse complements, the copies of a 1 MB signal -- runs on the host's threads; only the offset bookkeeping is serial.  A 500 x 50 kb batch
// read by read took 0.35-0.45 s of ONE thread, more than the GPU needs for it: the product driver's loader was what the GPU waited for
// (round 4: run_detect 497 Msamples/s against the bench's 720 on pre-built batches).  Returns the reads accepted; accepted[i] = 1 / 0.
size_t ReadBatch::addMany(const ReadInput *const *in, size_t n, uint8_t *accepted) { return addManyFromFile(in, n, accepted, -1, nullptr, nullptr); }

namespace {
bool preadAll(int fd, void *dst, size_t bytes, uint64_t off) {
    char *d = (char *)dst;
    while (bytes) {
        const ssize_t k = pread(fd, d, bytes, (off_t)off);
        if (k <= 0) return false;
        d += k; off += (uint64_t)k; bytes -= (size_t)k;
    }
    return true;
}
}  // namespace

size_t ReadBatch::addManyFromFile(const ReadInput *const *in, size_t n, uint8_t *accepted, int fd, const uint64_t *adcFileOff, bool *ioFailed) {
    std::vector<Prepared> P(n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(hostThreads()) if (n > 1)
    for (long i = 0; i < (long)n; i++) prepareRead(*in[i], P[(size_t)i]);
    // offsets of every accepted read in the flat arrays
    std::vector<size_t> o_adc(n), o_bc(n), o_rs(n), o_q2r(n), slot(n);
    size_t a = adc.size(), b = basecall.size(), r = refseq.size(), q = query2ref.size(), k = readID.size(), taken = 0;
    for (size_t i = 0; i < n; i++) {
        if (accepted) accepted[i] = P[i].ok ? 1 : 0;
        if (!P[i].ok) continue;
        o_adc[i] = a; o_bc[i] = b; o_rs[i] = r; o_q2r[i] = q; slot[i] = k + taken;
        a += P[i].hi - P[i].lo; b += P[i].bc.size(); r += P[i].rs.size(); q += P[i].q2r.size();
        adc_off.push_back(a); basecall_off.push_back(b); refseq_off.push_back(r);
        taken++;
    }
    if (!taken) return 0;
    readID.resize(k + taken); contig.resize(k + taken); cal_offset.resize(k + taken); cal_scale.resize(k + taken);
    ref_start.resize(k + taken); ref_end.resize(k + taken); is_reverse.resize(k + taken);
    adc.resize(a); basecall.resize(b); refseq.resize(r); ref2query.resize(r); ref2del.resize(r); query2ref.resize(q);     // ref2query / ref2del: one entry per reference base
    int bad_io = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(hostThreads()) reduction(| : bad_io) if (n > 1)
    for (long i = 0; i < (long)n; i++) {
        const Prepared &p = P[(size_t)i];
        if (!p.ok) continue;
        const ReadInput &x = *in[i];
        const size_t s_ = slot[(size_t)i];
        readID[s_] = x.readID; contig[s_] = x.contig; cal_offset[s_] = x.cal_offset; cal_scale[s_] = x.cal_scale;
        ref_start[s_] = x.refStart; ref_end[s_] = x.refStart + p.refLen; is_reverse[s_] = x.isReverse ? 1 : 0;
        if (fd >= 0) { if (!preadAll(fd, adc.data() + o_adc[(size_t)i], (p.hi - p.lo) * sizeof(int16_t), adcFileOff[i] + p.lo * sizeof(int16_t))) bad_io = 1; }
        else memcpy(adc.data() + o_adc[(size_t)i], x.adc + p.lo, (p.hi - p.lo) * sizeof(int16_t));
        memcpy(basecall.data() + o_bc[(size_t)i], p.bc.data(), p.bc.size());
        memcpy(refseq.data() + o_rs[(size_t)i], p.rs.data(), p.rs.size());
        memcpy(ref2query.data() + o_rs[(size_t)i], p.r2q.data(), p.r2q.size() * sizeof(uint32_t));
        memcpy(ref2del.data() + o_rs[(size_t)i], p.r2d.data(), p.r2d.size());
        memcpy(query2ref.data() + o_q2r[(size_t)i], p.q2r.data(), p.q2r.size() * sizeof(int32_t));               // queryLen + 1 entries
    }
    if (ioFailed) *ioFailed = bad_io != 0;
    return taken;
}

dn_batch_desc ReadBatch::desc() const {
    dn_batch_desc d;
    memset(&d, 0, sizeof d);
    d.n_reads = (uint32_t)readID.size();
    d.adc = adc.data(); d.adc_off = adc_off.data();
    d.cal_offset = cal_offset.data(); d.cal_scale = cal_scale.data();
    d.basecall = basecall.data(); d.basecall_off = basecall_off.data();
    d.refseq = refseq.data(); d.refseq_off = refseq_off.data();
    d.ref2query = ref2query.data(); d.query2ref = query2ref.data(); d.ref2del = ref2del.data();
    d.ref_start = ref_start.data(); d.ref_end = ref_end.data(); d.is_reverse = is_reverse.data();
    return d;
}

int normaliseEvents(dn_ctx *ctx, ReadBatch &batch) {
    if (batch.size() == 0) return DN_OK;
    const dn_batch_desc d = batch.desc();
    int rc = dn_batch_upload(ctx, &d);
    if (rc) return rc;
    if ((rc = dn_run_normalise(ctx))) return rc;
    batch.summary.resize(batch.size());
    return dn_get_summaries(ctx, batch.summary.data());
}

int eventalign(dn_ctx *ctx, ReadBatch &batch) {
    int rc = dn_run_eventalign(ctx);
    if (rc) return rc;
    batch.summary.resize(batch.size());
    return dn_get_summaries(ctx, batch.summary.data());
}

// ---- binary read container -----------------------------------------------------------------------------------------
namespace {
template <typename T> bool put(FILE *f, const T &v) { return fwrite(&v, sizeof(T), 1, f) == 1; }
template <typename T> bool get(FILE *f, T &v) { return fread(&v, sizeof(T), 1, f) == 1; }
bool putStr(FILE *f, const std::string &s) { const uint32_t n = (uint32_t)s.size(); return put(f, n) && (n == 0 || fwrite(s.data(), 1, n, f) == n); }
bool getStr(FILE *f, std::string &s, uint32_t limit) {
    uint32_t n; if (!get(f, n) || n > limit) return false;
    s.resize(n); return n == 0 || fread(&s[0], 1, n, f) == n;
}
}  // namespace

bool ReadContainerWriter::open(const std::string &path) {
    FILE *fp = fopen(path.c_str(), "wb"); if (!fp) return false;
    f = fp; n = 0;
    const uint32_t ver = 1; const uint64_t zero = 0;
    return fwrite("DNRC", 1, 4, fp) == 4 && put(fp, ver) && put(fp, zero);
}
bool ReadContainerWriter::add(const ReadInput &in) {
    FILE *fp = (FILE *)f; if (!fp || in.cigarOp.size() != in.cigarLen.size()) return false;
    const uint8_t split = in.isSplit, rev = in.isReverse; const int32_t sl = in.signalLength, st = in.signalTrim, sc = in.signalStartCoord, rs = in.refStart;
    const uint32_t nc = (uint32_t)in.cigarOp.size(); const uint64_t na = in.n_adc;
    bool ok = putStr(fp, in.readID) && putStr(fp, in.contig) && put(fp, in.cal_offset) && put(fp, in.cal_scale) && put(fp, sl) && put(fp, st) &&
              put(fp, sc) && put(fp, split) && put(fp, rev) && put(fp, rs) && putStr(fp, in.querySeq) && putStr(fp, in.refSlice) && put(fp, nc);
    ok = ok && (nc == 0 || (fwrite(in.cigarOp.data(), 4, nc, fp) == nc && fwrite(in.cigarLen.data(), 4, nc, fp) == nc));
    ok = ok && put(fp, na) && (na == 0 || fwrite(in.adc, 2, na, fp) == na);
    if (ok) n++;
    return ok;
}
bool ReadContainerWriter::close() {
    FILE *fp = (FILE *)f; if (!fp) return false;
    f = nullptr;
    const bool ok = fseek(fp, 8, SEEK_SET) == 0 && put(fp, n);
    return (fclose(fp) == 0) && ok;
}

bool ReadContainerReader::open(const std::string &path) {
    close();
    FILE *fp = fopen(path.c_str(), "rb"); if (!fp) return false;
    char magic[4]; uint32_t ver = 0;
    if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "DNRC", 4) != 0 || !get(fp, ver) || ver != 1 || !get(fp, n)) { fclose(fp); return false; }
    f = fp; seen = 0; bad = false;
    return true;
}
bool ReadContainerReader::skip(uint64_t *nSamples) {     // the next record's sample count; its payload is seeked over, not read
    FILE *fp = (FILE *)f; if (!fp || seen >= n) return false;
    auto skipStr = [&](uint32_t limit) { uint32_t k; return get(fp, k) && k <= limit && fseek(fp, (long)k, SEEK_CUR) == 0; };
    uint32_t nc = 0; uint64_t na = 0;
    bool ok = skipStr(1u << 16) && skipStr(1u << 16) && fseek(fp, 4 + 4 + 4 + 4 + 4 + 1 + 1 + 4, SEEK_CUR) == 0 && skipStr(1u << 30) && skipStr(1u << 30) &&
              get(fp, nc) && nc <= (1u << 28) && fseek(fp, (long)nc * 8, SEEK_CUR) == 0 && get(fp, na) && na <= (1ull << 33) &&
              fseek(fp, (long)(na * 2), SEEK_CUR) == 0;
    if (ok) { const int c = fgetc(fp); if (c == EOF) { if (seen + 1 < n) ok = false; } else ungetc(c, fp); }   // a truncated payload shows at the next read
    if (!ok) { bad = true; return false; }
    if (nSamples) *nSamples = na;
    seen++;
    return true;
}
bool ReadContainerReader::nextHeader(OwnedRead &o, uint64_t *adcFileOff) {
    FILE *fp = (FILE *)f; if (!fp || seen >= n) return false;
    ReadInput &in = o.in;
    uint8_t split = 0, rev = 0; int32_t sl = 0, st = 0, sc = 0, rs = 0; uint32_t nc = 0; uint64_t na = 0;
    bool ok = getStr(fp, in.readID, 1u << 16) && getStr(fp, in.contig, 1u << 16) && get(fp, in.cal_offset) && get(fp, in.cal_scale) && get(fp, sl) &&
              get(fp, st) && get(fp, sc) && get(fp, split) && get(fp, rev) && get(fp, rs) && getStr(fp, in.querySeq, 1u << 30) &&
              getStr(fp, in.refSlice, 1u << 30) && get(fp, nc) && nc <= (1u << 28);
    if (ok) { in.cigarOp.resize(nc); in.cigarLen.resize(nc); ok = nc == 0 || (fread(in.cigarOp.data(), 4, nc, fp) == nc && fread(in.cigarLen.data(), 4, nc, fp) == nc); }
    ok = ok && get(fp, na) && na <= (1ull << 33);
    if (!ok) { bad = true; return false; }
    in.signalLength = sl; in.signalTrim = st; in.signalStartCoord = sc; in.isSplit = split != 0; in.isReverse = rev != 0; in.refStart = rs;
    in.adc = nullptr; in.n_adc = (size_t)na;
    if (adcFileOff) {                                        // the caller fetches the samples itself (pread): step over them
        *adcFileOff = (uint64_t)ftello(fp);
        if (fseeko(fp, (off_t)(na * 2), SEEK_CUR) != 0) { bad = true; return false; }
        seen++;
    }
    return true;
}
bool ReadContainerReader::next(OwnedRead &o) {
    if (!nextHeader(o, nullptr)) return false;
    FILE *fp = (FILE *)f;
    const size_t na = o.in.n_adc;
    o.adc.resize(na);
    if (na && fread(o.adc.data(), 2, na, fp) != na) { bad = true; return false; }
    o.in.adc = o.adc.data();
    seen++;
    return true;
}
uint64_t ReadContainerReader::tell() const { return f ? (uint64_t)ftello((FILE *)f) : 0; }
bool ReadContainerReader::seek(uint64_t offset) {
    if (!f || fseeko((FILE *)f, (off_t)offset, SEEK_SET) != 0) { bad = true; return false; }
    seen = 0;                                                // the caller addresses records by offset from here on
    return true;
}
void ReadContainerReader::close() { if (f) fclose((FILE *)f); f = nullptr; }

// ---- output ----------------------------------------------------------------------------------------------------
// std::to_string(float) == printf("%f") of the value widened to double: 6 decimals, correctly rounded (glibc prints the exact
// binary value, ties to even).  For a probability the exact answer is cheap: p is a float (24 significant bits) and
// 10^6 = 2^6 * 15625 with 15625 < 2^14, so p * 1e6 is EXACT in double (<= 38 bits) and rint() -- nearest, ties to even on an
// exact value -- is the integer glibc prints.  Anything outside [0, 1] (or NaN) takes snprintf.  tests/test_output.py compares
// both on every float around the rounding boundaries.
// Round 6: two digits per table lookup and the rounding as ONE cvtsd2si (MXCSR's default mode is nearest, ties to even -- what rint() computes through a
// libm call on baseline x86-64): at N ranks per host a rank formats on its share of the cores (2 of 16 at N = 8), where the formatter's 86 ns per line were
// barely faster than one GPU produces lines (tests/test_format_budget.py).
static const char DIGIT_PAIRS[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
static inline char *put_prob(char *o, float pf) {
    const double p = (double)pf;
    if (!(p >= 0.0 && p <= 1.0) || signbit(p)) return o + snprintf(o, 48, "%f", p);
    const unsigned n = (unsigned)_mm_cvtsd_si32(_mm_set_sd(p * 1e6));
    if (n >= 1000000u) { memcpy(o, "1.000000", 8); return o + 8; }
    const unsigned a = n / 10000u, r = n - a * 10000u, b = r / 100u, c = r - b * 100u;
    o[0] = '0'; o[1] = '.';
    memcpy(o + 2, DIGIT_PAIRS + 2 * a, 2); memcpy(o + 4, DIGIT_PAIRS + 2 * b, 2); memcpy(o + 6, DIGIT_PAIRS + 2 * c, 2);
    return o + 8;
}
static inline size_t u32_len(uint32_t v);
static inline char *put_u32(char *o, uint32_t v) {
    const size_t len = u32_len(v);
    char *e = o + len;
    while (v >= 100u) { const uint32_t q = v / 100u; e -= 2; memcpy(e, DIGIT_PAIRS + 2 * (v - q * 100u), 2); v = q; }
    if (v >= 10u) memcpy(e - 2, DIGIT_PAIRS + 2 * v, 2); else e[-1] = (char)('0' + v);
    return o + len;
}

char *formatProbForTest(char *o, float p) { return put_prob(o, p); }

// one line of a record: "coord\tEdU\tBrdU\tkmer\n" (detect.cpp:716-721); km = the strand 9-mer, printed reverse-complemented for reverse reads (:699)
static inline char *put_call(char *o, uint32_t coord, float pEdU, float pBrdU, const char *km, bool isReverse, bool oriented = false) {
    o = put_u32(o, coord); *o++ = '\t';
    o = put_prob(o, pEdU); *o++ = '\t';
    o = put_prob(o, pBrdU); *o++ = '\t';
    if (isReverse && !oriented) {                                         // reverseComplement of the 9-mer (:699): A/C/G/T only reach here (T-centred, ACGT windows)
        for (int z = 0; z < 9; z++) {
            const char c = km[8 - z];
            o[z] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'G' ? 'C' : c == 'C' ? 'G' : c;
        }
    } else memcpy(o, km, 9);
    o[9] = '\n';
    return o + 10;
}
static std::string detectHeaderLine(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse) {
    return ">" + readID + " " + contig + " " + std::to_string(refStart) + " " + std::to_string(refEnd) + " " + (isReverse ? "rev" : "fwd") + "\n";
}

// the record of one read from its CALLS (positions whose strand 9-mer has 'T' in the middle, detect.cpp:690), creation order
static std::string formatDetectCalls(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                                     size_t n, const uint32_t *coord, const char *kmer9, const float *pEdU, const float *pBrdU) {
    std::string out = detectHeaderLine(readID, contig, refStart, refEnd, isReverse);
    const size_t head = out.size();
    out.resize(head + n * 128);                              // a line is at most 10 + 1 + 47 + 1 + 47 + 1 + 9 + 1 bytes
    char *o = &out[head];
    for (size_t q = 0; q < n; q++) {
        const size_t i = isReverse ? n - 1 - q : q;          // std::reverse of the line vector (:722)
        o = put_call(o, coord[i], pEdU[i], pBrdU[i], kmer9 + i * 9, isReverse);
    }
    out.resize((size_t)(o - out.data()));
    return out;
}

// ---- packed per-call results: what a rank sends to the writer rank instead of text (16 bytes per call against ~40 of text) ----
// per call {u32 coord, f32 P(EdU), f32 P(BrdU), u32 9-mer at 3 bits per base (A C G T N = 0 .. 4, first base in the low bits)}; per read a
// meta row {count, header bytes, flags} and in the payload its header line followed by the calls.  A read whose 9-mers hold anything
// else (IUPAC codes in a reference) travels as its formatted TEXT instead (flag DN_PACK_TEXT): the writer passes it through, so the file
// is the same bytes whatever the alphabet.
static inline int packBase(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : -1; }
void packCalls(const ReadBatch &batch, const dn_result_batch &res, std::vector<uint64_t> &meta, RawVec<uint8_t> &payload) {
    const long n = (long)batch.size();
    std::vector<std::string> hdr((size_t)n);
    std::vector<uint8_t> as_text((size_t)n, 0), ok((size_t)n, 0);
    std::vector<uint64_t> bytes((size_t)n, 0);
#pragma omp parallel for schedule(dynamic, 4) num_threads(hostThreads())
    for (long r = 0; r < n; r++) {
        if (res.summary[r].status != DN_READ_OK) continue;
        ok[(size_t)r] = 1;
        const uint64_t o = res.call_off[r], k = res.call_off[r + 1] - o;
        bool plain = true;
        for (uint64_t i = 0; i < 9 * k && plain; i++) plain = packBase(res.kmer9[9 * o + i]) >= 0;
        if (plain) {
            hdr[(size_t)r] = detectHeaderLine(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r], batch.is_reverse[r] != 0);
            bytes[(size_t)r] = hdr[(size_t)r].size() + 16 * k;
        } else {
            as_text[(size_t)r] = 1;
            hdr[(size_t)r] = formatDetectCalls(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r], batch.is_reverse[r] != 0, k,
                                               res.ref_coord + o, res.kmer9 + 9 * o, res.p_edu + o, res.p_brdu + o);
            bytes[(size_t)r] = hdr[(size_t)r].size();
        }
    }
    std::vector<uint64_t> off((size_t)n + 1, 0);
    size_t n_ok = 0;
    for (long r = 0; r < n; r++) { off[(size_t)r + 1] = off[(size_t)r] + bytes[(size_t)r]; n_ok += ok[(size_t)r]; }
    meta.assign(4 * n_ok, 0);
    payload.resize(off[(size_t)n]);
    size_t j = 0;
    for (long r = 0; r < n; r++) {
        if (!ok[(size_t)r]) continue;
        const uint64_t k = res.call_off[r + 1] - res.call_off[r];
        meta[4 * j] = (uint64_t)r;
        meta[4 * j + 1] = as_text[(size_t)r] ? hdr[(size_t)r].size() : k;
        meta[4 * j + 2] = as_text[(size_t)r] ? 0 : hdr[(size_t)r].size();
        meta[4 * j + 3] = (batch.is_reverse[r] ? DN_PACK_REVERSE : 0u) | (as_text[(size_t)r] ? DN_PACK_TEXT : 0u);
        j++;
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(hostThreads())
    for (long r = 0; r < n; r++) {
        if (!ok[(size_t)r]) continue;
        uint8_t *p = payload.data() + off[(size_t)r];
        memcpy(p, hdr[(size_t)r].data(), hdr[(size_t)r].size());
        if (as_text[(size_t)r]) continue;
        p += hdr[(size_t)r].size();
        const uint64_t o = res.call_off[r], k = res.call_off[r + 1] - o;
        for (uint64_t i = 0; i < k; i++) {
            uint32_t km = 0;
            for (int z = 0; z < 9; z++) km |= (uint32_t)packBase(res.kmer9[9 * (o + i) + z]) << (3 * z);
            uint32_t w[4];
            w[0] = res.ref_coord[o + i]; memcpy(&w[1], &res.p_edu[o + i], 4); memcpy(&w[2], &res.p_brdu[o + i], 4); w[3] = km;
            memcpy(p + 16 * i, w, 16);
        }
    }
}

// the writer rank's half: reads in the order given (pointers into the gathered payloads), formatted in parallel, laid end to end
// Two passes, nothing intermediate: the first sizes every record exactly (a line is digits(coord) + the two "%f" fields + 13 bytes), the second
// formats each record in its final place in ONE uninitialised buffer.  (Until round 4 every read went through a zero-filled 128-bytes-per-call
// string of its own and was copied once more into a zero-filled std::string of the window's 400 MB: three passes over memory the formatter did
// not need, on a host whose cgroup gives the process 16 CPUs.)  Returns false if a record did not come out at its computed size (cannot happen;
// checked, because the offsets of every later record depend on it).
static inline size_t prob_len(float pf) {
    const double p = (double)pf;
    if (!(p >= 0.0 && p <= 1.0) || signbit(p)) return (size_t)snprintf(nullptr, 0, "%f", p);
    return 8;
}
static inline size_t u32_len(uint32_t v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
// the first pass on its own: the exact text length of every packed read (what a rank announces before anybody formats: with the lengths of a window's
// reads every rank knows the file offset of each of its records -- round 5, the per-rank formatter of run_detect)
void packedSizes(size_t n, const uint64_t *meta3, const uint8_t *const *read_ptr, uint64_t *record_bytes /* [n] */) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(hostThreads())
    for (long r = 0; r < (long)n; r++) {
        const uint64_t cnt = meta3[3 * r], hb = meta3[3 * r + 1], fl = meta3[3 * r + 2];
        if (fl & DN_PACK_TEXT) { record_bytes[r] = cnt; continue; }
        const uint8_t *p = read_ptr[r] + hb;
        size_t len = (size_t)hb + (size_t)cnt * 13;
        for (uint64_t i = 0; i < cnt; i++) {
            uint32_t w[3]; float e, b;
            memcpy(w, p + 16 * i, 12); memcpy(&e, &w[1], 4); memcpy(&b, &w[2], 4);
            len += u32_len(w[0]) + prob_len(e) + prob_len(b);
        }
        record_bytes[r] = len;
    }
}
static const struct Km3Table {
    char t[512][4], rc[512][4];                             // three bases as packed; the same three reversed and complemented (detect.cpp:699 for reverse reads)
    Km3Table() {
        for (unsigned v = 0; v < 512; v++) {
            for (int z = 0; z < 3; z++) {
                const char c = "ACGTN???"[(v >> (3 * z)) & 7u];
                t[v][z] = c;
                rc[v][2 - z] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'G' ? 'C' : c == 'C' ? 'G' : c;
            }
            t[v][3] = rc[v][3] = 0;
        }
    }
} KM3;
bool formatPacked(size_t n, const uint64_t *meta3 /* [n][3]: count, header bytes, flags */, const uint8_t *const *read_ptr, RawVec<char> &text,
                  uint64_t *record_bytes /* [n] */) {
    std::vector<uint64_t> off(n + 1, 0);
    packedSizes(n, meta3, read_ptr, off.data() + 1);
    for (size_t r = 0; r < n; r++) { if (record_bytes) record_bytes[r] = off[r + 1]; off[r + 1] += off[r]; }
    text.resize(off[n] + 64);                                // + slack: put_prob's snprintf branch is given 48 bytes of room
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(hostThreads()) reduction(| : bad)
    for (long r = 0; r < (long)n; r++) {
        const uint64_t cnt = meta3[3 * r], hb = meta3[3 * r + 1], fl = meta3[3 * r + 2];
        const uint8_t *p = read_ptr[r];
        char *o = text.data() + off[(size_t)r];
        if (fl & DN_PACK_TEXT) { memcpy(o, p, (size_t)cnt); continue; }
        const bool rev = (fl & DN_PACK_REVERSE) != 0;
        memcpy(o, p, (size_t)hb);
        o += hb; p += hb;
        for (uint64_t q = 0; q < cnt; q++) {
            const uint64_t i = rev ? cnt - 1 - q : q;
            uint32_t w[4]; float e, b;
            memcpy(w, p + 16 * i, 16); memcpy(&e, &w[1], 4); memcpy(&b, &w[2], 4);
            char km[12];                                     // three bases per lookup (512 x 4 bytes, built once), already in the orientation the line prints
            if (rev) { memcpy(km, KM3.rc[(w[3] >> 18) & 511u], 4); memcpy(km + 3, KM3.rc[(w[3] >> 9) & 511u], 4); memcpy(km + 6, KM3.rc[w[3] & 511u], 4); }
            else { memcpy(km, KM3.t[w[3] & 511u], 4); memcpy(km + 3, KM3.t[(w[3] >> 9) & 511u], 4); memcpy(km + 6, KM3.t[(w[3] >> 18) & 511u], 4); }
            o = put_call(o, w[0], e, b, km, rev, true);
        }
        if ((uint64_t)(o - text.data()) != off[(size_t)r + 1]) bad = 1;
    }
    text.resize(off[n]);
    return !bad;
}

std::string formatDetectRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                               size_t n, const uint32_t *coord, const char *kmer9, const float *probs, uint32_t *nCalls) {
    std::vector<uint32_t> c; std::vector<char> km; std::vector<float> e, b;
    for (size_t i = 0; i < n; i++) {
        if (kmer9[i * 9 + 4] != 'T') continue;               // detect.cpp:690
        c.push_back(coord[i]); km.insert(km.end(), kmer9 + i * 9, kmer9 + i * 9 + 9);
        e.push_back(probs[i * 3 + 2]); b.push_back(probs[i * 3 + 1]);       // class 2 -> EdU, class 1 -> BrdU (:695)
    }
    if (nCalls) *nCalls = (uint32_t)c.size();
    return formatDetectCalls(readID, contig, refStart, refEnd, isReverse, c.size(), c.data(), km.data(), e.data(), b.data());
}

static void modBamCalls(size_t n, const uint32_t *queryIdx, const uint32_t *refIdx, const float *pEdU, const float *pBrdU,
                        const uint8_t *ref2del, std::string &MM, std::vector<uint8_t> &ML) {
    // queryIndexToCalls is a std::map<unsigned, pair<float, float>>: flat (key, position) pairs, stable sort by key, the LAST
    // entry of every key wins (operator[] assignment)
    std::vector<std::pair<uint32_t, uint32_t>> kv;
    for (size_t i = 0; i < n; i++) {
        if (ref2del[refIdx[i]]) continue;                    // detect.cpp:704
        kv.push_back({queryIdx[i], (uint32_t)i});
    }
    std::stable_sort(kv.begin(), kv.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first < b.first; });
    std::string fb = "N+b?", fe = "N+e?";
    std::vector<uint8_t> brdu, edu;
    unsigned prev = 0;
    for (size_t j = 0; j < kv.size(); j++) {
        if (j + 1 < kv.size() && kv[j + 1].first == kv[j].first) continue;
        const unsigned q = kv[j].first, i = kv[j].second;
        const std::string d = "," + std::to_string(q - prev);
        fb += d; fe += d;
        prev = q + 1;
        edu.push_back(static_cast<uint8_t>(pEdU[i] * 255.0));
        brdu.push_back(static_cast<uint8_t>(pBrdU[i] * 255.0));
    }
    MM = fb + ";" + fe + ";";
    ML = brdu;
    ML.insert(ML.end(), edu.begin(), edu.end());
}

void modBamFields(size_t n, const uint32_t *queryIdx, const uint32_t *refIdx, const char *kmer9, const float *probs,
                  const uint8_t *ref2del, std::string &MM, std::vector<uint8_t> &ML) {
    std::vector<uint32_t> q, r; std::vector<float> e, b;
    for (size_t i = 0; i < n; i++) {
        if (kmer9[i * 9 + 4] != 'T') continue;
        q.push_back(queryIdx[i]); r.push_back(refIdx[i]); e.push_back(probs[i * 3 + 2]); b.push_back(probs[i * 3 + 1]);
    }
    modBamCalls(q.size(), q.data(), r.data(), e.data(), b.data(), ref2del, MM, ML);
}

// the output half of runCNN for a collected batch: one record (or one MM / ML pair) per passing read, formatted in parallel
// (the reference formats inside its per-read OpenMP loop, detect.cpp:896; here the reads of a batch are spread over the host cores)
void formatCalls(const ReadBatch &batch, const dn_result_batch &res, bool humanReadable, std::vector<ReadCalls> &calls) {
    const long n = (long)batch.size();
    calls.assign((size_t)n, ReadCalls());
#pragma omp parallel for schedule(dynamic, 1) num_threads(hostThreads())
    for (long r = 0; r < n; r++) {
        if (res.summary[r].status != DN_READ_OK) continue;                 // detect.cpp:879-894: failed reads are counted, not written
        const uint64_t o = res.call_off[r], k = res.call_off[r + 1] - o;
        calls[r].nCalls = (uint32_t)k;
        if (humanReadable)
            calls[r].humanReadable_detectOut = formatDetectCalls(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r],
                                                                 batch.is_reverse[r] != 0, k, res.ref_coord + o, res.kmer9 + 9 * o, res.p_edu + o, res.p_brdu + o);
        else
            modBamCalls(k, res.query_idx + o, res.ref_idx + o, res.p_edu + o, res.p_brdu + o, batch.ref2del.data() + batch.refseq_off[r],
                        calls[r].MM, calls[r].ML);
    }
}

int runCNN(dn_ctx *ctx, ReadBatch &batch, bool humanReadable, std::vector<ReadCalls> &calls) {
    int rc = dn_run_cnn(ctx);
    if (rc) return rc;
    dn_result_batch res;
    if ((rc = dn_collect(ctx, &res))) return rc;            // ONE device-to-host transfer per output array for the whole batch
    if (res.n_reads != batch.size()) return DN_ERR_STATE;
    batch.summary.assign(res.summary, res.summary + res.n_reads);
    formatCalls(batch, res, humanReadable, calls);
    return DN_OK;
}

// ---- the buffer-of-reads loop of detect.cpp:821-907 with several batches in flight on one GPU, driven by ONE host thread ----
// Batch i runs on context i % n_ctx.  Nothing in the loop waits except dn_collect of the OLDEST batch in flight, and while the
// host formats and writes its records the GPU works on the n_ctx - 1 younger batches.  Records are written in input order.
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

DetectStream::DetectStream(dn_ctx **ctxs, int n_ctx, int emit_) : ctx(ctxs, ctxs + n_ctx), slot_batch((size_t)n_ctx, nullptr), slot_tag((size_t)n_ctx, 0),
                                                                 emit(emit_), t_open(now_s()) {}

int DetectStream::submit(ReadBatch *batch, uint64_t tag) {
    if (full() || !batch) return DN_ERR_STATE;
    const int slot = (head + inflight) % (int)ctx.size();
    const double a = now_s();
    const dn_batch_desc d = batch->desc();
    int rc = dn_batch_upload(ctx[(size_t)slot], &d);
    if (rc) return rc;
    const double b = now_s();
    S.seconds_upload += b - a;
    if ((rc = dn_run_detect(ctx[(size_t)slot]))) return rc;
    const double e = now_s();
    S.seconds_run += e - b;
    static const bool trace = [] { const char *t = getenv("DN_TRACE_SUBMIT"); return t && t[0] == '1'; }();     // when each batch went in and what its upload cost the host thread
    if (trace) fprintf(stderr, "submit tag %llu at %.3f s: upload %.1f ms, enqueue %.1f ms, %d in flight before it\n", (unsigned long long)tag, a - t_open, (b - a) * 1e3, (e - b) * 1e3, inflight);
    slot_batch[(size_t)slot] = batch; slot_tag[(size_t)slot] = tag;
    inflight++;
    return DN_OK;
}

int DetectStream::collect(Result &out) {
    if (inflight == 0) return DN_ERR_STATE;
    dn_ctx *c = ctx[(size_t)head];
    ReadBatch &B = *slot_batch[(size_t)head];
    out.tag = slot_tag[(size_t)head]; out.batch = &B;
    out.record_bytes.clear(); out.text.clear();
    const double a = now_s();
    int rc = dn_collect(c, &out.res);
    if (rc == DN_ERR_OVERFLOW && dn_ctx_get_event_bound(c) > 2) {
        // a read with more events than the tightened workspace bound holds (dn_ctx_set_event_bound): nothing was truncated silently -- the batch runs again on
        // this context with the detector's own bound (one peak per two samples: cannot overflow), and the context keeps it.  Rare by construction; the
        // regrown slab costs a device-wide wait once.
        S.overflow_retries++;
        const dn_batch_desc d = B.desc();
        if ((rc = dn_ctx_set_event_bound(c, 2)) || (rc = dn_batch_upload(c, &d)) || (rc = dn_run_detect(c))) return rc;
        rc = dn_collect(c, &out.res);
    }
    if (rc) return rc;
    const double b = now_s();
    const dn_result_batch &res = out.res;
    S.seconds_collect += b - a;
    S.reads += res.n_reads; S.calls += res.n_calls; S.samples += B.totalSamples();
    B.summary.assign(res.summary, res.summary + res.n_reads);
    for (uint32_t r = 0; r < res.n_reads; r++) if (res.summary[r].status == DN_READ_OK) { S.reads_ok++; S.positions += res.summary[r].n_positions; }
    out.record_bytes.assign(res.n_reads, 0);
    out.packed_meta.clear(); out.packed.clear();
    if (emit == EMIT_PACKED) {
        packCalls(B, res, out.packed_meta, out.packed);
        S.bytes_out += out.packed.size();
        S.seconds_emit += now_s() - b;
    } else if (emit == EMIT_TEXT) {
        formatCalls(B, res, true, calls);
        size_t total = 0;
        for (size_t r = 0; r < calls.size(); r++) if (res.summary[r].status == DN_READ_OK) total += calls[r].humanReadable_detectOut.size();
        out.text.reserve(total);
        for (size_t r = 0; r < calls.size(); r++) {
            if (res.summary[r].status != DN_READ_OK) continue;
            out.record_bytes[r] = calls[r].humanReadable_detectOut.size();
            out.text += calls[r].humanReadable_detectOut;
        }
        S.bytes_out += total;
        S.seconds_emit += now_s() - b;
    }
    head = (head + 1) % (int)ctx.size(); inflight--;
    S.seconds_total = now_s() - t_open;
    return DN_OK;
}

int streamDetect(dn_ctx **ctxs, int n_ctx, ReadBatch **batches, int n_batches, bool emit, const char *outPath, const char *header,
                 StreamStats *st, StreamKeep *keep) {
    FILE *f = nullptr;
    if (emit && outPath) { f = fopen(outPath, "wb"); if (!f) return DN_ERR_ARG; if (header) fwrite(header, 1, strlen(header), f); }
    const double t0 = now_s();
    DetectStream ds(ctxs, n_ctx, emit ? DetectStream::EMIT_TEXT : DetectStream::EMIT_NONE);
    DetectStream::Result R;
    int rc = DN_OK;
    auto drain_one = [&]() -> int {
        int e = ds.collect(R);
        if (e) return e;
        const dn_result_batch &res = R.res;
        if (keep) {
            for (uint32_t r = 0; r < res.n_reads; r++) keep->read_calls.push_back(res.call_off[r + 1] - res.call_off[r]);
            keep->record_bytes.insert(keep->record_bytes.end(), R.record_bytes.begin(), R.record_bytes.end());
            keep->coord.insert(keep->coord.end(), res.ref_coord, res.ref_coord + res.n_calls);
            keep->p_edu.insert(keep->p_edu.end(), res.p_edu, res.p_edu + res.n_calls);
            keep->p_brdu.insert(keep->p_brdu.end(), res.p_brdu, res.p_brdu + res.n_calls);
        }
        if (f && !R.text.empty()) fwrite(R.text.data(), 1, R.text.size(), f);
        return DN_OK;
    };
    for (int i = 0; i < n_batches && rc == DN_OK; i++) {
        if (ds.full()) rc = drain_one();
        if (rc == DN_OK) rc = ds.submit(batches[i], (uint64_t)i);
    }
    while (rc == DN_OK && ds.inFlight()) rc = drain_one();
    if (f) fclose(f);
    if (st) { *st = ds.stats(); st->seconds_total = now_s() - t0; }
    return rc;
}

static inline uint32_t kmer2index9(const char *k) {                     // data_IO.cpp:129-141: A0 T1 G2 C3, unknown -> 0
    uint32_t r = 0;
    for (int i = 0; i < 9; i++) { const char c = k[i]; r = r * 4u + (c == 'T' ? 1u : c == 'G' ? 2u : c == 'C' ? 3u : 0u); }
    return r;
}

std::string formatAlignRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                              const std::string &refseq, const double *poreModelMean, size_t nRows, const uint32_t *coord,
                              const uint32_t *refPos, const double *value, const uint8_t *kind) {
    std::string out = ">" + readID + " " + contig + " " + std::to_string(refStart) + " " + std::to_string(refEnd) + " " +
                      (isReverse ? "rev" : "fwd") + "\n";                                              // alignment.cpp:553
    out.reserve(out.size() + nRows * 48);
    char num[64];
    for (size_t i = 0; i < nRows; i++) {
        const std::string kmerStrand = refseq.substr(refPos[i], 9);                                     // :684
        const std::string kmerRef = isReverse ? reverseComplement(kmerStrand) : kmerStrand;             // :688-695
        int len = snprintf(num, sizeof num, "%u\t", coord[i]);
        out.append(num, (size_t)len); out += kmerRef;
        len = snprintf(num, sizeof num, "\t%f\t", value[i]);                                           // std::to_string(double)
        out.append(num, (size_t)len);
        if (kind[i] == 0) {
            out += kmerStrand;
            len = snprintf(num, sizeof num, "\t%f\n", poreModelMean[kmer2index9(kmerStrand.c_str())]);  // :701, :720
            out.append(num, (size_t)len);
        } else out += "NNNNNNNNN\t0\n";                                                                // :731
    }
    return out;
}

int alignWrite(dn_ctx *ctx, ReadBatch &batch, const double *poreModelMean, const std::string &path) {
    int rc = dn_set_align_table(ctx, 1);
    if (rc) return rc;
    rc = dn_run_eventalign(ctx);
    dn_set_align_table(ctx, 0);
    if (rc) return rc;
    const size_t n = batch.size();
    batch.summary.resize(n);
    if ((rc = dn_get_summaries(ctx, batch.summary.data()))) return rc;
    std::vector<uint32_t> rows(n);
    if (n && (rc = dn_get_align_rows(ctx, rows.data()))) return rc;
    FILE *f = fopen(path.c_str(), "ab");
    if (!f) return DN_ERR_ARG;
    int written = 0;
    std::vector<uint32_t> coord, rpos; std::vector<double> val; std::vector<uint8_t> kind;
    for (size_t r = 0; r < n; r++) {
        if (batch.summary[r].status != DN_READ_OK) continue;                                            // alignment.cpp:861-873
        const size_t k = rows[r];
        coord.resize(k); rpos.resize(k); val.resize(k); kind.resize(k);
        if ((rc = dn_get_align_table(ctx, (uint32_t)r, (uint32_t)k, coord.data(), rpos.data(), val.data(), kind.data()))) { fclose(f); return rc; }
        const std::string ref(batch.refseq.data() + batch.refseq_off[r], (size_t)(batch.refseq_off[r + 1] - batch.refseq_off[r]));
        const std::string rec = formatAlignRecord(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r], batch.is_reverse[r] != 0,
                                                  ref, poreModelMean, k, coord.data(), rpos.data(), val.data(), kind.data());
        fwrite(rec.data(), 1, rec.size(), f);                                                           // :876
        written++;
    }
    fclose(f);
    return written;
}

std::string formatHmmRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                            const std::string &basecall, const std::string &refseq, size_t n, const uint32_t *posOnRef,
                            const uint32_t *posOnQuery, const int32_t *globalPos, const double *llr) {
    std::string out = ">" + readID + " " + contig + " " + std::to_string(refStart) + " " + std::to_string(refEnd) + " " +
                      (isReverse ? "rev" : "fwd") + "\n";
    auto kmerAt = [&](const std::string &s, uint32_t pos) {            // s.substr(pos - k/2, k) (:534-536); out-of-range bases print N
        std::string k(9, 'N');
        for (int z = 0; z < 9; z++) { const long i = (long)pos - 4 + z; if (i >= 0 && (size_t)i < s.size()) k[z] = s[(size_t)i]; }
        return isReverse ? reverseComplement(k) : k;                    // :540-541
    };
    char num[64];
    for (size_t i = 0; i < n; i++) {
        const int len = snprintf(num, sizeof num, "%d\t%f\t", globalPos[i], llr[i]);                  // std::to_string: "%d", "%f"
        out.append(num, (size_t)len);
        out += kmerAt(refseq, posOnRef[i]); out += '\t'; out += kmerAt(basecall, posOnQuery[i]); out += '\n';
    }
    return out;
}

int llAcrossRead(dn_ctx *ctx, ReadBatch &batch, std::vector<ReadCalls> &calls) {
    int rc = dn_run_hmm(ctx);
    if (rc) return rc;
    const size_t n = batch.size();
    batch.summary.resize(n);
    if ((rc = dn_get_summaries(ctx, batch.summary.data()))) return rc;
    calls.assign(n, ReadCalls());
    std::vector<uint32_t> pr, pq; std::vector<int32_t> gp; std::vector<double> llr;
    for (size_t r = 0; r < n; r++) {
        const dn_read_summary &s = batch.summary[r];
        if (s.status != DN_READ_OK) continue;                            // detect.cpp:879-883
        const size_t k = s.n_hmm_calls;
        pr.resize(k); pq.resize(k); gp.resize(k); llr.resize(k);
        if ((rc = dn_get_hmm_calls(ctx, (uint32_t)r, k, pr.data(), pq.data(), gp.data(), nullptr, nullptr, nullptr, llr.data()))) return rc;
        const std::string bc(batch.basecall.data() + batch.basecall_off[r], batch.basecall.data() + batch.basecall_off[r + 1]);
        const std::string rf(batch.refseq.data() + batch.refseq_off[r], batch.refseq.data() + batch.refseq_off[r + 1]);
        calls[r].humanReadable_detectOut = formatHmmRecord(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r],
                                                           batch.is_reverse[r] != 0, bc, rf, k, pr.data(), pq.data(), gp.data(), llr.data());
        calls[r].nCalls = (uint32_t)k;
    }
    return DN_OK;
}

std::string writeDetectHeader(const std::string &alignmentFilename, const std::string &refFilename, const std::string &indexFn,
                              int threads, unsigned quality, unsigned length, bool useGPU, const std::string &startTime,
                              const std::string &software, const std::string &version, const std::string &commit) {
    std::string out;
    out += "#Alignment " + alignmentFilename + "\n";
    out += "#Genome " + refFilename + "\n";
    out += "#Index " + indexFn + "\n";
    out += "#Threads " + std::to_string(threads) + "\n";
    out += std::string("#Compute ") + (useGPU ? "GPU" : "CPU") + "\n";
    out += "#Mode CNN\n";
    out += "#MappingQuality " + std::to_string(quality) + "\n";
    out += "#MappingLength " + std::to_string(length) + "\n";
    out += "#SystemStartTime " + startTime + "\n";
    out += "#Software " + software + "\n";
    out += "#Version " + version + "\n";
    out += "#Commit " + commit + "\n";
    return out;
}

bool HumanReadableWriter::open(const std::string &filename) { close(); file = fopen(filename.c_str(), "w"); return file != nullptr; }
void HumanReadableWriter::writeHeader_HR(const std::string &header) { if (file) fwrite(header.data(), 1, header.size(), (FILE *)file); }
void HumanReadableWriter::write(const ReadCalls &r) { if (file) fwrite(r.humanReadable_detectOut.data(), 1, r.humanReadable_detectOut.size(), (FILE *)file); }
void HumanReadableWriter::close() { if (file) { fclose((FILE *)file); file = nullptr; } }

}  // namespace DNAscent

// ------------------------------------------------------------------------------------------------
// flat C wrappers for the Python test / bench harness (ctypes)
// ------------------------------------------------------------------------------------------------
using DNAscent::ReadBatch;
using DNAscent::ReadInput;

extern "C" {

void *dnh_batch_new(void) { return new ReadBatch(); }
void dnh_batch_free(void *b) { delete (ReadBatch *)b; }
void dnh_batch_clear(void *b) { ((ReadBatch *)b)->clear(); }
uint32_t dnh_batch_size(void *b) { return (uint32_t)((ReadBatch *)b)->size(); }
uint64_t dnh_batch_samples(void *b) { return ((ReadBatch *)b)->totalSamples(); }

// sequences are given in BAM / FASTA (reference-forward) orientation, exactly what the reference reads from disk
int dnh_batch_add(void *b, const char *read_id, const char *contig, const int16_t *adc, uint64_t n_adc, float cal_offset,
                  float cal_scale, int signal_length, int signal_trim, int signal_start, int is_split, const char *query_seq,
                  uint32_t n_query, const char *ref_slice, uint32_t n_ref, const uint32_t *cigar_op, const uint32_t *cigar_len,
                  uint32_t n_cigar, int ref_start, int is_reverse) {
    ReadInput in;
    in.readID = read_id; in.contig = contig;
    in.adc = adc; in.n_adc = (size_t)n_adc; in.cal_offset = cal_offset; in.cal_scale = cal_scale;
    in.signalLength = signal_length; in.signalTrim = signal_trim; in.signalStartCoord = signal_start; in.isSplit = is_split != 0;
    in.querySeq.assign(query_seq, n_query); in.refSlice.assign(ref_slice, n_ref);
    in.cigarOp.assign(cigar_op, cigar_op + n_cigar); in.cigarLen.assign(cigar_len, cigar_len + n_cigar);
    in.refStart = ref_start; in.isReverse = is_reverse != 0;
    return ((ReadBatch *)b)->add(in);
}

// ---- binary read container: writer handle, and "load reads [first, first + count) of a container into a batch" ----
void *dnh_container_create(const char *path) {
    DNAscent::ReadContainerWriter *w = new DNAscent::ReadContainerWriter();
    if (!w->open(path)) { delete w; return nullptr; }
    return w;
}
int dnh_container_add(void *w, const char *read_id, const char *contig, const int16_t *adc, uint64_t n_adc, float cal_offset, float cal_scale,
                      int signal_length, int signal_trim, int signal_start, int is_split, const char *query_seq, uint32_t n_query,
                      const char *ref_slice, uint32_t n_ref, const uint32_t *cigar_op, const uint32_t *cigar_len, uint32_t n_cigar, int ref_start,
                      int is_reverse) {
    ReadInput in;
    in.readID = read_id; in.contig = contig;
    in.adc = adc; in.n_adc = (size_t)n_adc; in.cal_offset = cal_offset; in.cal_scale = cal_scale;
    in.signalLength = signal_length; in.signalTrim = signal_trim; in.signalStartCoord = signal_start; in.isSplit = is_split != 0;
    in.querySeq.assign(query_seq, n_query); in.refSlice.assign(ref_slice, n_ref);
    in.cigarOp.assign(cigar_op, cigar_op + n_cigar); in.cigarLen.assign(cigar_len, cigar_len + n_cigar);
    in.refStart = ref_start; in.isReverse = is_reverse != 0;
    return ((DNAscent::ReadContainerWriter *)w)->add(in) ? 0 : -1;
}
int dnh_container_close(void *w) {
    DNAscent::ReadContainerWriter *W = (DNAscent::ReadContainerWriter *)w;
    const bool ok = W->close();
    delete W;
    return ok ? 0 : -1;
}
// returns the number of reads in the file (-1: cannot be opened / not a container)
int64_t dnh_container_count(const char *path) {
    DNAscent::ReadContainerReader r;
    return r.open(path) ? (int64_t)r.count() : -1;
}
// adds reads [first, first + count) to the batch; returns how many the batch accepted, or -1 on a malformed file
int64_t dnh_container_load(void *b, const char *path, uint64_t first, uint64_t count) {
    DNAscent::ReadContainerReader r;
    if (!r.open(path)) return -1;
    DNAscent::OwnedRead o;
    int64_t accepted = 0;
    for (uint64_t i = 0; i < first + count; i++) {
        if (!r.next(o)) { if (r.failed()) return -1; break; }
        if (i >= first && ((ReadBatch *)b)->add(o.in) >= 0) accepted++;
    }
    return accepted;
}

// stored sample count of every read of a container (its records are seeked over: cheap); returns the read count or -1
int64_t dnh_container_sizes(const char *path, uint64_t *out, uint64_t cap) {
    DNAscent::ReadContainerReader r;
    if (!r.open(path)) return -1;
    uint64_t i = 0, ns = 0;
    while (r.skip(&ns)) { if (i < cap) out[i] = ns; i++; }
    return r.failed() ? -1 : (int64_t)i;
}
// adds the reads with the given ASCENDING ordinals to the batch (the others are seeked over); returns how many were accepted or -1
int64_t dnh_container_load_list(void *b, const char *path, const uint64_t *ordinals, uint64_t n_ord) {
    DNAscent::ReadContainerReader r;
    if (!r.open(path)) return -1;
    DNAscent::OwnedRead o;
    int64_t accepted = 0;
    uint64_t at = 0;
    for (uint64_t j = 0; j < n_ord; j++) {
        if (ordinals[j] < at || ordinals[j] >= r.count()) return -1;
        while (at < ordinals[j]) { if (!r.skip(nullptr)) return -1; at++; }
        if (!r.next(o)) return -1;
        at++;
        if (((ReadBatch *)b)->add(o.in) >= 0) accepted++;
    }
    return accepted;
}

// index of a container: sample count and file offset of every record (one pass of seeks); returns the read count or -1
int64_t dnh_container_index(const char *path, uint64_t *sizes, uint64_t *offsets, uint64_t cap) {
    DNAscent::ReadContainerReader r;
    if (!r.open(path)) return -1;
    uint64_t i = 0, ns = 0;
    for (;;) {
        const uint64_t at = r.tell();
        if (!r.skip(&ns)) break;
        if (i < cap) { if (sizes) sizes[i] = ns; if (offsets) offsets[i] = at; }
        i++;
    }
    return r.failed() ? -1 : (int64_t)i;
}
// adds the records at the given file offsets (dnh_container_index) to the batch, in the given order: the records are read by all
// host cores (one FILE per thread), then packed in order.  accepted[j] = 1 if the batch took record j, 0 if it was REJECTED by the
// reference's own filters (ReadBatch::add: empty / too short signal, CIGAR that does not span the reference slice: detect.cpp:839,
// pod5.cpp:64) -- such a read counts as failed, it is not an error.  Returns the number accepted, or -1 on an I/O error / a
// truncated record (nothing usable: the batch is left as it was).
int64_t dnh_container_load_at(void *b, const char *path, const uint64_t *offsets, uint64_t n, uint8_t *accepted) {
    ReadBatch *B = (ReadBatch *)b;
    int64_t got = 0;
    bool io_bad = false;
    const uint64_t chunk = 1024;
    const size_t first = B->size();
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { if (fd >= 0) ::close(fd); return -1; }
    for (uint64_t c0 = 0; c0 < n && !io_bad; c0 += chunk) {
        const uint64_t m = std::min(chunk, n - c0);
        // the records' headers (ids, sequences, CIGAR: ~0.1 MB per 50 kb read) by all host cores, one FILE per thread; the samples stay in the file ...
        std::vector<DNAscent::OwnedRead> rd((size_t)m);
        std::vector<uint64_t> adc_at((size_t)m, 0);
        std::vector<uint8_t> ok((size_t)m, 0);
#pragma omp parallel num_threads(DNAscent::hostThreads())
        {
            DNAscent::ReadContainerReader r;
            const bool open_ok = r.open(path);
#pragma omp for schedule(dynamic, 4)
            for (long j = 0; j < (long)m; j++)
                ok[(size_t)j] = open_ok && r.seek(offsets[c0 + (uint64_t)j]) && r.nextHeader(rd[(size_t)j], &adc_at[(size_t)j]) &&
                                adc_at[(size_t)j] + rd[(size_t)j].in.n_adc * 2 <= (uint64_t)sb.st_size;          // a truncated payload is an I/O error, found before anything is appended
        }
        for (uint64_t j = 0; j < m; j++) if (!ok[(size_t)j]) { io_bad = true; break; }
        if (io_bad) break;
        // ... and go from the page cache straight to their place in the batch (addManyFromFile): a 500 x 50 kb batch was 0.23 s with an intermediate
        // 1 MB vector per read (fresh pages + a second copy), and three loaders at once competed with the engine's own host threads (round 4)
        std::vector<const ReadInput *> ptr((size_t)m);
        std::vector<uint8_t> took((size_t)m, 0);
        for (uint64_t j = 0; j < m; j++) ptr[(size_t)j] = &rd[(size_t)j].in;
        bool pread_bad = false;
        got += (int64_t)B->addManyFromFile(ptr.data(), (size_t)m, took.data(), fd, adc_at.data(), &pread_bad);
        if (pread_bad) { ::close(fd); B->clear(); return -1; }   // the file changed under us / EIO: nothing of this batch can be trusted
        if (accepted) memcpy(accepted + c0, took.data(), (size_t)m);
    }
    ::close(fd);
    if (io_bad) {                                            // all or nothing: a half-loaded batch would shift every later ordinal
        if (first == 0) B->clear();
        return -1;
    }
    return got;
}

// ---- DNAscent::DetectStream behind C signatures ----
void *dnh_stream_open(void **ctxs, int n_ctx, int emit /* 0 none, 1 text records, 2 packed calls */) { return new DNAscent::DetectStream((dn_ctx **)ctxs, n_ctx, emit); }
void dnh_stream_close(void *s) { delete (DNAscent::DetectStream *)s; }
int dnh_stream_full(void *s) { return ((DNAscent::DetectStream *)s)->full() ? 1 : 0; }
int dnh_stream_inflight(void *s) { return ((DNAscent::DetectStream *)s)->inFlight(); }
int dnh_stream_submit(void *s, void *batch, uint64_t tag) { return ((DNAscent::DetectStream *)s)->submit((ReadBatch *)batch, tag); }
void *dnh_result_new(void) { return new DNAscent::DetectStream::Result(); }
void dnh_result_free(void *r) { delete (DNAscent::DetectStream::Result *)r; }
// waits for the oldest batch in flight; *tag its tag, *n_reads its reads; record_bytes [n_reads] / text: owned by the result object,
// valid until its next use; status [n_reads] (DN_READ_*), calls: dn_result_batch arrays (valid until the context's next upload)
int dnh_stream_collect(void *s, void *result, uint64_t *tag, uint32_t *n_reads, const uint64_t **record_bytes, const char **text, uint64_t *text_bytes,
                       dn_result_batch *res) {
    DNAscent::DetectStream::Result &R = *(DNAscent::DetectStream::Result *)result;
    const int rc = ((DNAscent::DetectStream *)s)->collect(R);
    if (rc) return rc;
    if (tag) *tag = R.tag;
    if (n_reads) *n_reads = R.res.n_reads;
    if (record_bytes) *record_bytes = R.record_bytes.data();
    if (text) *text = R.text.data();
    if (text_bytes) *text_bytes = R.text.size();
    if (res) *res = R.res;
    return DN_OK;
}
// the packed form of the last collected batch (stream opened with emit == 2): meta [n_packed][4], payload bytes; owned by the result object
uint64_t dnh_result_packed(void *result, const uint64_t **meta, const uint8_t **payload, uint64_t *payload_bytes) {
    DNAscent::DetectStream::Result &R = *(DNAscent::DetectStream::Result *)result;
    if (meta) *meta = R.packed_meta.data();
    if (payload) *payload = R.packed.data();
    if (payload_bytes) *payload_bytes = R.packed.size();
    return R.packed_meta.size() / 4;
}
// packCalls for a batch and a result the caller supplies (CPU tests: no stream, no device); read back with dnh_result_packed
uint64_t dnh_pack_calls(void *batch, const dn_result_batch *res, void *result) {
    DNAscent::DetectStream::Result &R = *(DNAscent::DetectStream::Result *)result;
    DNAscent::packCalls(*(ReadBatch *)batch, *res, R.packed_meta, R.packed);
    return R.packed_meta.size() / 4;
}
// the writer rank's formatter: n reads in output order, read_ptr[i] = address of read i's payload; returns a text handle
// The formatter's text buffers are recycled (two at most): a window's 200-400 MB fresh from malloc are fresh from mmap, i.e. ~50-100 k page faults with the
// kernel zeroing every page -- a fifth of the formatter's time at two threads (round 6).  dnh_text_free hands a buffer back; its capacity stays.
static std::mutex text_pool_mu;
static std::vector<DNAscent::RawVec<char> *> text_pool;
void *dnh_format_packed(uint64_t n, const uint64_t *meta3, const uint64_t *read_ptr, uint64_t *record_bytes) {
    DNAscent::RawVec<char> *t = nullptr;
    {
        std::lock_guard<std::mutex> g(text_pool_mu);
        if (!text_pool.empty()) { t = text_pool.back(); text_pool.pop_back(); }
    }
    if (!t) t = new DNAscent::RawVec<char>();
    static_assert(sizeof(uint64_t) == sizeof(const uint8_t *), "64-bit host");
    if (!DNAscent::formatPacked((size_t)n, meta3, (const uint8_t *const *)read_ptr, *t, record_bytes)) { delete t; return nullptr; }
    return t;
}
// n bytes to file descriptor fd at offset off, in pieces written by the host's threads at once (pwrite): a single write() of a window's 400 MB of text
// runs at ~2 GB/s into the page cache, and the writer rank of an 8-GPU run has ~4 GB/s of .detect text to put down.  Returns 0, or -1 on a short / failed write.
int dnh_pwrite_parallel(int fd, const void *buf, uint64_t n, uint64_t off) {
    const uint64_t piece = 8ull << 20;
    const long pieces = (long)((n + piece - 1) / piece);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(16, DNAscent::hostThreads())) reduction(| : bad)
    for (long i = 0; i < pieces; i++) {
        uint64_t a = (uint64_t)i * piece, left = std::min(piece, n - a);
        const char *p = (const char *)buf + a;
        while (left) {
            const ssize_t w = pwrite(fd, p, (size_t)left, (off_t)(off + a));
            if (w <= 0) { bad = 1; break; }
            p += w; a += (uint64_t)w; left -= (uint64_t)w;
        }
    }
    return bad ? -1 : 0;
}
// the exact text length of n packed reads (DNAscent::packedSizes)
void dnh_packed_sizes(uint64_t n, const uint64_t *meta3, const uint64_t *read_ptr, uint64_t *record_bytes) {
    DNAscent::packedSizes((size_t)n, meta3, (const uint8_t *const *)read_ptr, record_bytes);
}
// n pieces of one buffer to their own places in the file: piece i = len[i] bytes at buf + src_off[i] -> file offset file_off[i] (pwrite, the host's threads at
// once, pieces of more than 8 MB cut up).  What every rank of run_detect does with the records it formatted itself: they interleave with the other ranks'
// records in input order, so a rank's text is contiguous in memory but not in the file.  Returns 0, or -1 on a short / failed write.
int dnh_pwrite_scatter(int fd, const void *buf, uint64_t n, const uint64_t *src_off, const uint64_t *len, const uint64_t *file_off) {
    const uint64_t piece = 8ull << 20;
    struct Job { uint64_t src, dst, len; };
    std::vector<Job> jobs;
    for (uint64_t i = 0; i < n; i++) {
        // neighbours in memory that are neighbours in the file too go out as one piece
        if (!jobs.empty() && jobs.back().src + jobs.back().len == src_off[i] && jobs.back().dst + jobs.back().len == file_off[i] && jobs.back().len + len[i] <= piece) {
            jobs.back().len += len[i];
            continue;
        }
        for (uint64_t a = 0; a < len[i]; a += piece) jobs.push_back({src_off[i] + a, file_off[i] + a, std::min(piece, len[i] - a)});
    }
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(16, DNAscent::hostThreads())) reduction(| : bad)
    for (long j = 0; j < (long)jobs.size(); j++) {
        const char *p = (const char *)buf + jobs[(size_t)j].src;
        uint64_t left = jobs[(size_t)j].len, at = jobs[(size_t)j].dst;
        while (left) {
            const ssize_t w = pwrite(fd, p, (size_t)left, (off_t)at);
            if (w <= 0) { bad = 1; break; }
            p += w; at += (uint64_t)w; left -= (uint64_t)w;
        }
    }
    return bad ? -1 : 0;
}
const char *dnh_text_data(void *t) { return ((DNAscent::RawVec<char> *)t)->data(); }
uint64_t dnh_text_size(void *t) { return ((DNAscent::RawVec<char> *)t)->size(); }
void dnh_text_free(void *t) {
    DNAscent::RawVec<char> *v = (DNAscent::RawVec<char> *)t;
    if (!v) return;
    {
        std::lock_guard<std::mutex> g(text_pool_mu);
        if (text_pool.size() < 2) { v->clear(); text_pool.push_back(v); return; }
    }
    delete v;
}
void dnh_stream_stats(void *s, DNAscent::StreamStats *st) { *st = ((DNAscent::DetectStream *)s)->stats(); }
int dnh_host_threads(void) { return DNAscent::hostThreads(); }                    // what the parallel loops of this library use (cgroup quota applied)
int dnh_batch_pin(void *b) { return ((ReadBatch *)b)->pin(); }
void dnh_batch_unpin(void *b) { ((ReadBatch *)b)->unpin(); }

void dnh_batch_desc(void *b, dn_batch_desc *out) { *out = ((ReadBatch *)b)->desc(); }

// flattened CIGAR maps of read i (for host-logic tests)
int dnh_batch_maps(void *b, uint32_t i, uint32_t *ref2query, int32_t *query2ref, uint8_t *ref2del) {
    ReadBatch *B = (ReadBatch *)b;
    if (i >= B->size()) return -1;
    const uint64_t f0 = B->refseq_off[i], f1 = B->refseq_off[i + 1], q0 = B->basecall_off[i] + i, q1 = B->basecall_off[i + 1] + i + 1;
    if (ref2query) memcpy(ref2query, B->ref2query.data() + f0, (f1 - f0) * 4);
    if (ref2del) memcpy(ref2del, B->ref2del.data() + f0, (f1 - f0));
    if (query2ref) memcpy(query2ref, B->query2ref.data() + q0, (q1 - q0) * 4);
    return 0;
}

// formatDetectRecord into a caller buffer; returns the record length (the required size if > cap)
uint64_t dnh_format_detect(const char *read_id, const char *contig, int ref_start, int ref_end, int is_reverse, uint32_t n,
                           const uint32_t *coord, const char *kmer9, const float *probs, char *buf, uint64_t cap) {
    const std::string s = DNAscent::formatDetectRecord(read_id, contig, ref_start, ref_end, is_reverse != 0, n, coord, kmer9, probs);
    if (s.size() <= cap) memcpy(buf, s.data(), s.size());
    return s.size();
}

uint64_t dnh_format_align(const char *read_id, const char *contig, int ref_start, int ref_end, int is_reverse, const char *refseq,
                          uint32_t n_ref, const double *model_mean, uint32_t n_rows, const uint32_t *coord, const uint32_t *ref_pos,
                          const double *value, const uint8_t *kind, char *buf, uint64_t cap) {
    const std::string s = DNAscent::formatAlignRecord(read_id, contig, ref_start, ref_end, is_reverse != 0, std::string(refseq, n_ref), model_mean,
                                                      n_rows, coord, ref_pos, value, kind);
    if (s.size() <= cap) memcpy(buf, s.data(), s.size());
    return s.size();
}

int dnh_align_write(void *ctx, void *b, const double *model_mean, const char *path) {
    return DNAscent::alignWrite((dn_ctx *)ctx, *(ReadBatch *)b, model_mean, path);
}

// modBamFields; returns the number of calls, MM text into mm (NUL-terminated if it fits), ML bytes into ml (2 * calls)
int dnh_modbam(uint32_t n, const uint32_t *query_idx, const uint32_t *ref_idx, const char *kmer9, const float *probs,
               const uint8_t *ref2del, char *mm, uint64_t mm_cap, uint8_t *ml, uint64_t ml_cap) {
    std::string MM; std::vector<uint8_t> ML;
    DNAscent::modBamFields(n, query_idx, ref_idx, kmer9, probs, ref2del, MM, ML);
    if (MM.size() + 1 <= mm_cap) memcpy(mm, MM.c_str(), MM.size() + 1);
    if (ML.size() <= ml_cap && !ML.empty()) memcpy(ml, ML.data(), ML.size());
    return (int)(ML.size() / 2);
}

// whole output step for an uploaded + aligned batch: runCNN for every read, records appended to `path` (header first if
// header != NULL).  Returns the number of reads written or a negative DN_* code.
int dnh_detect_write(void *ctx, void *b, const char *path, const char *header) {
    ReadBatch *B = (ReadBatch *)b;
    B->summary.resize(B->size());
    int rc = dn_get_summaries((dn_ctx *)ctx, B->summary.data());
    if (rc) return rc;
    std::vector<DNAscent::ReadCalls> calls;
    if ((rc = DNAscent::runCNN((dn_ctx *)ctx, *B, true, calls))) return rc;
    DNAscent::HumanReadableWriter w;
    if (!w.open(path)) return DN_ERR_ARG;
    if (header) w.writeHeader_HR(header);
    int written = 0;
    for (size_t i = 0; i < calls.size(); i++)
        if (B->summary[i].status == DN_READ_OK) { w.write(calls[i]); written++; }
    w.close();
    return written;
}

// n_reads synthetic reads (dn_synth.c: seeds seed0 .. seed0 + n - 1, every odd one on the reverse strand) generated on all host
// cores and added to the batch in seed order -- the bench's 10 000 x 50 kb stream would take minutes read by read from Python.
// Returns the number of reads the batch accepted.
}  // extern "C"
template <typename Sink>
static int synthReads(Sink &&sink, const double *model_mean, uint32_t n_reads, const uint64_t *seeds, const uint32_t *bases, const uint8_t *rev, double noise_pa,
                      double sub_rate, double ins_rate, double del_rate) {
    struct Gen { std::vector<char> ref, bc; std::vector<uint32_t> op, len; std::vector<int16_t> adc; dns_read_out o; int rc; };
    int accepted = 0;
    const uint32_t chunk = 64;                              // reads generated at a time (bounds the temporary memory)
    for (uint32_t c0 = 0; c0 < n_reads; c0 += chunk) {
        const uint32_t m = std::min(chunk, n_reads - c0);
        std::vector<Gen> g(m);
#pragma omp parallel for schedule(dynamic, 1) num_threads(DNAscent::hostThreads())
        for (long i = 0; i < (long)m; i++) {
            Gen &G = g[(size_t)i];
            const uint32_t n_bases = bases[c0 + i];
            dns_read_spec sp; memset(&sp, 0, sizeof sp);
            sp.seed = seeds[c0 + i]; sp.n_bases = n_bases; sp.ref_start = 1000; sp.is_reverse = (int)rev[c0 + i];
            sp.noise_pa = noise_pa; sp.mean_dwell = 11.5; sp.sub_rate = sub_rate; sp.ins_rate = ins_rate; sp.del_rate = del_rate;
            const size_t capq = 2 * (size_t)n_bases + 8;
            G.ref.resize(n_bases); G.bc.resize(capq); G.op.resize(capq); G.len.resize(capq); G.adc.resize(dns_max_samples(n_bases));
            memset(&G.o, 0, sizeof G.o);
            G.o.refseq = G.ref.data(); G.o.basecall = G.bc.data(); G.o.cigar_op = G.op.data(); G.o.cigar_len = G.len.data(); G.o.adc = G.adc.data();
            G.rc = dns_make_read(model_mean, &sp, &G.o);
        }
        for (uint32_t i = 0; i < m; i++) {
            Gen &G = g[i];
            if (G.rc) continue;
            char id[64]; snprintf(id, sizeof id, "synth-%016llx", (unsigned long long)seeds[c0 + i]);
            ReadInput in;
            in.readID = id; in.contig = "chrSynth";
            in.adc = G.adc.data(); in.n_adc = G.o.n_samples; in.cal_offset = G.o.cal_offset; in.cal_scale = G.o.cal_scale;
            // the generator emits strand-direction sequences; ReadInput takes what a BAM / FASTA hold (reference-forward)
            const std::string bc(G.bc.data(), G.o.n_base), rf(G.ref.data(), G.o.n_ref);
            in.querySeq = G.o.is_reverse ? DNAscent::reverseComplement(bc) : bc;
            in.refSlice = G.o.is_reverse ? DNAscent::reverseComplement(rf) : rf;
            in.cigarOp.assign(G.op.data(), G.op.data() + G.o.n_cigar); in.cigarLen.assign(G.len.data(), G.len.data() + G.o.n_cigar);
            in.refStart = G.o.ref_start; in.isReverse = G.o.is_reverse != 0;
            if (sink(in)) accepted++;
        }
    }
    return accepted;
}
extern "C" {
static int fillSynth(ReadBatch *B, const double *model_mean, uint32_t n_reads, const uint64_t *seeds, const uint32_t *bases, const uint8_t *rev, double noise_pa,
                     double sub_rate, double ins_rate, double del_rate) {
    return synthReads([B](const ReadInput &in) { return B->add(in) >= 0; }, model_mean, n_reads, seeds, bases, rev, noise_pa, sub_rate, ins_rate, del_rate);
}
// n_reads synthetic reads of n_bases straight into a binary read container (the product driver's input at BASELINE configs[2] size is 10 000 x
// 50 kb = 11.5 GB: generated on all host cores, 64 reads at a time).  Returns the reads written or -1.
int64_t dnh_container_write_synth(const char *path, const double *model_mean, uint64_t seed0, uint32_t n_reads, uint32_t n_bases) {
    DNAscent::ReadContainerWriter w;
    if (!w.open(path)) return -1;
    std::vector<uint64_t> seeds(n_reads); std::vector<uint32_t> bases(n_reads, n_bases); std::vector<uint8_t> rev(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) { seeds[i] = seed0 + i; rev[i] = (uint8_t)(i & 1u); }
    bool ok = true;
    const int n = synthReads([&](const ReadInput &in) { ok = ok && w.add(in); return ok; }, model_mean, n_reads, seeds.data(), bases.data(), rev.data(), 1.6, 0.002, 0.001, 0.001);
    return (w.close() && ok) ? n : -1;
}
int dnh_batch_fill_synth(void *b, const double *model_mean, uint64_t seed0, uint32_t n_reads, uint32_t n_bases, double noise_pa,
                         double sub_rate, double ins_rate, double del_rate) {
    std::vector<uint64_t> seeds(n_reads); std::vector<uint32_t> bases(n_reads, n_bases); std::vector<uint8_t> rev(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) { seeds[i] = seed0 + i; rev[i] = (uint8_t)(i & 1u); }      // every odd read of the call is reverse
    return fillSynth((ReadBatch *)b, model_mean, n_reads, seeds.data(), bases.data(), rev.data(), noise_pa, sub_rate, ins_rate, del_rate);
}
// the same with a seed and a length PER READ (mixed-length workloads: BASELINE configs[4]'s 1-200 kb law); strand = seed parity
int dnh_batch_fill_synth_list(void *b, const double *model_mean, uint32_t n_reads, const uint64_t *seeds, const uint32_t *bases, double noise_pa,
                              double sub_rate, double ins_rate, double del_rate) {
    std::vector<uint8_t> rev(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) rev[i] = (uint8_t)(seeds[i] & 1u);
    return fillSynth((ReadBatch *)b, model_mean, n_reads, seeds, bases, rev.data(), noise_pa, sub_rate, ins_rate, del_rate);
}

int dnh_stream_detect(void **ctxs, int n_ctx, void **batches, int n_batches, int emit, const char *out_path, const char *header,
                      DNAscent::StreamStats *st, void *keep) {
    return DNAscent::streamDetect((dn_ctx **)ctxs, n_ctx, (ReadBatch **)batches, n_batches, emit != 0, out_path, header, st,
                                  (DNAscent::StreamKeep *)keep);
}
void *dnh_keep_new(void) { return new DNAscent::StreamKeep(); }
void dnh_keep_free(void *k) { delete (DNAscent::StreamKeep *)k; }
// which: 0 read_calls (u64), 1 coord (u32), 2 p_edu (f32), 3 p_brdu (f32), 4 record_bytes (u64); returns the element count, *p the data
uint64_t dnh_keep_get(void *k, int which, const void **p) {
    DNAscent::StreamKeep *K = (DNAscent::StreamKeep *)k;
    switch (which) {
        case 0: *p = K->read_calls.data(); return K->read_calls.size();
        case 1: *p = K->coord.data(); return K->coord.size();
        case 2: *p = K->p_edu.data(); return K->p_edu.size();
        case 3: *p = K->p_brdu.data(); return K->p_brdu.size();
        case 4: *p = K->record_bytes.data(); return K->record_bytes.size();
    }
    *p = nullptr; return 0;
}

// --HMM output step for an uploaded + normalised batch (detect.cpp:885 + writer); returns reads written or a negative code
int dnh_hmm_write(void *ctx, void *b, const char *path, const char *header) {
    ReadBatch *B = (ReadBatch *)b;
    std::vector<DNAscent::ReadCalls> calls;
    int rc = DNAscent::llAcrossRead((dn_ctx *)ctx, *B, calls);
    if (rc) return rc;
    DNAscent::HumanReadableWriter w;
    if (!w.open(path)) return DN_ERR_ARG;
    if (header) w.writeHeader_HR(header);
    int written = 0;
    for (size_t i = 0; i < calls.size(); i++)
        if (B->summary[i].status == DN_READ_OK) { w.write(calls[i]); written++; }
    w.close();
    return written;
}

// writeDetectHeader into a caller buffer; returns the length (the required size if > cap)
uint64_t dnh_detect_header(const char *alignment, const char *genome, const char *index, int threads, unsigned quality, unsigned length,
                           int use_gpu, const char *start_time, const char *software, const char *version, const char *commit,
                           char *buf, uint64_t cap) {
    const std::string s = DNAscent::writeDetectHeader(alignment, genome, index, threads, quality, length, use_gpu != 0, start_time, software,
                                                      version, commit);
    if (s.size() <= cap) memcpy(buf, s.data(), s.size());
    return s.size();
}

// test hook: n probabilities through the fast formatter and through snprintf("%f"); returns the number of differing strings
uint64_t dnh_check_prob_format(const float *p, uint64_t n) {
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) {
        char a[64], b[64];
        char *e = DNAscent::formatProbForTest(a, p[i]); *e = 0;
        snprintf(b, sizeof b, "%f", (double)p[i]);
        bad += strcmp(a, b) != 0;
    }
    return bad;
}

int dnh_revcomp(const char *in, uint32_t n, char *out) {
    const std::string r = DNAscent::reverseComplement(std::string(in, n));
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}

}  // extern "C"
