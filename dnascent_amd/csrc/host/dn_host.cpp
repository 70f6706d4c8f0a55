// dn_host.cpp -- host side above the C-ABI: read model, CIGAR flattening, batch packing, stage drivers.
#include "dn_host.h"

#include <string.h>

#include <algorithm>

namespace DNAscent {

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

void ReadBatch::clear() { *this = ReadBatch(); }

int ReadBatch::add(const ReadInput &in) {
    // ---- signal slice (pod5.cpp:75-93) ----
    size_t lo = 0, hi = in.n_adc;
    if (in.signalLength > 0) {
        if (in.isSplit) { lo = (size_t)(in.signalStartCoord + in.signalTrim); hi = (size_t)(in.signalStartCoord + in.signalLength); }
        else { lo = (size_t)in.signalTrim; hi = (size_t)in.signalLength; }
        hi = std::min(hi, in.n_adc); lo = std::min(lo, hi);
    }
    if (hi - lo < 16 || in.querySeq.size() < DN_KMER + 1) return -1;
    std::vector<uint32_t> r2q; std::vector<int32_t> q2r; std::vector<uint8_t> r2d;
    const int refLen = parseCigar(in.cigarOp, in.cigarLen, in.isReverse, in.querySeq.size(), r2q, q2r, r2d);
    if (refLen < DN_KMER || (size_t)refLen != in.refSlice.size()) return -1;
    // ---- sequencing direction (reads.h:280-286) ----
    const std::string bc = in.isReverse ? reverseComplement(in.querySeq) : in.querySeq;
    const std::string rs = in.isReverse ? reverseComplement(in.refSlice) : in.refSlice;

    readID.push_back(in.readID); contig.push_back(in.contig);
    adc.insert(adc.end(), in.adc + lo, in.adc + hi); adc_off.push_back(adc.size());
    cal_offset.push_back(in.cal_offset); cal_scale.push_back(in.cal_scale);
    basecall.insert(basecall.end(), bc.begin(), bc.end()); basecall_off.push_back(basecall.size());
    refseq.insert(refseq.end(), rs.begin(), rs.end()); refseq_off.push_back(refseq.size());
    ref2query.insert(ref2query.end(), r2q.begin(), r2q.end());
    ref2del.insert(ref2del.end(), r2d.begin(), r2d.end());
    query2ref.insert(query2ref.end(), q2r.begin(), q2r.end());               // queryLen + 1 entries
    ref_start.push_back(in.refStart); ref_end.push_back(in.refStart + refLen);
    is_reverse.push_back(in.isReverse ? 1 : 0);
    return (int)readID.size() - 1;
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
bool ReadContainerReader::next(OwnedRead &o) {
    FILE *fp = (FILE *)f; if (!fp || seen >= n) return false;
    ReadInput &in = o.in;
    uint8_t split = 0, rev = 0; int32_t sl = 0, st = 0, sc = 0, rs = 0; uint32_t nc = 0; uint64_t na = 0;
    bool ok = getStr(fp, in.readID, 1u << 16) && getStr(fp, in.contig, 1u << 16) && get(fp, in.cal_offset) && get(fp, in.cal_scale) && get(fp, sl) &&
              get(fp, st) && get(fp, sc) && get(fp, split) && get(fp, rev) && get(fp, rs) && getStr(fp, in.querySeq, 1u << 30) &&
              getStr(fp, in.refSlice, 1u << 30) && get(fp, nc) && nc <= (1u << 28);
    if (ok) { in.cigarOp.resize(nc); in.cigarLen.resize(nc); ok = nc == 0 || (fread(in.cigarOp.data(), 4, nc, fp) == nc && fread(in.cigarLen.data(), 4, nc, fp) == nc); }
    ok = ok && get(fp, na) && na <= (1ull << 33);
    if (ok) { o.adc.resize((size_t)na); ok = na == 0 || fread(o.adc.data(), 2, (size_t)na, fp) == na; }
    if (!ok) { bad = true; return false; }
    in.signalLength = sl; in.signalTrim = st; in.signalStartCoord = sc; in.isSplit = split != 0; in.isReverse = rev != 0; in.refStart = rs;
    in.adc = o.adc.data(); in.n_adc = o.adc.size();
    seen++;
    return true;
}
void ReadContainerReader::close() { if (f) fclose((FILE *)f); f = nullptr; }

// ---- output ----------------------------------------------------------------------------------------------------
std::string formatDetectRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                               size_t n, const uint32_t *coord, const char *kmer9, const float *probs, uint32_t *nCalls) {
    std::string out = ">" + readID + " " + contig + " " + std::to_string(refStart) + " " + std::to_string(refEnd) + " " +
                      (isReverse ? "rev" : "fwd") + "\n";
    out.reserve(out.size() + n * 12);
    char line[96];
    uint32_t calls = 0;
    for (size_t q = 0; q < n; q++) {
        const size_t i = isReverse ? n - 1 - q : q;          // std::reverse of the line vector (:722)
        const char *km = kmer9 + i * 9;
        if (km[4] != 'T') continue;
        std::string ks(km, 9);
        if (isReverse) ks = reverseComplement(ks);
        // std::to_string(float) formats with "%f" (6 decimals) after promotion to double
        const int len = snprintf(line, sizeof line, "%u\t%f\t%f\t", coord[i], (double)probs[i * 3 + 2], (double)probs[i * 3 + 1]);
        out.append(line, (size_t)len);
        out += ks; out += '\n';
        calls++;
    }
    if (nCalls) *nCalls = calls;
    return out;
}

void modBamFields(size_t n, const uint32_t *queryIdx, const uint32_t *refIdx, const char *kmer9, const float *probs,
                  const uint8_t *ref2del, std::string &MM, std::vector<uint8_t> &ML) {
    // queryIndexToCalls is a std::map<unsigned, pair<float, float>>: flat (key, position) pairs, stable sort by key, the LAST
    // entry of every key wins (operator[] assignment)
    std::vector<std::pair<uint32_t, uint32_t>> kv;
    for (size_t i = 0; i < n; i++) {
        if (kmer9[i * 9 + 4] != 'T') continue;
        if (ref2del[refIdx[i]]) continue;
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
        edu.push_back(static_cast<uint8_t>(probs[i * 3 + 2] * 255.0));
        brdu.push_back(static_cast<uint8_t>(probs[i * 3 + 1] * 255.0));
    }
    MM = fb + ";" + fe + ";";
    ML = brdu;
    ML.insert(ML.end(), edu.begin(), edu.end());
}

int runCNN(dn_ctx *ctx, ReadBatch &batch, bool humanReadable, std::vector<ReadCalls> &calls) {
    int rc = dn_run_cnn(ctx);
    if (rc) return rc;
    const size_t n = batch.size();
    calls.assign(n, ReadCalls());
    std::vector<uint32_t> coord, qidx, ridx; std::vector<char> kmer; std::vector<float> probs;
    for (size_t r = 0; r < n; r++) {
        const dn_read_summary &s = batch.summary[r];
        if (s.status != DN_READ_OK) continue;               // detect.cpp:879-894: failed reads are counted, not written
        const size_t np = s.n_positions;
        coord.resize(np); qidx.resize(np); ridx.resize(np); kmer.resize(np * 9); probs.resize(np * 3);
        if ((rc = dn_get_positions(ctx, (uint32_t)r, coord.data(), qidx.data(), ridx.data(), nullptr, kmer.data(), nullptr, nullptr, nullptr, nullptr))) return rc;
        if ((rc = dn_get_probabilities(ctx, (uint32_t)r, probs.data()))) return rc;
        if (humanReadable)
            calls[r].humanReadable_detectOut = formatDetectRecord(batch.readID[r], batch.contig[r], batch.ref_start[r], batch.ref_end[r],
                                                                  batch.is_reverse[r] != 0, np, coord.data(), kmer.data(), probs.data(), &calls[r].nCalls);
        else
            modBamFields(np, qidx.data(), ridx.data(), kmer.data(), probs.data(), batch.ref2del.data() + batch.refseq_off[r], calls[r].MM, calls[r].ML);
    }
    return DN_OK;
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
        if ((rc = dn_get_hmm_calls(ctx, (uint32_t)r, pr.data(), pq.data(), gp.data(), nullptr, nullptr, nullptr, llr.data()))) return rc;
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

int dnh_revcomp(const char *in, uint32_t n, char *out) {
    const std::string r = DNAscent::reverseComplement(std::string(in, n));
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}

}  // extern "C"
