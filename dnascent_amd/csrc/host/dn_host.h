// dn_host.h -- C++ host side above the C-ABI (include/dnascent_hip.h).
//
// Mirrors the reference's per-read interface for the `detect` hot path with batch semantics:
//
//   reference (one read per OpenMP thread, detect.cpp:852-907)      here (one batch per call)
//   ------------------------------------------------------------    ------------------------------------------
//   DNAscent::read r(record, hdr, index, reference)  reads.h:210     DNAscent::ReadBatch::add(ReadInput)
//   pod5_getSignal(r)                                pod5.cpp:24     ReadInput::adc + calibration (+ ts/ns/sp trimming)
//   normaliseEvents(r, false)                  event_handling.h:13   DNAscent::normaliseEvents(ctx, batch)
//   r.eventAlignment.size() == 0  -> failed          detect.cpp:879  batch.summary[i].status != DN_READ_OK
//   eventalign(r, windowLength_align)                alignment.h:22   DNAscent::eventalign(ctx, batch)
//   runCNN(r, session, inputOps, humanReadable)      detect.h:120     DNAscent::runCNN(ctx, batch, humanReadable, calls)
//   writer->write(r)                                 detect.h:43      DNAscent::HumanReadableWriter::write(calls[i])
//
// The host keeps only flat arrays; every node-based container of the reference (std::map refToQuery,
// vector<event>, ...) is flattened once here and lives in HBM afterwards.
#pragma once
#include <stdint.h>

#include <memory>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "dnascent_hip.h"

namespace DNAscent {

// std::allocator whose construct() DEFAULT-initialises: resize() of the big sample / map arrays does not write zeros that the copies behind it
// overwrite at once (600 MB of int16 per batch: ~0.1 s of one thread per batch in the product driver's loader)
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

struct ReadInput {                     // what reads.h:210-287 + pod5.cpp:24-93 extract from BAM + POD5 + FASTA
    std::string readID, contig;
    const int16_t *adc = nullptr; size_t n_adc = 0;     // complete stored signal of the (parent) read
    float cal_offset = 0.f, cal_scale = 1.f;            // pod5 calibration
    // Dorado tags (reads.h:222-253); signalLength <= 0 means "no trimming" (pod5.cpp:76)
    int signalLength = -1, signalTrim = 0, signalStartCoord = 0; bool isSplit = false;
    std::string querySeq;                               // BAM query sequence, reference-forward orientation (htsInterface.cpp:160)
    std::string refSlice;                               // reference[contig].substr(refStart, refEnd-refStart), forward orientation
    std::vector<uint32_t> cigarOp, cigarLen;            // BAM order
    int refStart = 0; bool isReverse = false;
};

// ---- BAM without htslib (host/dn_bam.cpp: BGZF + BAM records from the SAM/BAM specification, over zlib) ----
struct BamRef { std::string name; uint32_t len; };
class BgzfReader {
public:
    ~BgzfReader();
    bool open(const std::string &path);
    size_t read(void *dst, size_t n);                   // bytes of the uncompressed stream; short only at the end of the file or on a malformed block
    bool failed() const { return bad; }
    void close();
private:
    bool fill();
    void *f = nullptr; std::vector<uint8_t> buf; size_t pos = 0; bool eof = false, bad = false;
};
class BgzfWriter {
public:
    ~BgzfWriter();
    bool open(const std::string &path);
    bool write(const void *src, size_t n);
    bool close();                                        // flushes, appends the EOF marker block
private:
    bool block(const uint8_t *p, size_t n);
    void *f = nullptr; std::vector<uint8_t> buf; bool ok = false;
};
struct BamRecord {                                       // one alignment record (SAMv1 s4.2)
    std::vector<uint8_t> raw;                            // the record as stored, behind its block_size (passed through unchanged by BamWriter)
    int32_t refID = -1, pos = -1, l_seq = 0; uint8_t mapq = 0; uint16_t flag = 0;
    std::string qname, seq;                              // seq decoded from the 4-bit codes "=ACMGRSVTWYHKDBN"
    std::vector<uint32_t> cigarOp, cigarLen;             // BAM operation codes M I D N S H P = X -> 0 .. 8; a CG:B,I tag (> 65 535 operations) is resolved
    size_t aux_off = 0;                                  // where the auxiliary fields start in raw
    long auxFind(const char *tag) const;                 // offset of the field's tag bytes in raw, -1 if absent
    bool auxInt(const char *tag, int64_t &v) const;      // bam_aux2i: any of c C s S i I
    bool auxStr(const char *tag, std::string &s) const;  // Z / H
    int64_t refLength() const;                           // reference bases the CIGAR spans (bam_endpos - pos)
};
class BamReader {
public:
    bool open(const std::string &path);                  // magic, header text, reference dictionary
    int next(BamRecord &r);                              // 1 a record, 0 end of file, -1 malformed
    const std::string &headerText() const { return text_; }
    const std::vector<BamRef> &refs() const { return refs_; }
private:
    BgzfReader z; std::string text_; std::vector<BamRef> refs_;
};
class BamWriter {                                        // detect.h:62-97 SamWriter: header, then records
public:
    bool open(const std::string &path, const std::string &headerText, const std::vector<BamRef> &refs);
    bool writeRaw(const std::vector<uint8_t> &raw);
    bool writeWithMods(const BamRecord &r, const std::string &mmFields, const std::vector<uint8_t> &ml);      // reads.h:453-512 writeModBamTag
    bool close();
private:
    BgzfWriter z;
};
struct ReadInput;
// reads.h:210-287: the record's fields -> ReadInput (no signal: in.adc stays null; fetchID = the POD5 read to fetch).  0 ok, -1 unmapped / empty, -2 not in
// the reference, -3 a base other than A C G T N (htsInterface.cpp:160-178 throws)
int readInputFromBam(const BamRecord &r, const std::vector<BamRef> &refs, const std::map<std::string, std::string> &reference, ReadInput &in, std::string &fetchID);

// htsInterface.cpp:59-157 flattened: ref2query[refLen], ref2del[refLen], query2ref[queryLen+1] (-1 = absent key)
int parseCigar(const std::vector<uint32_t> &ops, const std::vector<uint32_t> &lens, bool isReverse, size_t queryLen,
               std::vector<uint32_t> &ref2query, std::vector<int32_t> &query2ref, std::vector<uint8_t> &ref2del);

std::string reverseComplement(const std::string &s);    // common.h:91
int hostThreads();                                       // threads of the host-side parallel loops (DN_HOST_THREADS, default min(64, cores, the cgroup's CPU quota))

class ReadBatch {
public:
    void clear();
    // returns the index of the read in the batch, or -1 if the read is rejected (empty signal / too short), mirroring
    // the reference's filters (detect.cpp:839, pod5.cpp:64)
    int add(const ReadInput &in);
    // n reads at once, their per-read preparation (CIGAR flattening, reverse complements, copies) on the host's threads; accepted[i] = 1 / 0
    size_t addMany(const ReadInput *const *in, size_t n, uint8_t *accepted);
    // the same with the signals still IN A FILE: in[i]->adc is null, in[i]->n_adc the stored sample count, adcFileOff[i] the byte offset of
    // the read's first stored sample in fd.  The accepted reads' slices are pread() straight to their place in the batch (no intermediate copy
    // of a 1 MB signal per read).  *ioFailed is set when a pread came back short: the batch is then unusable (clear() it).
    size_t addManyFromFile(const ReadInput *const *in, size_t n, uint8_t *accepted, int fd, const uint64_t *adcFileOff, bool *ioFailed);
    size_t size() const { return readID.size(); }
    dn_batch_desc desc() const;
    uint64_t totalSamples() const { return adc_off.empty() ? 0 : adc_off.back(); }
    // Page-lock the arrays dn_batch_upload reads (dn_host_register): the upload then returns before its copies are done and the batch
    // must stay untouched until the context's next dn_collect / dn_sync.  unpin() before the batch is modified again (add / clear).
    int pin();
    void unpin();
    ~ReadBatch() { unpin(); }
    ReadBatch() = default;
    ReadBatch(const ReadBatch &) = delete;
    ReadBatch &operator=(const ReadBatch &) = delete;

    std::vector<std::string> readID, contig;
    RawVec<int16_t> adc; std::vector<uint64_t> adc_off{0};
    std::vector<float> cal_offset, cal_scale;
    std::vector<char> basecall; std::vector<uint64_t> basecall_off{0};
    std::vector<char> refseq; std::vector<uint64_t> refseq_off{0};
    RawVec<uint32_t> ref2query; RawVec<int32_t> query2ref; RawVec<uint8_t> ref2del;
    std::vector<int32_t> ref_start, ref_end; std::vector<uint8_t> is_reverse;
    std::vector<dn_read_summary> summary;               // filled by normaliseEvents / eventalign
private:
    std::vector<void *> pinned;
};

// ---- binary read container ----------------------------------------------------------------------------
// The reference ingests reads through htslib (BAM record, CIGAR, tags) and libpod5 / fast5 (signal, calibration); neither
// library is available to this build.  Hosts without them -- and the tests -- hand reads over in a flat little-endian file
// holding exactly the fields of ReadInput (what reads.h:210-287 + pod5.cpp:24-93 extract):
//   file    "DNRC" u32 version (1) u64 n_reads, then n_reads records
//   record  str readID, str contig (u32 length + bytes); f32 cal_offset, cal_scale; i32 signalLength, signalTrim,
//           signalStartCoord; u8 isSplit, isReverse; i32 refStart; str querySeq; str refSlice;
//           u32 n_cigar, u32 op[n], u32 len[n]; u64 n_adc, i16 adc[n]
struct OwnedRead { ReadInput in; RawVec<int16_t> adc; };                  // in.adc points into adc
class ReadContainerWriter {
public:
    bool open(const std::string &path);
    bool add(const ReadInput &in);
    bool close();                                                          // patches the read count into the header
    ~ReadContainerWriter() { if (f) close(); }
private:
    void *f = nullptr; uint64_t n = 0;
};
class ReadContainerReader {
public:
    bool open(const std::string &path);                                    // false: missing file / bad magic / version
    uint64_t count() const { return n; }
    bool next(OwnedRead &out);                                             // false at the end or on a truncated record
    bool nextHeader(OwnedRead &out, uint64_t *adcFileOff);                 // everything but the samples (out.in.adc = null, n_adc = their count), which are seeked over
    bool skip(uint64_t *nSamples);                                         // seek over the next record, reporting its sample count
    uint64_t tell() const;                                                 // file offset of the next record (for an index)
    bool seek(uint64_t offset);                                            // ... and back to it: next() then reads THAT record
    bool failed() const { return bad; }
    void close();
    ~ReadContainerReader() { close(); }
private:
    void *f = nullptr; uint64_t n = 0, seen = 0; bool bad = false;
};

// normaliseEvents for every read of the batch (event_handling.h:13).  Throws nothing: returns a DN_* code.
int normaliseEvents(dn_ctx *ctx, ReadBatch &batch);
int eventalign(dn_ctx *ctx, ReadBatch &batch);

// ---- output side of runCNN (detect.cpp:677-731) -------------------------------------------------------------------
struct ReadCalls {                     // what runCNN leaves on a DNAscent::read
    std::string humanReadable_detectOut;               // ">readID contig start end strand\n" + "coord\tEdU\tBrdU\tkmer\n" ...
    std::string MM;                                    // modbam: "N+b?,d0,d1,...;N+e?,d0,d1,...;"   (reads.h:464-488)
    std::vector<uint8_t> ML;                           // BrdU bytes then EdU bytes, uint8(p * 255.0)    (reads.h:482-507)
    uint32_t nCalls = 0;                               // T positions reported
};

// one record of the .detect file from the CNN outputs of a read (positions in creation order, probs [n][3]:
// 0 thymidine, 1 BrdU, 2 EdU).  Only strand 9-mers with 'T' in the middle are reported (detect.cpp:690); reverse reads
// print the reverse complement and their lines in reverse order (:699,722).
std::string formatDetectRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                               size_t n, const uint32_t *coord, const char *kmer9, const float *probs, uint32_t *nCalls = nullptr);
// the modbam branch (detect.cpp:704-707 + reads.h:453-512 with no pre-existing MM / ML tag): calls keyed by query index,
// deletions skipped, later positions overwrite earlier ones with the same query index
void modBamFields(size_t n, const uint32_t *queryIdx, const uint32_t *refIdx, const char *kmer9, const float *probs,
                  const uint8_t *ref2del, std::string &MM, std::vector<uint8_t> &ML);
// CNN for every read of the batch that passed eventalign (dn_run_cnn), the bulk result (dn_collect: one transfer per output
// array for the whole batch), then the per-read output above, formatted in parallel; failed reads get empty calls.
// batch.summary is refreshed from the collected result.
int runCNN(dn_ctx *ctx, ReadBatch &batch, bool humanReadable, std::vector<ReadCalls> &calls);
char *formatProbForTest(char *o, float p);               // the "%f" fast path of the record formatter (tests compare it with snprintf)
// the output half alone, for a batch whose result has been collected
void formatCalls(const ReadBatch &batch, const dn_result_batch &res, bool humanReadable, std::vector<ReadCalls> &calls);

// ---- packed per-call results for the gather to the writer rank (detect.cpp:902-906 is where the reference writes) ----
// What a rank of a multi-GPU run sends instead of text: per passing read a meta row {index in the batch, count, header bytes, flags} and, in
// the payload, the read's header line (">readID contig start end strand\n") followed by 16 bytes per call {u32 coord, f32 P(EdU),
// f32 P(BrdU), u32 strand 9-mer at 3 bits per base}.  A read whose 9-mers hold anything but A C G T N travels as its formatted text
// (DN_PACK_TEXT: count = text bytes, header bytes = 0).  formatPacked is the writer's half: same bytes as formatCalls' records.
enum { DN_PACK_REVERSE = 1, DN_PACK_TEXT = 2 };
void packCalls(const ReadBatch &batch, const dn_result_batch &res, std::vector<uint64_t> &meta /* [passing reads][4] */, RawVec<uint8_t> &payload);
void packedSizes(size_t n, const uint64_t *meta3 /* [n][3] */, const uint8_t *const *read_ptr /* [n] */, uint64_t *record_bytes /* [n]: exact text length of every read */);
bool formatPacked(size_t n, const uint64_t *meta3 /* [n][3]: count, header bytes, flags */, const uint8_t *const *read_ptr /* [n] */, RawVec<char> &text,
                  uint64_t *record_bytes /* [n] or null */);

// The buffer-of-reads loop of detect.cpp:821-907 as a stream: batch i is uploaded to context i % n_ctx and its whole per-read
// body (normaliseEvents -> eventalign -> runCNN) enqueued; ONE host thread keeps n_ctx batches in flight and only ever waits for
// the oldest one (dn_collect).  emit: format the .detect records of every collected batch (in parallel) and write them, in input
// order, to outPath (nullptr: formatted and counted, not written).  Every context needs its pore model and CNN loaded.
struct StreamKeep {                     // optional: the binary per-call results of the whole stream, kept for the gather to the writer rank
    std::vector<uint64_t> read_calls;  // calls of every read, in stream order (0 for failed reads)
    std::vector<uint64_t> record_bytes;// length of every read's .detect record in the output (0: failed read / emit off)
    std::vector<uint32_t> coord; std::vector<float> p_edu, p_brdu;
};
struct StreamStats {
    double seconds_total, seconds_upload, seconds_collect, seconds_emit;   // wall time of the call / spent inside uploads, collects, emission
    uint64_t reads, reads_ok, samples, calls, bytes_out, positions;        // positions: r.refCoordToAP entries of the passing reads (CNN rows)
    double seconds_run;                                                    // spent enqueueing the per-read body (dn_run_detect): launches + whatever the queue makes them wait for
    uint64_t overflow_retries;                                             // batches run a second time with the detector's own event bound (dn_ctx_set_event_bound)
};
int streamDetect(dn_ctx **ctxs, int n_ctx, ReadBatch **batches, int n_batches, bool emit, const char *outPath, const char *header,
                 StreamStats *st, StreamKeep *keep = nullptr);

// The same loop OPEN-ENDED, for hosts that do not have their batches up front (detect.cpp:821-907 reads its buffer of reads from the
// BAM as it goes, and writes records as reads complete): submit() uploads a batch to the free context and enqueues its whole per-read
// body, collect() waits for the OLDEST batch in flight and hands back its records.  At most n_ctx batches are in flight; a batch
// must stay alive and untouched between its submit() and its collect().  streamDetect() above is this class driven over a list.
class DetectStream {
public:
    struct Result {
        uint64_t tag = 0; ReadBatch *batch = nullptr;
        dn_result_batch res{};                              // valid until the context's next upload (= the submit() after next on a full stream)
        std::vector<uint64_t> record_bytes;                 // per read: length of its .detect record (0: failed read / emit off)
        std::string text;                                   // the records of the passing reads, batch order (emit == EMIT_TEXT)
        std::vector<uint64_t> packed_meta;                  // emit == EMIT_PACKED: packCalls' meta rows ...
        RawVec<uint8_t> packed;                             // ... and payload (the multi-rank driver gathers these; the writer rank formats)
    };
    enum { EMIT_NONE = 0, EMIT_TEXT = 1, EMIT_PACKED = 2 };
    DetectStream(dn_ctx **ctxs, int n_ctx, int emit);
    bool full() const { return inflight == (int)ctx.size(); }
    int inFlight() const { return inflight; }
    int submit(ReadBatch *batch, uint64_t tag);             // DN_ERR_STATE when full(): collect() first
    int collect(Result &out);                               // DN_ERR_STATE when nothing is in flight
    const StreamStats &stats() const { return S; }
private:
    std::vector<dn_ctx *> ctx; std::vector<ReadBatch *> slot_batch; std::vector<uint64_t> slot_tag;
    int head = 0, inflight = 0; int emit; StreamStats S{}; std::vector<ReadCalls> calls; double t_open;
};

// `detect --HMM` (detect.cpp:885): llAcrossRead for every read that passed normaliseEvents; fills
// calls[i].humanReadable_detectOut with ">readID contig start end strand" + "pos\tlogLR\tkmerRef\tkmerQuery" lines (:414, :571)
int llAcrossRead(dn_ctx *ctx, ReadBatch &batch, std::vector<ReadCalls> &calls);
std::string formatHmmRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                            const std::string &basecall, const std::string &refseq /* both in strand direction */, size_t n,
                            const uint32_t *posOnRef, const uint32_t *posOnQuery, const int32_t *globalPos, const double *llr);

// `DNAscent align` (alignment.cpp:747-898): the record eventalign builds in r.humanReadable_eventalignOut -- header (:553) and
// one line per raw sample (:697-733): "coord\tkmerRef\tscaled\tkmerStrand\tmodelMean" for matches,
// "coord\tkmerRef\tscaled\tNNNNNNNNN\t0" for insertions.  refseq: referenceSeqMappedTo (strand direction);
// poreModelMean: the 4^9 table in kmer2index order.  The rows come from dn_get_align_table.
std::string formatAlignRecord(const std::string &readID, const std::string &contig, int refStart, int refEnd, bool isReverse,
                              const std::string &refseq, const double *poreModelMean, size_t nRows, const uint32_t *coord,
                              const uint32_t *refPos, const double *value, const uint8_t *kind);
// eventalign with the table for an uploaded + normalised batch, records of the passing reads appended to `path`;
// returns the number of reads written or a negative DN_* code
int alignWrite(dn_ctx *ctx, ReadBatch &batch, const double *poreModelMean, const std::string &path);

// writeDetectHeader (detect.cpp:196-232); the time stamp / software strings are the caller's (they are not parity data)
std::string writeDetectHeader(const std::string &alignmentFilename, const std::string &refFilename, const std::string &indexFn,
                              int threads, unsigned quality, unsigned length, bool useGPU, const std::string &startTime,
                              const std::string &software, const std::string &version, const std::string &commit);

class HumanReadableWriter {            // detect.h:32-60
public:
    ~HumanReadableWriter() { close(); }
    bool open(const std::string &filename);
    void writeHeader_HR(const std::string &header);
    void write(const ReadCalls &r);
    void close();
private:
    void *file = nullptr;
};

}  // namespace DNAscent
