// NEVER COMPILED: neither htslib nor libpod5 exists in the image this project is built and tested in.  Kept under contrib/ (not in the product tree) as the
// starting point for a deployment that has both libraries; dnascent_amd/build.py: build_io() compiles it when DN_HTSLIB_INC / DN_POD5_INC point at the headers.
//
// dn_io_htslib.cpp -- OPTIONAL ingestion through the reference's own I/O libraries: htslib for the BAM records,
// libpod5 for the signal.  Compiled only where those libraries exist (dnascent_amd/build.py: build_io(); -DDN_WITH_HTSLIB /
// -DDN_WITH_POD5) -- this image and the reference checkout have neither (empty submodules, no network), so on the build and GPU
// boxes of this project the file is NOT compiled and the tested ingestion path is the binary read container (dn_host.h), which holds
// exactly the fields extracted here.  It produces DNAscent::ReadInput, i.e. what DNAscent::read's constructor (reads.h:210-287) and
// pod5_getSignal (pod5.cpp:24-105) take from the files; everything downstream is the tested path.
// Round 5: the BAM half exists WITHOUT htslib in dn_bam.cpp (BGZF + record + aux tags over zlib, compiled and tested: tests/test_bam.py); this file is what a
// deployment that has htslib / libpod5 would compile instead, and the only place the POD5 half can live.
//
// Differences from the reference that matter at >= 5 Msamples/s/GPU x 8: POD5 files stay open in a small cache instead
// of pod5_open_file / pod5_close_and_free_reader per read (pod5.cpp:30,101), and records are filtered before any signal is fetched.
#include "dn_host.h"

#include <map>
#include <string>
#include <vector>

#if defined(DN_WITH_HTSLIB)
#include <htslib/sam.h>

namespace DNAscent {

// the fields reads.h:210-287 extracts from a BAM record (+ the reference slice); false when the record is unusable
// (unmapped / no sequence / contig missing from the reference).  fetchID: the read id whose signal is needed (the parent's for a
// Dorado split read, reads.h:231-252).
bool readInputFromBam(const bam1_t *rec, const bam_hdr_t *hdr, const std::map<std::string, std::string> &reference, ReadInput &in,
                      std::string &fetchID) {
    if (!rec || !hdr || rec->core.tid < 0 || rec->core.l_qseq == 0) return false;              // detect.cpp:839
    const char *qn = bam_get_qname(rec);
    if (!qn) return false;
    in.readID = qn; fetchID = in.readID;
    in.contig = hdr->target_name[rec->core.tid];
    in.refStart = (int)rec->core.pos;
    in.isReverse = bam_is_rev(rec);
    // Dorado tags: ns (signal length), ts (trimmed samples), pi (parent id), sp (start in the parent signal)   reads.h:222-253
    in.signalLength = -1; in.signalTrim = 0; in.signalStartCoord = 0; in.isSplit = false;
    if (uint8_t *t = bam_aux_get(rec, "ns")) in.signalLength = (int)bam_aux2i(t);
    if (uint8_t *t = bam_aux_get(rec, "ts")) in.signalTrim = (int)bam_aux2i(t);
    if (uint8_t *t = bam_aux_get(rec, "pi")) {
        if (uint8_t *s = bam_aux_get(rec, "sp")) in.signalStartCoord = (int)bam_aux2i(s);
        const char *parent = bam_aux2Z(t);
        if (parent && parent[0]) { fetchID = parent; in.isSplit = fetchID != in.readID; }       // pod5.cpp:79 compares the two ids
    }
    // CIGAR in BAM order (parseCigar replays it, reversed for reverse reads: htsInterface.cpp:69)
    const uint32_t *cig = bam_get_cigar(rec);
    in.cigarOp.resize(rec->core.n_cigar); in.cigarLen.resize(rec->core.n_cigar);
    size_t refLen = 0;
    for (uint32_t i = 0; i < rec->core.n_cigar; i++) {
        in.cigarOp[i] = bam_cigar_op(cig[i]); in.cigarLen[i] = bam_cigar_oplen(cig[i]);
        if (in.cigarOp[i] == BAM_CMATCH || in.cigarOp[i] == BAM_CEQUAL || in.cigarOp[i] == BAM_CDIFF || in.cigarOp[i] == BAM_CDEL ||
            in.cigarOp[i] == BAM_CREF_SKIP) refLen += in.cigarLen[i];
    }
    // query sequence, reference-forward orientation, 4-bit codes 1 A 2 C 4 G 8 T 15 N   htsInterface.cpp:160-180
    const uint8_t *seq = bam_get_seq(rec);
    in.querySeq.resize((size_t)rec->core.l_qseq);
    for (int i = 0; i < rec->core.l_qseq; i++) {
        switch (bam_seqi(seq, i)) { case 1: in.querySeq[i] = 'A'; break; case 2: in.querySeq[i] = 'C'; break; case 4: in.querySeq[i] = 'G'; break;
                                    case 8: in.querySeq[i] = 'T'; break; default: in.querySeq[i] = 'N'; }
    }
    const auto it = reference.find(in.contig);
    if (it == reference.end() || (size_t)in.refStart + refLen > it->second.size()) return false;
    in.refSlice = it->second.substr((size_t)in.refStart, refLen);                               // reads.h:272
    return true;
}

}  // namespace DNAscent
#endif  // DN_WITH_HTSLIB

#if defined(DN_WITH_POD5)
#include <pod5_format/c_api.h>

namespace DNAscent {

// POD5 readers kept open across reads (the reference opens and closes the file for every read: pod5.cpp:30,101)
class Pod5Cache {
public:
    explicit Pod5Cache(size_t maxOpen = 64) : cap(maxOpen) { pod5_init(); }
    ~Pod5Cache() { for (auto &e : open) pod5_close_and_free_reader(e.second); pod5_terminate(); }
    // the complete stored signal of (file, batch, row) + its calibration   pod5.cpp:36-60
    bool fetch(const std::string &path, size_t batchIndex, size_t row, DNAscent::RawVec<int16_t> &adc, float &calOffset, float &calScale) {
        Pod5FileReader_t *f = reader(path);
        if (!f) return false;
        Pod5ReadRecordBatch_t *batch = nullptr;
        if (pod5_get_read_batch(&batch, f, batchIndex) != POD5_OK) return false;
        bool ok = false;
        uint16_t ver = 0; ReadBatchRowInfo_t info;
        size_t n = 0;
        if (pod5_get_read_batch_row_info_data(batch, row, READ_BATCH_ROW_INFO_VERSION, &info, &ver) == POD5_OK &&
            pod5_get_read_complete_sample_count(f, batch, row, &n) == POD5_OK) {
            adc.resize(n);
            ok = n > 0 && pod5_get_read_complete_signal(f, batch, row, adc.size(), adc.data()) == POD5_OK;
            calOffset = (float)info.calibration_offset; calScale = (float)info.calibration_scale;      // pod5.cpp:60 casts to float
        }
        pod5_free_read_batch(batch);
        return ok;
    }
private:
    Pod5FileReader_t *reader(const std::string &path) {
        auto it = open.find(path);
        if (it != open.end()) return it->second;
        if (open.size() >= cap) { pod5_close_and_free_reader(open.begin()->second); open.erase(open.begin()); }
        Pod5FileReader_t *f = pod5_open_file(path.c_str());
        if (f) open[path] = f;
        return f;
    }
    std::map<std::string, Pod5FileReader_t *> open; size_t cap;
};

}  // namespace DNAscent
#endif  // DN_WITH_POD5

#if defined(DN_WITH_HTSLIB) && defined(DN_WITH_POD5)
namespace DNAscent {

struct IndexEntry { std::string path; size_t batch, row; };                                     // index.cpp:306-308 line: readID \t batch \t row \t path

// BAM + reference + DNAscent index -> binary read container (one record per usable BAM record, BAM order, filters of detect.cpp:839).
// Returns the number of reads written, -1 on an unreadable input.
long containerFromBam(const std::string &bamPath, const std::map<std::string, std::string> &reference,
                      const std::map<std::string, IndexEntry> &index, unsigned minQuality, unsigned minLength, const std::string &outPath) {
    htsFile *bam = sam_open(bamPath.c_str(), "r");
    if (!bam) return -1;
    bam_hdr_t *hdr = sam_hdr_read(bam);
    if (!hdr) { sam_close(bam); return -1; }
    ReadContainerWriter w;
    if (!w.open(outPath)) { bam_hdr_destroy(hdr); sam_close(bam); return -1; }
    Pod5Cache pod5;
    bam1_t *rec = bam_init1();
    long written = 0;
    DNAscent::RawVec<int16_t> adc;
    while (sam_read1(bam, hdr, rec) >= 0) {
        if (rec->core.tid < 0 || rec->core.qual < minQuality || rec->core.l_qseq == 0) continue;
        if ((unsigned)(bam_endpos(rec) - rec->core.pos) < minLength) continue;                   // detect.cpp:839
        ReadInput in; std::string fetchID;
        if (!readInputFromBam(rec, hdr, reference, in, fetchID)) continue;
        const auto ie = index.find(fetchID);
        if (ie == index.end()) continue;                                                         // reads.h:263-266 "missing"
        if (!pod5.fetch(ie->second.path, ie->second.batch, ie->second.row, adc, in.cal_offset, in.cal_scale)) continue;
        in.adc = adc.data(); in.n_adc = adc.size();
        if (w.add(in)) written++;
    }
    bam_destroy1(rec); bam_hdr_destroy(hdr); sam_close(bam);
    return w.close() ? written : -1;
}

}  // namespace DNAscent
#endif
