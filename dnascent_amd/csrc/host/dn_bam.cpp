// dn_bam.cpp -- BAM ingestion and modbam emission WITHOUT htslib: a BGZF + BAM record reader / writer written from the SAM/BAM specification (SAMv1 s4:
// BGZF blocks = gzip members with a BC extra subfield; BAM = magic, header text, reference dictionary, records) over zlib's raw deflate, which IS in this image
// (htslib, the reference's I/O library, is not: its submodule is empty and there is no network -- contrib/dn_io_htslib.cpp stays the optional htslib /
// libpod5 path and has never been compiled).  What the reference takes from a BAM record lives in three places, mirrored here:
//   reads.h:210-287          DNAscent::read's constructor: qname, Dorado tags ns / ts / pi / sp, CIGAR, target name, query sequence, strand
//   htsInterface.cpp:59-180  parseCigar (host/dn_host.cpp parseCigar takes the CIGAR as read here), getQuerySequence (4-bit codes; anything but A C G T N throws)
//   reads.h:453-512          writeModBamTag: MM:Z = existing MM + "N+b?,<deltas>;N+e?,<deltas>;", ML:B:C = existing ML + BrdU bytes + EdU bytes, both re-appended
//                            at the end of the record; detect.h:62-97 SamWriter writes header + records (sam_hdr_write / sam_write1)
// CIGARs of more than 65 535 operations (50-200 kb nanopore reads reach that) are stored in the CG:B,I tag with a <l_seq>S<ref_len>N placeholder (SAMv1 s4.2.2):
// resolved on reading, produced on writing by passing records through unchanged.
// Not here: POD5 (Arrow IPC + VBZ/zstd: neither Arrow nor zstd.h is in the image) -- so real data still needs the signals from somewhere; the binary read
// container remains the tested ingestion path of the drivers.  Tested against BAM files written / read back by an independent Python encoder / decoder of the
// spec (tests/bam_codec.py, tests/test_bam.py).
#include "dn_host.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace DNAscent {

// ---------------------------------------------------------------------------------------------------------------------------------
// BGZF
// ---------------------------------------------------------------------------------------------------------------------------------
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
static inline void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }

bool BgzfReader::open(const std::string &path) {
    close();
    f = fopen(path.c_str(), "rb");
    buf.clear(); pos = 0; eof = false; bad = false;
    return f != nullptr;
}
void BgzfReader::close() { if (f) { fclose((FILE *)f); f = nullptr; } }
BgzfReader::~BgzfReader() { close(); }

// one BGZF block -> buf; false at end of file or on a malformed block (bad is set for the latter)
bool BgzfReader::fill() {
    if (!f || eof || bad) return false;
    uint8_t h[12];
    const size_t got = fread(h, 1, 12, (FILE *)f);
    if (got == 0) { eof = true; return false; }
    if (got != 12 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { bad = true; return false; }     // gzip magic, deflate, FEXTRA
    const unsigned xlen = le16(h + 10);
    std::vector<uint8_t> extra(xlen);
    if (fread(extra.data(), 1, xlen, (FILE *)f) != xlen) { bad = true; return false; }
    int bsize = -1;
    for (unsigned o = 0; o + 4 <= xlen;) {                 // subfields: SI1 SI2 SLEN data
        const unsigned sl = le16(&extra[o + 2]);
        if (extra[o] == 'B' && extra[o + 1] == 'C' && sl == 2 && o + 6 <= xlen) bsize = le16(&extra[o + 4]);
        o += 4 + sl;
    }
    if (bsize < 0) { bad = true; return false; }
    const long clen = (long)bsize + 1 - 12 - (long)xlen - 8;     // compressed payload
    if (clen < 0) { bad = true; return false; }
    std::vector<uint8_t> c((size_t)clen + 8);
    if (fread(c.data(), 1, c.size(), (FILE *)f) != c.size()) { bad = true; return false; }
    const uint32_t crc = le32(&c[(size_t)clen]), isize = le32(&c[(size_t)clen + 4]);
    if (isize > 65536) { bad = true; return false; }
    buf.resize(isize); pos = 0;
    if (isize) {
        z_stream z; memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) { bad = true; return false; }
        z.next_in = c.data(); z.avail_in = (uInt)clen; z.next_out = buf.data(); z.avail_out = isize;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END || z.avail_out != 0) { bad = true; return false; }
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data(), isize) != crc) { bad = true; return false; }
    }
    return true;                                           // (an empty block -- the EOF marker -- is skipped by read())
}

// n bytes of the uncompressed stream; returns how many could be had (short only at the end of the file or on an error)
size_t BgzfReader::read(void *dst, size_t n) {
    size_t done = 0;
    while (done < n) {
        if (pos == buf.size()) { if (!fill()) break; continue; }
        const size_t take = std::min(n - done, buf.size() - pos);
        memcpy((uint8_t *)dst + done, buf.data() + pos, take);
        pos += take; done += take;
    }
    return done;
}

bool BgzfWriter::open(const std::string &path) { close(); f = fopen(path.c_str(), "wb"); buf.clear(); ok = f != nullptr; return ok; }
BgzfWriter::~BgzfWriter() { close(); }
bool BgzfWriter::block(const uint8_t *p, size_t n) {     // one block of n <= 65280 uncompressed bytes
    std::vector<uint8_t> out(n + n / 1000 + 64);
    z_stream z; memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    z.next_in = const_cast<uint8_t *>(p); z.avail_in = (uInt)n; z.next_out = out.data(); z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    const size_t clen = out.size() - z.avail_out;
    deflateEnd(&z);
    if (rc != Z_STREAM_END || clen + 26 > 65536) return false;
    std::vector<uint8_t> h = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0 };
    put16(h, (uint16_t)(clen + 25));                       // BSIZE = total block size - 1
    std::vector<uint8_t> tail;
    put32(tail, (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n)); put32(tail, (uint32_t)n);
    return fwrite(h.data(), 1, h.size(), (FILE *)f) == h.size() && fwrite(out.data(), 1, clen, (FILE *)f) == clen && fwrite(tail.data(), 1, 8, (FILE *)f) == 8;
}
bool BgzfWriter::write(const void *src, size_t n) {
    if (!ok) return false;
    const uint8_t *p = (const uint8_t *)src;
    while (n) {
        const size_t take = std::min(n, (size_t)65280 - buf.size());
        buf.insert(buf.end(), p, p + take); p += take; n -= take;
        if (buf.size() == 65280) { ok = block(buf.data(), buf.size()); buf.clear(); if (!ok) return false; }
    }
    return true;
}
bool BgzfWriter::close() {
    if (!f) return ok;
    if (ok && !buf.empty()) { ok = block(buf.data(), buf.size()); buf.clear(); }
    if (ok) ok = block(nullptr, 0);                        // the 28-byte EOF marker: an empty block
    if (fclose((FILE *)f) != 0) ok = false;
    f = nullptr;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BAM records
// ---------------------------------------------------------------------------------------------------------------------------------
// walk the auxiliary fields of a record; returns the offset of the field `tag` (at its two tag bytes) or -1; *next = offset behind it
static long aux_size(const uint8_t *a, size_t n, size_t o) {       // bytes of the VALUE of the field whose type byte is at o, -1 if malformed
    if (o >= n) return -1;
    switch (a[o]) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'Z': case 'H': { size_t e = o + 1; while (e < n && a[e]) e++; return e < n ? (long)(e - o) : -1; }     // value + NUL
        case 'B': {
            if (o + 6 > n) return -1;
            const uint32_t cnt = le32(a + o + 2);
            const int w = (a[o + 1] == 'c' || a[o + 1] == 'C') ? 1 : (a[o + 1] == 's' || a[o + 1] == 'S') ? 2 : (a[o + 1] == 'i' || a[o + 1] == 'I' || a[o + 1] == 'f') ? 4 : 0;
            if (!w) return -1;
            return (long)(5 + (size_t)cnt * (size_t)w);
        }
        default: return -1;
    }
}
long BamRecord::auxFind(const char *tag) const {
    const uint8_t *a = raw.data(); const size_t n = raw.size();
    for (size_t o = aux_off; o + 3 <= n;) {
        const long vs = aux_size(a, n, o + 2);
        if (vs < 0 || o + 3 + (size_t)vs > n) return -1;
        if (a[o] == (uint8_t)tag[0] && a[o + 1] == (uint8_t)tag[1]) return (long)o;
        o += 3 + (size_t)vs;
    }
    return -1;
}
bool BamRecord::auxInt(const char *tag, int64_t &v) const {   // bam_aux2i: any integer type
    const long o = auxFind(tag);
    if (o < 0) return false;
    const uint8_t *p = raw.data() + o + 3;
    switch (raw[(size_t)o + 2]) {
        case 'c': v = (int8_t)p[0]; return true;
        case 'C': v = p[0]; return true;
        case 's': v = (int16_t)le16(p); return true;
        case 'S': v = le16(p); return true;
        case 'i': v = (int32_t)le32(p); return true;
        case 'I': v = le32(p); return true;
        default: return false;
    }
}
bool BamRecord::auxStr(const char *tag, std::string &s) const {
    const long o = auxFind(tag);
    if (o < 0 || (raw[(size_t)o + 2] != 'Z' && raw[(size_t)o + 2] != 'H')) return false;
    s.assign((const char *)raw.data() + o + 3);
    return true;
}
int64_t BamRecord::refLength() const {                   // bam_endpos - pos: M = X D N consume the reference
    int64_t n = 0;
    for (size_t i = 0; i < cigarOp.size(); i++) if (cigarOp[i] == 0 || cigarOp[i] == 2 || cigarOp[i] == 3 || cigarOp[i] == 7 || cigarOp[i] == 8) n += cigarLen[i];
    return n;
}

bool BamReader::open(const std::string &path) {
    refs_.clear(); text_.clear();
    if (!z.open(path)) return false;
    uint8_t m[8];
    if (z.read(m, 8) != 8 || memcmp(m, "BAM\1", 4) != 0) return false;
    const uint32_t lt = le32(m + 4);
    if (lt > (1u << 30)) return false;                    // a header text of more than 1 GB is a damaged file, not an allocation to attempt
    text_.resize(lt);
    if (lt && z.read(&text_[0], lt) != lt) return false;
    while (!text_.empty() && text_.back() == '\0') text_.pop_back();
    uint8_t q[4];
    if (z.read(q, 4) != 4) return false;
    const uint32_t nref = le32(q);
    if (nref > (1u << 24)) return false;
    for (uint32_t i = 0; i < nref; i++) {
        if (z.read(q, 4) != 4) return false;
        const uint32_t ln = le32(q);
        if (ln == 0 || ln > (1u << 16)) return false;
        std::string name(ln, '\0');
        if (z.read(&name[0], ln) != ln || z.read(q, 4) != 4) return false;
        name.resize(ln - 1);                              // NUL-terminated in the file
        refs_.push_back({name, le32(q)});
    }
    return true;
}

// 1: a record; 0: end of file; -1: malformed
int BamReader::next(BamRecord &r) {
    uint8_t q[4];
    const size_t got = z.read(q, 4);
    if (got == 0 && !z.failed()) return 0;
    if (got != 4) return -1;
    const uint32_t bs = le32(q);
    if (bs < 32 || bs > (1u << 30)) return -1;
    r.raw.resize(bs);
    if (z.read(r.raw.data(), bs) != bs) return -1;
    const uint8_t *p = r.raw.data();
    r.refID = (int32_t)le32(p); r.pos = (int32_t)le32(p + 4);
    const unsigned l_name = p[8]; r.mapq = p[9];
    const unsigned n_cig = le16(p + 12); r.flag = le16(p + 14);
    r.l_seq = (int32_t)le32(p + 16);
    if (r.l_seq < 0 || l_name == 0) return -1;
    size_t o = 32;
    if (o + l_name > bs) return -1;
    r.qname.assign((const char *)p + o, l_name - 1); o += l_name;
    if (o + 4ull * n_cig > bs) return -1;
    r.cigarOp.resize(n_cig); r.cigarLen.resize(n_cig);
    for (unsigned i = 0; i < n_cig; i++) { const uint32_t c = le32(p + o + 4 * i); r.cigarOp[i] = c & 15u; r.cigarLen[i] = c >> 4; }
    o += 4ull * n_cig;
    const size_t sb = ((size_t)r.l_seq + 1) / 2;
    if (o + sb + (size_t)r.l_seq > bs) return -1;
    r.seq.resize((size_t)r.l_seq);
    for (int32_t i = 0; i < r.l_seq; i++) r.seq[(size_t)i] = "=ACMGRSVTWYHKDBN"[(p[o + (size_t)i / 2] >> ((i & 1) ? 0 : 4)) & 15];
    o += sb + (size_t)r.l_seq;
    r.aux_off = o;
    // the real CIGAR of a record with more than 65 535 operations: CG:B,I, behind a <l_seq>S<ref_len>N placeholder (SAMv1 s4.2.2)
    if (n_cig == 2 && r.cigarOp[0] == 4 && (int64_t)r.cigarLen[0] == r.l_seq && r.cigarOp[1] == 3) {
        const long cg = r.auxFind("CG");
        if (cg >= 0 && r.raw[(size_t)cg + 2] == 'B' && r.raw[(size_t)cg + 3] == 'I') {
            const uint32_t cnt = le32(p + cg + 4);
            r.cigarOp.resize(cnt); r.cigarLen.resize(cnt);
            for (uint32_t i = 0; i < cnt; i++) { const uint32_t c = le32(p + cg + 8 + 4ull * i); r.cigarOp[i] = c & 15u; r.cigarLen[i] = c >> 4; }
        }
    }
    return 1;
}

// What DNAscent::read's constructor takes from a record (reads.h:210-287) -> ReadInput (the signal is not the BAM's business: in.adc stays null;
// fetchID names the POD5 read to fetch -- the parent's for a Dorado split read).  0 ok; -1 unmapped / no sequence (detect.cpp:839); -2 the contig or the
// slice is not in the reference; -3 a base other than A C G T N (getQuerySequence throws ParsingError, htsInterface.cpp:160-178).
int readInputFromBam(const BamRecord &r, const std::vector<BamRef> &refs, const std::map<std::string, std::string> &reference, ReadInput &in, std::string &fetchID) {
    if (r.refID < 0 || (size_t)r.refID >= refs.size() || r.l_seq == 0) return -1;
    in.readID = r.qname; fetchID = in.readID;
    in.contig = refs[(size_t)r.refID].name;
    in.refStart = r.pos;
    in.isReverse = (r.flag & 16) != 0;
    in.signalLength = -1; in.signalTrim = 0; in.signalStartCoord = 0; in.isSplit = false;
    int64_t v;
    if (r.auxInt("ns", v)) in.signalLength = (int)v;
    if (r.auxInt("ts", v)) in.signalTrim = (int)v;
    std::string parent;
    if (r.auxFind("pi") >= 0) {
        if (r.auxInt("sp", v)) in.signalStartCoord = (int)v;
        if (r.auxStr("pi", parent) && !parent.empty()) { fetchID = parent; in.isSplit = fetchID != in.readID; }     // pod5.cpp:79 compares the two ids
    }
    in.cigarOp = r.cigarOp; in.cigarLen = r.cigarLen;
    for (char c : r.seq) if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') return -3;
    in.querySeq = r.seq;
    const auto it = reference.find(in.contig);
    const int64_t refLen = r.refLength();
    if (it == reference.end() || in.refStart < 0 || (size_t)in.refStart + (size_t)refLen > it->second.size()) return -2;
    in.refSlice = it->second.substr((size_t)in.refStart, (size_t)refLen);          // reads.h:272
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// writer (detect.h:62-97 SamWriter, reads.h:453-512 writeModBamTag)
// ---------------------------------------------------------------------------------------------------------------------------------
bool BamWriter::open(const std::string &path, const std::string &headerText, const std::vector<BamRef> &refs) {
    if (!z.open(path)) return false;
    std::vector<uint8_t> h = { 'B', 'A', 'M', 1 };
    put32(h, (uint32_t)headerText.size());
    h.insert(h.end(), headerText.begin(), headerText.end());
    put32(h, (uint32_t)refs.size());
    for (const BamRef &r : refs) {
        put32(h, (uint32_t)r.name.size() + 1);
        h.insert(h.end(), r.name.begin(), r.name.end()); h.push_back(0);
        put32(h, r.len);
    }
    return z.write(h.data(), h.size());
}
bool BamWriter::writeRaw(const std::vector<uint8_t> &raw) {
    std::vector<uint8_t> bs; put32(bs, (uint32_t)raw.size());
    return z.write(bs.data(), 4) && z.write(raw.data(), raw.size());
}
// the record with its base-analogue calls: MM:Z = (the record's own MM) + mmFields, ML:B:C = (its own ML) + ml, both moved to the END of the record as
// bam_aux_del + bam_aux_append / bam_aux_update_array leave them (reads.h:456-510).  mmFields = "N+b?,<deltas>;N+e?,<deltas>;", ml = BrdU bytes then EdU bytes.
bool BamWriter::writeWithMods(const BamRecord &r, const std::string &mmFields, const std::vector<uint8_t> &ml) {
    std::vector<uint8_t> out(r.raw.begin(), r.raw.begin() + (long)r.aux_off);
    std::string mm; std::vector<uint8_t> mlAll;
    const uint8_t *a = r.raw.data(); const size_t n = r.raw.size();
    for (size_t o = r.aux_off; o + 3 <= n;) {
        const long vs = aux_size(a, n, o + 2);
        if (vs < 0 || o + 3 + (size_t)vs > n) return false;
        const bool isMM = a[o] == 'M' && a[o + 1] == 'M', isML = a[o] == 'M' && a[o + 1] == 'L';
        if (isMM && a[o + 2] == 'Z') mm.assign((const char *)a + o + 3);
        else if (isML && a[o + 2] == 'B') {
            const uint32_t cnt = le32(a + o + 4);
            const char st = (char)a[o + 3];
            for (uint32_t i = 0; i < cnt; i++)             // bam_auxB2i of every element, narrowed to the byte the reference pushes back
                mlAll.push_back(st == 'c' || st == 'C' ? a[o + 8 + i] : (st == 's' || st == 'S') ? (uint8_t)le16(a + o + 8 + 2ull * i) : (uint8_t)le32(a + o + 8 + 4ull * i));
        } else out.insert(out.end(), a + o, a + o + 3 + (size_t)vs);
        o += 3 + (size_t)vs;
    }
    mm += mmFields;
    out.push_back('M'); out.push_back('M'); out.push_back('Z');
    out.insert(out.end(), mm.begin(), mm.end()); out.push_back(0);
    mlAll.insert(mlAll.end(), ml.begin(), ml.end());
    out.push_back('M'); out.push_back('L'); out.push_back('B'); out.push_back('C');
    put32(out, (uint32_t)mlAll.size());
    out.insert(out.end(), mlAll.begin(), mlAll.end());
    return writeRaw(out);
}
bool BamWriter::close() { return z.close(); }

}  // namespace DNAscent

// ---------------------------------------------------------------------------------------------------------------------------------
// C wrappers for the Python tests (tests/test_bam.py)
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" {
struct dnh_bam_fields {                                    // scalars of the current record + of the ReadInput made from it
    int32_t ref_id, pos, mapq, flag, l_seq, n_cigar;
    int32_t input_rc;                                      // readInputFromBam's return code (-9: no reference given)
    int32_t ref_start, is_reverse, signal_length, signal_trim, signal_start, is_split;
    int64_t ref_len;
};
struct DnhBam { DNAscent::BamReader rd; DNAscent::BamRecord rec; DNAscent::ReadInput in; std::string fetch; std::map<std::string, std::string> reference; };
void *dnh_bam_open(const char *path) {
    DnhBam *b = new DnhBam();
    if (!b->rd.open(path)) { delete b; return nullptr; }
    return b;
}
void dnh_bam_close(void *h) { delete (DnhBam *)h; }
void dnh_bam_add_reference(void *h, const char *name, const char *seq) { ((DnhBam *)h)->reference[name] = seq; }
const char *dnh_bam_header(void *h) { return ((DnhBam *)h)->rd.headerText().c_str(); }
int dnh_bam_nref(void *h) { return (int)((DnhBam *)h)->rd.refs().size(); }
const char *dnh_bam_ref(void *h, int i, uint32_t *len) { const DNAscent::BamRef &r = ((DnhBam *)h)->rd.refs()[(size_t)i]; *len = r.len; return r.name.c_str(); }
int dnh_bam_next(void *h, dnh_bam_fields *f) {
    DnhBam *b = (DnhBam *)h;
    const int rc = b->rd.next(b->rec);
    if (rc != 1) return rc;
    const DNAscent::BamRecord &r = b->rec;
    f->ref_id = r.refID; f->pos = r.pos; f->mapq = r.mapq; f->flag = r.flag; f->l_seq = r.l_seq; f->n_cigar = (int32_t)r.cigarOp.size(); f->ref_len = r.refLength();
    b->in = DNAscent::ReadInput();
    f->input_rc = b->reference.empty() ? -9 : DNAscent::readInputFromBam(r, b->rd.refs(), b->reference, b->in, b->fetch);
    f->ref_start = b->in.refStart; f->is_reverse = b->in.isReverse; f->signal_length = b->in.signalLength; f->signal_trim = b->in.signalTrim;
    f->signal_start = b->in.signalStartCoord; f->is_split = b->in.isSplit;
    return 1;
}
const char *dnh_bam_str(void *h, int which) {              // 0 qname, 1 seq, 2 contig, 3 fetch id, 4 query sequence of the ReadInput, 5 its reference slice
    DnhBam *b = (DnhBam *)h;
    switch (which) { case 0: return b->rec.qname.c_str(); case 1: return b->rec.seq.c_str(); case 2: return b->in.contig.c_str(); case 3: return b->fetch.c_str();
                     case 4: return b->in.querySeq.c_str(); default: return b->in.refSlice.c_str(); }
}
void dnh_bam_cigar(void *h, uint32_t *op, uint32_t *len) {
    DnhBam *b = (DnhBam *)h;
    for (size_t i = 0; i < b->rec.cigarOp.size(); i++) { op[i] = b->rec.cigarOp[i]; len[i] = b->rec.cigarLen[i]; }
}
// copy a BAM: every record passed through; records whose index is in mod_idx get the base-analogue tags (mm[i], ml[i] of ml_len[i] bytes)
long dnh_bam_copy_with_mods(const char *src, const char *dst, const uint64_t *mod_idx, uint64_t n_mod, const char *const *mm, const uint8_t *const *ml, const uint64_t *ml_len) {
    DNAscent::BamReader rd; DNAscent::BamWriter wr;
    if (!rd.open(src) || !wr.open(dst, rd.headerText(), rd.refs())) return -1;
    DNAscent::BamRecord r;
    long n = 0; uint64_t k = 0;
    for (;;) {
        const int rc = rd.next(r);
        if (rc == 0) break;
        if (rc < 0) return -2;
        bool ok;
        if (k < n_mod && mod_idx[k] == (uint64_t)n) { ok = wr.writeWithMods(r, mm[k], std::vector<uint8_t>(ml[k], ml[k] + ml_len[k])); k++; }
        else ok = wr.writeRaw(r.raw);
        if (!ok) return -3;
        n++;
    }
    return wr.close() ? n : -4;
}
}  // extern "C"
