// dn_vbz.cpp -- the SIGNAL codec of POD5 (the half of pod5_getSignal that is arithmetic: pod5.cpp:57 pod5_get_read_complete_signal hands the
// reference int16 samples that libpod5 decoded from the signal table's `signal` column).  libpod5 and its Arrow dependency are absent from this image (empty
// submodule, no network), so the Arrow IPC container around the column is NOT read here; what IS here is the column's codec, "VBZ" as the POD5 format
// specification defines it (docs/SPECIFICATION.md of nanoporetech/pod5-file-format, signal table, `minknow.vbz` extension type; third-party, un-vendored):
//
//     samples (int16)  ->  delta against the previous sample (first against 0), 16-bit wrap-around
//                      ->  zig-zag: (d + d) ^ (d >> 15), so small magnitudes of either sign become small unsigned values
//                      ->  StreamVByte for 16-bit values ("svb16"): one KEY BIT per value (0: one data byte, 1: two data bytes, little endian), key bits
//                          packed LSB-first into ceil(n / 8) key bytes that PRECEDE the data bytes
//                      ->  one zstd frame around [key bytes][data bytes]
//
// The number of samples is not in the stream: it is the signal table's `samples` column, so the decoder takes it as an argument.
//
// zstd: libzstd.so.1 is in the image WITHOUT its header, so the three functions used are declared here with their published prototypes (zstd.h: stable API since
// 1.0) and bound with dlopen at first use; a host without the library gets a clear error, never a crash.  PARITY: written from the format's specification and
// tested against an independent Python encoder (tests/vbz_codec.py: numpy + pyarrow's bundled zstd); NOT checked against libpod5's own output -- there is no
// POD5 file and no libpod5 in this image.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

namespace {

typedef size_t (*zstd_decompress_t)(void *dst, size_t dstCapacity, const void *src, size_t compressedSize);
typedef size_t (*zstd_compress_t)(void *dst, size_t dstCapacity, const void *src, size_t srcSize, int compressionLevel);
typedef size_t (*zstd_bound_t)(size_t srcSize);
typedef unsigned (*zstd_iserror_t)(size_t code);
typedef unsigned long long (*zstd_framesize_t)(const void *src, size_t srcSize);

struct Zstd {
    void *h = nullptr;
    zstd_decompress_t decompress = nullptr; zstd_compress_t compress = nullptr; zstd_bound_t bound = nullptr; zstd_iserror_t is_error = nullptr;
    zstd_framesize_t frame_size = nullptr;
    std::string err;
};
static Zstd &zstd() {
    static Zstd z;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = { "libzstd.so.1", "libzstd.so" };
        for (const char *n : names) if ((z.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!z.h) { z.err = "libzstd.so.1 not found (dlopen): POD5's VBZ signal compression needs zstd"; return; }
        z.decompress = (zstd_decompress_t)dlsym(z.h, "ZSTD_decompress"); z.compress = (zstd_compress_t)dlsym(z.h, "ZSTD_compress");
        z.bound = (zstd_bound_t)dlsym(z.h, "ZSTD_compressBound"); z.is_error = (zstd_iserror_t)dlsym(z.h, "ZSTD_isError");
        z.frame_size = (zstd_framesize_t)dlsym(z.h, "ZSTD_getFrameContentSize");
        if (!z.decompress || !z.compress || !z.bound || !z.is_error || !z.frame_size) z.err = "libzstd.so.1 lacks ZSTD_decompress / ZSTD_compress / ZSTD_getFrameContentSize";
    });
    return z;
}
static thread_local std::string last_error;
static int fail(int code, const std::string &what) { last_error = what; return code; }

static inline size_t key_bytes(size_t n) { return (n + 7) / 8; }

}  // namespace

extern "C" {

const char *dnh_vbz_last_error(void) { return last_error.c_str(); }
int dnh_vbz_available(void) { return zstd().err.empty() ? 1 : 0; }

// svb16 + zig-zag + delta, no zstd (the two layers are separately testable): n int16 samples -> [keys][data]; returns the bytes written (dst holds >= key_bytes + 2 n)
uint64_t dnh_svb16_encode(const int16_t *samples, uint64_t n, uint8_t *dst) {
    uint8_t *keys = dst, *data = dst + key_bytes((size_t)n);
    memset(keys, 0, key_bytes((size_t)n));
    uint16_t prev = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint16_t cur = (uint16_t)samples[i];
        const uint16_t d = (uint16_t)(cur - prev);
        prev = cur;
        const uint16_t v = (uint16_t)((uint16_t)(d + d) ^ (uint16_t)((int16_t)d >> 15));
        if (v < 256u) *data++ = (uint8_t)v;
        else { *data++ = (uint8_t)(v & 0xFFu); *data++ = (uint8_t)(v >> 8); keys[i >> 3] |= (uint8_t)(1u << (i & 7u)); }
    }
    return (uint64_t)(data - dst);
}

// the inverse; 0 ok, -1: the stream is shorter / longer than its key bits say
int dnh_svb16_decode(const uint8_t *src, uint64_t n_src, uint64_t n, int16_t *samples) {
    const size_t kb = key_bytes((size_t)n);
    if (n_src < kb) return fail(-1, "svb16: stream shorter than its key bytes");
    const uint8_t *keys = src, *data = src + kb, *end = src + n_src;
    uint16_t prev = 0;
    for (uint64_t i = 0; i < n; i++) {
        const unsigned two = (keys[i >> 3] >> (i & 7u)) & 1u;
        if ((size_t)(end - data) < 1u + two) return fail(-1, "svb16: data bytes end before the last sample");
        uint16_t v = *data++;
        if (two) v |= (uint16_t)((uint16_t)*data++ << 8);
        const uint16_t d = (uint16_t)((v >> 1) ^ (uint16_t)(0u - (v & 1u)));      // zig-zag back
        prev = (uint16_t)(prev + d);
        samples[i] = (int16_t)prev;
    }
    if (data != end) return fail(-1, "svb16: bytes left over after the last sample");
    return 0;
}

// one VBZ chunk (a cell of the signal table's `signal` column) -> n_samples int16 (that row's `samples` cell).  0 ok; -1 malformed; -2 no zstd library
int dnh_vbz_decode(const uint8_t *src, uint64_t n_src, uint64_t n_samples, int16_t *samples) {
    Zstd &z = zstd();
    if (!z.err.empty()) return fail(-2, z.err);
    const unsigned long long raw = z.frame_size(src, (size_t)n_src);
    const size_t most = key_bytes((size_t)n_samples) + 2 * (size_t)n_samples;
    if (raw == (unsigned long long)-2 /* ZSTD_CONTENTSIZE_ERROR */) return fail(-1, "vbz: not a zstd frame");
    const size_t cap = raw == (unsigned long long)-1 /* unknown */ ? most : (size_t)raw;
    if (cap > most) return fail(-1, "vbz: the frame holds more bytes than n_samples values can take");
    std::vector<uint8_t> buf(cap ? cap : 1);
    const size_t got = z.decompress(buf.data(), cap, src, (size_t)n_src);
    if (z.is_error(got)) return fail(-1, "vbz: zstd could not decompress the frame");
    return dnh_svb16_decode(buf.data(), got, n_samples, samples);
}

// the encoder (tools that write test files; the reference only ever reads): returns the bytes written to dst, 0 on error.  dnh_vbz_bound: capacity dst needs.
uint64_t dnh_vbz_bound(uint64_t n_samples) {
    Zstd &z = zstd();
    const size_t most = key_bytes((size_t)n_samples) + 2 * (size_t)n_samples;
    return z.err.empty() ? (uint64_t)z.bound(most) : 0;
}
uint64_t dnh_vbz_encode(const int16_t *samples, uint64_t n_samples, uint8_t *dst, uint64_t cap, int level) {
    Zstd &z = zstd();
    if (!z.err.empty()) { fail(-2, z.err); return 0; }
    std::vector<uint8_t> buf(key_bytes((size_t)n_samples) + 2 * (size_t)n_samples + 1);
    const uint64_t nb = dnh_svb16_encode(samples, n_samples, buf.data());
    const size_t got = z.compress(dst, (size_t)cap, buf.data(), (size_t)nb, level > 0 ? level : 1);
    if (z.is_error(got)) { fail(-1, "vbz: zstd could not compress (dst too small?)"); return 0; }
    return (uint64_t)got;
}

}  // extern "C"
