"""An INDEPENDENT rendering of POD5's VBZ signal codec for the tests (numpy + pyarrow's bundled zstd -- not the libzstd.so.1 the host library binds): written
from the format description (delta -> zig-zag -> StreamVByte with one key bit per 16-bit value -> one zstd frame), vectorised, sharing no code with
csrc/host/dn_vbz.cpp.  TEST INFRASTRUCTURE."""
import numpy as np
import pyarrow as pa


def svb16_encode(samples):
    s = np.ascontiguousarray(samples, np.int16).astype(np.int64)
    n = s.shape[0]
    d = np.diff(np.concatenate([[0], s]))
    d = ((d + 32768) % 65536) - 32768                         # the difference wraps in 16 bits
    v = np.where(d >= 0, 2 * d, -2 * d - 1).astype(np.uint16)    # zig-zag: 0, -1, 1, -2, ... -> 0, 1, 2, 3, ...
    two = v > 255
    keys = np.packbits(two, bitorder="little")
    width = np.where(two, 2, 1)
    pos = np.concatenate([[0], np.cumsum(width)])
    data = np.zeros(int(pos[-1]), np.uint8)
    data[pos[:-1]] = (v & 0xFF).astype(np.uint8)
    data[pos[:-1][two] + 1] = (v[two] >> 8).astype(np.uint8)
    assert keys.shape[0] == (n + 7) // 8
    return keys.tobytes() + data.tobytes()


def svb16_decode(buf, n):
    b = np.frombuffer(buf, np.uint8)
    kb = (n + 7) // 8
    two = np.unpackbits(b[:kb], bitorder="little")[:n].astype(bool)
    width = np.where(two, 2, 1)
    pos = np.concatenate([[0], np.cumsum(width)])
    data = b[kb:]
    assert int(pos[-1]) == data.shape[0]
    v = data[pos[:-1]].astype(np.int64)
    v[two] |= data[pos[:-1][two] + 1].astype(np.int64) << 8
    d = np.where(v & 1, -((v + 1) >> 1), v >> 1)
    return (np.cumsum(d) % 65536).astype(np.uint16).view(np.int16)


def vbz_encode(samples, level=1):
    return pa.compress(svb16_encode(samples), codec="zstd", asbytes=True)


def vbz_decode(chunk, n):
    # pyarrow wants the decompressed size: the zstd frame header carries it
    size = _content_size(chunk, (n + 7) // 8 + 2 * n)
    return svb16_decode(pa.Codec("zstd").decompress(chunk, decompressed_size=size, asbytes=True), n)


def _content_size(chunk, most):
    """the zstd frame header's content size (RFC 8878 s3.1.1.1): magic, frame header descriptor, [window descriptor], [dict id], content size"""
    b = bytes(chunk)
    assert b[:4] == b"\x28\xb5\x2f\xfd"
    fhd = b[4]
    fcs_flag, single, did_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    o = 5 + (0 if single else 1) + (0, 1, 2, 4)[did_flag]
    if fcs_flag == 0:
        return b[o] if single else most
    if fcs_flag == 1:
        return int.from_bytes(b[o:o + 2], "little") + 256
    if fcs_flag == 2:
        return int.from_bytes(b[o:o + 4], "little")
    return int.from_bytes(b[o:o + 8], "little")
