#!/usr/bin/env python3
"""Decode a TensorFlow checkpoint index (`variables/variables.index`) into {name: dtype, shape, offset, size}.

The file is an SSTable in LevelDB's table format (table/format.cc, table/block.cc): data blocks of prefix-compressed
(key, value) entries, an index block of block handles, a 48-byte footer ending in the magic 0xdb4775248b80fb57.  Keys are
checkpoint variable names, values are serialized BundleEntryProto messages (tensor_bundle.proto: 1 dtype, 2 shape
{2 dim {1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c); the empty key holds the BundleHeaderProto.  Pure Python; no
TensorFlow, no protobuf runtime needed.

  python tools/parse_variables_index.py <variables.index> [out.json]

In the build container this is how tests/golden/cnn_variables_index.json was produced from the reference checkout
(dnn_models/detect_model_BrdUEdU_DNAr10_4_1/variables/variables.index: the only part of the SavedModel the checkout keeps).
"""
import json
import struct
import sys

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: "float32", 2: "float64", 3: "int32", 9: "int64", 7: "string", 10: "bool"}


def varint(b, i):
    x, s = 0, 0
    while True:
        c = b[i]; i += 1
        x |= (c & 0x7f) << s
        if c < 0x80:
            return x, i
        s += 7


def block(b, off, size):
    """entries of one block (uncompressed: type byte 0 after the contents)"""
    if b[off + size] != 0:
        raise ValueError("compressed block (type %d): not expected in a checkpoint index" % b[off + size])
    data = b[off:off + size]
    n_restarts = struct.unpack("<I", data[-4:])[0]
    end = len(data) - 4 - 4 * n_restarts
    i, key, out = 0, b"", []
    while i < end:
        shared, i = varint(data, i); non_shared, i = varint(data, i); vlen, i = varint(data, i)
        key = key[:shared] + data[i:i + non_shared]; i += non_shared
        out.append((key, data[i:i + vlen])); i += vlen
    return out


def proto_fields(b):
    i, out = 0, []
    while i < len(b):
        tag, i = varint(b, i)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 2:
            ln, i = varint(b, i); v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = struct.unpack("<I", b[i:i + 4])[0]; i += 4
        elif wt == 1:
            v = struct.unpack("<Q", b[i:i + 8])[0]; i += 8
        else:
            raise ValueError("wire type %d" % wt)
        out.append((f, v))
    return out


def parse_index(path):
    b = open(path, "rb").read()
    if struct.unpack("<Q", b[-8:])[0] != MAGIC:
        raise ValueError("not an SSTable: bad magic")
    foot = b[-48:]
    _, i = varint(foot, 0); _, i = varint(foot, i)          # metaindex handle
    ioff, i = varint(foot, i); isize, i = varint(foot, i)   # index handle
    entries = {}
    for _, handle in block(b, ioff, isize):
        off, j = varint(handle, 0); size, j = varint(handle, j)
        for key, val in block(b, off, size):
            if key == b"":
                continue                                     # BundleHeaderProto
            e = dict(dtype=None, shape=[], shard=0, offset=0, size=0)
            for f, v in proto_fields(val):
                if f == 1: e["dtype"] = DTYPES.get(v, str(v))
                elif f == 2: e["shape"] = [vv for ff, d in proto_fields(v) if ff == 2 for f3, vv in proto_fields(d) if f3 == 1]
                elif f == 3: e["shard"] = v
                elif f == 4: e["offset"] = v
                elif f == 5: e["size"] = v
            entries[key.decode()] = e
    return entries


if __name__ == "__main__":
    ent = parse_index(sys.argv[1])
    out = {"source": "dnn_models/detect_model_BrdUEdU_DNAr10_4_1/variables/variables.index", "generator": "tools/parse_variables_index.py",
           "entries": ent}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=0, sort_keys=True)
    f32 = sum(e["size"] for e in ent.values() if e["dtype"] == "float32")
    print("%d entries, %d float32 bytes (%d parameters)" % (len(ent), f32, f32 // 4))
