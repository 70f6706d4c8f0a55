"""An independent Python encoder / decoder of BGZF + BAM, written from the SAM/BAM specification (SAMv1 s4): the counterpart the C++ reader / writer of
csrc/host/dn_bam.cpp is tested against (tests/test_bam.py).  Test infrastructure only; nothing here is shared with the C++ code.

    write_bam(path, header_text, refs, records, block_size)     records: dicts (qname, ref_id, pos, mapq, flag, cigar [(op char, len)], seq, qual, tags)
    read_bam(path) -> (header_text, refs, records)              tags come back as [(tag, type char, value)] in file order
tags: (tag, type, value) with type in A c C s S i I f Z H, or B with value (subtype, [values]).  A CIGAR of more than 65 535 operations is stored as the
spec says: <l_seq>S<ref_len>N in the CIGAR field + the real operations in CG:B,I (write_bam does this itself, read_bam hands back what is in the file)."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    assert bsize < 65536
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bgzf_compress(payload, block_size=0xff00):
    out = []
    for a in range(0, len(payload), block_size):
        out.append(_bgzf_block(payload[a:a + block_size]))
    out.append(_bgzf_block(b""))                            # EOF marker
    return b"".join(out)


def bgzf_decompress(raw):
    out, o = [], 0
    while o < len(raw):
        id1, id2, cm, flg, _mt, _xfl, _os, xlen = struct.unpack_from("<BBBBIBBH", raw, o)
        assert (id1, id2, cm) == (31, 139, 8) and flg & 4
        x, bsize = o + 12, None
        while x < o + 12 + xlen:
            si1, si2, slen = struct.unpack_from("<BBH", raw, x)
            if (si1, si2) == (66, 67):
                bsize = struct.unpack_from("<H", raw, x + 4)[0]
            x += 4 + slen
        assert bsize is not None
        cdata = raw[o + 12 + xlen:o + bsize + 1 - 8]
        crc, isize = struct.unpack_from("<II", raw, o + bsize + 1 - 8)
        data = zlib.decompress(cdata, -15) if isize else b""
        assert len(data) == isize and (zlib.crc32(data) & 0xffffffff) == crc
        out.append(data)
        o += bsize + 1
    return b"".join(out)


def _aux(tag, typ, val):
    b = tag.encode()
    if typ == "A":
        return b + b"A" + val.encode()
    if typ in "cCsSiIf":
        return b + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ], val)
    if typ in "ZH":
        return b + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub, vals = val
        return b + b"B" + sub.encode() + struct.pack("<I", len(vals)) + b"".join(
            struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub], v) for v in vals)
    raise ValueError(typ)


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def encode_record(r):
    cigar = list(r["cigar"])
    tags = list(r.get("tags", []))
    seq = r["seq"]
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    if len(cigar) > 65535:                                   # SAMv1 s4.2.2
        tags.append(("CG", "B", ("I", [n << 4 | CIGAR_OPS.index(op) for op, n in cigar])))
        cigar = [("S", len(seq)), ("N", ref_len)]
    name = r["qname"].encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i // 2] |= SEQ_CODES.index(ch) << (0 if i & 1 else 4)
    qual = bytes(r.get("qual", [0xff] * len(seq)))
    body = struct.pack("<iiBBHHHiiii", r["ref_id"], r["pos"], len(name), r["mapq"], _reg2bin(r["pos"], r["pos"] + max(ref_len, 1)), len(cigar), r["flag"],
                       len(seq), -1, -1, 0) + name + b"".join(struct.pack("<I", n << 4 | CIGAR_OPS.index(op)) for op, n in cigar) + bytes(packed) + qual + \
        b"".join(_aux(*t) for t in tags)
    return struct.pack("<I", len(body)) + body


def write_bam(path, header_text, refs, records, block_size=0xff00):
    payload = b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", len(refs))
    for name, ln in refs:
        payload += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    payload += b"".join(encode_record(r) for r in records)
    with open(path, "wb") as f:
        f.write(bgzf_compress(payload, block_size))


def _parse_aux(b):
    tags, o = [], 0
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}
    while o < len(b):
        tag, typ = b[o:o + 2].decode(), chr(b[o + 2]); o += 3
        if typ == "A":
            tags.append((tag, typ, chr(b[o]))); o += 1
        elif typ in fmt:
            n = struct.calcsize(fmt[typ]); tags.append((tag, typ, struct.unpack_from("<" + fmt[typ], b, o)[0])); o += n
        elif typ in "ZH":
            e = b.index(b"\0", o); tags.append((tag, typ, b[o:e].decode())); o = e + 1
        elif typ == "B":
            sub = chr(b[o]); cnt = struct.unpack_from("<I", b, o + 1)[0]; n = struct.calcsize(fmt[sub])
            tags.append((tag, typ, (sub, list(struct.unpack_from("<%d%s" % (cnt, fmt[sub]), b, o + 5))))); o += 5 + cnt * n
        else:
            raise ValueError(typ)
    return tags


def read_bam(path):
    p = bgzf_decompress(open(path, "rb").read())
    assert p[:4] == b"BAM\1"
    lt = struct.unpack_from("<I", p, 4)[0]
    text = p[8:8 + lt].decode()
    o = 8 + lt
    nref = struct.unpack_from("<I", p, o)[0]; o += 4
    refs = []
    for _ in range(nref):
        ln = struct.unpack_from("<I", p, o)[0]; o += 4
        name = p[o:o + ln - 1].decode(); o += ln
        refs.append((name, struct.unpack_from("<I", p, o)[0])); o += 4
    recs = []
    while o < len(p):
        bs = struct.unpack_from("<I", p, o)[0]; o += 4
        b = p[o:o + bs]; o += bs
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl = struct.unpack_from("<iiBBHHHiiii", b, 0)
        q = 32
        qname = b[q:q + l_name - 1].decode(); q += l_name
        cigar = [(CIGAR_OPS[c & 15], c >> 4) for c in struct.unpack_from("<%dI" % n_cig, b, q)]; q += 4 * n_cig
        seq = "".join(SEQ_CODES[(b[q + i // 2] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq)); q += (l_seq + 1) // 2
        qual = list(b[q:q + l_seq]); q += l_seq
        recs.append(dict(qname=qname, ref_id=ref_id, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual, tags=_parse_aux(b[q:])))
    return text, refs, recs
