"""CPU: the from-scratch BGZF + BAM reader / modbam writer (csrc/host/dn_bam.cpp; htslib is not in the image) against an INDEPENDENT Python encoder / decoder
of the SAM/BAM specification (tests/bam_codec.py).  What is checked is what the reference takes from a record -- reads.h:210-287 (qname, ns / ts / pi / sp,
CIGAR, strand, target, query sequence), htsInterface.cpp:160-180 (4-bit sequence codes; anything but A C G T N rejects the read), detect.cpp:839 (unmapped /
empty records) -- and what it writes back: reads.h:453-512 (MM / ML appended behind whatever MM / ML the record already had)."""
import ctypes as C
import os

import numpy as np
import pytest

import bam_codec as bc
from dnascent_amd import host


class Fields(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ref_id", "pos", "mapq", "flag", "l_seq", "n_cigar", "input_rc", "ref_start", "is_reverse", "signal_length",
                                         "signal_trim", "signal_start", "is_split")] + [("ref_len", C.c_int64)]


def _lib():
    L = host.lib()
    L.dnh_bam_open.restype = C.c_void_p; L.dnh_bam_open.argtypes = [C.c_char_p]
    L.dnh_bam_close.argtypes = [C.c_void_p]
    L.dnh_bam_add_reference.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.dnh_bam_header.restype = C.c_char_p; L.dnh_bam_header.argtypes = [C.c_void_p]
    L.dnh_bam_nref.argtypes = [C.c_void_p]
    L.dnh_bam_ref.restype = C.c_char_p; L.dnh_bam_ref.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
    L.dnh_bam_next.argtypes = [C.c_void_p, C.POINTER(Fields)]
    L.dnh_bam_str.restype = C.c_char_p; L.dnh_bam_str.argtypes = [C.c_void_p, C.c_int]
    L.dnh_bam_cigar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.dnh_bam_copy_with_mods.restype = C.c_long
    L.dnh_bam_copy_with_mods.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def _read_all(path, reference=None):
    L = _lib()
    h = L.dnh_bam_open(path.encode())
    assert h
    try:
        for name, seq in (reference or {}).items():
            L.dnh_bam_add_reference(h, name.encode(), seq.encode())
        n = C.c_uint32()
        refs = []
        for i in range(L.dnh_bam_nref(h)):
            nm = L.dnh_bam_ref(h, i, C.byref(n)).decode()
            refs.append((nm, n.value))
        text = L.dnh_bam_header(h).decode()
        out = []
        f = Fields()
        while True:
            rc = L.dnh_bam_next(h, C.byref(f))
            if rc != 1:
                break
            op = np.zeros(max(f.n_cigar, 1), np.uint32); ln = np.zeros(max(f.n_cigar, 1), np.uint32)
            L.dnh_bam_cigar(h, op.ctypes.data, ln.ctypes.data)
            out.append(dict({k: getattr(f, k) for k, _ in Fields._fields_}, qname=L.dnh_bam_str(h, 0).decode(), seq=L.dnh_bam_str(h, 1).decode(),
                            contig=L.dnh_bam_str(h, 2).decode(), fetch=L.dnh_bam_str(h, 3).decode(), query=L.dnh_bam_str(h, 4).decode(),
                            ref_slice=L.dnh_bam_str(h, 5).decode(), cigar=[(bc.CIGAR_OPS[int(o)], int(l)) for o, l in zip(op[:f.n_cigar], ln[:f.n_cigar])]))
        return text, refs, out, rc
    finally:
        L.dnh_bam_close(h)


def _records(rng):
    bases = "ACGT"
    ref = {"chrI": "".join(rng.choice(list(bases), 5000)), "chrII": "".join(rng.choice(list(bases), 3000))}
    recs = []

    def seq(n, alphabet=bases):
        return "".join(rng.choice(list(alphabet), n))
    # forward read with every consuming and non-consuming operation, Dorado tags as the integer types basecallers use
    recs.append(dict(qname="read-fwd", ref_id=0, pos=100, mapq=60, flag=0, cigar=[("S", 5), ("M", 50), ("I", 3), ("M", 20), ("D", 4), ("=", 10), ("X", 2), ("N", 7), ("M", 30), ("H", 9)],
                     seq=seq(5 + 50 + 3 + 20 + 10 + 2 + 30), tags=[("ns", "i", 123456), ("ts", "C", 17), ("NM", "c", -3), ("xx", "A", "q"), ("fl", "f", 1.5)]))
    # reverse, split read: parent id + start in the parent, odd sequence length, an N in the sequence, unsigned 16 / 32-bit tags
    recs.append(dict(qname="read-rev-split", ref_id=1, pos=7, mapq=33, flag=16, cigar=[("M", 41)], seq=seq(20) + "N" + seq(20),
                     tags=[("pi", "Z", "parent-read"), ("sp", "I", 4000000000 % 2 ** 31), ("ns", "S", 65000), ("ts", "s", 12), ("MM", "Z", "C+m,1,2;"), ("ML", "B", ("C", [10, 20]))]))
    # pi equal to the own id: not a split read (pod5.cpp:79 compares the ids); pi without sp
    recs.append(dict(qname="self-parent", ref_id=0, pos=0, mapq=1, flag=0, cigar=[("M", 12)], seq=seq(12), tags=[("pi", "Z", "self-parent")]))
    # unmapped, and a record with an IUPAC base: both unusable (detect.cpp:839; htsInterface.cpp:160-178 throws)
    recs.append(dict(qname="unmapped", ref_id=-1, pos=-1, mapq=0, flag=4, cigar=[], seq=seq(10), tags=[]))
    recs.append(dict(qname="iupac", ref_id=0, pos=10, mapq=60, flag=0, cigar=[("M", 8)], seq="ACGTRYAC", tags=[]))
    # beyond the end of its contig: not in the reference
    recs.append(dict(qname="off-the-end", ref_id=1, pos=2990, mapq=60, flag=0, cigar=[("M", 30)], seq=seq(30), tags=[]))
    # more than 65 535 CIGAR operations: the CG tag (a 200 kb nanopore read with an indel every other base reaches that)
    big = []
    for i in range(33000):
        big += [("M", 1), ("I", 1)] if i & 1 else [("M", 1), ("D", 1)]
    nq = sum(n for o, n in big if o in "MIS=X")
    recs.append(dict(qname="long-cigar", ref_id=0, pos=3, mapq=60, flag=16, cigar=big, seq=seq(nq), tags=[("ns", "i", 99)]))
    ref["chrI"] = ref["chrI"] + seq(70000)                  # the long read needs reference under it
    return ref, recs


def test_reader_against_the_python_encoder(tmp_path):
    rng = np.random.default_rng(7)
    ref, recs = _records(rng)
    refs = [(k, len(v)) for k, v in ref.items()]
    header = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    for block in (0xff00, 333):                             # ordinary blocks, and blocks so small that every record straddles several
        path = str(tmp_path / ("t%d.bam" % block))
        bc.write_bam(path, header, refs, recs, block_size=block)
        text, got_refs, got, rc = _read_all(path, ref)
        assert rc == 0 and text == header and got_refs == refs and len(got) == len(recs)
        for g, r in zip(got, recs):
            assert g["qname"] == r["qname"] and g["ref_id"] == r["ref_id"] and g["pos"] == r["pos"] and g["mapq"] == r["mapq"] and g["flag"] == r["flag"]
            assert g["seq"] == r["seq"] and g["l_seq"] == len(r["seq"])
            assert g["cigar"] == r["cigar"]                 # the CG tag resolved
            assert g["ref_len"] == sum(n for o, n in r["cigar"] if o in "MDN=X")
        by = {g["qname"]: g for g in got}
        a = by["read-fwd"]
        assert a["input_rc"] == 0 and (a["signal_length"], a["signal_trim"], a["signal_start"], a["is_split"], a["is_reverse"]) == (123456, 17, 0, 0, 0)
        assert a["contig"] == "chrI" and a["ref_start"] == 100 and a["fetch"] == "read-fwd" and a["query"] == recs[0]["seq"]
        assert a["ref_slice"] == ref["chrI"][100:100 + a["ref_len"]]
        b = by["read-rev-split"]
        assert b["input_rc"] == 0 and b["is_reverse"] == 1 and b["is_split"] == 1 and b["fetch"] == "parent-read" and b["signal_start"] == 4000000000 % 2 ** 31
        assert (b["signal_length"], b["signal_trim"]) == (65000, 12) and b["ref_slice"] == ref["chrII"][7:48] and "N" in b["query"]
        c = by["self-parent"]
        assert c["input_rc"] == 0 and c["is_split"] == 0 and c["fetch"] == "self-parent" and c["signal_length"] == -1
        assert by["unmapped"]["input_rc"] == -1 and by["iupac"]["input_rc"] == -3 and by["off-the-end"]["input_rc"] == -2
        lc = by["long-cigar"]
        assert lc["input_rc"] == 0 and lc["n_cigar"] == 66000 and lc["signal_length"] == 99 and lc["ref_slice"] == ref["chrI"][3:3 + lc["ref_len"]]


def test_malformed_input_is_an_error_not_a_crash(tmp_path):
    rng = np.random.default_rng(8)
    ref, recs = _records(rng)
    refs = [(k, len(v)) for k, v in ref.items()]
    path = str(tmp_path / "ok.bam")
    bc.write_bam(path, "@HD\tVN:1.6\n", refs, recs[:3], block_size=120)      # small blocks: header and records in blocks of their own
    raw = open(path, "rb").read()
    L = _lib()
    trunc = str(tmp_path / "trunc.bam")
    open(trunc, "wb").write(raw[:len(raw) // 2])
    text, got_refs, got, rc = _read_all(trunc)
    assert rc == -1                                         # a truncated file ends in an error, not in a silent short read
    flip = bytearray(raw); flip[2 * len(raw) // 3] ^= 0x55
    bad = str(tmp_path / "flip.bam")
    open(bad, "wb").write(bytes(flip))
    h = L.dnh_bam_open(bad.encode())
    if h:                                                   # the flipped byte may sit in the header block (open fails) or in a record block (next fails)
        f = Fields(); rcs = []
        for _ in range(5):
            rcs.append(L.dnh_bam_next(h, C.byref(f)))
            if rcs[-1] != 1:
                break
        L.dnh_bam_close(h)
        assert rcs[-1] == -1
    assert not L.dnh_bam_open(str(tmp_path / "absent.bam").encode())
    notbam = str(tmp_path / "not.bam")
    open(notbam, "wb").write(bc.bgzf_compress(b"SAM\1 nothing here"))
    assert not L.dnh_bam_open(notbam.encode())


def test_modbam_writer_against_the_python_decoder(tmp_path):
    """reads.h:453-512: MM / ML of the base-analogue calls appended behind whatever MM / ML the record carried, both at the END of the record; records
    without calls pass through byte for byte.  The MM fields and ML bytes come from the tested host.modbam (reads.h:469-487 deltas, p * 255 truncated)."""
    rng = np.random.default_rng(9)
    ref, recs = _records(rng)
    refs = [(k, len(v)) for k, v in ref.items()]
    header = "@HD\tVN:1.6\n@PG\tID:test\n"
    src = str(tmp_path / "src.bam"); dst = str(tmp_path / "dst.bam")
    bc.write_bam(src, header, refs, recs, block_size=4000)
    mods = {0: ("N+b?,3,0,12;N+e?,3,0,12;", [200, 0, 17, 1, 255, 3]), 1: ("N+b?,5;N+e?,5;", [128, 64]), 6: ("N+b?;N+e?;", [])}
    idx = np.array(sorted(mods), np.uint64)
    mm = (C.c_char_p * len(idx))(*[mods[int(i)][0].encode() for i in idx])
    mls = [np.array(mods[int(i)][1], np.uint8) for i in idx]
    ml = (C.c_void_p * len(idx))(*[m.ctypes.data if m.size else None for m in mls])
    mll = np.array([m.size for m in mls], np.uint64)
    n = _lib().dnh_bam_copy_with_mods(src.encode(), dst.encode(), idx.ctypes.data, len(idx), mm, ml, mll.ctypes.data)
    assert n == len(recs)
    text, got_refs, got = bc.read_bam(dst)
    _, _, orig = bc.read_bam(src)
    assert text == header and got_refs == refs and len(got) == len(orig)
    for i, (g, o) in enumerate(zip(got, orig)):
        for k in ("qname", "ref_id", "pos", "mapq", "flag", "cigar", "seq", "qual"):
            assert g[k] == o[k], (i, k)
        if i not in mods:
            assert g["tags"] == o["tags"]
            continue
        keep = [t for t in o["tags"] if t[0] not in ("MM", "ML")]
        old_mm = "".join(t[2] for t in o["tags"] if t[0] == "MM")
        old_ml = [v for t in o["tags"] if t[0] == "ML" for v in t[2][1]]
        assert g["tags"][:-2] == keep
        assert g["tags"][-2] == ("MM", "Z", old_mm + mods[i][0])
        assert g["tags"][-1] == ("ML", "B", ("C", old_ml + mods[i][1]))
    # and the copy reads back through the C++ reader: same records, the long CIGAR still resolved
    _, _, again, rc = _read_all(dst, ref)
    assert rc == 0 and [a["qname"] for a in again] == [r["qname"] for r in recs] and again[6]["n_cigar"] == 66000
