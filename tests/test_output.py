"""Output side of runCNN (detect.cpp:677-731, reads.h:453-512): host C++ formatting vs the oracle restatement.

CPU only: the positions come from the ORACLE's eventalign of small synthetic reads (forward / reverse / indels), the
probabilities are random, so the whole text path -- T filter, EdU / BrdU column order, "%f" formatting, reverse-complement
and line reversal for reverse reads, modbam deltas / deletions / byte quantisation -- is exercised without a GPU."""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import host, synth

SPECS = [
    (301, 2500, dict()),
    (302, 2500, dict(is_reverse=True)),
    (303, 2600, dict(sub_rate=0.003, ins_rate=0.001, del_rate=0.002)),
    (304, 2600, dict(is_reverse=True, sub_rate=0.003, ins_rate=0.001, del_rate=0.002, soft_clip_head=12, soft_clip_tail=9)),
]


@pytest.fixture(scope="module")
def aligned(model):
    out = []
    for seed, n, kw in SPECS:
        sr = synth.make_read(seed, n, model=model, **kw)
        o = po.OracleRead(sr, model)
        assert o.normalise() == 0
        assert o.eventalign() == 0
        rng = np.random.default_rng(seed)
        pr = rng.dirichlet((1.0, 1.0, 1.0), int(o.align.n_pos)).astype(np.float32)
        pr[::7] = np.float32([1.0, 0.0, 0.0])                       # exact 0 / 1 and tiny values through "%f"
        pr[3::11] = np.float32([0.0, 1.0, 0.0])
        pr[5::13] = np.float32([1e-7, 0.9999995, 4e-7])
        out.append((sr, o, pr))
    yield out
    for _, o, _ in out:
        o.free()


def test_detect_record_matches_oracle(aligned):
    for sr, o, pr in aligned:
        pos = o.positions()
        want = o.format_detect(pr)
        got = host.format_detect(sr.read_id, sr.contig, sr.ref_start, sr.ref_end, sr.is_reverse, pos["coord"], pos["kmer"], pr)
        assert got == want
        lines = got.decode().splitlines()
        assert lines[0] == ">%s %s %d %d %s" % (sr.read_id, sr.contig, sr.ref_start, sr.ref_end, "rev" if sr.is_reverse else "fwd")
        coords = [int(l.split("\t")[0]) for l in lines[1:]]
        assert coords == sorted(coords) and len(coords) > 100        # both strands ascend in the file (detect.cpp:722)
        for l in lines[1:20]:
            f = l.split("\t")
            assert len(f) == 4 and len(f[3]) == 9 and len(f[1].split(".")[1]) == 6
            assert f[3][4] == ("A" if sr.is_reverse else "T")         # reverse reads print the reverse complement


def test_modbam_fields_match_oracle(aligned):
    for sr, o, pr in aligned:
        pos = o.positions()
        n_want, mm_want, ml_want = o.modbam(pr)
        n_got, mm_got, ml_got = host.modbam(pos["query_idx"], pos["ref_idx"], pos["kmer"], pr, o.r2d)
        assert n_got == n_want and n_got > 100
        assert mm_got == mm_want
        assert np.array_equal(ml_got, ml_want)
        b, e = mm_got.rstrip(";").split(";")
        assert b.startswith("N+b?,") and e.startswith("N+e?,") and b[4:] == e[4:]
        assert len(b[5:].split(",")) == n_got and ml_got.shape[0] == 2 * n_got


def test_empty_read():
    got = host.format_detect("r", "chr", 5, 9, False, np.zeros(0, np.uint32), np.zeros(0, "S9"), np.zeros((0, 3), np.float32))
    assert got == b">r chr 5 9 fwd\n"
    n, mm, ml = host.modbam(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, "S9"), np.zeros((0, 3), np.float32), np.zeros(1, np.uint8))
    assert n == 0 and mm == "N+b?;N+e?;" and ml.shape[0] == 0


def test_header_matches_the_documented_example():
    """docs/source/detect.rst:44-57 of the reference prints a complete .detect header; our writeDetectHeader, fed the values
    of that example, reproduces it byte for byte (tests/golden/doc_detect_header.txt holds the documented lines)."""
    import os
    want = open(os.path.join(os.path.dirname(__file__), "golden", "doc_detect_header.txt"), "rb").read()
    got = host.detect_header("/path/to/alignment.bam", "/path/to/reference.fasta", "/path/to/index.dnascent", 1, 20, 5000, False,
                             "09/02/2024 12:45:29", "/path/to/DNAscent", "4.0.3", "4cf80a7b89bdf510a91b54572f8f94d3daf9b167")
    assert got == want
    # the documented record lines: ">readID contig start end strand", then "coord<TAB>p<TAB>p" with six decimals (:88-97);
    # the code adds the strand 9-mer as a fourth column (detect.cpp:698-701)
    import re
    rec = host.format_detect("a4ea2872-9cb6-4218-afad-905f79204eb1", "14", 992440, 996846, True, np.array([992448], np.uint32),
                             np.array([b"ACGTTCGTA"], "S9"), np.array([[0.7, 0.131483, 0.125751]], np.float32))
    lines = rec.decode().splitlines()
    assert lines[0] == ">a4ea2872-9cb6-4218-afad-905f79204eb1 14 992440 996846 rev"
    assert re.fullmatch(r"992448\t0\.125751\t0\.131483\t[ACGT]{9}", lines[1])


def test_align_record_matches_oracle(aligned, model):
    """`DNAscent align` record text (alignment.cpp:553, :697-733): host C++ formatter vs the oracle's, from the oracle's own table."""
    for sr, o, _ in aligned:
        t = o.align_table()
        want = o.format_align()
        got = host.format_align(sr.read_id, sr.contig, sr.ref_start, sr.ref_end, sr.is_reverse, sr.refseq.tobytes(), model, t)
        assert got == want
        lines = got.decode().splitlines()
        assert lines[0] == ">%s %s %d %d %s" % (sr.read_id, sr.contig, sr.ref_start, sr.ref_end, "rev" if sr.is_reverse else "fwd")
        assert len(lines) == 1 + t["coord"].shape[0]
        f = lines[1].split("\t")
        assert len(f) == 5 and len(f[1]) == 9 and len(f[3]) == 9 and len(f[2].split(".")[1]) == 6
        ins = [l for l in lines[1:] if l.endswith("NNNNNNNNN\t0")]
        assert len(ins) == int((t["kind"] == 1).sum())


def test_fast_probability_format_equals_printf():
    """The record formatter prints probabilities without snprintf (p * 1e6 is exact in double for a float p, so rint() is
    glibc's correctly rounded "%f").  Every float next to a 6-decimal rounding boundary, random floats and the special values
    (negative zero, out of range, NaN, inf) must give the same text as printf("%f")."""
    import ctypes as C
    L = host.lib()
    L.dnh_check_prob_format.restype = C.c_uint64
    L.dnh_check_prob_format.argtypes = [C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(1)
    near = ((np.arange(0, 1000001) + 0.5) * 1e-6).astype(np.float32)          # every boundary k + 0.5 ppm, and its float neighbours
    cand = np.concatenate([near, np.nextafter(near, np.float32(0)), np.nextafter(near, np.float32(2)), rng.random(1000000, dtype=np.float32),
                           np.array([0, 1, -0.0, 1e-7, 5e-7, 4.9999997e-7, 0.9999995, 0.99999994, 1.0000001, 2.5, -1, np.nan, np.inf, 1e-30,
                                     0.5, 0.1234565, 0.0000005, 0.0000015, 0.0000025], np.float32)])
    cand = np.ascontiguousarray(cand)
    assert L.dnh_check_prob_format(cand.ctypes.data, cand.shape[0]) == 0
