"""SURVEY s8(f).4 -- the `.detect` text this framework writes must be readable by the reference's own consumers.  Their
parsers are restated here line by line (the reference sources need htslib / TensorFlow to compile) and fed a file produced by
the host C++ writer (header + records of forward / reverse / indel reads):

  forkSense.cpp:1459-1536   callFractions_HR          '#'/blank skipped, '>' starts a read, data = TAB columns: pos, EdU, BrdU
  forkSense.cpp:1618-1706   iterateOnHumanReadable    '>' line = exactly 5 SPACE-separated fields (else DetectParsing)
  seeBreaks.cpp:164-190     detectUnpack              '>' line splits into exactly 5 columns (assert), [2] / [3] integers
  utils/dnascent2bedgraph.py:91-99  makeDetectLine    whitespace split: [0] int position, [1] EdU, [2] BrdU

Round trip: what the parsers recover (read ids, contigs, coordinates, strands, positions, probabilities) is what went in,
to the 6 decimals of "%f".  CPU only."""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import host, synth

SPECS = [(401, 2400, dict()), (402, 2600, dict(is_reverse=True)), (403, 2500, dict(sub_rate=0.003, ins_rate=0.002, del_rate=0.002))]


class DetectParsing(Exception):
    pass


def call_fractions_hr(text):                                # forkSense.cpp:1459-1536 (counting part)
    reads = 0; rows = []
    for line in text.split("\n"):
        if line[:1] == "#" or len(line) == 0:
            continue
        if line[:1] == ">":
            reads += 1
            continue
        position, E, B = -1, 0.0, 0.0
        for c, column in enumerate(line.split("\t")):
            if c == 0: position = int(column)
            elif c == 1: E = float(np.float32(column))      # std::stof
            elif c == 2: B = float(np.float32(column))
        rows.append((position, E, B))
    return reads, rows


def iterate_on_human_readable(text):                         # forkSense.cpp:1618-1706
    reads = []; cur = None
    for line in text.split("\n"):
        if line[:1] == "#" or len(line) == 0:
            continue
        if line[:1] == ">":
            cols = line.split(" ")
            if len(cols) != 5:
                raise DetectParsing(line)                    # the 6th column throws in the reference
            cur = dict(readID=cols[0], chromosome=cols[1], lower=int(cols[2]), upper=int(cols[3]), strand=cols[4], pos=[], B=[], E=[])
            reads.append(cur)
        else:
            position, E, B = -1, 0.0, 0.0
            for c, column in enumerate(line.split("\t")):
                if c == 0: position = int(column)
                elif c == 1: E = float(np.float32(column))
                elif c == 2: B = float(np.float32(column))
            assert position != -1                            # :1696
            cur["pos"].append(position); cur["B"].append(B); cur["E"].append(E)
    return reads


def detect_unpack(text):                                     # seeBreaks.cpp:164-190
    out = []
    for line in text.split("\n"):
        if line[:1] == "#" or len(line) == 0:
            continue
        if line[:1] == ">":
            columns = line.split()
            assert len(columns) == 5                         # "well-formed detect header"
            out.append((int(columns[2]), int(columns[3])))
    return out


def make_detect_line(line, chromosome):                      # utils/dnascent2bedgraph.py:91-99
    s = line.rstrip().split()
    return int(s[0]), float(s[2]), float(s[1])               # pos, BrdU, EdU


def detect_text_for_fixture(model):
    """The .detect text of the committed fixture (tests/golden/bedgraph_input.detect): header + the records of SPECS as the host
    writer formats them from the oracle's positions and seeded probabilities.  Deterministic; CPU only."""
    text = host.detect_header("aln.bam", "genome.fa", "index.dnascent", 4, 20, 1000, False, "2026-01-01 00:00:00", "/opt/dnascent", "4.1.1", "deadbeef").decode()
    for seed, n, kw in SPECS:
        sr = synth.make_read(seed, n, model=model, **kw)
        o = po.OracleRead(sr, model)
        assert o.normalise() == 0 and o.eventalign() == 0
        pr = np.random.default_rng(seed).dirichlet((1.0, 1.0, 1.0), int(o.align.n_pos)).astype(np.float32)
        pos = o.positions()
        text += host.format_detect(sr.read_id, sr.contig, sr.ref_start, sr.ref_end, sr.is_reverse, pos["coord"], pos["kmer"], pr).decode()
        o.free()
    return text


def bedgraphs_restated(text):
    """utils/dnascent2bedgraph.py parseBaseFile (:107-260) for a detect file with default arguments: per read two bedgraph files
    (BrdU, EdU) in directory 1, a track line + one line per data line: chromosome pos pos+1 probability (Python float repr)."""
    files, cur, cid, chrom = {}, None, None, None

    def flush():
        if cid is None:
            return
        head = 'track type=bedGraph name="%s" description="BedGraph format" visibility=full color=%s altColor=0,100,200 priority=20 viewLimits=0.0:1.0\n'
        files["1/%s.BrdUdetect.bedgraph" % cid] = head % (cid, "200,100,0") + "".join(l[0] for l in cur)
        files["1/%s.EdUdetect.bedgraph" % cid] = head % (cid, "93,197,186") + "".join(l[1] for l in cur)
    for line in text.split("\n"):
        if not line.rstrip() or line[0] == "#":
            continue
        if line[0] == ">":
            flush()
            cols = line.rstrip().split(" ")
            cid, chrom, cur = cols[0][1:], cols[1], []
            continue
        pos, bd, ed = make_detect_line(line, chrom)
        cur.append(("%s %d %d %s\n" % (chrom, pos, pos + 1, str(bd)), "%s %d %d %s\n" % (chrom, pos, pos + 1, str(ed))))
    flush()
    return files


def test_bedgraphs_match_what_the_reference_script_extracted(model):
    """tests/golden/bedgraph_expected.json holds every file the REFERENCE's utils/dnascent2bedgraph.py wrote when it was run (in
    the build container, tests/golden/make_bedgraph_golden.py) on tests/golden/bedgraph_input.detect -- a file written by this
    framework's host layer.  Here: the writer still produces that file byte for byte, and the restated parser extracts exactly
    what the reference's parser extracted."""
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    committed = open(os.path.join(g, "bedgraph_input.detect")).read()
    assert detect_text_for_fixture(model) == committed
    want = json.load(open(os.path.join(g, "bedgraph_expected.json")))["files"]
    got = bedgraphs_restated(committed)
    assert sorted(got) == sorted(want) and len(want) == 2 * len(SPECS)
    for k in want:
        assert got[k] == want[k], k


def test_detect_file_round_trips_through_the_reference_consumers(model, tmp_path):
    text = host.detect_header("aln.bam", "genome.fa", "index.dnascent", 4, 20, 1000, False, "2026-01-01 00:00:00", "/opt/dnascent", "4.1.1", "deadbeef").decode()
    truth = []
    for seed, n, kw in SPECS:
        sr = synth.make_read(seed, n, model=model, **kw)
        o = po.OracleRead(sr, model)
        assert o.normalise() == 0 and o.eventalign() == 0
        pr = np.random.default_rng(seed).dirichlet((1.0, 1.0, 1.0), int(o.align.n_pos)).astype(np.float32)
        pos = o.positions()
        rec = host.format_detect(sr.read_id, sr.contig, sr.ref_start, sr.ref_end, sr.is_reverse, pos["coord"], pos["kmer"], pr).decode()
        text += rec
        # what the record should carry: thymidine positions only (detect.cpp:711), ascending in the file
        keep = [(int(c), float(p[2]), float(p[1])) for c, k, p in zip(pos["coord"], pos["kmer"], pr) if bytes(k)[4:5] == b"T"]
        truth.append((sr, sorted(keep)))
        o.free()
    assert text.startswith("#")
    # ---- forkSense: read count, per-read header fields, per-line columns ----
    n_reads, rows = call_fractions_hr(text)
    assert n_reads == len(SPECS) and len(rows) == sum(len(t[1]) for t in truth)
    parsed = iterate_on_human_readable(text)
    assert len(parsed) == len(SPECS)
    for got, (sr, want) in zip(parsed, truth):
        assert got["readID"] == ">" + sr.read_id and got["chromosome"] == sr.contig
        assert (got["lower"], got["upper"]) == (sr.ref_start, sr.ref_end) and got["strand"] == ("rev" if sr.is_reverse else "fwd")
        assert got["pos"] == [w[0] for w in want]
        assert np.allclose(got["E"], [w[1] for w in want], atol=5.1e-7) and np.allclose(got["B"], [w[2] for w in want], atol=5.1e-7)
    # ---- seeBreaks ----
    assert detect_unpack(text) == [(sr.ref_start, sr.ref_end) for sr, _ in truth]
    # ---- dnascent2bedgraph ----
    data = [l for l in text.split("\n") if l and l[0] not in "#>"]
    first = make_detect_line(data[0], truth[0][0].contig)
    assert first[0] == truth[0][1][0][0] and abs(first[1] - truth[0][1][0][2]) < 5.1e-7 and abs(first[2] - truth[0][1][0][1]) < 5.1e-7
    # a sixth header column is what the reference rejects: make sure the check has teeth
    with pytest.raises(DetectParsing):
        iterate_on_human_readable(text.replace(" fwd\n", " fwd extra\n", 1))
