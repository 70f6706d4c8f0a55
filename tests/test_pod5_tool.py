"""tools/pod5_to_container.py end to end on CPU: a BAM written by the independent Python encoder (tests/bam_codec.py) + a FASTA + a STAND-IN `pod5` module (the
real package is not in this image: only its three calls the tool uses are imitated -- DatasetReader(files).get_read(id) -> .signal / .calibration.offset /
.calibration.scale; the stand-in keeps every signal as a VBZ chunk and decodes it through csrc/host/dn_vbz.cpp, the way libpod5 would) -> container -> ReadBatch.
Checked: the reference's record filter (detect.cpp:839), the parent look-up of a split read (pod5.cpp:79), the Dorado trimming applied when the batch is
built (pod5.cpp:75-93), and that what comes out is what went in."""
import importlib.util
import os
import sys
import types

import numpy as np

import bam_codec as bc
from dnascent_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("pod5_to_container", os.path.join(ROOT, "tools", "pod5_to_container.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake_pod5(signals):
    """signals: read id -> (int16 samples, offset, scale); stored VBZ-compressed, decoded on access"""
    chunks = {k: (host.vbz_encode(v[0]), v[0].shape[0], v[1], v[2]) for k, v in signals.items()}
    mod = types.ModuleType("pod5")

    class Rec:
        def __init__(self, c):
            self._c = c
            self.calibration = types.SimpleNamespace(offset=c[2], scale=c[3])

        @property
        def signal(self):
            return host.vbz_decode(self._c[0], self._c[1])

    class DatasetReader:
        def __init__(self, files):
            self.files = list(files)

        def get_read(self, read_id):
            return Rec(chunks[read_id]) if read_id in chunks else None
    mod.DatasetReader = DatasetReader
    return mod


def test_bam_fasta_pod5_to_container_to_batch(tmp_path, monkeypatch):
    rng = np.random.default_rng(5)
    bases = list("ACGT")
    ref = {"chrI": "".join(rng.choice(bases, 9000)), "chrII": "".join(rng.choice(bases, 4000))}

    def seq(n):
        return "".join(rng.choice(bases, n))
    recs = [dict(qname="whole", ref_id=0, pos=200, mapq=60, flag=0, cigar=[("S", 4), ("M", 1500), ("D", 3), ("M", 700)], seq=seq(4 + 2200), tags=[("ns", "i", 30000), ("ts", "i", 250)]),
            dict(qname="child", ref_id=1, pos=50, mapq=45, flag=16, cigar=[("M", 1800)], seq=seq(1800), tags=[("pi", "Z", "parent"), ("sp", "i", 5000), ("ns", "i", 21000), ("ts", "i", 100)]),
            dict(qname="low-mapq", ref_id=0, pos=10, mapq=5, flag=0, cigar=[("M", 1500)], seq=seq(1500), tags=[]),
            dict(qname="too-short", ref_id=0, pos=10, mapq=60, flag=0, cigar=[("M", 400)], seq=seq(400), tags=[]),
            dict(qname="unmapped", ref_id=-1, pos=-1, mapq=0, flag=4, cigar=[], seq=seq(50), tags=[]),
            dict(qname="no-signal", ref_id=0, pos=3000, mapq=60, flag=0, cigar=[("M", 1200)], seq=seq(1200), tags=[])]
    refs = [(k, len(v)) for k, v in ref.items()]
    bam = str(tmp_path / "calls.bam"); fasta = str(tmp_path / "genome.fasta"); out = str(tmp_path / "reads.dnc")
    bc.write_bam(bam, "@HD\tVN:1.6\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs), refs, recs, block_size=4000)
    with open(fasta, "w") as f:
        for k, v in ref.items():
            f.write(">%s some description\n" % k)
            for i in range(0, len(v), 60):
                f.write(v[i:i + 60].lower() + "\n")                                         # lower case: upper-cased on the way in
    signals = {"whole": (rng.integers(200, 1200, 31000).astype(np.int16), -240.0, 0.1755),
               "parent": (rng.integers(200, 1200, 40000).astype(np.int16), -237.0, 0.18)}    # "child" has no signal of its own
    (tmp_path / "p5").mkdir(); open(str(tmp_path / "p5" / "a.pod5"), "wb").close()
    monkeypatch.setitem(sys.modules, "pod5", _fake_pod5(signals))
    tool = _tool()
    assert tool.main(["--bam", bam, "--reference", fasta, "--pod5", str(tmp_path / "p5"), "--out", out, "-q", "20", "-l", "1000"]) == 0
    assert host.container_count(out) == 2
    assert host.container_sizes(out).tolist() == [31000, 40000]                            # untrimmed in the file
    b = host.ReadBatch()
    assert b.add_container(out) == 2
    d = b.desc()
    off = np.frombuffer((np.ctypeslib.ctypes.c_uint64 * 3).from_address(d.adc_off), np.uint64)
    # pod5.cpp:75-93: a whole read keeps [ts, ns), a split read [sp + ts, sp + ns) of its PARENT's signal
    assert (int(off[1] - off[0]), int(off[2] - off[1])) == (30000 - 250, 21000 - 100)
    adc = np.frombuffer((np.ctypeslib.ctypes.c_int16 * int(off[2])).from_address(d.adc), np.int16)
    assert np.array_equal(adc[:int(off[1])], signals["whole"][0][250:30000])
    assert np.array_equal(adc[int(off[1]):], signals["parent"][0][5000 + 100:5000 + 21000])
    cal = np.frombuffer((np.ctypeslib.ctypes.c_float * 2).from_address(d.cal_offset), np.float32)
    assert cal.tolist() == [-240.0, -237.0]


def test_without_the_pod5_package_the_tool_says_so(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "pod5", None)                                          # import pod5 -> ImportError
    tool = _tool()
    try:
        tool.Pod5Source([str(tmp_path)])
    except SystemExit as e:
        assert "pod5" in str(e) and "not installed" in str(e)
    else:
        raise AssertionError("expected SystemExit")
