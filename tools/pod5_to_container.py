"""BAM + reference FASTA + POD5 signal -> the binary read container `python -m dnascent_amd.run_detect` ingests.

The reference reads these three through htslib and libpod5 (detect.cpp:821-872, reads.h:210-287, pod5.cpp:24-105); this image has neither, so:
  * the BAM half runs on this repository's own reader (csrc/host/dn_bam.cpp: BGZF + records + Dorado tags ns / ts / pi / sp over zlib), with the reference's
    record filter `mapQ >= q && refEnd - refStart >= l && l_qseq != 0` (detect.cpp:839);
  * the SIGNAL half needs the `pod5` Python package of the host this tool runs on (pip install pod5): `pod5.DatasetReader(paths).get_read(read_id)` gives the
    int16 samples and the calibration that pod5.cpp:57-61 takes from libpod5.  Where the package is absent the tool says so and stops -- the Arrow IPC container
    of a POD5 file is not parsed here (csrc/host/dn_vbz.cpp holds only the column's codec).
A split read (BAM tag pi = the parent's id) fetches its PARENT's signal (pod5.cpp:79-86); the trimming by ts / ns / sp happens later, on the device's host
side, exactly as pod5.cpp:75-93 does it -- the container stores the untrimmed signal plus the three tags.

    python tools/pod5_to_container.py --bam calls.bam --reference genome.fasta --pod5 pod5_dir [more.pod5 ...] --out reads.dnc [-q 20] [-l 1000]

TESTED HERE only against a stand-in `pod5` module (tests/test_pod5_tool.py): BAM reading, filtering, parent look-up and the container are real, the package is not.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def read_fasta(path):
    """name (up to the first blank) -> upper-case sequence (pfasta semantics of the reference's import_reference: data_IO.cpp:44-88)"""
    ref, name, parts = {}, None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    ref[name] = "".join(parts).upper()
                name, parts = line[1:].split()[0], []
            elif name is not None:
                parts.append(line.strip())
    if name is not None:
        ref[name] = "".join(parts).upper()
    return ref


class Pod5Source:
    """read id -> (int16 samples, calibration offset, calibration scale) through the `pod5` package"""

    def __init__(self, paths):
        try:
            import pod5
        except ImportError:
            raise SystemExit("pod5_to_container: the `pod5` Python package is not installed on this host (pip install pod5): it is what reads the Arrow IPC tables of a "
                             "POD5 file; this repository only carries the signal column's codec (csrc/host/dn_vbz.cpp)")
        files = []
        for p in paths:
            if os.path.isdir(p):
                files += sorted(os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs if f.endswith(".pod5"))
            else:
                files.append(p)
        if not files:
            raise SystemExit("pod5_to_container: no .pod5 file under %s" % ", ".join(paths))
        self.reader = pod5.DatasetReader(files)

    def get(self, read_id):
        r = self.reader.get_read(read_id)
        if r is None:
            return None
        return np.ascontiguousarray(r.signal, np.int16), float(r.calibration.offset), float(r.calibration.scale)


class _BamFields(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ref_id", "pos", "mapq", "flag", "l_seq", "n_cigar", "input_rc", "ref_start", "is_reverse", "signal_length", "signal_trim",
                                          "signal_start", "is_split")] + [("ref_len", C.c_int64)]


def convert(bam, reference, source, out, min_mapq=20, min_len=1000, log=sys.stderr):
    """-> dict of counters.  `source`: an object with get(read_id) -> (adc, offset, scale) or None"""
    from dnascent_amd import host
    L = host.lib()
    L.dnh_bam_open.restype = C.c_void_p; L.dnh_bam_open.argtypes = [C.c_char_p]
    L.dnh_bam_close.argtypes = [C.c_void_p]
    L.dnh_bam_add_reference.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.dnh_bam_next.argtypes = [C.c_void_p, C.POINTER(_BamFields)]
    L.dnh_bam_str.restype = C.c_char_p; L.dnh_bam_str.argtypes = [C.c_void_p, C.c_int]
    L.dnh_bam_cigar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.dnh_container_create.restype = C.c_void_p; L.dnh_container_create.argtypes = [C.c_char_p]
    L.dnh_container_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                    C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    L.dnh_container_close.argtypes = [C.c_void_p]
    h = L.dnh_bam_open(bam.encode())
    if not h:
        raise SystemExit("pod5_to_container: cannot open %s as a BAM file" % bam)
    for name, seq in reference.items():
        L.dnh_bam_add_reference(h, name.encode(), seq.encode())
    w = L.dnh_container_create(out.encode())
    if not w:
        raise SystemExit("pod5_to_container: cannot create %s" % out)
    n = dict(records=0, filtered=0, unusable=0, no_signal=0, written=0, samples=0)
    f = _BamFields()
    while True:
        rc = L.dnh_bam_next(h, C.byref(f))
        if rc == 0:
            break
        if rc < 0:
            raise SystemExit("pod5_to_container: %s is truncated or malformed (record %d)" % (bam, n["records"]))
        n["records"] += 1
        if not (f.mapq >= min_mapq and f.ref_len >= min_len and f.l_seq != 0):            # detect.cpp:839
            n["filtered"] += 1
            continue
        if f.input_rc != 0:                                                                  # unmapped / contig not in the FASTA / a base outside A C G T N
            n["unusable"] += 1
            continue
        qname, contig, fetch = (L.dnh_bam_str(h, k).decode() for k in (0, 2, 3))
        sig = source.get(fetch)
        if sig is None or sig[0].shape[0] == 0:                                              # pod5.cpp:64: an empty signal ends the reference; here the read is skipped and said
            n["no_signal"] += 1
            print("pod5_to_container: no signal for %s (fetched as %s)" % (qname, fetch), file=log)
            continue
        adc, off, scale = sig
        query, ref_slice = L.dnh_bam_str(h, 4), L.dnh_bam_str(h, 5)
        op = np.zeros(max(f.n_cigar, 1), np.uint32); ln = np.zeros(max(f.n_cigar, 1), np.uint32)
        L.dnh_bam_cigar(h, op.ctypes.data, ln.ctypes.data)
        if L.dnh_container_add(w, qname.encode(), contig.encode(), adc.ctypes.data, adc.shape[0], off, scale, f.signal_length, f.signal_trim, f.signal_start, f.is_split,
                               query, len(query), ref_slice, len(ref_slice), op.ctypes.data, ln.ctypes.data, f.n_cigar, f.ref_start, f.is_reverse) != 0:
            raise SystemExit("pod5_to_container: writing %s failed" % out)
        n["written"] += 1; n["samples"] += int(adc.shape[0])
    L.dnh_bam_close(h)
    if L.dnh_container_close(w) != 0:
        raise SystemExit("pod5_to_container: closing %s failed" % out)
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--bam", required=True); ap.add_argument("--reference", required=True)
    ap.add_argument("--pod5", nargs="+", required=True, help=".pod5 files and / or directories searched for them")
    ap.add_argument("--out", required=True)
    ap.add_argument("-q", "--quality", type=int, default=20, help="minimum mapping quality (detect.cpp:40, default 20)")
    ap.add_argument("-l", "--length", type=int, default=1000, help="minimum mapped length (detect.cpp:41, default 1000)")
    a = ap.parse_args(argv)
    n = convert(a.bam, read_fasta(a.reference), Pod5Source(a.pod5), a.out, a.quality, a.length)
    print("pod5_to_container: %(records)d BAM records: %(filtered)d below -q / -l, %(unusable)d unusable (unmapped / contig missing / IUPAC base), %(no_signal)d without "
          "signal; %(written)d reads, %(samples)d samples written" % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
