"""ctypes binding of the C++ host side (csrc/host/dn_host.cpp): DNAscent::ReadBatch and CIGAR flattening."""
import ctypes as C
import os

import numpy as np

from . import build as _build
from . import hip as _hip

_lib = None


class StreamStats(C.Structure):
    """DNAscent::StreamStats"""
    _fields_ = [("seconds_total", C.c_double), ("seconds_upload", C.c_double), ("seconds_collect", C.c_double), ("seconds_emit", C.c_double),
                ("reads", C.c_uint64), ("reads_ok", C.c_uint64), ("samples", C.c_uint64), ("calls", C.c_uint64), ("bytes_out", C.c_uint64),
                ("positions", C.c_uint64), ("seconds_run", C.c_double), ("overflow_retries", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.HOST_SO):
            raise RuntimeError("libdnascent_host.so missing: run `python -m dnascent_amd.build`")
        L = C.CDLL(_build.HOST_SO)
        L.dnh_batch_new.restype = C.c_void_p
        L.dnh_batch_free.argtypes = [C.c_void_p]
        L.dnh_batch_clear.argtypes = [C.c_void_p]
        L.dnh_batch_size.restype = C.c_uint32
        L.dnh_batch_size.argtypes = [C.c_void_p]
        L.dnh_batch_samples.restype = C.c_uint64
        L.dnh_batch_samples.argtypes = [C.c_void_p]
        L.dnh_batch_add.restype = C.c_int
        L.dnh_batch_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.dnh_batch_desc.argtypes = [C.c_void_p, C.POINTER(_hip.BatchDesc)]
        L.dnh_batch_maps.restype = C.c_int
        L.dnh_batch_maps.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dnh_format_detect.restype = C.c_uint64
        L.dnh_format_detect.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_char_p, C.c_uint64]
        L.dnh_modbam.restype = C.c_int
        L.dnh_modbam.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64,
                                 C.c_void_p, C.c_uint64]
        L.dnh_detect_write.restype = C.c_int
        L.dnh_detect_write.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.dnh_format_align.restype = C.c_uint64
        L.dnh_format_align.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.dnh_align_write.restype = C.c_int
        L.dnh_align_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
        L.dnh_container_create.restype = C.c_void_p
        L.dnh_container_create.argtypes = [C.c_char_p]
        L.dnh_container_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.dnh_container_close.argtypes = [C.c_void_p]
        L.dnh_container_count.restype = C.c_int64
        L.dnh_container_count.argtypes = [C.c_char_p]
        L.dnh_container_sizes.restype = C.c_int64
        L.dnh_container_sizes.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        L.dnh_container_load_list.restype = C.c_int64
        L.dnh_container_load_list.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.dnh_container_load.restype = C.c_int64
        L.dnh_container_load.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64]
        L.dnh_hmm_write.restype = C.c_int
        L.dnh_hmm_write.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.dnh_detect_header.restype = C.c_uint64
        L.dnh_detect_header.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_char_p,
                                        C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        L.dnh_batch_fill_synth.restype = C.c_int
        L.dnh_batch_fill_synth.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double]
        L.dnh_batch_fill_synth_list.restype = C.c_int
        L.dnh_batch_fill_synth_list.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.dnh_container_write_synth.restype = C.c_int64
        L.dnh_container_write_synth.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.dnh_stream_detect.restype = C.c_int
        L.dnh_stream_detect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(StreamStats), C.c_void_p]
        L.dnh_keep_new.restype = C.c_void_p
        L.dnh_keep_free.argtypes = [C.c_void_p]
        L.dnh_keep_get.restype = C.c_uint64
        L.dnh_keep_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.dnh_container_index.restype = C.c_int64
        L.dnh_container_index.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.dnh_container_load_at.restype = C.c_int64
        L.dnh_container_load_at.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.dnh_stream_open.restype = C.c_void_p
        L.dnh_stream_open.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dnh_stream_close.argtypes = [C.c_void_p]
        L.dnh_stream_full.argtypes = [C.c_void_p]
        L.dnh_stream_inflight.argtypes = [C.c_void_p]
        L.dnh_stream_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.dnh_result_new.restype = C.c_void_p
        L.dnh_result_free.argtypes = [C.c_void_p]
        L.dnh_stream_collect.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_uint64), C.POINTER(_hip.ResultBatch)]
        L.dnh_stream_stats.argtypes = [C.c_void_p, C.POINTER(StreamStats)]
        L.dnh_result_packed.restype = C.c_uint64
        L.dnh_result_packed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.dnh_pack_calls.restype = C.c_uint64
        L.dnh_pack_calls.argtypes = [C.c_void_p, C.POINTER(_hip.ResultBatch), C.c_void_p]
        L.dnh_format_packed.restype = C.c_void_p
        L.dnh_format_packed.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dnh_text_data.restype = C.c_void_p
        L.dnh_text_data.argtypes = [C.c_void_p]
        L.dnh_text_size.restype = C.c_uint64
        L.dnh_text_size.argtypes = [C.c_void_p]
        L.dnh_text_free.argtypes = [C.c_void_p]
        L.dnh_pwrite_parallel.restype = C.c_int
        L.dnh_pwrite_parallel.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
        L.dnh_pwrite_scatter.restype = C.c_int
        L.dnh_pwrite_scatter.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dnh_packed_sizes.restype = None
        L.dnh_packed_sizes.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dnh_batch_pin.argtypes = [C.c_void_p]
        L.dnh_batch_unpin.argtypes = [C.c_void_p]
        L.dnh_revcomp.restype = C.c_int
        L.dnh_revcomp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def revcomp(seq_u8):
    s = np.ascontiguousarray(seq_u8, np.uint8)
    out = np.zeros_like(s)
    lib().dnh_revcomp(s.ctypes.data, s.shape[0], out.ctypes.data)
    return out


def detect_header(alignment, genome, index, threads, quality, length, use_gpu, start_time, software, version, commit):
    """DNAscent::writeDetectHeader (detect.cpp:196-232) -> bytes."""
    buf = C.create_string_buffer(4096)
    n = lib().dnh_detect_header(alignment.encode(), genome.encode(), index.encode(), threads, quality, length, int(use_gpu),
                                start_time.encode(), software.encode(), version.encode(), commit.encode(), buf, len(buf))
    return buf.raw[:n]


def format_align(read_id, contig, ref_start, ref_end, is_reverse, refseq_strand, model_mean, table):
    """DNAscent::formatAlignRecord -> bytes of one `DNAscent align` record (table: coord / ref_pos / value / kind arrays)."""
    ref = bytes(refseq_strand)
    mm = np.ascontiguousarray(model_mean, np.float64)
    c = np.ascontiguousarray(table["coord"], np.uint32); p = np.ascontiguousarray(table["ref_pos"], np.uint32)
    v = np.ascontiguousarray(table["value"], np.float64); k = np.ascontiguousarray(table["kind"], np.uint8)
    n = c.shape[0]
    cap = 256 + 80 * max(1, n)
    buf = C.create_string_buffer(cap)
    ln = lib().dnh_format_align(read_id.encode(), contig.encode(), ref_start, ref_end, int(is_reverse), ref, len(ref), mm.ctypes.data, n,
                                c.ctypes.data, p.ctypes.data, v.ctypes.data, k.ctypes.data, buf, cap)
    assert ln <= cap
    return buf.raw[:ln]


def format_detect(read_id, contig, ref_start, ref_end, is_reverse, coord, kmer_s9, probs):
    """DNAscent::formatDetectRecord -> bytes of one .detect record."""
    coord = np.ascontiguousarray(coord, np.uint32); probs = np.ascontiguousarray(probs, np.float32)
    km = np.ascontiguousarray(kmer_s9, "S9"); n = coord.shape[0]
    cap = 128 + 64 * max(1, n)
    buf = C.create_string_buffer(cap)
    ln = lib().dnh_format_detect(read_id.encode(), contig.encode(), ref_start, ref_end, int(is_reverse), n, coord.ctypes.data,
                                 km.ctypes.data, probs.ctypes.data, buf, cap)
    assert ln <= cap
    return buf.raw[:ln]


def modbam(query_idx, ref_idx, kmer_s9, probs, ref2del):
    """DNAscent::modBamFields -> (n_calls, MM text, ML bytes)."""
    q = np.ascontiguousarray(query_idx, np.uint32); r = np.ascontiguousarray(ref_idx, np.uint32)
    km = np.ascontiguousarray(kmer_s9, "S9"); probs = np.ascontiguousarray(probs, np.float32); d = np.ascontiguousarray(ref2del, np.uint8)
    n = q.shape[0]
    mm = C.create_string_buffer(16 * n + 64); ml = np.zeros(2 * n + 1, np.uint8)
    k = lib().dnh_modbam(n, q.ctypes.data, r.ctypes.data, km.ctypes.data, probs.ctypes.data, d.ctypes.data, mm, len(mm), ml.ctypes.data,
                         ml.shape[0])
    return k, mm.value.decode(), ml[:2 * k].copy()


def write_container(path, synth_reads, signal_length=-1, signal_trim=0, signal_start=0, is_split=False):
    """DNAscent::ReadContainerWriter: the binary read container hosts without htslib / libpod5 (and the tests) ingest from."""
    w = lib().dnh_container_create(path.encode())
    if not w:
        raise IOError(path)
    for sr in synth_reads:
        q = np.ascontiguousarray(revcomp(sr.basecall) if sr.is_reverse else sr.basecall)
        f = np.ascontiguousarray(revcomp(sr.refseq) if sr.is_reverse else sr.refseq)
        adc = np.ascontiguousarray(sr.adc)
        rc = lib().dnh_container_add(w, sr.read_id.encode(), sr.contig.encode(), adc.ctypes.data, adc.shape[0], sr.cal_offset, sr.cal_scale,
                                     signal_length, signal_trim, signal_start, int(is_split), q.ctypes.data, q.shape[0], f.ctypes.data,
                                     f.shape[0], sr.cigar_op.ctypes.data, sr.cigar_len.ctypes.data, sr.cigar_op.shape[0], sr.ref_start,
                                     int(sr.is_reverse))
        assert rc == 0
    if lib().dnh_container_close(w) != 0:
        raise IOError(path)


def write_synth_container(path, model, seed0, n_reads, n_bases):
    """n_reads synthetic reads (seeds seed0 .., every odd one reverse: ReadBatch.fill_synth's reads) into a container, generated on all host cores"""
    m = np.ascontiguousarray(model, np.float64)
    n = int(lib().dnh_container_write_synth(path.encode(), m.ctypes.data, seed0, n_reads, n_bases))
    if n < 0:
        raise IOError(path)
    return n


def container_count(path):
    return int(lib().dnh_container_count(path.encode()))


def container_sizes(path):
    """stored sample count of every read of a container (uint64 array); the records are seeked over, not read"""
    n = container_count(path)
    if n < 0:
        raise IOError(path)
    out = np.zeros(max(n, 1), np.uint64)
    got = int(lib().dnh_container_sizes(path.encode(), out.ctypes.data, n))
    if got != n:
        raise IOError("%s: malformed container" % path)
    return out[:n]


def container_index(path):
    """(sample counts, file offsets) of every record of a container, uint64 arrays: ONE pass of seeks; the offsets let any subset of
    the records be read directly afterwards (ReadBatch.add_container_at)"""
    n = container_count(path)
    if n < 0:
        raise IOError(path)
    sizes = np.zeros(max(n, 1), np.uint64); offs = np.zeros(max(n, 1), np.uint64)
    got = int(lib().dnh_container_index(path.encode(), sizes.ctypes.data, offs.ctypes.data, n))
    if got != n:
        raise IOError("%s: malformed container" % path)
    return sizes[:n], offs[:n]


class ReadBatch:
    """DNAscent::ReadBatch: reads packed as SoA, ready for dn_batch_upload."""

    def __init__(self):
        self.h = C.c_void_p(lib().dnh_batch_new())
        self.reads = []

    def __del__(self):
        try:
            if self.h:
                lib().dnh_batch_free(self.h)
        except Exception:
            pass

    def add_synth(self, sr, signal_length=-1, signal_trim=0, signal_start=0, is_split=False):
        """Add a synth.SynthRead.  The generator emits strand-direction sequences; the host API takes what a BAM / FASTA
        hold (reference-forward orientation), so reverse reads are reverse-complemented back first."""
        q = revcomp(sr.basecall) if sr.is_reverse else sr.basecall
        f = revcomp(sr.refseq) if sr.is_reverse else sr.refseq
        q = np.ascontiguousarray(q); f = np.ascontiguousarray(f)
        adc = np.ascontiguousarray(sr.adc)
        rc = lib().dnh_batch_add(self.h, sr.read_id.encode(), sr.contig.encode(), adc.ctypes.data, adc.shape[0],
                                 sr.cal_offset, sr.cal_scale, signal_length, signal_trim, signal_start, int(is_split),
                                 q.ctypes.data, q.shape[0], f.ctypes.data, f.shape[0], sr.cigar_op.ctypes.data,
                                 sr.cigar_len.ctypes.data, sr.cigar_op.shape[0], sr.ref_start, int(sr.is_reverse))
        if rc >= 0:
            self.reads.append(sr)
        return rc

    def fill_synth(self, model, seed0, n_reads, n_bases, noise_pa=1.6, sub_rate=0.002, ins_rate=0.001, del_rate=0.001):
        """n_reads synthetic reads (seeds seed0 .., every odd one reverse), generated on all host cores inside the host library.
        The batch does not keep SynthRead objects for them (self.reads stays as it was)."""
        m = np.ascontiguousarray(model, np.float64)
        return int(lib().dnh_batch_fill_synth(self.h, m.ctypes.data, seed0, n_reads, n_bases, noise_pa, sub_rate, ins_rate, del_rate))

    def fill_synth_list(self, model, seeds, bases, noise_pa=1.6, sub_rate=0.002, ins_rate=0.001, del_rate=0.001):
        """one synthetic read per (seed, length in bases) pair, strand by seed parity; generated on all host cores"""
        m = np.ascontiguousarray(model, np.float64)
        sd = np.ascontiguousarray(seeds, np.uint64); nb = np.ascontiguousarray(bases, np.uint32)
        assert sd.shape == nb.shape
        return int(lib().dnh_batch_fill_synth_list(self.h, m.ctypes.data, sd.shape[0], sd.ctypes.data, nb.ctypes.data, noise_pa, sub_rate, ins_rate, del_rate))

    def add_container(self, path, first=0, count=1 << 62):
        """reads [first, first + count) of a binary read container; returns how many were accepted (-1: malformed file)"""
        return int(lib().dnh_container_load(self.h, path.encode(), first, count))

    def add_container_list(self, path, ordinals):
        """the reads with the given ascending ordinals of a container; returns how many were accepted (-1: malformed / bad list)"""
        o = np.ascontiguousarray(ordinals, np.uint64)
        return int(lib().dnh_container_load_list(self.h, path.encode(), o.ctypes.data, o.shape[0]))

    def add_container_at(self, path, offsets):
        """the records at the given file offsets (container_index), in that order, read by all host cores.  Returns a uint8 array:
        1 where the batch accepted the record, 0 where the reference's own filters reject it (a failed read, not an error).
        Raises IOError on a truncated / unreadable record."""
        o = np.ascontiguousarray(offsets, np.uint64)
        acc = np.zeros(o.shape[0], np.uint8)
        got = int(lib().dnh_container_load_at(self.h, path.encode(), o.ctypes.data, o.shape[0], acc.ctypes.data))
        if got < 0:
            raise IOError("%s: truncated or unreadable record" % path)
        return acc

    def clear(self):
        """empty the batch, keeping its buffers (a streaming host reuses its batch objects)"""
        lib().dnh_batch_clear(self.h)
        self.reads = []

    def pin(self):
        """page-lock the arrays dn_batch_upload reads: the upload then returns before its copies are done"""
        if lib().dnh_batch_pin(self.h) != 0:
            raise _hip.DnError("dnh_batch_pin failed")

    def unpin(self):
        lib().dnh_batch_unpin(self.h)

    def size(self):
        return int(lib().dnh_batch_size(self.h))

    def samples(self):
        return int(lib().dnh_batch_samples(self.h))

    def desc(self):
        d = _hip.BatchDesc()
        lib().dnh_batch_desc(self.h, C.byref(d))
        return d

    def maps(self, i, n_ref, n_query):
        r2q = np.zeros(n_ref, np.uint32); q2r = np.zeros(n_query + 1, np.int32); r2d = np.zeros(n_ref, np.uint8)
        rc = lib().dnh_batch_maps(self.h, i, r2q.ctypes.data, q2r.ctypes.data, r2d.ctypes.data)
        assert rc == 0
        return r2q, q2r, r2d

    def upload(self, ctx):
        ctx.upload(self.desc(), self.size(), keep=self)

    def hmm_write(self, ctx, path, header=None):
        """detect --HMM: llAcrossRead for the normalised batch + HumanReadableWriter; returns reads written."""
        rc = lib().dnh_hmm_write(ctx.h, self.h, path.encode(), header.encode() if header is not None else None)
        if rc < 0:
            raise _hip.DnError("dnh_hmm_write failed (%d): %s" % (rc, _hip.lib().dn_last_error(ctx.h).decode()))
        return rc

    def align_write(self, ctx, path, model_mean):
        """`DNAscent align`: eventalign with the per-sample table for the normalised batch, records appended to path."""
        mm = np.ascontiguousarray(model_mean, np.float64)
        rc = lib().dnh_align_write(ctx.h, self.h, mm.ctypes.data, path.encode())
        if rc < 0:
            raise _hip.DnError("dnh_align_write failed (%d): %s" % (rc, _hip.lib().dn_last_error(ctx.h).decode()))
        return rc

    def detect_write(self, ctx, path, header=None):
        """runCNN for the aligned batch + HumanReadableWriter: writes the .detect records; returns reads written."""
        rc = lib().dnh_detect_write(ctx.h, self.h, path.encode(), header.encode() if header is not None else None)
        if rc < 0:
            raise _hip.DnError("dnh_detect_write failed (%d): %s" % (rc, _hip.lib().dn_last_error(ctx.h).decode()))
        return rc


def stream_detect(ctxs, batches, emit=True, out_path=None, header=None, keep=False):
    """DNAscent::streamDetect: the batches through len(ctxs) contexts in flight, driven by one host thread (this one, inside the
    host library).  Returns StreamStats, or (StreamStats, dict of numpy arrays: read_calls / coord / p_edu / p_brdu of the whole
    stream) with keep=True."""
    hc = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    hb = (C.c_void_p * len(batches))(*[b.h for b in batches])
    st = StreamStats()
    kh = C.c_void_p(lib().dnh_keep_new()) if keep else None
    try:
        rc = lib().dnh_stream_detect(hc, len(ctxs), hb, len(batches), int(emit), out_path.encode() if out_path else None,
                                     header.encode() if header is not None else None, C.byref(st), kh)
        if rc != 0:
            msgs = [_hip.lib().dn_last_error(c.h).decode() for c in ctxs]
            raise _hip.DnError("dnh_stream_detect failed (%d): %s" % (rc, "; ".join(m for m in msgs if m)))
        for i, b in enumerate(batches):
            if i >= len(batches) - len(ctxs):
                ctxs[i % len(ctxs)].n_reads = b.size()
        if not keep:
            return st
        out = {}
        for which, (name, dt) in enumerate((("read_calls", np.uint64), ("coord", np.uint32), ("p_edu", np.float32), ("p_brdu", np.float32),
                                             ("record_bytes", np.uint64))):
            p = C.c_void_p()
            n = int(lib().dnh_keep_get(kh, which, C.byref(p)))
            out[name] = np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(p.value), dtype=dt).copy() if n else np.zeros(0, dt)
        return st, out
    finally:
        if kh:
            lib().dnh_keep_free(kh)


PACK_REVERSE, PACK_TEXT = 1, 2          # DNAscent::DN_PACK_* (flags of a packed read's meta row)


def _buffer_address(buf):
    """(address, keep-alive object) of a bytes / bytearray / memoryview / ctypes buffer: the object returned must stay referenced for as long as the
    address is used (round-4 advisor: the bytearray branch used to take the address of a temporary copy that nothing kept alive)"""
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p).value, buf
    if isinstance(buf, bytearray):
        hold = (C.c_char * len(buf)).from_buffer(buf)
        return C.addressof(hold), hold
    mv = memoryview(buf)
    if mv.readonly:
        arr = np.frombuffer(mv, np.uint8)                    # wraps the caller's buffer (no copy); `arr` keeps the view, the view keeps the buffer
        return arr.ctypes.data, arr
    hold = C.c_char.from_buffer(mv)
    return C.addressof(hold), (hold, mv)


def pwrite_parallel(fd, buf, offset):
    """buf (bytes or a memoryview / ctypes view of the formatter's text) to file descriptor fd at `offset`, written in pieces by several host threads"""
    n = len(buf)
    if n == 0:
        return
    addr, hold = _buffer_address(buf)
    rc = lib().dnh_pwrite_parallel(int(fd), C.c_void_p(addr), n, int(offset))
    del hold
    if rc != 0:
        raise IOError("short write to the output file")


def pwrite_scatter(fd, buf, src_off, lens, file_off):
    """pieces of one buffer to their own file offsets (dnh_pwrite_scatter): piece i = lens[i] bytes at buf[src_off[i]:] -> file offset file_off[i]"""
    so = np.ascontiguousarray(src_off, np.uint64); ln = np.ascontiguousarray(lens, np.uint64); fo = np.ascontiguousarray(file_off, np.uint64)
    n = int(ln.shape[0])
    assert so.shape[0] == n and fo.shape[0] == n
    if n == 0 or int(ln.sum()) == 0:
        return
    assert int((so + ln).max()) <= len(buf)
    addr, hold = _buffer_address(buf)
    rc = lib().dnh_pwrite_scatter(int(fd), C.c_void_p(addr), n, so.ctypes.data, ln.ctypes.data, fo.ctypes.data)
    del hold
    if rc != 0:
        raise IOError("short write to the output file")


def packed_sizes(meta3, read_ptr):
    """DNAscent::packedSizes: the exact text length of every packed read (uint64 [n]), without formatting anything"""
    m = np.ascontiguousarray(meta3, np.uint64); p = np.ascontiguousarray(read_ptr, np.uint64)
    n = p.shape[0]
    rb = np.zeros(max(n, 1), np.uint64)
    if n:
        lib().dnh_packed_sizes(n, m.ctypes.data, p.ctypes.data, rb.ctypes.data)
    return rb[:n]


class _TextOwner:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                lib().dnh_text_free(self.h); self.h = None
        except Exception:
            pass


def vbz_available():
    """is libzstd.so.1 loadable (POD5's VBZ signal compression = svb16 + zstd)?"""
    return bool(lib().dnh_vbz_available())


def vbz_decode(chunk, n_samples):
    """one VBZ chunk (a cell of a POD5 signal table's `signal` column) -> int16 samples (csrc/host/dn_vbz.cpp; pod5.cpp:57 receives them from libpod5)"""
    src = np.frombuffer(bytes(chunk), np.uint8)
    out = np.zeros(int(n_samples), np.int16)
    L = lib()
    L.dnh_vbz_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.dnh_vbz_last_error.restype = C.c_char_p
    rc = L.dnh_vbz_decode(src.ctypes.data, src.shape[0], int(n_samples), out.ctypes.data)
    if rc != 0:
        raise ValueError("vbz_decode (%d): %s" % (rc, L.dnh_vbz_last_error().decode()))
    return out


def vbz_encode(samples, level=1):
    """int16 samples -> one VBZ chunk (bytes)"""
    a = np.ascontiguousarray(samples, np.int16)
    L = lib()
    L.dnh_vbz_bound.restype = C.c_uint64; L.dnh_vbz_bound.argtypes = [C.c_uint64]
    L.dnh_vbz_encode.restype = C.c_uint64; L.dnh_vbz_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    L.dnh_vbz_last_error.restype = C.c_char_p
    cap = int(L.dnh_vbz_bound(a.shape[0]))
    dst = np.zeros(max(cap, 64), np.uint8)
    n = int(L.dnh_vbz_encode(a.ctypes.data, a.shape[0], dst.ctypes.data, dst.shape[0], int(level)))
    if n == 0:
        raise ValueError("vbz_encode: %s" % L.dnh_vbz_last_error().decode())
    return dst[:n].tobytes()


def host_threads():
    """threads of the library's parallel loops: min(64, cores, the cgroup CPU quota), or DN_HOST_THREADS"""
    return int(lib().dnh_host_threads())


def usable_cpus():
    """CPUs this process can actually keep busy: the affinity mask, cut to the cgroup's CPU quota (cpu.max) when there is one.  The hosts of the GPU
    pool show 256 hardware threads and grant 16 CPUs of quota: anything wider is frozen by the kernel for the rest of each 100 ms period."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def format_packed(meta3, read_ptr):
    """DNAscent::formatPacked: the writer rank's formatter.  meta3 uint64 [n][3] (count, header bytes, flags), read_ptr uint64 [n] (address
    of every read's payload, output order) -> (text of the n records laid end to end -- a memoryview of the C++ buffer, bytes when empty --,
    record_bytes uint64 [n])."""
    m = np.ascontiguousarray(meta3, np.uint64); p = np.ascontiguousarray(read_ptr, np.uint64)
    n = p.shape[0]
    rb = np.zeros(max(n, 1), np.uint64)
    if n == 0:
        return b"", rb[:0]
    t = C.c_void_p(lib().dnh_format_packed(n, m.ctypes.data, p.ctypes.data, rb.ctypes.data))
    if not t:
        raise _hip.DnError("dnh_format_packed: a record did not come out at its computed size")
    size = int(lib().dnh_text_size(t))
    if size == 0:
        lib().dnh_text_free(t)
        return b"", rb[:n]
    # a VIEW of the C++ string, not a copy: C.string_at would memcpy 200-400 MB per window while holding the GIL, i.e. with the thread that drives
    # the GPU locked out (round 4: ~1 s of a 9 s run); file.write() takes the view as it is and releases the GIL for the system call
    buf = (C.c_char * size).from_address(lib().dnh_text_data(t))
    buf._owner = _TextOwner(t)                           # freed when the last view of the buffer goes
    return memoryview(buf).cast("B"), rb[:n]


class DetectStream:
    """DNAscent::DetectStream: the buffer-of-reads loop of detect.cpp:821-907 OPEN-ENDED.  submit() uploads a batch to the free
    context and enqueues its whole per-read body; collect() waits for the OLDEST batch in flight and returns its records.  At most
    len(ctxs) batches are in flight; a batch stays alive and untouched between its submit() and its collect().
    emit: False / 0 nothing, True / 1 the text records, "packed" / 2 the packed per-call results (16 bytes per call: what a rank of a
    multi-GPU run hands to the writer rank, which formats them -- shard.StreamDriver)."""

    def __init__(self, ctxs, emit=True):
        self.ctxs = list(ctxs)
        hc = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        self.emit = 2 if emit in ("packed", 2) else int(bool(emit))
        self.h = C.c_void_p(lib().dnh_stream_open(hc, len(ctxs), self.emit))
        self.res = C.c_void_p(lib().dnh_result_new())
        self._batches = {}

    def close(self):
        if self.h:
            lib().dnh_stream_close(self.h); lib().dnh_result_free(self.res)
            self.h = C.c_void_p(); self.res = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def full(self):
        return bool(lib().dnh_stream_full(self.h))

    def in_flight(self):
        return int(lib().dnh_stream_inflight(self.h))

    def _fail(self, what, rc):
        msgs = [_hip.lib().dn_last_error(c.h).decode() for c in self.ctxs]
        raise _hip.DnError("%s failed (%d): %s" % (what, rc, "; ".join(m for m in msgs if m)))

    def submit(self, batch, tag):
        rc = lib().dnh_stream_submit(self.h, batch.h, int(tag))
        if rc != 0:
            self._fail("DetectStream.submit", rc)
        self._batches[int(tag)] = batch

    class _Owner:
        """keeps a C++ Result (the packed payload of one collected batch) alive while numpy views point into it"""

        def __init__(self, h):
            self.h = h

        def __del__(self):
            try:
                if self.h:
                    lib().dnh_result_free(self.h); self.h = None
            except Exception:
                pass

    def collect(self, calls=False):
        """-> dict(tag, batch, status [n_reads], record_bytes [n_reads], text (bytes: the records of the passing reads, batch order));
        calls=True adds read_calls / coord / p_edu / p_brdu (copies of the dn_result_batch arrays).  emit == "packed": packed_meta [k][4] and
        packed (uint8) are VIEWS into a C++ result object of their own, kept alive by out["owner"] (no copy of ~190 MB per batch)."""
        tag = C.c_uint64(); n = C.c_uint32(); rb = C.c_void_p(); tx = C.c_void_p(); tb = C.c_uint64(); res = _hip.ResultBatch()
        resh = C.c_void_p(lib().dnh_result_new()) if self.emit == 2 else self.res
        rc = lib().dnh_stream_collect(self.h, resh, C.byref(tag), C.byref(n), C.byref(rb), C.byref(tx), C.byref(tb), C.byref(res))
        owner = DetectStream._Owner(resh) if self.emit == 2 else None
        if rc != 0:
            self._fail("DetectStream.collect", rc)
        nr = int(n.value)

        def arr(ptr, dtype, cnt):
            if cnt == 0 or not ptr:
                return np.zeros(0, dtype)
            return np.frombuffer((C.c_char * (cnt * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()
        summ = arr(res.summary, _hip.SUMMARY_DTYPE, nr)
        out = dict(tag=int(tag.value), batch=self._batches.pop(int(tag.value)), status=summ["status"].copy() if nr else np.zeros(0, np.int32),
                   n_positions=summ["n_positions"].copy() if nr else np.zeros(0, np.uint32),
                   record_bytes=arr(rb.value, np.uint64, nr), text=C.string_at(tx.value, tb.value) if tb.value else b"")
        if self.emit == 2:
            mp = C.c_void_p(); pp = C.c_void_p(); pb = C.c_uint64()
            k = int(lib().dnh_result_packed(resh, C.byref(mp), C.byref(pp), C.byref(pb)))

            def view(ptr, dtype, cnt):
                if cnt == 0 or not ptr:
                    return np.zeros(0, dtype)
                return np.frombuffer((C.c_char * (cnt * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype)
            out["packed_meta"] = view(mp.value, np.uint64, 4 * k).reshape(k, 4)       # rows: index in the batch, count, header bytes, flags
            out["packed"] = view(pp.value, np.uint8, int(pb.value))
            out["owner"] = owner
        if calls:
            k = int(res.n_calls)
            off = arr(res.call_off, np.uint64, nr + 1 if nr else 0)
            out.update(read_calls=np.diff(off) if nr else np.zeros(0, np.uint64), coord=arr(res.ref_coord, np.uint32, k), p_edu=arr(res.p_edu, np.float32, k),
                       p_brdu=arr(res.p_brdu, np.float32, k))
        return out

    def stats(self):
        st = StreamStats()
        lib().dnh_stream_stats(self.h, C.byref(st))
        return st
