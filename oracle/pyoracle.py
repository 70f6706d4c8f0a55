"""ctypes binding of oracle/liboracle.so (our CPU restatement) and oracle/_ref/libref.so (the reference's
own event_detection.c + probability.cpp compiled in place).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref.so")

u8p, u32p, i32p, f32p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_float, C.c_double))


class Model(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("sigma", C.c_double)]


class Read(C.Structure):
    _fields_ = [("raw", C.c_void_p), ("n_raw", C.c_size_t), ("basecall", C.c_void_p), ("n_base", C.c_size_t),
                ("refseq", C.c_void_p), ("n_ref", C.c_size_t), ("ref2query", C.c_void_p), ("query2ref", C.c_void_p),
                ("ref2del", C.c_void_p), ("ref_start", C.c_int32), ("ref_end", C.c_int32), ("is_reverse", C.c_int)]


class Event(C.Structure):
    _fields_ = [("mean", C.c_double), ("raw_start", C.c_uint32), ("raw_len", C.c_uint32)]


class Norm(C.Structure):
    _fields_ = [("status", C.c_int), ("n_scrappie", C.c_size_t), ("events", C.POINTER(Event)), ("n_events", C.c_size_t),
                ("rank_q", u32p), ("n_kq", C.c_size_t), ("rank_r", u32p), ("n_kr", C.c_size_t),
                ("q_shift", C.c_double), ("q_scale", C.c_double), ("n_bands", C.c_size_t), ("fills", C.c_uint64),
                ("end_event", C.c_int), ("aln_event", u32p), ("aln_kmer", u32p), ("n_aln", C.c_size_t),
                ("avg_log_emission", C.c_double), ("spanned", C.c_int), ("max_gap", C.c_int),
                ("cleaned_sig", f64p), ("cleaned_rank", u32p), ("n_cleaned", C.c_size_t),
                ("ts_slope", C.c_double), ("ts_intercept", C.c_double),
                ("shift", C.c_double), ("scale", C.c_double), ("events_per_base", C.c_double)]


class Align(C.Structure):
    _fields_ = [("n_pos", C.c_size_t), ("coord", u32p), ("query_idx", u32p), ("ref_idx", u32p), ("indel_score", i32p),
                ("kmer", C.POINTER(C.c_char)), ("n_signal", u32p), ("signal", f32p), ("core", f32p), ("residual", f32p),
                ("n_windows", C.c_size_t), ("sum_TN", C.c_uint64), ("score_sum", C.c_double),
                ("win_ref", u32p), ("win_len", u32p), ("win_T", u32p), ("win_score", f64p),
                ("n_rows", C.c_size_t), ("row_coord", u32p), ("row_rpos", u32p), ("row_val", f64p), ("row_kind", C.POINTER(C.c_uint8))]


class FitModels(C.Structure):
    _fields_ = [("unl_mean", C.c_void_p), ("unl_std", C.c_void_p), ("ana_mean", C.c_void_p), ("ana_std", C.c_void_p)]


class Hmm(C.Structure):
    _fields_ = [("n", C.c_size_t), ("pos_on_ref", C.POINTER(C.c_uint32)), ("pos_on_query", C.POINTER(C.c_uint32)),
                ("global_pos", C.POINTER(C.c_int32)), ("n_events", C.POINTER(C.c_uint32)), ("log_analogue", C.POINTER(C.c_double)),
                ("log_thymidine", C.POINTER(C.c_double)), ("llr", C.POINTER(C.c_double))]


class SEvent(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_float), ("mean", C.c_float), ("stdv", C.c_float)]


_o = None
_r = None


def oracle():
    global _o
    if _o is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        L = C.CDLL(ORACLE_SO)
        for n in ("dno_eexp",):
            getattr(L, n).restype = C.c_double
            getattr(L, n).argtypes = [C.c_double]
        L.dno_eln.restype = C.c_double
        L.dno_eln.argtypes = [C.c_double, C.POINTER(C.c_int)]
        for n in ("dno_lnSum", "dno_lnProd"):
            getattr(L, n).restype = C.c_double
            getattr(L, n).argtypes = [C.c_double, C.c_double]
        L.dno_lnGreaterThan.restype = C.c_int
        L.dno_lnGreaterThan.argtypes = [C.c_double, C.c_double]
        L.dno_normalPDF.restype = C.c_double
        L.dno_normalPDF.argtypes = [C.c_double] * 3
        L.dno_kmer2index.restype = C.c_uint32
        L.dno_kmer2index.argtypes = [C.c_char_p, C.c_uint]
        L.dno_detect_events.restype = C.c_size_t
        L.dno_detect_events.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(SEvent)), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.POINTER(C.c_size_t)]
        L.dno_adc_to_pa.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
        L.dno_normalise.restype = C.c_int
        L.dno_normalise.argtypes = [C.POINTER(Model), C.POINTER(Read), C.POINTER(Norm)]
        L.dno_norm_free.argtypes = [C.POINTER(Norm)]
        L.dno_quantile_scaling.argtypes = [C.POINTER(Model), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, f64p, f64p]
        L.dno_theil_sen.restype = C.c_int
        L.dno_theil_sen.argtypes = [C.POINTER(Model), C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double,
                                    f64p, f64p, f64p, f64p]
        L.dno_viterbi.restype = C.c_size_t
        L.dno_viterbi.argtypes = [C.POINTER(Model), C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_double, C.c_double,
                                  C.c_double, f64p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.dno_eventalign.restype = C.c_int
        L.dno_eventalign.argtypes = [C.POINTER(Model), C.POINTER(Read), C.POINTER(Norm), C.POINTER(Align)]
        L.dno_align_free.argtypes = [C.POINTER(Align)]
        L.dno_sequence_probability.restype = C.c_double
        L.dno_sequence_probability.argtypes = [C.POINTER(FitModels), C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_double,
                                               C.c_double, C.c_double, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
        L.dno_ll_across_read.restype = C.c_int
        L.dno_ll_across_read.argtypes = [C.POINTER(FitModels), C.POINTER(Read), C.POINTER(Norm), C.c_uint, C.POINTER(Hmm)]
        L.dno_hmm_free.argtypes = [C.POINTER(Hmm)]
        L.dno_format_hmm.restype = C.c_size_t
        L.dno_format_hmm.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Read), C.POINTER(Hmm), C.c_void_p, C.c_size_t]
        L.dno_reverse_complement.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.dno_vector_mean.restype = C.c_double
        L.dno_vector_mean.argtypes = [C.c_void_p, C.c_size_t]
        L.dno_modbam_tags.restype = C.c_size_t
        L.dno_modbam_tags.argtypes = [C.POINTER(Read), C.POINTER(Align), C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.dno_format_detect.restype = C.c_size_t
        L.dno_format_detect.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Read), C.POINTER(Align), C.c_void_p, C.c_void_p,
                                        C.c_size_t]
        L.dno_format_align.restype = C.c_size_t
        L.dno_format_align.argtypes = [C.POINTER(Model), C.c_char_p, C.c_char_p, C.POINTER(Read), C.POINTER(Align), C.c_void_p, C.c_size_t]
        L.dno_parse_cigar.restype = C.c_int
        L.dno_parse_cigar.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t]
        _o = L
    return _o


def ref():
    """The compiled reference (None when oracle/_ref/libref.so is absent)."""
    global _r
    if _r is None:
        if not os.path.exists(REF_SO):
            return None
        L = C.CDLL(REF_SO)
        L.ref_eexp.restype = C.c_double
        L.ref_eexp.argtypes = [C.c_double]
        L.ref_eln.restype = C.c_double
        L.ref_eln.argtypes = [C.c_double, C.POINTER(C.c_int)]
        for n in ("ref_lnSum", "ref_lnProd"):
            getattr(L, n).restype = C.c_double
            getattr(L, n).argtypes = [C.c_double, C.c_double]
        L.ref_lnGreaterThan.restype = C.c_int
        L.ref_lnGreaterThan.argtypes = [C.c_double, C.c_double]
        L.ref_normalPDF.restype = C.c_double
        L.ref_normalPDF.argtypes = [C.c_double] * 3
        L.ref_detect_events.restype = C.c_size_t
        L.ref_detect_events.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        if hasattr(L, "ref_reverseComplement"):
            L.ref_reverseComplement.restype = C.c_size_t
            L.ref_reverseComplement.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
            L.ref_vectorMean.restype = C.c_double
            L.ref_vectorMean.argtypes = [C.c_void_p, C.c_size_t]
        _r = L
    return _r


# ---------------------------------------------------------------------------------------------
# convenience layer over the raw structs
# ---------------------------------------------------------------------------------------------
def reverse_complement(seq_bytes):
    out = C.create_string_buffer(len(seq_bytes))
    oracle().dno_reverse_complement(seq_bytes, len(seq_bytes), out)
    return out.raw


def vector_mean(v):
    v = np.ascontiguousarray(v, np.float64)
    return float(oracle().dno_vector_mean(v.ctypes.data, v.shape[0]))


def ref_reverse_complement(seq_bytes):
    out = C.create_string_buffer(len(seq_bytes))
    n = ref().ref_reverseComplement(seq_bytes, len(seq_bytes), out)
    return out.raw[:n]


def ref_vector_mean(v):
    v = np.ascontiguousarray(v, np.float64)
    return float(ref().ref_vectorMean(v.ctypes.data, v.shape[0]))


def adc_to_pa(adc, offset, scale):
    out = np.empty(adc.shape[0], dtype=np.float64)
    a = np.ascontiguousarray(adc, dtype=np.int16)
    oracle().dno_adc_to_pa(a.ctypes.data, a.shape[0], offset, scale, out.ctypes.data)
    return out


def detect_events(raw, want_intermediates=False):
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    n = raw.shape[0]
    p = C.POINTER(SEvent)()
    if want_intermediates:
        t1 = np.zeros(n, np.float32); t2 = np.zeros(n, np.float32); pk = np.zeros(n, np.uint64); npk = C.c_size_t(0)
        ne = oracle().dno_detect_events(raw.ctypes.data, n, C.byref(p), t1.ctypes.data, t2.ctypes.data, pk.ctypes.data, C.byref(npk))
    else:
        ne = oracle().dno_detect_events(raw.ctypes.data, n, C.byref(p), None, None, None, None)
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(ne * C.sizeof(SEvent),)).copy()
    C.CDLL(None).free(p)
    ev = arr.view(np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4")], align=True))
    if want_intermediates:
        return ev, t1, t2, pk[:npk.value]
    return ev


def ref_detect_events(raw):
    raw = np.ascontiguousarray(raw, dtype=np.float64).copy()
    n = raw.shape[0]
    cap = n + 1
    st = np.zeros(cap, np.uint64); ln = np.zeros(cap, np.float32); mn = np.zeros(cap, np.float32); sd = np.zeros(cap, np.float32)
    ne = ref().ref_detect_events(raw.ctypes.data, n, st.ctypes.data, ln.ctypes.data, mn.ctypes.data, sd.ctypes.data, cap)
    return st[:ne], ln[:ne], mn[:ne], sd[:ne]


def parse_cigar(ops, lens, is_reverse, n_query):
    ops = np.ascontiguousarray(ops, np.uint32); lens = np.ascontiguousarray(lens, np.uint32)
    ref_len = int(sum(int(l) for o, l in zip(ops, lens) if o in (0, 2, 3, 7, 8)))
    capr = ref_len + int(lens.max()) + 1
    r2q = np.zeros(capr, np.uint32); r2d = np.zeros(capr, np.uint8); q2r = np.full(n_query + 1, -1, np.int32)
    rl = oracle().dno_parse_cigar(ops.ctypes.data, lens.ctypes.data, ops.shape[0], int(is_reverse), r2q.ctypes.data,
                                  r2d.ctypes.data, capr, q2r.ctypes.data, n_query + 1)
    assert rl == ref_len
    return r2q[:ref_len].copy(), q2r, r2d[:ref_len].copy()


class OracleRead:
    """Holds the numpy buffers behind a dno_read so they outlive the ctypes struct."""

    def __init__(self, sr, model_mean, sigma=0.14):
        self.sr = sr
        self.raw = adc_to_pa(sr.adc, sr.cal_offset, sr.cal_scale)
        self.r2q, self.q2r, self.r2d = parse_cigar(sr.cigar_op, sr.cigar_len, sr.is_reverse, sr.basecall.shape[0])
        assert self.r2q.shape[0] == sr.refseq.shape[0]
        self.model_mean = np.ascontiguousarray(model_mean, np.float64)
        self.model = Model(self.model_mean.ctypes.data, sigma)
        self.c = Read(self.raw.ctypes.data, self.raw.shape[0], sr.basecall.ctypes.data, sr.basecall.shape[0],
                      sr.refseq.ctypes.data, sr.refseq.shape[0], self.r2q.ctypes.data, self.q2r.ctypes.data,
                      self.r2d.ctypes.data, sr.ref_start, sr.ref_end, int(sr.is_reverse))
        self.norm = None
        self.align = None

    def normalise(self):
        self.norm = Norm()
        return oracle().dno_normalise(C.byref(self.model), C.byref(self.c), C.byref(self.norm))

    def eventalign(self):
        self.align = Align()
        return oracle().dno_eventalign(C.byref(self.model), C.byref(self.c), C.byref(self.norm), C.byref(self.align))

    # numpy views of results -------------------------------------------------------------
    def events(self):
        n = self.norm.n_events
        a = np.ctypeslib.as_array(C.cast(self.norm.events, C.POINTER(C.c_uint8)), shape=(n * 16,))
        return a.view(np.dtype([("mean", "<f8"), ("raw_start", "<u4"), ("raw_len", "<u4")])).copy()

    def _arr(self, p, n, dt):
        if n == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True)

    def alignment(self):
        n = self.norm.n_aln
        return self._arr(self.norm.aln_event, n, np.uint32), self._arr(self.norm.aln_kmer, n, np.uint32)

    def cleaned(self):
        n = self.norm.n_cleaned
        return self._arr(self.norm.cleaned_sig, n, np.float64), self._arr(self.norm.cleaned_rank, n, np.uint32)

    def ranks(self):
        return self._arr(self.norm.rank_q, self.norm.n_kq, np.uint32), self._arr(self.norm.rank_r, self.norm.n_kr, np.uint32)

    def positions(self):
        a = self.align
        n = a.n_pos
        d = dict(coord=self._arr(a.coord, n, np.uint32), query_idx=self._arr(a.query_idx, n, np.uint32),
                 ref_idx=self._arr(a.ref_idx, n, np.uint32), indel=self._arr(a.indel_score, n, np.int32),
                 n_signal=self._arr(a.n_signal, n, np.uint32), core=self._arr(a.core, n, np.float32),
                 residual=self._arr(a.residual, n, np.float32))
        d["kmer"] = np.frombuffer(C.string_at(a.kmer, n * 9), dtype="S9").copy() if n else np.zeros(0, "S9")
        d["signal"] = np.ctypeslib.as_array(a.signal, shape=(n * 20,)).reshape(n, 20).copy() if n else np.zeros((0, 20), np.float32)
        return d

    def windows(self):
        a = self.align
        n = a.n_windows
        return (self._arr(a.win_ref, n, np.uint32), self._arr(a.win_len, n, np.uint32), self._arr(a.win_T, n, np.uint32),
                self._arr(a.win_score, n, np.float64))

    def align_table(self):
        """rows of the `DNAscent align` output (alignment.cpp:697-733), in emission order"""
        a = self.align
        n = a.n_rows
        return dict(coord=self._arr(a.row_coord, n, np.uint32), ref_pos=self._arr(a.row_rpos, n, np.uint32),
                    value=self._arr(a.row_val, n, np.float64), kind=self._arr(a.row_kind, n, np.uint8))

    def format_align(self):
        cap = 256 + 80 * max(1, self.align.n_rows)
        buf = C.create_string_buffer(cap)
        n = oracle().dno_format_align(C.byref(self.model), self.sr.read_id.encode(), self.sr.contig.encode(), C.byref(self.c),
                                      C.byref(self.align), buf, cap)
        assert n <= cap
        return buf.raw[:n]

    def format_detect(self, probs):
        probs = np.ascontiguousarray(probs, np.float32)
        cap = 64 + 64 * max(1, self.align.n_pos)
        buf = C.create_string_buffer(cap)
        n = oracle().dno_format_detect(self.sr.read_id.encode(), self.sr.contig.encode(), C.byref(self.c),
                                       C.byref(self.align), probs.ctypes.data, buf, cap)
        assert n <= cap
        return buf.raw[:n]

    def hmm(self, fit, window=12):
        """llAcrossRead (detect.cpp:393): dict of per-call arrays, in emission order."""
        um, us, am, as_ = (np.ascontiguousarray(a, np.float64) for a in fit)
        self._fit = (um, us, am, as_)
        fm = FitModels(um.ctypes.data, us.ctypes.data, am.ctypes.data, as_.ctypes.data)
        self.hmm_c = Hmm()
        rc = oracle().dno_ll_across_read(C.byref(fm), C.byref(self.c), C.byref(self.norm), window, C.byref(self.hmm_c))
        assert rc == 0
        h = self.hmm_c; n = h.n
        return dict(pos_on_ref=self._arr(h.pos_on_ref, n, np.uint32), pos_on_query=self._arr(h.pos_on_query, n, np.uint32),
                    global_pos=self._arr(h.global_pos, n, np.int32), n_events=self._arr(h.n_events, n, np.uint32),
                    log_analogue=self._arr(h.log_analogue, n, np.float64), log_thymidine=self._arr(h.log_thymidine, n, np.float64),
                    llr=self._arr(h.llr, n, np.float64))

    def format_hmm(self):
        cap = 128 + 64 * max(1, self.hmm_c.n)
        buf = C.create_string_buffer(cap)
        n = oracle().dno_format_hmm(self.sr.read_id.encode(), self.sr.contig.encode(), C.byref(self.c), C.byref(self.hmm_c), buf, cap)
        assert n <= cap
        return buf.raw[:n]

    def modbam(self, probs):
        probs = np.ascontiguousarray(probs, np.float32)
        mm = C.create_string_buffer(16 * int(self.align.n_pos) + 64)
        ml = np.zeros(2 * int(self.align.n_pos) + 1, np.uint8)
        n = oracle().dno_modbam_tags(C.byref(self.c), C.byref(self.align), probs.ctypes.data, mm, len(mm), ml.ctypes.data, ml.shape[0])
        return int(n), mm.value.decode(), ml[:2 * n].copy()

    def free(self):
        if self.norm is not None:
            oracle().dno_norm_free(C.byref(self.norm)); self.norm = None
        if self.align is not None:
            oracle().dno_align_free(C.byref(self.align)); self.align = None
        if getattr(self, "hmm_c", None) is not None:
            oracle().dno_hmm_free(C.byref(self.hmm_c)); self.hmm_c = None


def viterbi(model_mean, obs, seq, shift, scale, epb, sigma=0.14):
    mm = np.ascontiguousarray(model_mean, np.float64)
    m = Model(mm.ctypes.data, sigma)
    obs = np.ascontiguousarray(obs, np.float64)
    T = obs.shape[0]; N = len(seq) - 8
    st = np.zeros(T + N + 8, np.uint8); ps = np.zeros(T + N + 8, np.uint32)
    sc = C.c_double(0); err = C.c_int(0)
    n = oracle().dno_viterbi(C.byref(m), obs.ctypes.data, T, seq if isinstance(seq, bytes) else seq.encode(), len(seq),
                             shift, scale, epb, C.byref(sc), st.ctypes.data, ps.ctypes.data, C.byref(err))
    return sc.value, st[:n].copy(), ps[:n].copy(), err.value


def bench_reads(synth_reads, model_mean, full, threads):
    """dno_bench_reads: the reference's OpenMP loop (one read per thread, schedule(dynamic)) over the given reads.
    Returns (seconds, samples, positions, reads passing)."""
    ors = [OracleRead(r, model_mean) for r in synth_reads]          # int16 -> pA and CIGAR flattening happen here, outside the timing
    arr = (Read * len(ors))(*[o.c for o in ors])
    st = (C.c_int * len(ors))()
    npos = (C.c_uint64 * len(ors))()
    L = oracle()
    L.dno_bench_reads.restype = C.c_double
    L.dno_bench_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    secs = L.dno_bench_reads(C.byref(ors[0].model), arr, len(ors), int(full), int(threads), st, npos)
    return float(secs), sum(r.n_samples() for r in synth_reads), int(sum(npos)), int(sum(1 for x in st if x == 0))
