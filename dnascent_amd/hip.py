"""ctypes binding of the C-ABI (include/dnascent_hip.h -> dnascent_amd/lib/libdnascent_hip.so).

There is no fallback: if the library is missing, or no gfx950 device is usable, creating a Context raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

DN_OK = 0
# dn_read_status (include/dnascent_hip.h)
READ_OK, READ_FAIL_BANDED_QC, READ_FAIL_SCALING, READ_FAIL_NO_END_CELL, READ_FAIL_NEGATIVE_LOG, READ_FAIL_TOO_SHORT, READ_FAIL_WINDOW_EVENTS = range(7)
K_NAMES = ["DN_K_SCAN", "DN_K_TSTAT", "DN_K_DETECT", "DN_K_EVENTS", "DN_K_RANKS", "DN_K_QUANTILE", "DN_K_PREP",
           "DN_K_BAND_FILL", "DN_K_BAND_TRACE", "DN_K_THEILSEN", "DN_K_VITERBI", "DN_K_CNN", "DN_K_HMM"]
DN_K_COUNT = len(K_NAMES)
for _i, _n in enumerate(K_NAMES):
    globals()[_n] = _i

# every symbol include/dnascent_hip.h declares (tests/test_abi.py checks the export list against the header)
SYMBOLS = ["dn_abi_version", "dn_device_count", "dn_ctx_create", "dn_ctx_destroy", "dn_last_error", "dn_sync",
           "dn_load_pore_model", "dn_batch_upload", "dn_host_alloc", "dn_host_free", "dn_host_register", "dn_host_unregister", "dn_run_detect", "dn_collect", "dn_batch_workspace_bytes", "dn_ctx_reserve", "dn_cnn_reserve", "dn_ctx_set_event_bound", "dn_ctx_get_event_bound", "dn_debug_emission", "dn_debug_keep_k1", "dn_debug_seg_warm", "dn_run_segment", "dn_run_rough_scaling", "dn_run_banded",
           "dn_run_theilsen", "dn_run_normalise", "dn_run_eventalign", "dn_set_align_table", "dn_get_align_rows", "dn_get_align_table", "dn_load_cnn", "dn_cnn_set_math", "dn_cnn_range_escalations", "dn_cnn_canaries", "dn_run_cnn", "dn_get_probabilities", "dn_cnn_infer", "dn_load_fit_models", "dn_run_hmm", "dn_get_hmm_calls", "dn_get_summaries", "dn_get_prefix_sums",
           "dn_get_tstats", "dn_get_scrappie_events", "dn_get_events", "dn_get_kmer_ranks", "dn_get_alignment",
           "dn_get_cleaned", "dn_get_trace", "dn_get_positions", "dn_get_windows", "dn_profile_enable", "dn_profile_get",
           "dn_profile_reset", "dn_profile_get_layer", "dn_kernel_name", "dn_device_bytes", "dn_shutdown"]


class BatchDesc(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("adc", C.c_void_p), ("adc_off", C.c_void_p), ("cal_offset", C.c_void_p),
                ("cal_scale", C.c_void_p), ("basecall", C.c_void_p), ("basecall_off", C.c_void_p), ("refseq", C.c_void_p),
                ("refseq_off", C.c_void_p), ("ref2query", C.c_void_p), ("query2ref", C.c_void_p), ("ref2del", C.c_void_p),
                ("ref_start", C.c_void_p), ("ref_end", C.c_void_p), ("is_reverse", C.c_void_p)]


class ResultBatch(C.Structure):
    """dn_result_batch"""
    _fields_ = [("n_reads", C.c_uint32), ("summary", C.c_void_p), ("call_off", C.c_void_p), ("n_calls", C.c_uint64),
                ("ref_coord", C.c_void_p), ("query_idx", C.c_void_p), ("ref_idx", C.c_void_p), ("p_edu", C.c_void_p),
                ("p_brdu", C.c_void_p), ("kmer9", C.c_void_p)]


class CnnOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("k", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32), ("reserved", C.c_int32), ("w", C.c_int64),
                ("scale", C.c_int64), ("shift", C.c_int64), ("aux", C.c_int64 * 6)]


CNN_OPCODE = {"encode_gru": 0, "conv": 1, "dwconv": 2, "add_relu": 3, "dense_softmax": 4, "conv_add": 5}


def cnn_ops_from_description(desc):
    """dnascent_amd.cnn_model description (dict) -> array of dn_cnn_op."""
    ops = (CnnOp * len(desc["ops"]))()
    for i, o in enumerate(desc["ops"]):
        c = ops[i]
        c.op = CNN_OPCODE[o["op"]]
        if o["op"] == "encode_gru":
            c.dst = o["dst"]; c.cout = o["cout"]
            for j, k in enumerate(("g1_kernel", "g1_recurrent", "g1_bias", "g2_kernel", "g2_recurrent", "g2_bias")):
                c.aux[j] = o[k]
        elif o["op"] == "conv":
            c.src, c.dst, c.k, c.cin, c.cout, c.relu = o["src"], o["dst"], o["k"], o["cin"], o["cout"], int(o["relu"])
            if o.get("add", -1) >= 0:
                c.op, c.a = CNN_OPCODE["conv_add"], o["add"]
            c.w, c.scale, c.shift = o["w"], o["scale"], o["shift"]
        elif o["op"] == "dwconv":
            c.src, c.dst, c.k, c.cin, c.cout, c.w = o["src"], o["dst"], o["k"], o["c"], o["c"], o["w"]
        elif o["op"] == "add_relu":
            c.a, c.b, c.dst, c.cin, c.cout = o["a"], o["b"], o["dst"], o["c"], o["c"]
        elif o["op"] == "dense_softmax":
            c.src, c.cin, c.cout, c.w, c.shift = o["src"], o["cin"], o["cout"], o["w"], o["b"]
    return ops


SUMMARY_DTYPE = np.dtype([
    ("status", "<i4"), ("n_samples", "<u4"), ("n_scrappie", "<u4"), ("n_events", "<u4"), ("n_kmers_query", "<u4"),
    ("n_kmers_ref", "<u4"), ("n_bands", "<u4"), ("band_cells", "<u8"), ("rough_shift", "<f8"), ("rough_scale", "<f8"),
    ("end_event", "<i4"), ("n_aligned", "<u4"), ("avg_log_emission", "<f8"), ("spanned", "<i4"), ("max_gap", "<i4"),
    ("n_cleaned", "<u4"), ("ts_slope", "<f8"), ("ts_intercept", "<f8"), ("shift", "<f8"), ("scale", "<f8"),
    ("events_per_base", "<f8"), ("n_positions", "<u4"), ("n_windows", "<u4"), ("detector_rechecks", "<u4"),
    ("n_hmm_calls", "<u4")], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.HIP_SO):
            raise RuntimeError("libdnascent_hip.so missing: run `python -m dnascent_amd.build` (there is no CPU fallback)")
        L = C.CDLL(_build.HIP_SO)
        L.dn_abi_version.restype = C.c_int
        L.dn_device_count.restype = C.c_int
        L.dn_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.dn_ctx_destroy.argtypes = [C.c_void_p]
        L.dn_last_error.restype = C.c_char_p
        L.dn_last_error.argtypes = [C.c_void_p]
        L.dn_load_pore_model.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.dn_batch_upload.argtypes = [C.c_void_p, C.POINTER(BatchDesc)]
        for n in ("dn_sync", "dn_run_segment", "dn_run_rough_scaling", "dn_run_banded", "dn_run_theilsen", "dn_run_normalise",
                  "dn_run_eventalign", "dn_run_detect", "dn_set_align_table", "dn_get_align_rows", "dn_get_align_table", "dn_profile_reset"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.dn_load_cnn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32]
        L.dn_run_cnn.argtypes = [C.c_void_p]
        L.dn_cnn_set_math.argtypes = [C.c_void_p, C.c_int]
        L.dn_cnn_range_escalations.argtypes = [C.c_void_p]; L.dn_cnn_range_escalations.restype = C.c_uint64
        L.dn_cnn_canaries.argtypes = [C.c_void_p]; L.dn_cnn_canaries.restype = C.c_uint64
        L.dn_load_fit_models.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.dn_run_hmm.argtypes = [C.c_void_p]
        L.dn_get_hmm_calls.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 7
        L.dn_cnn_infer.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dn_get_probabilities.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.dn_get_summaries.argtypes = [C.c_void_p, C.c_void_p]
        L.dn_collect.argtypes = [C.c_void_p, C.POINTER(ResultBatch)]
        L.dn_batch_workspace_bytes.argtypes = [C.c_void_p, C.POINTER(BatchDesc), C.POINTER(C.c_uint64)]
        L.dn_ctx_reserve.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.dn_cnn_reserve.argtypes = [C.c_void_p, C.c_uint64]
        L.dn_ctx_set_event_bound.argtypes = [C.c_void_p, C.c_uint32]
        L.dn_ctx_get_event_bound.restype = C.c_uint32
        L.dn_ctx_get_event_bound.argtypes = [C.c_void_p]
        L.dn_debug_emission.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dn_debug_keep_k1.argtypes = [C.c_void_p, C.c_int]
        L.dn_debug_seg_warm.argtypes = [C.c_void_p, C.c_uint32]
        L.dn_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.dn_host_free.argtypes = [C.c_void_p]
        L.dn_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.dn_host_unregister.argtypes = [C.c_void_p]
        L.dn_get_prefix_sums.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dn_get_tstats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dn_get_scrappie_events.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dn_get_events.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dn_get_kmer_ranks.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dn_get_alignment.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dn_get_cleaned.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dn_get_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dn_get_positions.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 9
        L.dn_get_windows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 4
        L.dn_set_align_table.argtypes = [C.c_void_p, C.c_int]
        L.dn_get_align_rows.argtypes = [C.c_void_p, C.c_void_p]
        L.dn_get_align_table.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4
        L.dn_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.dn_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        L.dn_profile_get_layer.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
        L.dn_kernel_name.restype = C.c_char_p
        L.dn_kernel_name.argtypes = [C.c_int]
        L.dn_device_bytes.restype = C.c_size_t
        L.dn_device_bytes.argtypes = [C.c_void_p]
        L.dn_shutdown.restype = C.c_int
        _lib = L
    return _lib


class DnError(RuntimeError):
    pass


class Context:
    """One device context (one per GPU / per rank)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        rc = lib().dn_ctx_create(device, stream, C.byref(self.h))
        if rc != DN_OK:
            raise DnError("dn_ctx_create(device=%d) failed with %d: no usable gfx950 device (no CPU fallback)" % (device, rc))
        self.n_reads = 0
        self._keep = None

    def _chk(self, rc, what):
        if rc != DN_OK:
            raise DnError("%s failed (%d): %s" % (what, rc, lib().dn_last_error(self.h).decode()))

    def close(self):
        if self.h:
            lib().dn_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_pore_model(self, mean, sigma=0.14):
        m = np.ascontiguousarray(mean, np.float64)
        assert m.shape[0] == 262144
        self._chk(lib().dn_load_pore_model(self.h, m.ctypes.data, sigma), "dn_load_pore_model")

    def load_cnn(self, desc, blob):
        ops = cnn_ops_from_description(desc)
        w = np.ascontiguousarray(blob, np.float32)
        self._chk(lib().dn_load_cnn(self.h, ops, len(desc["ops"]), w.ctypes.data, w.shape[0], desc["n_buffers"]), "dn_load_cnn")

    def cnn_set_math(self, mode):
        """'fp32' (exact fp32 MFMA), 'bf16x6' (three-piece bf16 split on the bf16 matrix cores) or 'f16x3' (default: two-piece fp16
        split, half the matrix work; a pass with an activation outside fp16's range is repeated in bf16x6 automatically)."""
        self._chk(lib().dn_cnn_set_math(self.h, {"fp32": 0, "bf16x6": 1, "f16x3": 2}[mode]), "dn_cnn_set_math")

    def cnn_range_escalations(self):
        return int(lib().dn_cnn_range_escalations(self.h))

    def cnn_canaries(self):
        """how many canaries ran (the first sequences of a batch repeated with bf16 pieces and compared on the device)"""
        return int(lib().dn_cnn_canaries(self.h))

    def load_fit_models(self, unl_mean, unl_std, ana_mean, ana_std):
        a = [np.ascontiguousarray(x, np.float64) for x in (unl_mean, unl_std, ana_mean, ana_std)]
        assert all(x.shape == (262144,) for x in a)
        self._chk(lib().dn_load_fit_models(self.h, *[x.ctypes.data for x in a]), "dn_load_fit_models")

    def hmm_calls(self, r, n):
        d = dict(pos_on_ref=np.zeros(n, np.uint32), pos_on_query=np.zeros(n, np.uint32), global_pos=np.zeros(n, np.int32),
                 n_events=np.zeros(n, np.uint32), log_analogue=np.zeros(n), log_thymidine=np.zeros(n), llr=np.zeros(n))
        self._chk(lib().dn_get_hmm_calls(self.h, r, n, *[d[k].ctypes.data for k in ("pos_on_ref", "pos_on_query", "global_pos", "n_events",
                                                                                "log_analogue", "log_thymidine", "llr")]), "dn_get_hmm_calls")
        return d

    def cnn_infer(self, lens, core, resid, signal):
        """dn_cnn_infer: host tensors of several sequences -> probabilities [sum(lens), 3]."""
        lens = np.ascontiguousarray(lens, np.uint32); L = int(lens.sum())
        core = np.ascontiguousarray(core, np.float32); resid = np.ascontiguousarray(resid, np.float32)
        signal = np.ascontiguousarray(signal, np.float32)
        assert core.shape == (L,) and resid.shape == (L,) and signal.shape == (L, 20)
        p = np.zeros((L, 3), np.float32)
        self._chk(lib().dn_cnn_infer(self.h, lens.shape[0], lens.ctypes.data, core.ctypes.data, resid.ctypes.data, signal.ctypes.data,
                                     p.ctypes.data), "dn_cnn_infer")
        return p

    def probabilities(self, r, n):
        p = np.zeros((n, 3), np.float32)
        self._chk(lib().dn_get_probabilities(self.h, r, n, p.ctypes.data), "dn_get_probabilities")
        return p

    def upload(self, desc, n_reads, keep=None):
        self._keep = keep
        self._chk(lib().dn_batch_upload(self.h, C.byref(desc)), "dn_batch_upload")
        self.n_reads = n_reads

    def run(self, stage):
        self._chk(getattr(lib(), "dn_run_" + stage)(self.h), "dn_run_" + stage)

    def sync(self):
        self._chk(lib().dn_sync(self.h), "dn_sync")

    def summaries(self):
        out = np.zeros(self.n_reads, SUMMARY_DTYPE)
        self._chk(lib().dn_get_summaries(self.h, out.ctypes.data), "dn_get_summaries")
        return out

    def workspace_bytes(self, desc):
        """dn_batch_workspace_bytes: the device workspace a batch of this shape needs (sizing pass only; the context holds no batch afterwards)"""
        b = C.c_uint64(0)
        self._chk(lib().dn_batch_workspace_bytes(self.h, C.byref(desc), C.byref(b)), "dn_batch_workspace_bytes")
        return int(b.value)

    def reserve(self, workspace_bytes, collect_bytes=0):
        """dn_ctx_reserve: make the workspace slab hold at least workspace_bytes now (no regrowth -- a device-wide wait -- in the middle of a stream)"""
        self._chk(lib().dn_ctx_reserve(self.h, C.c_uint64(int(workspace_bytes)), C.c_uint64(int(collect_bytes))), "dn_ctx_reserve")

    def cnn_reserve(self, rows=0):
        """dn_cnn_reserve: the activation buffers of this context's CNN lane now (rows = 0: a full pass) instead of inside the first pass's enqueue"""
        self._chk(lib().dn_cnn_reserve(self.h, C.c_uint64(int(rows))), "dn_cnn_reserve")

    def set_event_bound(self, samples_per_event):
        """dn_ctx_set_event_bound: a read's workspace holds samples / samples_per_event + 64 events (default 2 = the detector's own bound; tighter bounds need
        a host that retries on DN_ERR_OVERFLOW: host.DetectStream does)"""
        self._chk(lib().dn_ctx_set_event_bound(self.h, C.c_uint32(int(samples_per_event))), "dn_ctx_set_event_bound")

    def collect(self):
        """dn_collect: the bulk result of the batch as numpy COPIES: summary [n_reads], call_off [n_reads + 1], and per call
        ref_coord / query_idx / ref_idx / p_edu / p_brdu / kmer (S9)."""
        rb = ResultBatch()
        self._chk(lib().dn_collect(self.h, C.byref(rb)), "dn_collect")
        n, k = int(rb.n_reads), int(rb.n_calls)

        def arr(ptr, dtype, cnt):
            if cnt == 0:
                return np.zeros(0, dtype)
            return np.frombuffer((C.c_char * (cnt * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()
        return dict(summary=arr(rb.summary, SUMMARY_DTYPE, n), call_off=arr(rb.call_off, np.uint64, n + 1 if n else 0),
                    ref_coord=arr(rb.ref_coord, np.uint32, k), query_idx=arr(rb.query_idx, np.uint32, k), ref_idx=arr(rb.ref_idx, np.uint32, k),
                    p_edu=arr(rb.p_edu, np.float32, k), p_brdu=arr(rb.p_brdu, np.float32, k), kmer=arr(rb.kmer9, "S9", k))

    def keep_k1(self, on=True):
        """prefix sums / t-statistics also go to HBM for the batches uploaded from now on (taps prefix_sums() / tstats())"""
        self._chk(lib().dn_debug_keep_k1(self.h, int(on)), "dn_debug_keep_k1")

    def seg_warm(self, samples):
        """dn_debug_seg_warm: warm-up of the speculative peak detector (default 192 samples; 0 sends every chunk with a pending peak through the exact redo)"""
        self._chk(lib().dn_debug_seg_warm(self.h, C.c_uint32(int(samples))), "dn_debug_seg_warm")

    def debug_emission(self, x, mu):
        x = np.ascontiguousarray(x, np.float64); mu = np.ascontiguousarray(mu, np.float64)
        out = np.zeros(x.shape[0], np.float64)
        self._chk(lib().dn_debug_emission(self.h, x.shape[0], x.ctypes.data, mu.ctypes.data, out.ctypes.data), "dn_debug_emission")
        return out

    # ---- taps -------------------------------------------------------------------------------
    def prefix_sums(self, r, n):
        a = np.zeros(n + 1); b = np.zeros(n + 1)
        self._chk(lib().dn_get_prefix_sums(self.h, r, n, a.ctypes.data, b.ctypes.data), "dn_get_prefix_sums")
        return a, b

    def tstats(self, r, n):
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        self._chk(lib().dn_get_tstats(self.h, r, n, a.ctypes.data, b.ctypes.data), "dn_get_tstats")
        return a, b

    def scrappie_events(self, r, n):
        st = np.zeros(n, np.uint32); ln = np.zeros(n, np.float32); mn = np.zeros(n, np.float32)
        self._chk(lib().dn_get_scrappie_events(self.h, r, n, st.ctypes.data, ln.ctypes.data, mn.ctypes.data), "dn_get_scrappie_events")
        return st, ln, mn

    def events(self, r, n):
        mean = np.zeros(n); st = np.zeros(n, np.uint32); ln = np.zeros(n, np.uint32)
        self._chk(lib().dn_get_events(self.h, r, n, mean.ctypes.data, st.ctypes.data, ln.ctypes.data), "dn_get_events")
        return mean, st, ln

    def kmer_ranks(self, r, nq, nr):
        q = np.zeros(nq, np.uint32); f = np.zeros(nr, np.uint32)
        self._chk(lib().dn_get_kmer_ranks(self.h, r, nq, nr, q.ctypes.data, f.ctypes.data), "dn_get_kmer_ranks")
        return q, f

    def alignment(self, r, n):
        e = np.zeros(n, np.uint32); k = np.zeros(n, np.uint32)
        self._chk(lib().dn_get_alignment(self.h, r, n, e.ctypes.data, k.ctypes.data), "dn_get_alignment")
        return e, k

    def cleaned(self, r, n):
        s = np.zeros(n); k = np.zeros(n, np.uint32)
        self._chk(lib().dn_get_cleaned(self.h, r, n, s.ctypes.data, k.ctypes.data), "dn_get_cleaned")
        return s, k

    def trace(self, r, n_bands):
        t = np.zeros(n_bands * 100, np.uint8); e = np.zeros(n_bands, np.int32); k = np.zeros(n_bands, np.int32)
        self._chk(lib().dn_get_trace(self.h, r, n_bands, t.ctypes.data, e.ctypes.data, k.ctypes.data), "dn_get_trace")
        return t.reshape(n_bands, 100), e, k

    def positions(self, r, n):
        d = dict(coord=np.zeros(n, np.uint32), query_idx=np.zeros(n, np.uint32), ref_idx=np.zeros(n, np.uint32),
                 indel=np.zeros(n, np.int32), n_signal=np.zeros(n, np.uint32), signal=np.zeros((n, 20), np.float32),
                 core=np.zeros(n, np.float32), residual=np.zeros(n, np.float32))
        km = np.zeros(n * 9, np.uint8)
        self._chk(lib().dn_get_positions(self.h, r, n, d["coord"].ctypes.data, d["query_idx"].ctypes.data, d["ref_idx"].ctypes.data,
                                         d["indel"].ctypes.data, km.ctypes.data, d["n_signal"].ctypes.data, d["signal"].ctypes.data,
                                         d["core"].ctypes.data, d["residual"].ctypes.data), "dn_get_positions")
        d["kmer"] = np.frombuffer(km.tobytes(), dtype="S9").copy() if n else np.zeros(0, "S9")
        return d

    def set_align_table(self, on=True):
        """`DNAscent align`: make the next run("eventalign") also materialise the per-sample event table."""
        self._chk(lib().dn_set_align_table(self.h, int(on)), "dn_set_align_table")

    def align_rows(self, n_reads):
        n = np.zeros(n_reads, np.uint32)
        self._chk(lib().dn_get_align_rows(self.h, n.ctypes.data), "dn_get_align_rows")
        return n

    def align_table(self, r, n):
        d = dict(coord=np.zeros(n, np.uint32), ref_pos=np.zeros(n, np.uint32), value=np.zeros(n, np.float64), kind=np.zeros(n, np.uint8))
        self._chk(lib().dn_get_align_table(self.h, r, n, d["coord"].ctypes.data, d["ref_pos"].ctypes.data, d["value"].ctypes.data,
                                           d["kind"].ctypes.data), "dn_get_align_table")
        return d

    def windows(self, r, n):
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32); t = np.zeros(n, np.uint32); s = np.zeros(n)
        self._chk(lib().dn_get_windows(self.h, r, n, a.ctypes.data, b.ctypes.data, t.ctypes.data, s.ctypes.data), "dn_get_windows")
        return a, b, t, s

    # ---- measurement ------------------------------------------------------------------------
    def profile(self, on=True):
        lib().dn_profile_enable(self.h, int(on))

    def profile_reset(self):
        lib().dn_profile_reset(self.h)

    def profile_get(self):
        out = {}
        for k in range(DN_K_COUNT):
            ms = C.c_double(0); n = C.c_uint32(0)
            lib().dn_profile_get(self.h, k, C.byref(ms), C.byref(n))
            out[lib().dn_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def profile_layers(self, n_ops):
        """[(ms, launches, kernel name)] per op of the loaded CNN description (dn_profile_get_layer)"""
        out = []
        for i in range(n_ops):
            ms = C.c_double(0); n = C.c_uint32(0); buf = C.create_string_buffer(96)
            self._chk(lib().dn_profile_get_layer(self.h, i, C.byref(ms), C.byref(n), buf, len(buf)), "dn_profile_get_layer")
            out.append((ms.value, n.value, buf.value.decode()))
        return out

    def device_bytes(self):
        return int(lib().dn_device_bytes(self.h))
