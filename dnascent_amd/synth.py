"""ctypes binding of the deterministic synthetic-read generator (csrc/host/dn_synth.c)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


class _Spec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_bases", C.c_uint32), ("ref_start", C.c_uint32),
                ("is_reverse", C.c_int32), ("noise_pa", C.c_double), ("mean_dwell", C.c_double),
                ("sub_rate", C.c_double), ("ins_rate", C.c_double), ("del_rate", C.c_double),
                ("soft_clip_head", C.c_uint32), ("soft_clip_tail", C.c_uint32), ("n_unknown", C.c_uint32)]


class _Out(C.Structure):
    _fields_ = [("refseq", C.c_void_p), ("n_ref", C.c_uint32),
                ("basecall", C.c_void_p), ("n_base", C.c_uint32),
                ("cigar_op", C.c_void_p), ("cigar_len", C.c_void_p), ("n_cigar", C.c_uint32),
                ("adc", C.c_void_p), ("n_samples", C.c_size_t),
                ("cal_offset", C.c_float), ("cal_scale", C.c_float),
                ("ref_start", C.c_int32), ("ref_end", C.c_int32), ("is_reverse", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.HOST_SO):
            raise RuntimeError("libdnascent_host.so missing: run `python -m dnascent_amd.build`")
        _lib = C.CDLL(_build.HOST_SO)
        _lib.dns_pore_model.argtypes = [C.c_uint64, C.c_void_p]
        _lib.dns_fit_models.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.dns_max_samples.restype = C.c_size_t
        _lib.dns_max_samples.argtypes = [C.c_uint32]
        _lib.dns_make_read.argtypes = [C.c_void_p, C.POINTER(_Spec), C.POINTER(_Out)]
        _lib.dns_make_read.restype = C.c_int
    return _lib


_model_cache = {}


def pore_model(seed=12345):
    """Synthetic static pore-model means, [262144] float64 in kmer2index order (sigma is 0.14, data_IO.cpp:173)."""
    if seed not in _model_cache:
        m = np.empty(262144, dtype=np.float64)
        lib().dns_pore_model(seed, m.ctypes.data)
        _model_cache[seed] = m
    return _model_cache[seed]


def fit_models(seed=12345):
    """Synthetic fit tables for the --HMM path: (unl_mean, unl_std, ana_mean, ana_std), each [262144] float64."""
    key = ("fit", seed)
    if key not in _model_cache:
        m = pore_model(seed)
        us = np.empty(262144); am = np.empty(262144); as_ = np.empty(262144)
        lib().dns_fit_models(seed, m.ctypes.data, us.ctypes.data, am.ctypes.data, as_.ctypes.data)
        _model_cache[key] = (m, us, am, as_)
    return _model_cache[key]


class SynthRead:
    """One synthetic read: what a BAM record + POD5 row + reference slice give `DNAscent::read` (reads.h:210)."""
    __slots__ = ("read_id", "contig", "refseq", "basecall", "cigar_op", "cigar_len", "adc", "cal_offset",
                 "cal_scale", "ref_start", "ref_end", "is_reverse")

    def n_samples(self):
        return int(self.adc.shape[0])


def make_read(seed, n_bases, model=None, is_reverse=False, noise_pa=1.6, mean_dwell=11.5, sub_rate=0.0,
              ins_rate=0.0, del_rate=0.0, soft_clip_head=0, soft_clip_tail=0, n_unknown=0, ref_start=1000):
    model = pore_model() if model is None else model
    sp = _Spec(seed, n_bases, ref_start, int(is_reverse), noise_pa, mean_dwell, sub_rate, ins_rate, del_rate,
               soft_clip_head, soft_clip_tail, n_unknown)
    cap_q = 2 * n_bases + soft_clip_head + soft_clip_tail + 8
    refseq = np.zeros(n_bases, dtype=np.uint8)
    basecall = np.zeros(cap_q, dtype=np.uint8)
    cop = np.zeros(cap_q, dtype=np.uint32)
    clen = np.zeros(cap_q, dtype=np.uint32)
    adc = np.zeros(lib().dns_max_samples(n_bases), dtype=np.int16)
    out = _Out(refseq.ctypes.data, 0, basecall.ctypes.data, 0, cop.ctypes.data, clen.ctypes.data, 0,
               adc.ctypes.data, 0, 0.0, 0.0, 0, 0, 0)
    rc = lib().dns_make_read(model.ctypes.data, C.byref(sp), C.byref(out))
    if rc != 0:
        raise ValueError("dns_make_read failed (n_bases too small?)")
    r = SynthRead()
    r.read_id = "synth-%016x" % seed
    r.contig = "chrSynth"
    r.refseq = refseq[:out.n_ref].copy()
    r.basecall = basecall[:out.n_base].copy()
    r.cigar_op = cop[:out.n_cigar].copy()
    r.cigar_len = clen[:out.n_cigar].copy()
    r.adc = adc[:out.n_samples].copy()
    r.cal_offset = float(out.cal_offset)
    r.cal_scale = float(out.cal_scale)
    r.ref_start, r.ref_end, r.is_reverse = int(out.ref_start), int(out.ref_end), bool(out.is_reverse)
    return r
