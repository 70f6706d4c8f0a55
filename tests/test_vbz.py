"""POD5's signal codec (csrc/host/dn_vbz.cpp: delta -> zig-zag -> svb16 -> zstd through dlopen("libzstd.so.1")) against an independent Python rendering
(tests/vbz_codec.py: numpy + pyarrow's bundled zstd): each side decodes what the other encoded, on synthetic reads, hostile signals (full-range jumps that wrap
the 16-bit difference, constant runs, every key-byte remainder) and a 50 kb read; malformed streams are refused.  SURVEY.md s8 f1 (pod5.cpp:57): the
Arrow IPC container around the column is NOT read -- there is no libpod5 and no POD5 file in this image."""
import ctypes as C

import numpy as np
import pytest

import adversarial_signals as adv
import vbz_codec as vz
from dnascent_amd import host, synth


def _svb_c(samples):
    a = np.ascontiguousarray(samples, np.int16)
    L = host.lib()
    L.dnh_svb16_encode.restype = C.c_uint64; L.dnh_svb16_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    dst = np.zeros((a.shape[0] + 7) // 8 + 2 * a.shape[0] + 8, np.uint8)
    n = int(L.dnh_svb16_encode(a.ctypes.data, a.shape[0], dst.ctypes.data))
    return dst[:n].tobytes()


def _signals(model):
    s = {"synthetic_5kb": synth.make_read(8801, 5000, model=model).adc, "synthetic_50kb": synth.make_read(8802, 50000, model=model).adc}
    s.update({k: v for k, v in adv.cases(model).items() if k in ("spikes", "uniform_int16", "bare_ramp", "stall6000_flat", "steps_3", "tiny_16", "saturated_plateaus")})
    s["extremes"] = np.array([0, 32767, -32768, -1, 1, -32768, 32767, 0, 127, 128, -128, -129, 255, 256], np.int16)     # differences that wrap; zig-zag 255 | 256 boundary
    for n in range(0, 18):
        s["ramp_%d" % n] = (np.arange(n) * 300 - 2000).astype(np.int16)                # every key-byte remainder, n = 0 included
    return s


def test_svb16_layer_is_byte_identical_to_the_independent_encoder(model):
    for name, x in _signals(model).items():
        assert _svb_c(x) == vz.svb16_encode(x), name


def test_vbz_round_trips_against_the_independent_codec(model):
    assert host.vbz_available()
    total_raw = total_vbz = 0
    for name, x in _signals(model).items():
        theirs = vz.vbz_encode(x)                                   # pyarrow's zstd frame
        assert np.array_equal(host.vbz_decode(theirs, x.shape[0]), x), name
        ours = host.vbz_encode(x)                                   # libzstd.so.1's frame
        assert np.array_equal(vz.vbz_decode(ours, x.shape[0]), x), name
        assert np.array_equal(host.vbz_decode(ours, x.shape[0]), x), name
        total_raw += 2 * x.shape[0]; total_vbz += len(ours)
    print("VBZ: %.2f MB of int16 -> %.2f MB (%.2f x)" % (total_raw / 1e6, total_vbz / 1e6, total_raw / max(total_vbz, 1)))
    assert total_vbz < 0.8 * total_raw


def test_vbz_refuses_malformed_chunks(model):
    x = synth.make_read(8803, 800, model=model).adc
    good = host.vbz_encode(x)
    with pytest.raises(ValueError):
        host.vbz_decode(good, x.shape[0] + 1)                       # one sample more than the stream holds
    with pytest.raises(ValueError):
        host.vbz_decode(good, x.shape[0] - 9)                       # bytes left over
    with pytest.raises(ValueError):
        host.vbz_decode(good[:len(good) // 2], x.shape[0])          # truncated frame
    with pytest.raises(ValueError):
        host.vbz_decode(b"not a zstd frame at all", 4)
