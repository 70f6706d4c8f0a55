"""Golden vectors generated from the compiled REFERENCE (tests/golden/make_golden.py) -- these travel to the GPU box,
where neither /root/reference nor (necessarily) oracle/_ref exists.

CPU: the oracle reproduces them bit for bit.  GPU (-m gpu): the HIP segmentation reproduces them bit for bit.
"""
import ctypes as C
import os

import numpy as np
import pytest

import pyoracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def test_oracle_segmentation_matches_reference_goldens():
    g = np.load(os.path.join(G, "ref_segmentation.npz"))
    for i in range(int(g["n_cases"])):
        cal = g["cal_%d" % i]
        raw = po.adc_to_pa(g["adc_%d" % i], float(cal[0]), float(cal[1]))
        ev = po.detect_events(raw)
        assert np.array_equal(ev["start"], g["start_%d" % i].astype(np.uint64))
        for f in ("length", "mean", "stdv"):
            assert np.array_equal(_bits(ev[f]), _bits(g["%s_%d" % (f, i)])), (i, f)


SEG50 = (950, 50000, dict(sub_rate=0.002, ins_rate=0.001, del_rate=0.001))      # tests/golden/make_golden.py SEG50


def _seg50_read(model):
    import hashlib
    from dnascent_amd import synth
    g = np.load(os.path.join(G, "ref_segmentation_50kb.npz"))
    r = synth.make_read(*SEG50[:2], model=model, **SEG50[2])
    assert r.adc.shape[0] == int(g["n_samples"]) and hashlib.sha256(r.adc.tobytes()).digest() == g["adc_sha256"].tobytes()    # the signal the reference saw
    assert np.array_equal(np.array([r.cal_offset, r.cal_scale], np.float32), g["cal"])
    return r, g


def test_oracle_segmentation_matches_reference_golden_at_50kb(model):
    """the headline read length: one 50 kb read (575 k samples), 110 746 events of the reference's own detect_events"""
    r, g = _seg50_read(model)
    ev = po.detect_events(po.adc_to_pa(r.adc, r.cal_offset, r.cal_scale))
    assert ev["start"].shape[0] == g["start"].shape[0] > 100000
    assert np.array_equal(ev["start"], g["start"].astype(np.uint64))
    assert np.array_equal(_bits(ev["length"]), _bits(g["length"])) and np.array_equal(_bits(ev["mean"]), _bits(g["mean"]))


def test_oracle_logspace_matches_reference_goldens():
    g = np.load(os.path.join(G, "ref_logspace.npz"))
    o = po.oracle()
    assert np.array_equal(_bits(np.array([o.dno_eexp(float(x)) for x in g["xs"]])), _bits(g["eexp"]))
    for x, want, wneg in zip(g["xs"], g["eln"], g["eln_neg"]):
        neg = C.c_int(0)
        got = o.dno_eln(float(x), C.byref(neg))
        assert np.float64(got).tobytes() == np.float64(want).tobytes() and neg.value == wneg
    for i, u in enumerate(g["a"]):
        for j, v in enumerate(g["b"]):
            assert np.float64(o.dno_lnSum(float(u), float(v))).tobytes() == np.float64(g["lnsum"][i, j]).tobytes()
            assert np.float64(o.dno_lnProd(float(u), float(v))).tobytes() == np.float64(g["lnprod"][i, j]).tobytes()
            assert o.dno_lnGreaterThan(float(u), float(v)) == g["lngt"][i, j]
    got = np.array([o.dno_normalPDF(float(m), 0.14, float(v)) for m, v in zip(g["mu"], g["x"])])
    assert np.array_equal(_bits(got), _bits(g["npdf"]))


def test_oracle_emission_matches_reference_goldens():
    """builtinViterbi's emission eln(normalPDF(mu, 0.14, x)) (alignment.cpp:273,347): the oracle's restatement against values the
    reference's probability.cpp produced (tests/golden/ref_emission.npz, 6 000 pairs from the mode out to the exp() underflow)."""
    g = np.load(os.path.join(G, "ref_emission.npz"))
    o = po.oracle()
    got = []
    for m, v in zip(g["mu"], g["x"]):
        neg = C.c_int(0)
        got.append(o.dno_eln(o.dno_normalPDF(float(m), 0.14, float(v)), C.byref(neg)))
    assert np.array_equal(_bits(np.array(got)), _bits(g["emission"]))
    assert 100 < np.isnan(g["emission"]).sum() < 1000 and np.nanmax(g["emission"]) > 1.04          # log 0 and the mode are both covered


@pytest.mark.gpu
def test_hip_emission_matches_reference_goldens(model):
    """The emission term as the DEVICE lattice evaluates it (dn_debug_emission: log c + arg where exp(arg) is a normal number,
    the literal exp -> log chain below that) against the reference's own values.  Where exp() is normal the two agree to a few
    ulps of the result (1e-12 relative: glibc vs the device library); in the band where the reference's exp() is subnormal
    (|x - mu| > 37 sigma) a one-ulp difference of a subnormal is a large relative difference, so the bar there is 0.5 absolute on
    a value near -720; log 0 (NaN) must appear for the same pairs except within that band's last 5 units."""
    from dnascent_amd import hip
    g = np.load(os.path.join(G, "ref_emission.npz"))
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    got, want = ctx.debug_emission(g["x"], g["mu"]), g["emission"]
    ctx.close()
    normal = np.isfinite(want) & (want > -700.0)
    assert normal.sum() > 5000
    rel = np.abs(got[normal] - want[normal]) / np.maximum(1.0, np.abs(want[normal]))
    assert rel.max() < 1e-12, rel.max()
    edge = np.isfinite(want) & ~normal
    assert np.all(np.isfinite(got[edge]) | (want[edge] < -740.0)) and np.nanmax(np.abs(got[edge] - want[edge])) < 0.5
    zero = np.isnan(want)
    assert np.all(np.isnan(got[zero]) | (got[zero] < -740.0))
    print("emission: max relative difference %.2e over %d pairs; %d subnormal-band pairs; %d log-0 pairs" % (rel.max(), normal.sum(), edge.sum(), zero.sum()))


@pytest.mark.gpu
def test_hip_segmentation_matches_reference_goldens(model):
    """The device segmentation (through the C-ABI) against the reference's own detect_events output."""
    from dnascent_amd import hip, host, synth
    g = np.load(os.path.join(G, "ref_segmentation.npz"))
    n = int(g["n_cases"])
    # same generator calls as make_golden.py (the adc arrays are also stored in the fixture and compared)
    cases = [(901, 1200, dict()), (902, 2500, dict(noise_pa=3.0)), (903, 4000, dict(is_reverse=True)),
             (904, 1500, dict(noise_pa=6.5)), (905, 6000, dict(mean_dwell=8.0))]
    reads = [synth.make_read(s, nb, model=model, **kw) for s, nb, kw in cases]
    b = host.ReadBatch()
    for i, r in enumerate(reads):
        assert np.array_equal(r.adc, g["adc_%d" % i])
        b.add_synth(r)
    for taps in (True, False):
        ctx = hip.Context(0)
        ctx.load_pore_model(model)
        ctx.keep_k1(taps)                  # the scrappie table itself leaves the kernel only as a tap (same registers, one more store)
        b.upload(ctx)
        ctx.run("segment")
        s = ctx.summaries()
        for i in range(n):
            gs, gm = g["start_%d" % i], g["mean_%d" % i]
            assert s["n_scrappie"][i] == gs.shape[0]
            if taps:
                st, ln, mn = ctx.scrappie_events(i, int(s["n_scrappie"][i]))
                assert np.array_equal(st, gs)
                assert np.array_equal(_bits(ln), _bits(g["length_%d" % i]))
                assert np.array_equal(_bits(mn), _bits(gm))
            # what the product path keeps: the DNAscent events built from the reference's table (event_handling.cpp:549-575:
            # kept = index > 0 and mean > 0; an event carries mean / start of the previous kept index, 0.0 / 0 for the first)
            kept = np.flatnonzero((np.arange(gs.shape[0]) > 0) & (gm.astype(np.float64) > 0))
            prev = np.concatenate([[-1], kept[:-1]])
            want_mean = np.where(prev >= 0, gm[np.maximum(prev, 0)].astype(np.float64), 0.0)
            want_start = np.where(prev >= 0, gs[np.maximum(prev, 0)], 0).astype(np.uint32)
            last = np.minimum(gs[kept].astype(np.int64) - 1, int(s["n_samples"][i]) - 1)
            want_len = np.maximum(last - want_start.astype(np.int64) + 1, 0).astype(np.uint32)
            assert s["n_events"][i] == kept.shape[0]
            em, es, el = ctx.events(i, int(s["n_events"][i]))
            assert np.array_equal(_bits(em), _bits(want_mean)) and np.array_equal(es, want_start) and np.array_equal(el, want_len)
        ctx.close()


@pytest.mark.gpu
def test_hip_segmentation_matches_reference_golden_at_50kb(model):
    """The device segmentation of a 50 kb read -- the bench's read length -- against the REFERENCE's event table (not the oracle's): the
    scrappie table through the tap, and the DNAscent events the product path keeps (event_handling.cpp:549-575), bit for bit; the read
    sits between two others so that its chunks are not the first of the batch."""
    from dnascent_amd import hip, host, synth
    r, g = _seg50_read(model)
    others = [synth.make_read(951, 3000, model=model), synth.make_read(952, 7000, model=model, is_reverse=True)]
    b = host.ReadBatch()
    for x in (others[0], r, others[1]):
        assert b.add_synth(x) >= 0
    gs, gm = g["start"], g["mean"]
    for taps in (True, False):
        ctx = hip.Context(0)
        ctx.load_pore_model(model)
        ctx.keep_k1(taps)
        b.upload(ctx)
        ctx.run("segment")
        s = ctx.summaries()
        assert s["n_scrappie"][1] == gs.shape[0] and s["detector_rechecks"][1] == 0
        if taps:
            st, ln, mn = ctx.scrappie_events(1, int(s["n_scrappie"][1]))
            assert np.array_equal(st, gs) and np.array_equal(_bits(ln), _bits(g["length"])) and np.array_equal(_bits(mn), _bits(gm))
        kept = np.flatnonzero((np.arange(gs.shape[0]) > 0) & (gm.astype(np.float64) > 0))
        prev = np.concatenate([[-1], kept[:-1]])
        want_mean = np.where(prev >= 0, gm[np.maximum(prev, 0)].astype(np.float64), 0.0)
        want_start = np.where(prev >= 0, gs[np.maximum(prev, 0)], 0).astype(np.uint32)
        last = np.minimum(gs[kept].astype(np.int64) - 1, int(s["n_samples"][1]) - 1)
        want_len = np.maximum(last - want_start.astype(np.int64) + 1, 0).astype(np.uint32)
        assert s["n_events"][1] == kept.shape[0]
        em, es, el = ctx.events(1, int(s["n_events"][1]))
        assert np.array_equal(_bits(em), _bits(want_mean)) and np.array_equal(es, want_start) and np.array_equal(el, want_len)
        ctx.close()


def test_common_helpers_match_reference_goldens():
    """common.h reverseComplement / vectorMean: the oracle's restatements and the host C++ layer's reverseComplement
    against what the reference's own common.h returned (tests/golden/ref_common.npz)."""
    from dnascent_amd import host
    g = np.load(os.path.join(G, "ref_common.npz"))
    for i in range(int(g["n_seq"])):
        q, want = g["seq_%d" % i].tobytes(), g["rc_%d" % i].tobytes()
        assert po.reverse_complement(q) == want
        assert host.revcomp(np.frombuffer(q, np.uint8)).tobytes() == want
    for i in range(int(g["n_vm"])):
        got = po.vector_mean(g["vm_in_%d" % i])
        assert _bits(np.float64(got)) == _bits(np.float64(g["vm"][i]))
