"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only where /root/reference exists (this container): oracle/_ref/libref.so is the reference's
scrappie/event_detection.c, probability.cpp and common.cpp compiled in place (oracle/Makefile).  The outputs are data only:
inputs (seeded synthetic int16 signals + calibration) and the reference's results for them.

    python tests/golden/make_golden.py

Files:
  ref_segmentation.npz   for each case: adc (int16), cal_offset, cal_scale -> reference detect_events (start, length,
                         mean, stdv) with the reference's default detector parameters (event_detection.h:19-25)
  ref_segmentation_50kb.npz  the same for ONE 50 kb read (the headline read length): the reference's event table only, the signal by seed + SHA-256
  ref_segmentation_adversarial.npz   hostile signals (tests/adversarial_signals.py: stalls, runs of identical samples, full-scale spikes, steps of
                         2 / 3 / 4 samples, square waves, drift, ramps, uniform noise, 16- and 40-sample reads, a 50 kb read with five stalls) -> the
                         reference's event table (start, mean, stdv; length = the starts' differences) per case, the signal by SHA-256; and the list of
                         no-peak signals with what the reference did on each (it aborts: event_detection.c:215) -- run in a forked child
  ref_common.npz         IUPAC sequences -> reference reverseComplement; fp64 vectors -> reference vectorMean (common.h)
  ref_logspace.npz       argument grids -> reference eexp / eln / lnSum / lnProd / lnGreaterThan / normalPDF (bit patterns)
  ref_emission.npz       (level, observation) pairs -> reference eln(normalPDF(level, 0.14, observation)): builtinViterbi's emission
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from dnascent_amd import synth  # noqa: E402


SEG50 = (950, 50000, dict(sub_rate=0.002, ins_rate=0.001, del_rate=0.001))      # the 50 kb segmentation golden (tests/test_golden.py uses the same tuple)


def _ref_child(conn, raw):
    conn.send(po.ref_detect_events(raw))
    conn.close()


def ref_detect_events_forked(raw):
    """the reference's detect_events in a forked child: None when the child died (assert -> abort)"""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    rx, tx = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_ref_child, args=(tx, raw))
    p.start()
    tx.close()
    try:
        out = rx.recv()
    except EOFError:
        out = None
    p.join()
    return out if p.exitcode == 0 else None


def adversarial(model):
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adversarial_signals as adv
    sigs = dict(adv.cases(model))
    sigs["read50kb_with_stalls"] = adv.read50kb_with_stalls(model)
    out = {"names": np.array(sorted(sigs)), "cal": np.array(adv.CAL, np.float32)}
    for name in sorted(sigs):
        adc = sigs[name]
        raw = ((adc.astype(np.float32) + np.float32(adv.CAL[0])) * np.float32(adv.CAL[1])).astype(np.float64)      # pod5.cpp:60
        st, ln, mn, sd = po.ref_detect_events(raw)
        assert np.array_equal(ln, np.diff(np.concatenate([st, [adc.shape[0]]])).astype(np.float32))                 # length is redundant: not stored
        out["sha_" + name] = np.frombuffer(hashlib.sha256(adc.tobytes()).digest(), np.uint8)
        out["n_" + name] = np.int64(adc.shape[0])
        out["start_" + name] = st.astype(np.uint32)
        out["mean_" + name] = mn
        out["stdv_" + name] = sd
        print("  %-24s %7d samples -> %6d events (one per %.2f samples)" % (name, adc.shape[0], st.shape[0], adc.shape[0] / st.shape[0]))
    nop = adv.no_peak_cases()
    out["nopeak_names"] = np.array(sorted(nop))
    fate = []
    for name in sorted(nop):
        adc = nop[name]
        raw = ((adc.astype(np.float32) + np.float32(adv.CAL[0])) * np.float32(adv.CAL[1])).astype(np.float64)
        r = ref_detect_events_forked(raw)
        fate.append(-1 if r is None else int(r[0].shape[0]))                      # -1: the reference aborted
        out["nopeak_sha_" + name] = np.frombuffer(hashlib.sha256(adc.tobytes()).digest(), np.uint8)
        print("  %-24s %7d samples -> reference %s" % (name, adc.shape[0], "ABORTS (event_detection.c:215)" if r is None else "%d events" % r[0].shape[0]))
    out["nopeak_reference_events"] = np.array(fate, np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_segmentation_adversarial.npz"), **out)
    print("ref_segmentation_adversarial.npz", os.path.getsize(os.path.join(HERE, "ref_segmentation_adversarial.npz")), "bytes")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "adversarial":      # only this fixture (the others do not change when it does)
        if po.ref() is None:
            raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists")
        return adversarial(synth.pore_model())
    ref = po.ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists")
    model = synth.pore_model()
    out = {}
    cases = [(901, 1200, dict()), (902, 2500, dict(noise_pa=3.0)), (903, 4000, dict(is_reverse=True)),
             (904, 1500, dict(noise_pa=6.5)), (905, 6000, dict(mean_dwell=8.0))]
    out["n_cases"] = np.int64(len(cases))
    for i, (seed, nb, kw) in enumerate(cases):
        r = synth.make_read(seed, nb, model=model, **kw)
        raw = ((r.adc.astype(np.float32) + np.float32(r.cal_offset)) * np.float32(r.cal_scale)).astype(np.float64)  # pod5.cpp:60
        st, ln, mn, sd = po.ref_detect_events(raw)
        out["adc_%d" % i] = r.adc
        out["cal_%d" % i] = np.array([r.cal_offset, r.cal_scale], np.float32)
        out["start_%d" % i] = st.astype(np.uint32)
        out["length_%d" % i] = ln
        out["mean_%d" % i] = mn
        out["stdv_%d" % i] = sd
    np.savez_compressed(os.path.join(HERE, "ref_segmentation.npz"), **out)
    # one read of the HEADLINE length (50 kb, ~575 k samples, ~110 k events): the reference-pinned GPU test then covers the length the
    # bench is quoted on (round-2 verdict).  The signal is regenerated from its seed (its SHA-256 is stored), only the reference's event
    # table is kept: start (uint32), length, mean (fp32 bit patterns)
    import hashlib
    r = synth.make_read(*SEG50[:2], model=model, **SEG50[2])
    raw = ((r.adc.astype(np.float32) + np.float32(r.cal_offset)) * np.float32(r.cal_scale)).astype(np.float64)
    st, ln, mn, sd = po.ref_detect_events(raw)
    np.savez_compressed(os.path.join(HERE, "ref_segmentation_50kb.npz"), adc_sha256=np.frombuffer(hashlib.sha256(r.adc.tobytes()).digest(), np.uint8),
                        n_samples=np.int64(r.adc.shape[0]), cal=np.array([r.cal_offset, r.cal_scale], np.float32), start=st.astype(np.uint32), length=ln, mean=mn)

    rng = np.random.default_rng(20251002)
    xs = np.concatenate([[0.0, -0.0, 1.0, 1e-320, 5e-324, 1e308, np.inf, -np.inf, np.nan, 0.14, 745.2, -745.2, -708.5, 709.9],
                         rng.normal(0, 50, 200), -np.abs(rng.normal(0, 400, 100))])
    eexp = np.array([ref.ref_eexp(float(x)) for x in xs])
    eln = []
    eln_neg = []
    for x in xs:
        neg = C.c_int(0)
        eln.append(ref.ref_eln(float(x), C.byref(neg))); eln_neg.append(neg.value)
    a = xs[:64]; b = xs[32:96]
    lnsum = np.array([[ref.ref_lnSum(float(u), float(v)) for v in b] for u in a])
    lnprod = np.array([[ref.ref_lnProd(float(u), float(v)) for v in b] for u in a])
    lngt = np.array([[ref.ref_lnGreaterThan(float(u), float(v)) for v in b] for u in a], np.int8)
    mu = rng.normal(0, 1, 400); x = rng.normal(0, 3, 400)
    x[:8] = [5.3, 5.4, 5.5, 6.0, 40.0, -40.0, 1e6, 0.0]
    npdf = np.array([ref.ref_normalPDF(float(m), 0.14, float(v)) for m, v in zip(mu, x)])
    np.savez_compressed(os.path.join(HERE, "ref_logspace.npz"), xs=xs, eexp=eexp, eln=np.array(eln), eln_neg=np.array(eln_neg, np.int8),
                        a=a, b=b, lnsum=lnsum, lnprod=lnprod, lngt=lngt, mu=mu, x=x, npdf=npdf)
    # emission term of builtinViterbi (alignment.cpp:273,347): eln(normalPDF(mu, 0.14, x)) from the reference's probability.cpp,
    # on levels like the pore table's and observations from the mode out to where exp() underflows (+- 40 sigma and beyond)
    rng2 = np.random.default_rng(20251003)                  # its own stream: the draws below must not shift the other fixtures
    emu = rng2.normal(0, 1, 6000)
    dev = np.concatenate([rng2.normal(0, 0.14, 2000), rng2.normal(0, 0.6, 2000), rng2.uniform(-6.0, 6.0, 1500),
                          rng2.uniform(5.0, 5.6, 250) * rng2.choice([-1, 1], 250), rng2.uniform(-60, 60, 250)])
    ex = emu + dev
    eem = []
    for m, v in zip(emu, ex):
        neg = C.c_int(0)
        eem.append(ref.ref_eln(ref.ref_normalPDF(float(m), 0.14, float(v)), C.byref(neg)))
    np.savez_compressed(os.path.join(HERE, "ref_emission.npz"), mu=emu, x=ex, emission=np.array(eem))
    # common.h: reverseComplement over the IUPAC alphabet the reference accepts, vectorMean on buffers of event means
    alpha = np.frombuffer(b"ATGCUYRKMBDHVNWS", np.uint8)
    seqs = [bytes(alpha[rng.integers(0, 16, n)]) for n in (0, 1, 9, 33, 250)] + [b"ATGCATGCN", b"TTTTTTTTT"]
    rc = [po.ref_reverse_complement(q) for q in seqs]
    vm_in = [rng.normal(95, 14, n) for n in (1, 2, 3, 7, 50)] + [np.array([1e300, 1e300, -1e300]), np.array([0.1] * 10)]
    vm = np.array([po.ref_vector_mean(v) for v in vm_in])
    np.savez_compressed(os.path.join(HERE, "ref_common.npz"), n_seq=np.int64(len(seqs)), n_vm=np.int64(len(vm_in)), vm=vm,
                        **{"seq_%d" % i: np.frombuffer(q, np.uint8) for i, q in enumerate(seqs)},
                        **{"rc_%d" % i: np.frombuffer(q, np.uint8) for i, q in enumerate(rc)},
                        **{"vm_in_%d" % i: v for i, v in enumerate(vm_in)})
    adversarial(model)
    for f in ("ref_segmentation.npz", "ref_segmentation_50kb.npz", "ref_logspace.npz", "ref_emission.npz", "ref_common.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
