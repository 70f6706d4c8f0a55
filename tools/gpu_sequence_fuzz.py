"""Exploration: LOW-COMPLEXITY reference sequences (homopolymer runs of 5-40, di- and tri-nucleotide repeats, two-letter stretches, the odd N) with matching signals, forward
reads, device against oracle through normaliseEvents + eventalign.  Inside identical or periodically repeating k-mers the Viterbi has exact ties that the last bits of the
emission decide, so the comparison is made twice:
  * against the oracle with the DEVICE's emission formula (dno_set_device_emission(1): log c + arg instead of log(c exp(arg)); everything else the reference's arithmetic):
    every label, count and feature must agree bit for bit -- anything else is a bug and is printed;
  * against the oracle as the reference computes it: reads whose labels differ are COUNTED (ties decided by libm's last bits), with the window scores required to agree.
    python tools/gpu_sequence_fuzz.py [reads] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import pyoracle as po
from dnascent_amd import hip, host, synth

import adversarial_signals as _adv  # noqa: E402
make = _adv.big_indel_read if os.environ.get("DN_FUZZ_KIND") == "indel" else _adv.low_complexity_read      # DN_FUZZ_KIND=indel: large indels / skips / clips in the mapping


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    model = synth.pore_model()
    reads = [make(model, seed0 + i, [1500, 3000, 5000, 8000][i % 4]) for i in range(n_reads)]
    ctx = hip.Context(0); ctx.load_pore_model(model, 0.14)
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx); ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    L = po.oracle()
    L.dno_set_device_emission.argtypes = [__import__("ctypes").c_int]
    strict_bad = tie_reads = ok = failed = 0
    for i, r in enumerate(reads):
        res = {}
        for mode in (1, 0):
            L.dno_set_device_emission(mode)
            o = po.OracleRead(r, model)
            st = o.normalise(); n = o.norm
            if st == 0:
                st = o.eventalign()
            msg = []
            if s["status"][i] != st: msg.append("status %d vs %d" % (s["status"][i], st))
            if (s["n_scrappie"][i], s["n_events"][i], s["n_aligned"][i], s["n_cleaned"][i]) != (n.n_scrappie, n.n_events, n.n_aln, n.n_cleaned): msg.append("counts")
            labels_differ = False
            if st == 0 and not msg:
                if np.float64(s["shift"][i]).tobytes() != np.float64(n.shift).tobytes() or np.float64(s["scale"][i]).tobytes() != np.float64(n.scale).tobytes(): msg.append("scaling")
                ae, ak = ctx.alignment(i, int(s["n_aligned"][i])); we, wk = o.alignment()
                if not (np.array_equal(ae, we) and np.array_equal(ak, wk)): msg.append("pairs")
                wr, wl, wt, ws = o.windows()
                if int(s["n_windows"][i]) != wr.shape[0]: labels_differ = True; msg.append("window count %d vs %d" % (s["n_windows"][i], wr.shape[0]))
                else:
                    gr, gl, gt, gs = ctx.windows(i, wr.shape[0])
                    if not (np.array_equal(gr, wr) and np.array_equal(gl, wl) and np.array_equal(gt, wt)): labels_differ = True; msg.append("windows")
                    elif not np.allclose(gs, ws, rtol=1e-11, atol=0, equal_nan=True): msg.append("window scores beyond 1e-11")
                if not msg:
                    got, want = ctx.positions(i, int(s["n_positions"][i])), o.positions()
                    if got["coord"].shape != want["coord"].shape: labels_differ = True; msg.append("position count")
                    else:
                        for f in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual", "kmer"):
                            if not np.array_equal(got[f], want[f]): labels_differ = True; msg.append(f)
                        if got["signal"].tobytes() != want["signal"].tobytes(): labels_differ = True; msg.append("signal")
            res[mode] = (st, msg, labels_differ)
            o.free()
        L.dno_set_device_emission(0)
        st, msg, _ = res[1]
        if msg:
            strict_bad += 1; print("read %d (seed %d) DIFFERS FROM THE ORACLE WITH THE DEVICE'S EMISSION: %s" % (i, seed0 + i, "; ".join(msg[:5])), flush=True)
        elif st == 0: ok += 1
        else: failed += 1
        st0, msg0, ld0 = res[0]
        if msg0:
            hard = [m for m in msg0 if m in ("counts", "scaling", "pairs", "window scores beyond 1e-11") or m.startswith("status")]
            if hard: print("read %d (seed %d) vs the reference's emission, NOT a label tie: %s" % (i, seed0 + i, "; ".join(hard)), flush=True)
            tie_reads += 1
    print("%d reads: with the device's emission formula in the oracle %d pass and agree bit for bit, %d fail alike, %d DIFFER; against the reference's emission %d reads "
          "have labels that differ (ties decided by the emission's last bits)" % (n_reads, ok, failed, strict_bad, tie_reads))


if __name__ == "__main__":
    main()
