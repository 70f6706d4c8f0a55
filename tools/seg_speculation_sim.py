"""CPU model of k1_detect's speculation (csrc/k1_segment.hip): for each hostile signal of tests/adversarial_signals.py, run the ORACLE's
t-statistics through the detector state machine twice -- once from the true start (what the reference does), once per 1 024-sample
chunk from the default state `warm` samples before the chunk -- and count the chunks whose state at the chunk start differs
(= the chunks k1_events has to redo).  Also compares oracle and reference event tables where oracle/_ref/libref.so exists (each
reference call in a forked child: on no-peak signals the reference aborts).

    python tools/seg_speculation_sim.py [warm]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po  # noqa: E402
import adversarial_signals as adv  # noqa: E402
from dnascent_amd import synth  # noqa: E402

FLT_MAX = np.float32(3.4028234663852886e38)


class Det:
    __slots__ = ("masked_to", "peak_pos", "peak_val", "valid", "thr", "win")

    def __init__(self, thr, win):
        self.masked_to = 0; self.peak_pos = -1; self.peak_val = FLT_MAX; self.valid = False; self.thr = np.float32(thr); self.win = win

    def key(self, at):
        return (self.peak_pos, float(self.peak_val), self.valid, self.masked_to if self.masked_to >= at else -1)


def walk(t1, t2, lo, hi, s=None, l=None, peaks=None):
    """event_detection.c:136-195 over samples [lo, hi)"""
    s = s or Det(1.4, 3); l = l or Det(9.0, 6)
    ph = np.float32(0.2)
    for i in range(lo, hi):
        for d, t in ((s, t1), (l, t2)):
            if d.masked_to >= i:
                continue
            v = t[i]
            if d.peak_pos == -1:
                if v < d.peak_val:
                    d.peak_val = v
                elif v - d.peak_val > ph:
                    d.peak_val = v; d.peak_pos = i
            else:
                if v > d.peak_val:
                    d.peak_val = v; d.peak_pos = i
                if d is s and d.peak_val > d.thr:
                    l.masked_to = d.peak_pos + d.win; l.peak_pos = -1; l.peak_val = FLT_MAX; l.valid = False
                if d.peak_val - v > ph and d.peak_val > d.thr:
                    d.valid = True
                if d.valid and (i - d.peak_pos) > d.win // 2:
                    if peaks is not None:
                        peaks.append(d.peak_pos)
                    d.peak_pos = -1; d.peak_val = v; d.valid = False
    return s, l


def rechecks(t1, t2, n, warm, chunk=1024):
    import copy
    s, l = Det(1.4, 3), Det(9.0, 6)
    miss = 0
    nch = (n + chunk - 1) // chunk
    for c in range(nch):
        lo = c * chunk
        if c > 0:
            ss, sl = walk(t1, t2, max(1, lo - warm), lo)
            if ss.key(lo) != s.key(lo) or sl.key(lo) != l.key(lo):
                miss += 1
        walk(t1, t2, lo, min(n, lo + chunk), s, l)
    return miss, nch


def _ref_child(conn, raw):
    conn.send(po.ref_detect_events(raw))
    conn.close()


def ref_events(raw):
    """the reference's detect_events in a forked child; None when it aborts"""
    if po.ref() is None:
        return "absent"
    ctx = mp.get_context("fork")
    rx, tx = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_ref_child, args=(tx, raw))
    p.start()
    tx.close()
    try:
        out = rx.recv()             # EOFError when the child died (assert -> abort) before sending
    except EOFError:
        out = None
    p.join()
    return out if p.exitcode == 0 else None


def main():
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    model = synth.pore_model()
    sigs = dict(adv.cases(model))
    sigs["read50kb_with_stalls"] = adv.read50kb_with_stalls(model)
    for k, v in adv.no_peak_cases().items():
        sigs["nopeak:" + k] = v
    for name, adc in sigs.items():
        raw = po.adc_to_pa(adc, *adv.CAL)
        ev, t1, t2, pk = po.detect_events(raw, want_intermediates=True)
        r = ref_events(raw)
        if r is None:
            same = "REFERENCE ABORTS"
        elif r == "absent":
            same = "no libref"
        else:
            same = "oracle == reference" if (np.array_equal(r[0], ev["start"]) and r[1].tobytes() == ev["length"].tobytes() and r[2].tobytes() == ev["mean"].tobytes() and r[3].tobytes() == ev["stdv"].tobytes()) else "ORACLE DIFFERS"
        miss, nch = rechecks(t1, t2, adc.shape[0], warm) if adc.shape[0] <= 700000 else (-1, -1)
        print("%-26s n=%7d events=%6d (1 per %.2f samples)  %s  chunks=%d mis-speculated(warm=%d)=%d" % (name, adc.shape[0], ev.shape[0], adc.shape[0] / max(1, ev.shape[0]), same, nch, warm, miss), flush=True)


if __name__ == "__main__":
    main()
