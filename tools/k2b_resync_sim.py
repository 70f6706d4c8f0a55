"""Would a SPECULATIVE multi-segment eventalign re-synchronise?  (round-5 verdict, next-round item 4; CPU only, the oracle's window chain.)

eventalign's windows are a serial chain per read (alignment.cpp:556-740): the state between two windows is (ri, readHead).  The idea: cut a read into segments,
start each at a GUESSED state -- a reference index ri0 taken from nowhere in particular, readHead0 = the first rough-alignment pair whose query k-mer is at or
beyond refToQuery[ri0] -- and keep a segment's results from the first window on whose (ri, readHead-after-the-scan) equals a window of the true chain.  This tool
measures, on synthetic reads, how many windows a guessed start needs until it falls into step with the true chain (window ends follow the SEQUENCE's break-points,
:574-594, so two chains in the same region tend to pick the same end), and how often it never does within a bound.

    python tools/k2b_resync_sim.py [n_reads] [bases] [starts per read]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from dnascent_amd import synth  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bases = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    starts = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    L = po.oracle()
    L.dno_eventalign_chain.restype = C.c_size_t
    L.dno_eventalign_chain.argtypes = [C.POINTER(po.Model), C.POINTER(po.Read), C.POINTER(po.Norm), C.c_uint, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    model = synth.pore_model()
    need = []
    never = 0
    for i in range(n_reads):
        r = synth.make_read(9100 + i, bases, model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001)
        o = po.OracleRead(r, model)
        if o.normalise() != 0:
            print("read %d failed normalise" % i); continue
        cap = bases // 8 + 64
        ri = np.zeros(cap, np.uint32); rh = np.zeros(cap, np.int32)
        nw = int(L.dno_eventalign_chain(C.byref(o.model), C.byref(o.c), C.byref(o.norm), 0, 0, cap, ri.ctypes.data, rh.ctypes.data, 0))
        true = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(ri[:nw], rh[:nw]))}
        true_ri = set(int(a) for a in ri[:nw])
        ak = np.ctypeslib.as_array(o.norm.aln_kmer, shape=(o.norm.n_aln,))
        # (b) the chain the SEQUENCE alone predicts (no Viterbi: every window's last match at its last position): how far does it stay in step with the true one?
        pri = np.zeros(cap, np.uint32); prh = np.zeros(cap, np.int32)
        pw = int(L.dno_eventalign_chain(C.byref(o.model), C.byref(o.c), C.byref(o.norm), 0, 0, cap, pri.ctypes.data, prh.ctypes.data, 1))
        same = np.array([int(pri[k]) in true_ri for k in range(pw)])
        steps = np.diff(ri[:nw].astype(np.int64)); wl = None
        first_dev = int(np.argmin(same)) if not same.all() else pw
        at = [int(same[min(pw - 1, (q * pw) // 4)]) for q in (1, 2, 3)]
        print("   sequence-only chain: %d windows (true %d); first window NOT a true window start: #%d; fraction of its windows that are true window starts: %.3f; "
              "in step at 1/4, 1/2, 3/4 of the read: %s" % (pw, nw, first_dev, same.mean(), at))
        n_ref = r.refseq.shape[0]
        rng = np.random.default_rng(i)
        sri = np.zeros(64, np.uint32); srh = np.zeros(64, np.int32)
        for ri0 in rng.integers(100, n_ref - 300, starts):
            ri0 = int(ri0)
            qlo = int(o.r2q[ri0])
            rh0 = int(np.searchsorted(ak, qlo, side="left"))                       # first pair whose query k-mer >= qlo (the pairs are sorted by k-mer)
            k = int(L.dno_eventalign_chain(C.byref(o.model), C.byref(o.c), C.byref(o.norm), ri0, rh0, 64, sri.ctypes.data, srh.ctypes.data, 0))
            hit = next((w for w in range(k) if (int(sri[w]), int(srh[w])) in true), None)
            hit_ri = next((w for w in range(k) if int(sri[w]) in true_ri), None)
            if hit is None:
                never += 1
            else:
                need.append((hit, hit_ri))
        o.free()
        print("read %d (%s, %d windows): so far %d starts, never in step within 64 windows: %d" % (i, "rev" if i & 1 else "fwd", nw, len(need) + never, never), flush=True)
    a = np.array([h for h, _ in need]); b = np.array([h for _, h in need])
    print("windows until (ri, readHead) equals the true chain's: mean %.2f, median %d, 90%% %d, 99%% %d, max %d; same ri alone: mean %.2f; never (of %d): %d" % (
        a.mean(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max(), b.mean(), len(need) + never, never))
    print("histogram of windows-to-sync:", np.bincount(a)[:20].tolist())


if __name__ == "__main__":
    main()
