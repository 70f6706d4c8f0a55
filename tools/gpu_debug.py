"""First-contact GPU debugging: run normaliseEvents stage by stage and report, per read and per tap, where the device
result first departs from the oracle.  Test infrastructure (imports oracle/)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from dnascent_amd import hip, host, synth  # noqa: E402


def first_diff(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    if a.dtype.kind == "f":
        d = np.nonzero(a.view(np.uint32 if a.dtype == np.float32 else np.uint64) != b.view(np.uint32 if b.dtype == np.float32 else np.uint64))[0]
    else:
        d = np.nonzero(a != b)[0]
    if d.size == 0:
        return None
    i = int(d[0])
    return "%d diffs, first at %d: gpu %r vs oracle %r" % (d.size, i, a[i], b[i])


def main():
    specs = [(101, 1500, {}), (102, 3000, dict(is_reverse=True)), (103, 5000, dict(sub_rate=0.02, ins_rate=0.01, del_rate=0.01)),
             (106, 3000, dict(noise_pa=6.5)), (107, 900, {}), (108, 20000, dict(sub_rate=0.01))]
    if len(sys.argv) > 1:
        specs = specs[:int(sys.argv[1])]
    model = synth.pore_model()
    reads = [synth.make_read(s, n, model=model, **kw) for s, n, kw in specs]
    ctx = hip.Context(0)
    ctx.load_pore_model(model)
    ctx.profile(True)
    b = host.ReadBatch()
    for r in reads:
        b.add_synth(r)
    b.upload(ctx)
    print("device bytes after upload: %.1f MB" % (ctx.device_bytes() / 1e6))
    oracles = []
    for r in reads:
        o = po.OracleRead(r, model); o.normalise(); oracles.append(o)

    def stage(name):
        t = time.time(); ctx.run(name); ctx.sync(); print("== stage %s: %.2f ms" % (name, (time.time() - t) * 1e3), flush=True)

    stage("segment")
    summ = ctx.summaries()
    for i, o in enumerate(oracles):
        n = o.raw.shape[0]
        s, q = ctx.prefix_sums(i, n)
        ws = np.concatenate([[0.0], np.cumsum(o.raw)]); wq = np.concatenate([[0.0], np.cumsum(o.raw * o.raw)])
        ev, t1, t2, pk = po.detect_events(o.raw, want_intermediates=True)
        g1, g2 = ctx.tstats(i, n)
        print("read %d n=%d: sum %s | sumsq %s | t1 %s | t2 %s" % (i, n, first_diff(s, ws), first_diff(q, wq), first_diff(g1, t1), first_diff(g2, t2)))
        print("   n_scrappie gpu %d oracle %d  rechecks %d" % (summ["n_scrappie"][i], ev.shape[0], summ["detector_rechecks"][i]))
        m = min(int(summ["n_scrappie"][i]), ev.shape[0])
        st, ln, mn = ctx.scrappie_events(i, int(summ["n_scrappie"][i]))
        print("   et.start %s | et.mean %s" % (first_diff(st[:m].astype(np.uint64), ev["start"][:m]), first_diff(mn[:m], ev["mean"][:m])))
        oe = o.events()
        print("   n_events gpu %d oracle %d" % (summ["n_events"][i], oe.shape[0]))
        m = min(int(summ["n_events"][i]), oe.shape[0])
        mean, es, el = ctx.events(i, int(summ["n_events"][i]))
        print("   ev.mean %s | raw_start %s | raw_len %s" % (first_diff(mean[:m], oe["mean"][:m]), first_diff(es[:m], oe["raw_start"][:m]), first_diff(el[:m], oe["raw_len"][:m])))
        rq, rr = o.ranks(); gq, gr = ctx.kmer_ranks(i, rq.shape[0], rr.shape[0])
        print("   rank_q %s | rank_r %s" % (first_diff(gq, rq), first_diff(gr, rr)))
    stage("rough_scaling")
    summ = ctx.summaries()
    for i, o in enumerate(oracles):
        print("read %d rough shift gpu %.17g oracle %.17g | scale gpu %.17g oracle %.17g" % (i, summ["rough_shift"][i], o.norm.q_shift, summ["rough_scale"][i], o.norm.q_scale))
    stage("banded")
    summ = ctx.summaries()
    for i, o in enumerate(oracles):
        n = o.norm
        print("read %d status gpu %d | bands %d/%d end_event %d/%d n_aligned %d/%d avg %.17g/%.17g spanned %d/%d gap %d/%d ncl %d/%d" % (
            i, summ["status"][i], summ["n_bands"][i], n.n_bands, summ["end_event"][i], n.end_event, summ["n_aligned"][i], n.n_aln,
            summ["avg_log_emission"][i], n.avg_log_emission, summ["spanned"][i], n.spanned, summ["max_gap"][i], n.max_gap,
            summ["n_cleaned"][i], n.n_cleaned))
        oe, ok = o.alignment()
        if summ["n_aligned"][i]:
            ge, gk = ctx.alignment(i, int(summ["n_aligned"][i]))
            m = min(ge.shape[0], oe.shape[0])
            print("   aln (from the end) event %s | kmer %s" % (first_diff(ge[::-1][:m], oe[::-1][:m]), first_diff(gk[::-1][:m], ok[::-1][:m])))
        cs, cr = o.cleaned()
        if summ["n_cleaned"][i]:
            gs, gr = ctx.cleaned(i, int(summ["n_cleaned"][i]))
            m = min(gs.shape[0], cs.shape[0])
            print("   cleaned sig %s | rank %s" % (first_diff(gs[:m], cs[:m]), first_diff(gr[:m], cr[:m])))
    stage("theilsen")
    summ = ctx.summaries()
    for i, o in enumerate(oracles):
        n = o.norm
        print("read %d final status %d/%d slope %.17g/%.17g icpt %.17g/%.17g shift %.17g/%.17g scale %.17g/%.17g epb %.17g/%.17g" % (
            i, summ["status"][i], n.status, summ["ts_slope"][i], n.ts_slope, summ["ts_intercept"][i], n.ts_intercept, summ["shift"][i], n.shift,
            summ["scale"][i], n.scale, summ["events_per_base"][i], n.events_per_base))
    print("kernel times (ms, launches):")
    for k, v in ctx.profile_get().items():
        print("   %-20s %9.3f  %d" % (k, v[0], v[1]))


if __name__ == "__main__":
    main()
