"""First-contact debugging of the eventalign kernel: per read, where do windows / positions first differ from the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
from dnascent_amd import hip, host, synth

specs = [(201, 3000, {}), (202, 5000, dict(is_reverse=True)), (203, 5000, dict(sub_rate=0.003, ins_rate=0.001, del_rate=0.001)),
         (205, 4000, dict(n_unknown=3)), (207, 20000, dict(sub_rate=0.002))]
model = synth.pore_model()
reads = [synth.make_read(s, n, model=model, **kw) for s, n, kw in specs]
ctx = hip.Context(0); ctx.load_pore_model(model); ctx.profile(True)
b = host.ReadBatch()
for r in reads: b.add_synth(r)
b.upload(ctx); ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
s = ctx.summaries()
for i, r in enumerate(reads):
    o = po.OracleRead(r, model); st = o.normalise(); o.eventalign()
    wr, wl, wt, ws = o.windows()
    print("read %d status gpu %d oracle %d windows %d/%d positions %d/%d" % (i, s["status"][i], st, s["n_windows"][i], wr.shape[0], s["n_positions"][i], o.align.n_pos))
    m = min(int(s["n_windows"][i]), wr.shape[0])
    gr, gl, gt, gs = ctx.windows(i, int(s["n_windows"][i]))
    for name, a, bb in (("ref", gr, wr), ("len", gl, wl), ("T", gt, wt)):
        d = np.nonzero(a[:m] != bb[:m])[0]
        print("   win.%s: %s" % (name, "ok" if d.size == 0 else "first diff at %d gpu %d oracle %d" % (d[0], a[d[0]], bb[d[0]])))
    rel = np.abs(gs[:m] - ws[:m]) / np.maximum(np.abs(ws[:m]), 1e-300)
    print("   score max rel diff %.3g  (bit-identical %d of %d)" % (np.nanmax(rel) if m else 0, int(np.sum(gs[:m] == ws[:m])), m))
    p = o.positions(); n = min(int(s["n_positions"][i]), p["coord"].shape[0])
    g = ctx.positions(i, int(s["n_positions"][i]))
    for f in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual"):
        d = np.nonzero(g[f][:n] != p[f][:n])[0]
        print("   pos.%s: %s" % (f, "ok" if d.size == 0 else "%d diffs first at %d gpu %r oracle %r" % (d.size, d[0], g[f][d[0]], p[f][d[0]])))
    d = np.nonzero(np.any(g["signal"][:n].view(np.uint32) != p["signal"][:n].view(np.uint32), axis=1))[0]
    print("   pos.signal: %s" % ("ok" if d.size == 0 else "%d rows differ, first %d: gpu %r oracle %r" % (d.size, d[0], g["signal"][d[0]][:6], p["signal"][d[0]][:6])))
    o.free()
for k, v in ctx.profile_get().items():
    if v[1]: print("   %-20s %9.3f ms" % (k, v[0]))
