"""GPU, experiment build only (libdnascent_hip.so compiled with -DDN_K2B_TRACE): shader-clock ticks per phase of k2b_eventalign's window
walk, summed over the reads of one batch (argv: reads, bases), and the stage's time alone on the chip (any build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dnascent_amd import hip, host, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
model = synth.pore_model()
ctx = hip.Context(0); ctx.load_pore_model(model, 0.14); ctx.profile(True)
b = host.ReadBatch()
b.fill_synth(model, 777000, n_reads, bases)
b.upload(ctx)
ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
traced = hasattr(hip.lib(), "dn_debug_k2b_trace")
if traced:
    hip.lib().dn_debug_k2b_trace(None, 1)
ctx.profile_reset()
for _ in range(3):
    ctx.run("normalise"); ctx.run("eventalign")
ctx.sync()
for k, v in ctx.profile_get().items():
    if v[1]: print("%-20s %9.3f ms per launch (%d launches)" % (k, v[0] / v[1], v[1]))
if traced:
    t = np.zeros(12, np.uint64)
    assert hip.lib().dn_debug_k2b_trace(C.c_void_p(t.ctypes.data), 0) == 0
    nw = float(t[7])
    names = ["window setup", "event gather", "lattice", "termination + traceback", "fill: position records", "align table", "between windows"]
    names += [None, "fill: label pass", "fill: samples"]
    tot = float(t[:7].sum() + t[8:].sum())
    for i, n in enumerate(names):
        if n is None: continue
        print("%-24s %9.0f ticks per window  %5.1f %%" % (n, float(t[i]) / nw, 100.0 * float(t[i]) / tot))
    print("%-24s %9.0f ticks per window, %d windows" % ("total", tot / nw, int(nw) // 3))
