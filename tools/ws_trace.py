"""GPU, experiment build only (libdnascent_hip.so compiled with -DDN_WS_TRACE=<workgroup id>): shader-clock stamps of the phases of
the third tile of one persistent k3_sep_ws workgroup (its last launch = a 256 -> 256 layer).  Prints cycles between stamps for every wavefront."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dnascent_amd import cnn_model, hip, host, synth

model = synth.pore_model()
desc, blob, _ = cnn_model.default_model()
ctx = hip.Context(0)
ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob)
b = host.ReadBatch()
for i in range(64):
    b.add_synth(synth.make_read(9000 + i, 20000, model=model, sub_rate=0.002))
b.upload(ctx)
ctx.run("normalise"); ctx.run("eventalign"); ctx.run("cnn"); ctx.sync(); ctx.run("cnn"); ctx.sync()
t = np.zeros((16, 64), np.uint64)
rc = hip.lib().dn_debug_ws_trace(C.c_void_p(t.ctypes.data))
assert rc == 0, rc
t0 = int(t[t > 0].min())
names_p = {}
for cb in range(0, 8, 2):
    names_p.update({3 + 4 * cb: "[step %d]" % cb, 4 + 4 * cb: "dw", 5 + 4 * cb: "store+gload", 6 + 4 * cb: "barrier", 7 + 4 * cb: "dw", 8 + 4 * cb: "store+gload", 9 + 4 * cb: "barrier"})
names_c = {3: "[tile start]", 40: "epilogue"}
for cb in range(8):
    names_c.update({4 + 3 * cb: "mma0(%d)" % cb, 5 + 3 * cb: "mma1", 6 + 3 * cb: "barrier"})
ws16 = os.environ.get("DN_CNN_WS16", "0") != "0"       # the 16-wavefront form was an experiment (tools/k3_sep_ws16_experiment.hip); the product kernel has 4 consumers + 4 producers
ncons = 8 if ws16 else 4
for w in range(16 if ws16 else 8):
    names = names_c if w < ncons else names_p
    idx = [i for i in sorted(names) if t[w, i] > 0]
    if not idx:
        continue
    line = []
    prev = None
    for i in idx:
        v = int(t[w, i]) - t0
        line.append("%s %d" % (names[i], v if prev is None else v - prev))
        prev = v
    print("wave %d (%s) total %d: " % (w, "consumer" if w < ncons else "producer", prev - (int(t[w, idx[0]]) - t0)) + " | ".join(line))
