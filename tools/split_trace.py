"""GPU, experiment build only (-DDN_SPLIT_TRACE=<workgroup id>): shader-clock stamps of the phases of the third tile of one persistent
k3_sep_split<128, 9> workgroup (its last launch = the last 9-tap 128 -> 128 layer of the network).  Prints ticks between stamps for its four wavefronts."""
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
names = {40: "epilogue"}
for cb in range(4):
    names.update({2 + 6 * cb: "[step %d]" % cb, 3 + 6 * cb: "barrier A", 4 + 6 * cb: "raw tile -> LDS + next request", 5 + 6 * cb: "12 MFMAs", 6 + 6 * cb: "barrier B", 7 + 6 * cb: "B tile -> LDS + depthwise"})
for w in range(4):
    idx = [i for i in sorted(names) if t[w, i] > 0]
    if not idx:
        continue
    line, prev = [], None
    for i in idx:
        v = int(t[w, i])
        line.append("%s %s" % (names[i], "" if prev is None else v - prev))
        prev = v
    print("wave %d: tile %d ticks: " % (w, int(t[w, idx[-1]]) - int(t[w, idx[0]])) + " | ".join(line))
