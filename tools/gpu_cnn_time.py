"""GPU: time dn_run_cnn on a batch of synthetic reads (K3 is outside the banded-HMM bench scope, so it has its own timer)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dnascent_amd import cnn_model, hip, host, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
model = synth.pore_model()
desc, blob, _ = cnn_model.default_model()
ctx = hip.Context(0)
ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob)
b = host.ReadBatch()
for i in range(n_reads):
    b.add_synth(synth.make_read(9000 + i, n_bases, model=model, sub_rate=0.002))
b.upload(ctx)
ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
s = ctx.summaries()
npos = int(s["n_positions"].sum())
mac = sum(o["k"] * o["cin"] * o["cout"] for o in desc["ops"] if o["op"] == "conv")
for math in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["f16x3"]):
    ctx.cnn_set_math(math)
    ctx.run("cnn"); ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ctx.run("cnn"); ctx.sync(); best = min(best, time.perf_counter() - t0)
    print("math %s reads %d positions %d  cnn %.2f ms  %.2f Mpos/s  network %.2f TFLOP/s algorithmic (peaks: fp32 MFMA 157, 6 x bf16 417, 3 x fp16 833)" %
          (math, n_reads, npos, best * 1e3, npos / best / 1e6, 2.0 * mac * npos / best / 1e12))
