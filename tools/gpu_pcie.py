"""GPU: host -> HBM upload time of the bench batch (the PCIe-inclusive rate DESIGN.md quotes next to the HBM-resident one)."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import bench
from dnascent_amd import hip, synth
model = synth.pore_model()
batch, reads = bench.make_batch(1000, 20000, 1000003, model)
ctx = hip.Context(0); ctx.load_pore_model(model, 0.14)
batch.upload(ctx); ctx.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); batch.upload(ctx); ctx.sync(); best = min(best, time.perf_counter() - t0)
t0 = time.perf_counter(); ctx.run("normalise"); ctx.sync(); step = time.perf_counter() - t0
print("upload (incl. workspace sizing + allocation) %.1f ms for %.1f M samples; one solo step %.1f ms -> PCIe-inclusive %.0f Msamples/s (resident %.0f)" %
      (best * 1e3, batch.samples() / 1e6, step * 1e3, batch.samples() / (best + step) / 1e6, batch.samples() / step / 1e6))
