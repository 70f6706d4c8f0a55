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
ctx.run("normalise"); ctx.sync()                           # first step after a context is created sizes the slabs
t0 = time.perf_counter(); ctx.run("normalise"); ctx.sync(); step = time.perf_counter() - t0
print("upload (incl. workspace sizing + allocation) %.1f ms for %.1f M samples; one solo step %.1f ms -> PCIe-inclusive %.0f Msamples/s (resident %.0f)" %
      (best * 1e3, batch.samples() / 1e6, step * 1e3, batch.samples() / (best + step) / 1e6, batch.samples() / step / 1e6))

# sustained: N slots in flight, every step uploads its batch again (host -> HBM) before it runs -- what a host feeding the GPU sees
import threading
for nctx in (4, 8):
    ctxs = [ctx] + [hip.Context(0) for _ in range(nctx - 1)]
    for c in ctxs[1:]:
        c.load_pore_model(model, 0.14)
    for c in ctxs:
        batch.upload(c); c.run("normalise"); c.sync()
    steps = 4 * nctx

    def worker(j):
        for _ in range(j, steps, nctx):
            batch.upload(ctxs[j]); ctxs[j].run("normalise")
        ctxs[j].sync()
    th = [threading.Thread(target=worker, args=(j,)) for j in range(nctx)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%d in flight, upload every step: %.1f ms/step -> %.0f Msamples/s PCIe-inclusive (%.1f GB/s host -> HBM)" %
          (nctx, dt / steps * 1e3, batch.samples() * steps / dt / 1e6, batch.samples() * 2 * steps / dt / 1e9))
    for c in ctxs[1:]:
        c.close()
