"""GPU: how long the host spends ENQUEUEING each stage of a batch (the calls only enqueue; they should cost launches, not waits), for a batch of
many short reads against one of few long reads.  Round 4: bench.py --scope mixed showed 120-170 ms per batch inside dn_run_detect."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dnascent_amd import cnn_model, hip, host, synth
model = synth.pore_model()
desc, blob, _ = cnn_model.default_model()
ctx = hip.Context(0); ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob)
for name, n, bases in (("500 x 50 kb", 500, 50000), ("4000 x 3 kb", 4000, 3000), ("4000 x 3 kb again", 4000, 3000), ("1500 x 15 kb", 1500, 15000), ("120 x 200 kb", 120, 200000)):
    b = host.ReadBatch()
    got = b.fill_synth(model, 77000, n, bases)
    t0 = time.perf_counter(); b.upload(ctx); t_up = time.perf_counter() - t0
    ts = []
    for stage in ("normalise", "eventalign", "cnn"):
        t0 = time.perf_counter(); ctx.run(stage); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ctx.sync(); t_sync = time.perf_counter() - t0
    print("%-20s reads %5d  upload %.1f ms | enqueue normalise %.1f ms, eventalign %.1f ms, cnn %.1f ms | then sync %.1f ms" % (
        name, got, t_up * 1e3, ts[0] * 1e3, ts[1] * 1e3, ts[2] * 1e3, t_sync * 1e3), flush=True)
