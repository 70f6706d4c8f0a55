"""GPU: time dn_run_hmm (detect --HMM) on a batch of synthetic reads."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from dnascent_amd import hip, host, synth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
model = synth.pore_model()
ctx = hip.Context(0)
ctx.load_pore_model(model, 0.14); ctx.load_fit_models(*synth.fit_models())
b = host.ReadBatch()
for i in range(n_reads):
    b.add_synth(synth.make_read(9000 + i, n_bases, model=model, sub_rate=0.002))
b.upload(ctx)
ctx.run("normalise"); ctx.run("hmm"); ctx.sync()
ctx.profile(True)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); ctx.run("hmm"); ctx.sync(); best = min(best, time.perf_counter() - t0)
s = ctx.summaries()
print("reads %d calls %d  hmm %.2f ms (incl. host count pass)  %.2f Mcalls/s  samples %.1f M" %
      (n_reads, int(s["n_hmm_calls"].sum()), best * 1e3, s["n_hmm_calls"].sum() / best / 1e6, b.samples() / 1e6))
print({k: v for k, v in ctx.profile_table().items() if "hmm" in k} if hasattr(ctx, "profile_table") else "")
