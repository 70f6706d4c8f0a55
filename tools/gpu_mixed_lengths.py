"""GPU: SURVEY config 5 in miniature -- reads with log-normal lengths (1-200 kb).  Batches in arrival order vs batches of similar
length (shard.make_batches), same reads, same number of in-flight slots: Msamples/s of the banded-HMM scope."""
import os, sys, threading, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
from dnascent_amd import hip, host, shard, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = synth.pore_model()
rng = np.random.default_rng(2025)
lens = np.clip(np.exp(rng.normal(np.log(20000), 0.9, N)), 1000, 200000).astype(int)
reads = [synth.make_read(500000 + i, int(l), model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001) for i, l in enumerate(lens)]
samples = np.array([r.adc.shape[0] for r in reads], dtype=np.int64)
budget = 230_000_000
arrival, cur, load = [], [], 0
for i in range(N):
    if cur and (load + samples[i] > budget or len(cur) >= 1000):
        arrival.append(np.array(cur)); cur, load = [], 0
    cur.append(i); load += int(samples[i])
if cur: arrival.append(np.array(cur))
bucketed = shard.make_batches(samples, budget, max_reads=1000)
ctxs = [hip.Context(0) for _ in range(NCTX)]
for c in ctxs: c.load_pore_model(model, 0.14)


def run(plan, label):
    batches = []
    for idx in plan:
        b = host.ReadBatch()
        for i in idx: b.add_synth(reads[int(i)])
        batches.append(b)
    def worker(j):
        for k in range(j, len(batches), NCTX):
            batches[k].upload(ctxs[j]); ctxs[j].run("normalise")
        ctxs[j].sync()
    best = 1e9
    for _ in range(2):
        th = [threading.Thread(target=worker, args=(j,)) for j in range(NCTX)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        best = min(best, time.perf_counter() - t0)
    print("%-22s %2d batches  %.0f ms  %.0f Msamples/s (uploads included, %d slots in flight)" % (label, len(batches), best * 1e3, samples.sum() / best / 1e6, NCTX))


print("reads %d, samples %.0f M, longest %d kb, median %d kb" % (N, samples.sum() / 1e6, lens.max() // 1000, int(np.median(lens)) // 1000))
run(arrival, "arrival order")
run(bucketed, "bucketed by length")
