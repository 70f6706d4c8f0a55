"""GPU: does a PAGEABLE dn_batch_upload on one context slow down while another context's batch is in its per-read stages?  (Round 4: in the first 0.6 s of
the bench's timed region the uploads of batches 1-4 take 100-155 ms each, 25 ms afterwards -- DN_TRACE_SUBMIT.)  Context A gets a batch and a given set of
stages enqueued (nothing waits), then context B uploads a different batch of the same shape; the upload is timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from dnascent_amd import cnn_model, hip, host, synth
model = synth.pore_model()
desc, blob, _ = cnn_model.default_model()
ctxs = [hip.Context(0) for _ in range(3)]
for c in ctxs:
    c.load_pore_model(model, 0.14); c.load_cnn(desc, blob)
bs = []
for i in range(3):
    b = host.ReadBatch(); b.fill_synth(model, 5000 + 1000 * i, 500, 50000); bs.append(b)
for c, b in zip(ctxs, bs):                                      # every context sized and warm (a full batch each)
    b.upload(c); [c.run(s) for s in ("normalise", "eventalign", "cnn")]; c.sync()
A, B, Cc = ctxs


def timed_upload(c, b):
    t = time.perf_counter(); b.upload(c); return (time.perf_counter() - t) * 1e3


print("GPU idle:                                   upload %.1f ms" % timed_upload(B, bs[1])); B.sync()
for what, stages in (("A in segment only", ("segment",)), ("A in normalise (segment, scaling, banded fill + host function, Theil-Sen)", ("normalise",)),
                     ("A in normalise + eventalign", ("normalise", "eventalign")), ("A in normalise + eventalign + cnn", ("normalise", "eventalign", "cnn"))):
    bs[0].upload(A)
    for s in stages:
        A.run(s)
    t1 = timed_upload(B, bs[1])
    t2 = timed_upload(Cc, bs[2])
    t = time.perf_counter(); A.sync(); rest = (time.perf_counter() - t) * 1e3
    print("%-90s upload on B %.1f ms, then on C %.1f ms (A needed %.0f ms more)" % (what + ":", t1, t2, rest))
    B.sync(); Cc.sync()
bs[0].upload(A); [A.run(s) for s in ("normalise", "eventalign", "cnn")]
time.sleep(0.25)                                                 # A is in its network by now
print("A in its network (0.25 s after the enqueue):  upload on B %.1f ms" % timed_upload(B, bs[1]))
A.sync(); B.sync()
