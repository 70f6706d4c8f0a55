"""GPU: how long do eight contexts' workspace reservations take one after the other, and from several host threads at once?  (run_detect's set-up)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnascent_amd import hip
GB = 14.5e9
for nthreads in (1, 2, 4, 8):
    ctxs = [hip.Context(0) for _ in range(8)]
    t0 = time.time()
    def work(ids):
        for i in ids:
            ctxs[i].reserve(int(GB), collect_bytes=1 << 28)
    ths = [threading.Thread(target=work, args=(list(range(k, 8, nthreads)),)) for k in range(nthreads)]
    for t in ths: t.start()
    for t in ths: t.join()
    t1 = time.time()
    for c in ctxs: c.close()
    print("%d thread(s): 8 x %.1f GB reserved in %.2f s, freed in %.2f s" % (nthreads, GB / 1e9, t1 - t0, time.time() - t1), flush=True)
