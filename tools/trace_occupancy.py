"""From a rocprofv3 --kernel-trace CSV of a bench.py run: is the GPU ever idle, how many network kernels (one per CNN lane at most) run at once, and at
what rate batches complete in the steady part of the run -- against the average over the whole timed region, which also pays for filling and draining
the pipeline.  The steady part is the span between the 33rd and the 83rd percentile of the network's pass completions (k3_dense_softmax ends a pass;
warm-up, fill, drain and the solo batch bench.py times afterwards lie outside it).
    python tools/trace_occupancy.py <kernel_trace.csv> [passes_per_batch=7]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ppb = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")) for r in rows)
dense = sorted(e for s, e, n in ks if n.startswith("k3_dense_softmax"))
t0 = min(s for s, e, n in ks if n.startswith("k3_"))
t1 = max(e for s, e, n in ks)
BIN = 0.25e9
nb = int((t1 - t0) / BIN) + 1
idle = [BIN] * nb; n3 = [0.0] * nb


def spread(a, b, arr, sign):
    i = int((a - t0) / BIN)
    while a < b and i < nb:
        e = min(b, t0 + (i + 1) * BIN)
        arr[i] += sign * (e - a); a = e; i += 1


reach = t0
for s, e, n in ks:
    if e <= t0:
        continue
    s2 = max(s, reach, t0)
    if e > s2:
        spread(s2, e, idle, -1.0); reach = e
    if n.startswith("k3_"):
        spread(max(s, t0), e, n3, 1.0)
print("from the network's first kernel, per 0.25 s:  %% of the bin with NO kernel running | mean number of network kernels in flight")
print("  ".join("%2.0f|%.1f" % (100.0 * max(0.0, idle[i]) / BIN, n3[i] / BIN) for i in range(nb)))
lo, hi = dense[len(dense) // 3], dense[len(dense) * 5 // 6]
npass = len(dense) * 5 // 6 - len(dense) // 3
print("steady part: %.2f s, %d passes: one every %.1f ms = %.0f ms per batch of %d passes" % ((hi - lo) / 1e9, npass, (hi - lo) / 1e6 / npass, ppb * (hi - lo) / 1e6 / npass, ppb))


def fam(n):
    for p in ("k3_conv", "k3_sep_ws", "k3_sep_split", "k3_encode", "k3_", "k2b_", "k2_", "k1_", "kc_", "k_"):
        if n.startswith(p):
            return p
    return "other"


ev = []
for s, e, n in ks:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        f = fam(n)
        ev.append((s, 1, f)); ev.append((e, -1, f))
ev.sort()
depth = 0; last = lo
by_depth = collections.Counter(); live = collections.Counter(); k3_depth = collections.Counter(); fam_time = collections.Counter()
for t, d, f in ev:
    dt = t - last
    if dt > 0:
        by_depth[depth] += dt
        k3_depth[sum(v for k, v in live.items() if k.startswith("k3_"))] += dt
        for k, v in live.items():
            if v:
                fam_time[k] += dt
    live[f] += d; depth += d; last = t
by_depth[depth] += hi - last
span = hi - lo
print("in it -- no kernel running: %.2f %%" % (100.0 * by_depth[0] / span))
print("network kernels in flight:  " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(k3_depth.items())))
print("all kernels in flight:  " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(by_depth.items()) if v / span > 0.002))
print("share of it with at least one kernel of the family running:  " + "  ".join("%s %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(fam_time.items(), key=lambda x: -x[1])))
