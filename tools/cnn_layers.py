"""Per-layer table of the last CNN run in a rocprofv3 kernel trace: python tools/cnn_layers.py <trace.csv> <n_positions>
A fused SeparableConv1D (k3_sep_*) covers a depthwise op and the pointwise conv after it."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnascent_amd import cnn_model
desc, _, _ = cnn_model.default_model()
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k3_" in r["Kernel_Name"]]
is_sort = lambda r: "k3_encode_len" in r["Kernel_Name"] or "k3_encode_perm" in r["Kernel_Name"]
sort_us = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if is_sort(r))
rows = [r for r in rows if not is_sort(r) and "k3_layout" not in r["Kernel_Name"] and "k3_range_check" not in r["Kernel_Name"]]      # the encoder's counting sort (two small kernels) is reported on its own line
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
npos = int(sys.argv[2]); ops = desc["ops"]
# kernels of one run: walk the op list backwards from the end of the trace
runs = max(1, sum(1 for r in rows if "k3_encode(" in r["Kernel_Name"] or "k3_encode_mfma(" in r["Kernel_Name"]))
per_run = len(rows) // runs
last = rows[-per_run:]
tot = 0; agg = {}; i = 0
for r in last:
    o = ops[i]
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += dt
    if "k3_block64" in r["Kernel_Name"]:                    # a whole residual block (13 ops) or its six separable layers (12) in one launch
        span = 13 if "k3_block64<true>" in r["Kernel_Name"] or "ILb1E" in r["Kernel_Name"] else 12
        fl = sum(2 * (q["k"] * q["c"] if q["op"] == "dwconv" else q["k"] * q["cin"] * q["cout"]) for q in ops[i:i + span]) * npos
        key = "block k5 64 (%d ops)" % span
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += fl; a[3] += npos * (64 + 64 + (64 if span == 13 else 0)) * 4
        i += span; continue
    if "k3_pair128" in r["Kernel_Name"]:                    # two separable layers (4 ops) in one launch
        p0, p1 = ops[i + 1], ops[i + 3]
        fl = sum(2 * (q["k"] * q["c"] if q["op"] == "dwconv" else q["k"] * q["cin"] * q["cout"]) for q in ops[i:i + 4]) * npos
        key = "pair  k9 %3d->128->128" % p0["cin"]
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += fl; a[3] += npos * (p0["cin"] + p1["cout"]) * 4
        i += 4; continue
    if "k3_sep" in r["Kernel_Name"]:
        pw = ops[i + 1]
        kind = "ws" if "k3_sep_ws" in r["Kernel_Name"] else "un" if "k3_sep_uni" in r["Kernel_Name"] else "  "
        key = "sep%s k%-2d %3d->%3d" % (kind, o["k"], pw["cin"], pw["cout"]); fl = 2 * (pw["cin"] * pw["cout"] + o["k"] * o["c"]) * npos
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += fl; a[3] += npos * (pw["cin"] + pw["cout"]) * 4
        i += 2; continue
    if o["op"] == "conv":
        key = "conv  k%-2d %3d->%3d" % (o["k"], o["cin"], o["cout"]); fl = 2 * o["k"] * o["cin"] * o["cout"] * npos
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += fl; a[3] += npos * (o["cin"] + o["cout"]) * 4
    elif o["op"] == "dwconv":
        key = "dw    k%-2d %3d     " % (o["k"], o["c"]); a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[3] += npos * o["c"] * 8
    else:
        a = agg.setdefault(o["op"], [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dt
    i += 1
assert i == len(ops), (i, len(ops))
for k, (c, dt, fl, by) in agg.items():
    extra = ""
    if fl: extra += "%7.1f TFLOP/s" % (fl / dt / 1e6)
    if by: extra += "  %5.0f GB/s of layer I/O" % (by / dt / 1e3)
    print("%-22s x%-2d %9.1f us %s" % (k, c, dt, extra))
print("%-20s x%-2d %9.1f us" % ("encoder sort", 1, sort_us / runs)); tot += sort_us / runs
print("total %.2f ms" % (tot / 1e3))
