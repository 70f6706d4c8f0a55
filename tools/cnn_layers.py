"""Per-layer table of the last CNN run in a rocprofv3 kernel trace: python tools/cnn_layers.py <trace.csv> <n_positions>"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnascent_amd import cnn_model
desc, _, _ = cnn_model.default_model()
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k3_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
npos = int(sys.argv[2]); n = len(desc["ops"]); tot = 0; agg = {}
for o, r in zip(desc["ops"], rows[-n:]):
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += dt
    if o["op"] == "conv":
        key = "conv k%-2d %3d->%3d" % (o["k"], o["cin"], o["cout"]); fl = 2 * o["k"] * o["cin"] * o["cout"] * npos
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += fl
    elif o["op"] == "dwconv":
        key = "dw   k%-2d %3d     " % (o["k"], o["c"]); a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += dt; a[2] += npos * o["c"] * 8
    else:
        a = agg.setdefault(o["op"], [0, 0.0, 0.0]); a[0] += 1; a[1] += dt
for k, (c, dt, w) in agg.items():
    extra = ("%7.1f TFLOP/s" % (w / dt / 1e6)) if k.startswith("conv") else (("%7.0f GB/s" % (w / dt / 1e3)) if k.startswith("dw") else "")
    print("%-18s x%-2d %9.1f us %s" % (k, c, dt, extra))
print("total %.2f ms" % (tot / 1e3))
