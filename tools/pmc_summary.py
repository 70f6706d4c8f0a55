"""Summarise a rocprofv3 --pmc counter_collection.csv of the CNN timing run into profiles/<out>.json (MFMA utilisation per kernel)."""
import csv, collections, json, glob, sys
src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "k3_" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); n[k] += 1
res = {"command": cmd, "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "kernels": {}}
for k, c in agg.items():
    d = dict(c); d["launches"] = n[k]
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        d["MfmaUtil"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    res["kernels"][k] = d
json.dump(res, open(out, "w"), indent=1)
for k, d in res["kernels"].items(): print("%-40s x%-4d MfmaUtil %s" % (k[:40], d["launches"], d.get("MfmaUtil")))
