#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "<counters>" <python script + args...>  -> gpurun_out/pmc_<tag>/  (own run: --pmc + kernel trace only)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
grep -v "rocprofv3\|SQLite\|generateRocpd\|tool.cpp\|output_stream" gpurun_out/pmc_$tag.log | tail -2
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)
if not f: sys.exit("no counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
for k in agg:
    print("%-42s x%-4d " % (k, n[k]) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(agg[k].items())))
PY
