#!/bin/bash
# usage: tools/prof_stats.sh <tag> <python script + args...>   -> gpurun_out/prof_<tag>/ + a top-20 kernel table on stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
grep -v "rocprofv3\|SQLite\|generateRocpd\|tool.cpp" gpurun_out/prof_$tag.log | tail -3
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
if not f: sys.exit("no kernel_stats.csv")
for i, r in enumerate(csv.DictReader(open(f[0]))):
    if i < 20: print("%-70s calls %6s total %9.3f ms avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
