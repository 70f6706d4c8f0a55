#!/bin/bash
# balanced CNN passes against the old fill-to-the-cap partition (DN_CNN_GREEDY_PASSES=1), and the cap
OUT=gpurun_out/r7j; mkdir -p $OUT; rm -f $OUT/ab.txt
python3 tools/time_run_detect.py --reads 2000 --sha --keep 2>&1 | grep sha256 | sed 's/^/balanced /' | tee -a $OUT/ab.txt
DN_CNN_GREEDY_PASSES=1 python3 tools/time_run_detect.py --reads 2000 --sha --reuse 2>&1 | grep sha256 | sed 's/^/greedy   /' | tee -a $OUT/ab.txt
rm -f /tmp/bench_reads.dnrc /tmp/bench_reads.detect
for cfg in "4 1" "4 0" "6 0" "8 0" "4 1" "4 0" "6 0" "8 0"; do set -- $cfg
  v=$(DN_CNN_ROWS=$(($1 << 20)) DN_CNN_GREEDY_PASSES=$2 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "rows_per_pass ${1}Mi greedy $2 $v" | tee -a $OUT/ab.txt
done
