#!/bin/bash
# stream priorities of the per-batch streams (DN_STREAM_PRIO: 1 highest = default, 0 plain, 2 lowest) against the CNN lanes' (DN_LANE_PRIO: 0 lowest = default, 1 middle, 2 highest)
OUT=gpurun_out/r7e; mkdir -p $OUT; rm -f $OUT/ab.txt
for rep in 1 2; do
for cfg in "1 0" "0 1" "2 2" "1 2" "0 0" "2 0"; do set -- $cfg
  v=$(DN_STREAM_PRIO=$1 DN_LANE_PRIO=$2 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "rep $rep stream_prio $1 lane_prio $2 $v" | tee -a $OUT/ab.txt
done; done
