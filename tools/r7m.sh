#!/bin/bash
# the product driver at its own default (4 Mi rows per pass), cold and right behind its predecessor
OUT=gpurun_out/r05; mkdir -p $OUT
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats.json --keep > $OUT/run_detect.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats_warm.json --reuse > $OUT/run_detect_warm.log 2>&1
grep -h "process\|wall" $OUT/run_detect.log $OUT/run_detect_warm.log
