#!/bin/bash
# the product driver with a RCCL process group of one beside a plain 1-rank run: same SHA-256
OUT=gpurun_out/r7b; mkdir -p $OUT
python3 tools/time_run_detect.py --reads 2000 --sha --keep > $OUT/plain.log 2>&1
python3 tools/time_run_detect.py --reads 2000 --sha --reuse --rccl-group-of-one --stats $OUT/rccl_stats.json > $OUT/rccl.log 2>&1
tail -6 $OUT/plain.log; tail -12 $OUT/rccl.log
