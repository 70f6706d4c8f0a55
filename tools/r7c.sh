#!/bin/bash
OUT=gpurun_out/r7c; mkdir -p $OUT
export DN_RUN_DETECT_TIMING=1
python3 tools/time_run_detect.py --reads 2000 --keep > $OUT/plain.log 2>&1
python3 tools/time_run_detect.py --reads 2000 --reuse --rccl-group-of-one > $OUT/rccl.log 2>&1
python3 tools/time_run_detect.py --reads 2000 --reuse --ranks 1 > $OUT/x.log 2>&1
grep -h "set-up\|first batch\|process" $OUT/plain.log; echo ==; grep -h "set-up\|first batch\|process" $OUT/rccl.log
