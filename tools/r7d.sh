#!/bin/bash
OUT=gpurun_out/r7d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_run_detect.py tests/test_gpu_mixed.py tests/test_gpu_rccl_group_of_one.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log
export DN_RUN_DETECT_TIMING=1
python3 tools/time_run_detect.py --reads 10000 --sha --keep > $OUT/plain.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --sha --reuse --rccl-group-of-one --stats $OUT/rccl_stats.json > $OUT/rccl.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --sha --reuse --rccl-group-of-one --central-writer > $OUT/rccl_central.log 2>&1
tail -4 $OUT/tests.log; grep -h "process\|sha256\|wall" $OUT/plain.log; echo ==; grep -h "process\|sha256\|wall" $OUT/rccl.log; echo ==; grep -h "process\|sha256\|wall" $OUT/rccl_central.log
