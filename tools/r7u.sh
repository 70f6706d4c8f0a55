#!/bin/bash
OUT=gpurun_out/r7u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_rccl_group_of_one.py tests/test_gpu_run_detect.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
python3 tools/time_run_detect.py --reads 10000 --sha --keep > $OUT/plain.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --sha --reuse --rccl-group-of-one --stats $OUT/rccl_stats.json > $OUT/rccl.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --sha --reuse > $OUT/plain2.log 2>&1
for f in plain rccl plain2; do grep -h "process \|sha256" $OUT/$f.log; done
