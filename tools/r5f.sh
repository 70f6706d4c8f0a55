#!/bin/bash
OUT=gpurun_out/r5f; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_run_detect.py tests/test_gpu_cnn.py tests/test_gpu_cnn_fuzz.py tests/test_gpu_mixed.py tests/test_gpu_collect.py -x -q -s > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
grep -E "seed 1[12]" $OUT/pytest.log | sort -u > $OUT/cnn_fuzz_table.txt; cat $OUT/cnn_fuzz_table.txt
timeout 900 python tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats.json --reuse > $OUT/run_detect.log 2>&1; tail -6 $OUT/run_detect.log
timeout 600 python tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats2.json --reuse > $OUT/run_detect2.log 2>&1; tail -4 $OUT/run_detect2.log
