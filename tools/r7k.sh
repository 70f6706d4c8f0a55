#!/bin/bash
# 8 Mi rows per CNN pass as the default: the GPU suite, the default and mixed bench lines, the product driver at configs[2] size
OUT=gpurun_out/r7k; mkdir -p $OUT
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json; cut -c1-260 $OUT/bench_default.json; grep -o '"hbm": {[^}]*}' $OUT/bench_default.json
python3 bench.py --scope mixed --no-cpu-baseline > $OUT/bench_mixed.log 2>&1; tail -1 $OUT/bench_mixed.log > $OUT/bench_mixed.json; cut -c1-260 $OUT/bench_mixed.json; grep -o '"hbm": {[^}]*}' $OUT/bench_mixed.json
python3 tools/time_run_detect.py --reads 10000 --sha --keep --stats $OUT/run_detect_stats.json > $OUT/run_detect.log 2>&1; grep -h "process\|sha256\|wall\|reads ok" $OUT/run_detect.log
DN_CNN_ROWS=$((4 << 20)) python3 tools/time_run_detect.py --reads 10000 --sha --reuse > $OUT/run_detect_4mi.log 2>&1; grep -h "process\|sha256\|wall" $OUT/run_detect_4mi.log
