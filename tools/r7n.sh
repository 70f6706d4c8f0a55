#!/bin/bash
# run_detect with the lanes' buffers taken by the helper thread (dn_cnn_reserve): cold, then runs right behind each other, 4 Mi and 8 Mi rows per pass
OUT=gpurun_out/r7n; mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_run_detect.py tests/test_gpu_mixed.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
export DN_RUN_DETECT_TIMING=1
run() { python3 tools/time_run_detect.py --reads 10000 --stats $OUT/stats_$1.json $2 > $OUT/run_$1.log 2>&1
  echo "$1: $(grep -h 'process ' $OUT/run_$1.log | sed 's/run_detect: //')  $(python3 -c "import json; r=json.load(open('$OUT/stats_$1.json'))['ranks'][0]; print('upload', r['upload_s'], 'enqueue', r['enqueue_s'])")" | tee -a $OUT/ab.txt; }
run cold_4mi --keep
run warm_4mi --reuse
DN_CNN_ROWS=$((8 << 20)) run warm_8mi_a --reuse
DN_CNN_ROWS=$((8 << 20)) run warm_8mi_b --reuse
run warm_4mi_b --reuse
grep -h "set-up" $OUT/run_warm_8mi_b.log | head -6
