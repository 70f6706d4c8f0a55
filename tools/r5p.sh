#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
python3 tools/time_run_detect.py --reads 10000 --keep > gpurun_out/r05/run_detect_again.log 2>&1
sleep 20
python3 tools/time_run_detect.py --reads 10000 --ranks 2 --inflight 3 --sha --reuse --stats $OUT/run_detect_2ranks_gloo_stats.json > $OUT/run_detect_2ranks.log 2>&1
sleep 20
python3 tools/time_run_detect.py --reads 10000 --sha --reuse > $OUT/run_detect_1rank_sha.log 2>&1
grep -E "sha256|wall|stream itself" $OUT/run_detect_2ranks.log $OUT/run_detect_1rank_sha.log gpurun_out/r05/run_detect_again.log | cut -c1-300
