#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json; tail -1 $OUT/bench_default.json | cut -c1-400
