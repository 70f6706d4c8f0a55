#!/bin/bash
OUT=gpurun_out/r6n; mkdir -p $OUT; rm -f $OUT/check.txt
timeout 300 tools/_bin/k3_one256_trace 1200128 256 3 >> $OUT/check.txt 2>&1
cat $OUT/check.txt
