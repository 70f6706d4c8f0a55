#!/bin/bash
OUT=gpurun_out/r6d; mkdir -p $OUT; rm -f $OUT/check.txt
for args in "4096 128 2" "25600 64 2" "1200128 128 4" "1200128 64 4" "1200128 128 3 6"; do echo "== $args" >> $OUT/check.txt; timeout 300 tools/_bin/k3_pair128_check $args >> $OUT/check.txt 2>&1; done
timeout 300 tools/_bin/k3_pair128_trace 1200128 128 3 >> $OUT/check.txt 2>&1
cat $OUT/check.txt
