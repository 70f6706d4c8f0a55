#!/bin/bash
OUT=gpurun_out/r6d; mkdir -p $OUT; rm -f $OUT/check.txt
for rep in 1 2; do for b in k3_pair128_check k3_pair128_check_s2; do echo "== $b" >> $OUT/check.txt; timeout 300 tools/_bin/$b 1200128 128 4 >> $OUT/check.txt 2>&1; done; done
timeout 300 tools/_bin/k3_pair128_check_s2 1200128 64 4 >> $OUT/check.txt 2>&1
cat $OUT/check.txt
