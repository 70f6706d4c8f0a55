#!/bin/bash
OUT=gpurun_out/r5c; mkdir -p $OUT
for b in k3_block64_check k3_block64_check_ns; do
  for args in "4096 2" "25600 2 3" "1200128 5"; do
    echo "== $b $args" >> $OUT/check.txt
    timeout 300 tools/_bin/$b $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt
  done
done
for b in k3_block64_trace k3_block64_trace_ns; do echo "== $b" >> $OUT/check.txt; timeout 300 tools/_bin/$b 1200128 3 >> $OUT/check.txt 2>&1; done
cat $OUT/check.txt
