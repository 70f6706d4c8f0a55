#!/bin/bash
# round 5: third version of k3_block64 (filter in the accumulator layout) against the layer-by-layer kernels
OUT=gpurun_out/r5m; mkdir -p $OUT
for args in "4096 2" "25600 2 3" "1200128 5"; do echo "== check $args" >> $OUT/check.txt; timeout 300 tools/_bin/k3_block64_check3 $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done
tail -40 $OUT/check.txt
