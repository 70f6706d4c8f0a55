#!/bin/bash
OUT=gpurun_out/r6a; mkdir -p $OUT; rm -f $OUT/check.txt
for b in k3_conv_ws_p1a0 k3_conv_ws_p1a1; do for args in "25600 17 128 256 1 2" "1200128 17 128 256 1 4" "1200128 9 128 128 1 4"; do echo "== $b $args" >> $OUT/check.txt; timeout 300 tools/_bin/$b $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done; done
echo "== trace" >> $OUT/check.txt; timeout 300 tools/_bin/k3_conv_ws_trace 1200128 17 128 256 1 3 >> $OUT/check.txt 2>&1
cat $OUT/check.txt
