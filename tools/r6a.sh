#!/bin/bash
OUT=gpurun_out/r6a; mkdir -p $OUT; rm -f $OUT/check.txt
for args in "1200128 17 128 256 1 4" "1200128 9 128 128 1 4" "1200128 9 64 128 1 4" "1200128 3 128 64 0 4"; do echo "== $args" >> $OUT/check.txt; timeout 300 tools/_bin/k3_conv_ws_check $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done
cat $OUT/check.txt
