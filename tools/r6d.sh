#!/bin/bash
OUT=gpurun_out/r6d; mkdir -p $OUT; rm -f $OUT/check.txt
for b in k3_pair128_check_sh k3_pair128_check k3_pair128_check_sh0; do for args in "25600 128 2" "1200128 128 4" "1200128 64 4"; do echo "== $b $args" >> $OUT/check.txt; timeout 300 tools/_bin/$b $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done; done
echo "== chain of six" >> $OUT/check.txt; timeout 300 tools/_bin/k3_pair128_check_sh 1200128 128 3 6 >> $OUT/check.txt 2>&1
echo "== trace" >> $OUT/check.txt; timeout 300 tools/_bin/k3_pair128_trace_sh 1200128 128 3 >> $OUT/check.txt 2>&1
cat $OUT/check.txt
