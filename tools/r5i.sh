#!/bin/bash
OUT=gpurun_out/r5i; mkdir -p $OUT
for args in "4096 2" "25600 2 3" "1200128 5"; do echo "== check $args" >> $OUT/check.txt; timeout 300 tools/_bin/k3_block64_check $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done
for b in k3_block64_trace k3_block64_trace_prio; do echo "== $b" >> $OUT/check.txt; timeout 300 tools/_bin/$b 1200128 3 >> $OUT/check.txt 2>&1; done
grep -E "^==|BLOCK64=[12]|differ|stage|conv|RESULT|exit" $OUT/check.txt
