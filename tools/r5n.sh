#!/bin/bash
OUT=gpurun_out/r5n; mkdir -p $OUT; rm -f $OUT/check.txt
for b in k3_b64_v2t k3_b64_v2t90 k3_b64_v3t; do echo "== $b" >> $OUT/check.txt; timeout 300 tools/_bin/$b 1200128 5 >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt; done
grep -E "^==|BLOCK64=[2]|stage|conv |workgroup|RESULT" $OUT/check.txt
