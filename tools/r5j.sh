#!/bin/bash
OUT=gpurun_out/r5j; mkdir -p $OUT
for b in b64_narrow b64_wide b64_wide_yp b64_wide_yp_cp b64_narrow b64_wide; do echo "== $b" >> $OUT/check.txt; timeout 300 tools/_bin/$b 1200128 5 >> $OUT/check.txt 2>&1; done
echo "== trace (wide)" >> $OUT/check.txt; timeout 300 tools/_bin/k3_block64_trace 1200128 3 >> $OUT/check.txt 2>&1
grep -E "^==|BLOCK64=2|stage|conv|RESULT" $OUT/check.txt
