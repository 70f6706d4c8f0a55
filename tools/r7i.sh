#!/bin/bash
# activation rows per CNN pass and lane, re-measured with the persistent whole-CU kernels in the network (default 4 Mi)
OUT=gpurun_out/r7i; mkdir -p $OUT; rm -f $OUT/ab.txt
for mi in 4 3 5 6 8 4; do
  v=$(DN_CNN_ROWS=$((mi << 20)) timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "rows_per_pass ${mi}Mi $v" | tee -a $OUT/ab.txt
done
