#!/bin/bash
OUT=gpurun_out/r7y; mkdir -p $OUT; rm -f $OUT/ab.txt
for q in 8 12 16 8; do
  line=$(GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1)
  echo "GPU_MAX_HW_QUEUES $q $(echo "$line" | grep -o '"value": [0-9.]*' | head -1)" | tee -a $OUT/ab.txt
done
