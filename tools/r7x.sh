#!/bin/bash
# the round's two fused kernel families switched off one at a time, in the pipeline at its final defaults (same session)
OUT=gpurun_out/r7x; mkdir -p $OUT; rm -f $OUT/ab.txt
for cfg in "2 1" "0 1" "2 0" "0 0" "2 1"; do set -- $cfg
  line=$(DN_CNN_BLOCK64=$1 DN_CNN_PAIR128=$2 timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1)
  echo "block64 $1 pair128 $2 $(echo "$line" | grep -o '"value": [0-9.]*' | head -1)" | tee -a $OUT/ab.txt
done
