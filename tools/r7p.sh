#!/bin/bash
# two passes per batch (12 Mi rows) on fewer lanes against the default (8 Mi, 4 lanes)
OUT=gpurun_out/r7p; mkdir -p $OUT; rm -f $OUT/ab.txt
for cfg in "8 4 6" "12 3 6" "12 2 6" "8 3 6" "12 3 5" "8 4 6" "12 3 6"; do set -- $cfg
  line=$(DN_CNN_ROWS=$(($1 << 20)) DN_CNN_LANES=$2 timeout 600 python bench.py --no-cpu-baseline --inflight $3 2>/dev/null | tail -1)
  echo "rows ${1}Mi lanes $2 inflight $3 $(echo "$line" | grep -o '"value": [0-9.]*' | head -1) $(echo "$line" | grep -o '"hbm": {[^}]*}')" | tee -a $OUT/ab.txt
done
