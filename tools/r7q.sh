#!/bin/bash
# lanes x batches in flight at 8 Mi rows per pass
OUT=gpurun_out/r7q; mkdir -p $OUT; rm -f $OUT/ab.txt
for cfg in "8 4 6" "8 4 5" "8 4 7" "8 5 6" "8 5 7" "8 4 6"; do set -- $cfg
  line=$(DN_CNN_ROWS=$(($1 << 20)) DN_CNN_LANES=$2 timeout 600 python bench.py --no-cpu-baseline --inflight $3 2>/dev/null | tail -1)
  echo "rows ${1}Mi lanes $2 inflight $3 $(echo "$line" | grep -o '"value": [0-9.]*' | head -1) $(echo "$line" | grep -o '"hbm": {[^}]*}')" | tee -a $OUT/ab.txt
done
