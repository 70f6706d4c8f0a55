#!/bin/bash
# reads per batch (the same 10 000 x 50 kb in fewer, larger or more, smaller steps; contexts in flight scaled to the same workspace total)
OUT=gpurun_out/r7r; mkdir -p $OUT; rm -f $OUT/ab.txt
for cfg in "500 6 20 5" "1000 3 10 3" "750 4 14 4" "334 9 30 8" "1000 4 10 3" "500 6 20 5"; do set -- $cfg
  line=$(timeout 900 python bench.py --no-cpu-baseline --fp32-steps 0 --reads-per-step $1 --inflight $2 --steps $3 --warmup $4 2>/dev/null | tail -1)
  echo "reads_per_step $1 inflight $2 steps $3 $(echo "$line" | grep -o '"value": [0-9.]*' | head -1) $(echo "$line" | grep -o '"hbm": {[^}]*}')" | tee -a $OUT/ab.txt
done
