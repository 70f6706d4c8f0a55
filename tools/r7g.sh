#!/bin/bash
# CNN lanes x batches in flight, re-measured with the whole-CU kernels of round 5 in the network
OUT=gpurun_out/r7g; mkdir -p $OUT; rm -f $OUT/ab.txt
for cfg in "4 6" "3 6" "2 6" "5 6" "3 5" "4 5" "4 7" "3 7" "4 6"; do set -- $cfg
  v=$(DN_CNN_LANES=$1 timeout 600 python bench.py --no-cpu-baseline --inflight $2 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "lanes $1 inflight $2 $v" | tee -a $OUT/ab.txt
done
