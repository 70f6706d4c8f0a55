#!/bin/bash
OUT=gpurun_out/r7t; mkdir -p $OUT; rm -f $OUT/ab.txt
for rep in 1 2; do for pin in 0 1; do
  echo "pin $pin rep $rep $(timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 --pin $pin 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | head -1)" | tee -a $OUT/ab.txt
done; done
