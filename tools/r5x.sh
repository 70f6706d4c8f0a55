#!/bin/bash
# round 5: what the contract's two synchronisations cost: the same bench at --steps 10 / 20 / 40 (the pipeline starts empty and drains inside the timed region)
OUT=gpurun_out/r5x; mkdir -p $OUT; rm -f $OUT/steps.txt
for k in 20 10 40; do
  timeout 1200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --fp32-steps 0 > $OUT/bench_$k.log 2>&1
  tail -1 $OUT/bench_$k.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $k: value %.1f Msamples/s  %.2f ms/step  total %.0f ms' % (d['value'], d['ms_per_step'], d['ms_per_step']*$k))" | tee -a $OUT/steps.txt
done
