#!/bin/bash
# round 5: bench.py's depth (contexts in flight): 8 (default) / 6 / 5 / 4, alternating, one session
OUT=gpurun_out/r5w; mkdir -p $OUT; rm -f $OUT/ab.txt
for rep in 1 2; do for d in 8 5 4 6; do
  timeout 900 python bench.py --steps 14 --warmup 4 --inflight $d --no-cpu-baseline --fp32-steps 0 > $OUT/bench_$d.log 2>&1
  tail -1 $OUT/bench_$d.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_launch']
print('inflight $d: value %.1f Msamples/s  %.2f ms/step  k3 in flight %.0f ms  hbm %s' % (d['value'], d['ms_per_step'], k.get('k3_cnn',0), d.get('hbm')))" | tee -a $OUT/ab.txt
done; done
