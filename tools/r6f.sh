#!/bin/bash
# round 5: does a CU partition pay now that a third of the network's time is in kernels that need whole CUs (k3_block64, k3_pair128)?
OUT=gpurun_out/r6f; mkdir -p $OUT; rm -f $OUT/ab.txt
run() { # name, env...
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 14 --warmup 4 --no-cpu-baseline --fp32-steps 0 > $OUT/bench_$name.log 2>&1
  tail -1 $OUT/bench_$name.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_launch']
print('$name: value %.1f Msamples/s  %.2f ms/step  k3 in flight %.0f ms  k2_fill %.0f  k2b %.0f' % (d['value'], d['ms_per_step'], k.get('k3_cnn',0), k.get('k2_fill',0), k.get('k2b_viterbi',0)))" | tee -a $OUT/ab.txt
}
for rep in 1 2; do
  run default DN_X=0
  run front32 DN_FRONT_CUS=32 DN_CNN_WS_WGS=224
  run front64 DN_FRONT_CUS=64 DN_CNN_WS_WGS=192
  run front16 DN_FRONT_CUS=16 DN_CNN_WS_WGS=240
done
