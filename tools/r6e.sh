#!/bin/bash
# round 5: k3_pair128 in the product library: digests of every variant, the network per layer on / off, the pipeline A / B
OUT=gpurun_out/r6e; mkdir -p $OUT; rm -f $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python tools/variant_check.py > $OUT/variant_check.txt 2>&1; tail -14 $OUT/variant_check.txt
for v in 1 0; do
  rm -rf $OUT/k3trace
  DN_CNN_PAIR128=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time_$v.log 2>&1
  python3 tools/cnn_layers.py $(find $OUT/k3trace -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time_$v.log | head -1 | cut -d" " -f2) > $OUT/k3_layers_pair$v.txt
  echo "-- PAIR128=$v: $(grep -E '^total' $OUT/k3_layers_pair$v.txt) | $(grep -E 'pair|sep   k9' $OUT/k3_layers_pair$v.txt | tr -s ' ' | tr '\n' ';')" | tee -a $OUT/ab.txt
  rm -rf $OUT/k3trace
done
for rep in 1 2; do for v in 1 0; do
  DN_CNN_PAIR128=$v timeout 900 python bench.py --steps 14 --warmup 4 --no-cpu-baseline --fp32-steps 0 > $OUT/bench_$v.log 2>&1
  tail -1 $OUT/bench_$v.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PAIR128=$v: value %.1f Msamples/s  %.2f ms/step  network alone %.1f ms' % (d['value'], d['ms_per_step'], d['kernel_ms_solo'].get('k3_cnn',0)))" | tee -a $OUT/ab.txt
done; done
