#!/bin/bash
OUT=gpurun_out/r6g; mkdir -p $OUT; rm -f $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python tools/variant_check.py > $OUT/variant_check.txt 2>&1; tail -3 $OUT/variant_check.txt
timeout 1200 python -m pytest tests -x -q -m gpu -k "cnn or golden or smoke or edges or run_detect" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
rm -rf $OUT/k3trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time.log 2>&1
python3 tools/cnn_layers.py $(find $OUT/k3trace -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time.log | head -1 | cut -d" " -f2) > $OUT/k3_layers.txt; rm -rf $OUT/k3trace
grep -E "pair|total" $OUT/k3_layers.txt
for rep in 1 2; do
  timeout 900 python bench.py --steps 14 --warmup 4 --no-cpu-baseline --fp32-steps 0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f Msamples/s  %.2f ms/step  network alone %.1f ms' % (d['value'], d['ms_per_step'], d['kernel_ms_solo'].get('k3_cnn',0)))" | tee -a $OUT/ab.txt
done
