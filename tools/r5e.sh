#!/bin/bash
# round 5: the full GPU suite on the fused-block default, K3 alone layer by layer, and the pipeline A / B (DN_CNN_BLOCK64 = 2 against 0) in one session
OUT=gpurun_out/r5e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/stats.log 2>&1
python3 tools/cnn_layers.py $OUT/stats/k3_kernel_trace.csv 1200057 > $OUT/k3_layers.txt; cat $OUT/k3_layers.txt
for v in 2 0 2 0; do
  DN_CNN_BLOCK64=$v timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $OUT/bench_b$v.log 2>&1
  tail -1 $OUT/bench_b$v.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BLOCK64=$v value %.1f Msamples/s  %.2f ms/step' % (d['value'], d['ms_per_step']))" || tail -5 $OUT/bench_b$v.log
done
