#!/bin/bash
# quick GPU check: parity tests + short bench, prints the per-kernel table
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
tail -1 gpurun_out/bench_quick.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f Msamples/s  %.2f ms/step  roofline frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
for k,v in d['kernel_ms_per_launch'].items(): print('  %-18s %8.3f' % (k,v))
" || tail -20 gpurun_out/bench_quick.log
