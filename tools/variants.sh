#!/bin/bash
# A/B of fill kernel variants in separate short bench runs (same box)
mkdir -p gpurun_out
DN_FILL_VARIANT=$1 timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for v in "$@"; do
  DN_FILL_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_launch']
print('variant $v: %.1f Msamples/s %.2f ms/step fill %.2f ms chase+post %.2f frac %.4f' % (d['value'], d['ms_per_step'], k['k2_fill'], k['k2_chase+k2_post'], d['roofline']['frac']))"
done
