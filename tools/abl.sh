#!/bin/bash
for a in 0 1 2 4 8 16 31; do
  DN_FILL_VARIANT=5 DN_FILL_ABL=$a timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $a: fill %.2f ms' % d['kernel_ms_per_launch']['k2_fill'])"
done
