#!/bin/bash
OUT=gpurun_out/r5u; mkdir -p $OUT
python3 tools/time_run_detect.py --reads 10000 --keep --stats $OUT/s0.json > $OUT/run0.log 2>&1
for i in 1 2 3 4; do DN_TRACE_SUBMIT=1 python3 tools/time_run_detect.py --reads 10000 --reuse --stats $OUT/s$i.json > $OUT/run$i.log 2> $OUT/err$i.log; done
for i in 1 2 3 4; do python3 -c "
import json
d=json.load(open('$OUT/s$i.json')); r=d['ranks'][0]; print('run $i stream %.2f upload %.2f submit %.2f collect_wait %.2f' % (d['stream_s'], r['upload_s'], r['driver_submit_s'], r['collect_wait_s']))"; grep "submit tag" $OUT/err$i.log | awk '{printf "%s/%s ", \$8, \$11}' ; echo; done | tee $OUT/summary.txt
