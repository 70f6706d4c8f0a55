#!/bin/bash
OUT=gpurun_out/r5t; mkdir -p $OUT
python3 tools/h2d_probe.py > $OUT/probe1.txt 2>&1; cat $OUT/probe1.txt
python3 tools/time_run_detect.py --reads 10000 --keep --stats $OUT/s0.json > $OUT/run0.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --reuse --stats $OUT/s1.json > $OUT/run1.log 2>&1
python3 -c "
import json
for i in (0,1):
    d=json.load(open('$OUT/s%d.json'%i)); r=d['ranks'][0]; print('run %d stream %.2f upload %.2f submit %.2f collect_wait %.2f' % (i, d['stream_s'], r['upload_s'], r['driver_submit_s'], r['collect_wait_s']))" | tee $OUT/runs.txt
python3 tools/h2d_probe.py > $OUT/probe2.txt 2>&1; cat $OUT/probe2.txt
