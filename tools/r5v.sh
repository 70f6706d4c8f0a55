#!/bin/bash
# round 5: run_detect's depth: fewer contexts = fewer GB to hipMalloc while the stream is already running
OUT=gpurun_out/r5v; mkdir -p $OUT; rm -f $OUT/summary.txt
python3 tools/time_run_detect.py --reads 10000 --keep --stats $OUT/s.json > $OUT/run.log 2>&1
for rep in 1 2; do for d in 8 6 5 4; do
  python3 tools/time_run_detect.py --reads 10000 --reuse --inflight $d --stats $OUT/s.json > $OUT/run.log 2>&1
  python3 -c "
import json
d=json.load(open('$OUT/s.json')); r=d['ranks'][0]
w=[l for l in open('$OUT/run.log') if 'run_detect wall' in l][0].split('):')[1].split(',')[0]
print('inflight $d: stream %.2f s, wall%s, submit %.2f (upload %.2f) collect_wait %.2f, hbm %s' % (d['stream_s'], w, r['driver_submit_s'], r['upload_s'], r['collect_wait_s'], d['hbm']))" | tee -a $OUT/summary.txt
done; done
