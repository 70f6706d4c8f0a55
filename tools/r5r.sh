#!/bin/bash
# round 5: why is run_detect's stream 7.6 s in some runs and 11 s in others?  six runs of the same command, host load around each
OUT=gpurun_out/r5r; mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/time_run_detect.py --reads 10000 --keep --stats $OUT/s0.json > $OUT/run0.log 2>&1
for i in 1 2 3 4 5; do
  echo "== before run $i: loadavg $(cat /proc/loadavg) | cpu.stat $(grep -E 'nr_throttled|throttled_usec' /sys/fs/cgroup/cpu.stat | tr '\n' ' ') | mem pressure $(head -1 /proc/pressure/memory 2>/dev/null) | thp $(cat /sys/kernel/mm/transparent_hugepage/enabled) | $(grep -E 'AnonHugePages|MemFree|^Cached' /proc/meminfo | tr -s ' ' | tr '\n' ' ')" >> $OUT/log.txt
  python3 tools/time_run_detect.py --reads 10000 --reuse --stats $OUT/s$i.json > $OUT/run$i.log 2>&1
  sleep 5
done
python3 - <<'PY' >> gpurun_out/r5r/log.txt
import json
for i in range(6):
    try:
        d=json.load(open('gpurun_out/r5r/s%d.json'%i)); r=d['ranks'][0]
        print('run %d stream %.2f upload %.2f submit %.2f collect_wait %.2f load %.2f setup %.2f' % (i, d['stream_s'], r['upload_s'], r['driver_submit_s'], r['collect_wait_s'], r['load_s'], d['setup_s']))
    except Exception as e: print(i, e)
PY
cat $OUT/log.txt
