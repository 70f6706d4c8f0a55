#!/bin/bash
# round 5: run_detect's host teams against the cgroup CPU quota: DN_HOST_THREADS = 16 (the quota: today's default) / 8 / 6 / 4, alternating, throttling counted per run
OUT=gpurun_out/r5s; mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/time_run_detect.py --reads 10000 --keep --stats $OUT/warm.json > $OUT/warm.log 2>&1
for rep in 1 2 3; do for t in 16 8 6 4; do
  a=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); n0=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  DN_HOST_THREADS=$t python3 tools/time_run_detect.py --reads 10000 --reuse --stats $OUT/s.json > $OUT/run.log 2>&1
  b=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); n1=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  python3 - $t $a $b $n0 $n1 <<'PY' >> gpurun_out/r5s/log.txt
import json,sys
t,a,b,n0,n1=sys.argv[1:]
d=json.load(open('gpurun_out/r5s/s.json')); r=d['ranks'][0]
w=[l for l in open('gpurun_out/r5s/run.log') if 'run_detect wall' in l][0].split('):')[1].split(',')[0]
print('threads %2s: stream %.2f s  wall%s  submit %.2f collect_wait %.2f load %.2f format %.2f write %.2f | throttled %.1f s in %d periods' % (t, d['stream_s'], w, r['driver_submit_s'], r['collect_wait_s'], r['load_s'], r['format_s'], r['write_s'], (int(b)-int(a))/1e6, int(n1)-int(n0)))
PY
done; done
cat $OUT/log.txt
