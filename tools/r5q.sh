#!/bin/bash
mkdir -p gpurun_out/r5q
python3 tools/numa_probe.py > gpurun_out/r5q/numa_probe.txt 2>&1; cat gpurun_out/r5q/numa_probe.txt
lscpu | grep -i -E "numa|socket|model name|^CPU\(s\)" >> gpurun_out/r5q/numa_probe.txt
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset.mems.effective >> gpurun_out/r5q/numa_probe.txt 2>&1
tail -8 gpurun_out/r5q/numa_probe.txt
