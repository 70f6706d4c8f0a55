#!/bin/bash
mkdir -p gpurun_out/r5l
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r5l/pytest_gpu.log 2>&1; tail -3 gpurun_out/r5l/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5l/smoke.log 2>&1; tail -1 gpurun_out/r5l/smoke.log
bash tools/r05_profile.sh > gpurun_out/r5l/profile.log 2>&1; tail -5 gpurun_out/r5l/profile.log
