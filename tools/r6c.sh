#!/bin/bash
mkdir -p gpurun_out/r6c gpurun_out/r05
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r6c/pytest_gpu.log 2>&1; tail -3 gpurun_out/r6c/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c/smoke.log 2>&1; tail -1 gpurun_out/r6c/smoke.log
python3 bench.py > gpurun_out/r05/bench_default.log 2>&1; tail -1 gpurun_out/r05/bench_default.log > gpurun_out/r05/bench_default.json; tail -1 gpurun_out/r05/bench_default.json | cut -c1-300
