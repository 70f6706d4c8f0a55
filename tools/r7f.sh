#!/bin/bash
# final-tree check: the whole GPU suite, smoke, the default bench line
OUT=gpurun_out/r7f; mkdir -p $OUT
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python3 bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json; cut -c1-330 $OUT/bench_default.json
