#!/bin/bash
# End-of-round evidence: (1) the default bench under rocprofv3 kernel stats, (2) HBM traffic of every kernel (two PMC passes),
# (3) instruction mix.  Everything lands in gpurun_out/final/.
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/final/bench_under_rocprof.log 2>&1
grep '^{"metric' gpurun_out/final/bench_under_rocprof.log | tail -1 > gpurun_out/final/bench_under_rocprof.json
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/final/pmc_$c -o p -- python3 bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > gpurun_out/final/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/final/pmc_insts -o p -- python3 bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > gpurun_out/final/pmc_insts.log 2>&1
python3 bench.py > gpurun_out/final/bench_default.log 2>&1; tail -1 gpurun_out/final/bench_default.log > gpurun_out/final/bench_default.json
python3 bench.py --scope full --steps 8 --warmup 2 --inflight 2 > gpurun_out/final/bench_full.log 2>&1; tail -1 gpurun_out/final/bench_full.log > gpurun_out/final/bench_full.json
ls gpurun_out/final
