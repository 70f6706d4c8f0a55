#!/bin/bash
# Round-2 evidence, everything into gpurun_out/r02/ (tools/r02_collect.py turns it into profiles/r02_*):
#   (1) the default bench (BASELINE configs[2]: 10 000 x 50 kb, full pipeline, CPU baseline beside it)
#   (2) the same workload under rocprofv3 --kernel-trace --stats (fewer steps: the trace of 22 batches is large)
#   (3) configs[1] (banded scope) bench, and its kernels' HBM traffic: two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
#   (4) the CNN in fp32-MFMA math for comparison
OUT=gpurun_out/r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
python3 bench.py --scope banded --no-cpu-baseline --steps 32 --warmup 8 > $OUT/bench_banded.log 2>&1; tail -1 $OUT/bench_banded.log > $OUT/bench_banded.json
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --scope banded --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
python3 bench.py --cnn-math fp32 --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_fp32.log 2>&1; tail -1 $OUT/bench_fp32.log > $OUT/bench_fp32.json
ls $OUT $OUT/stats | head -40
