#!/bin/bash
# Round-2 evidence, everything into gpurun_out/r02/ (tools/r02_collect.py turns it into profiles/r02_*):
#   (1) the default bench exactly as the driver runs it (BASELINE configs[2]: 10 000 x 50 kb, full pipeline, CPU baseline beside it)
#   (2) the same workload under rocprofv3 --kernel-trace --stats (fewer steps: the trace of 25 batches is large)
#   (3) configs[1] (banded scope) bench, and its kernels' HBM traffic: two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
#   (4) the CNN in fp32-MFMA math for comparison
#   (5) K3 alone on 64 x 20 kb reads (1.2 M positions): per-layer table from the kernel trace, HBM traffic of its kernels (PMC, two passes)
OUT=gpurun_out/r02; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 6 > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
python3 bench.py --scope banded --no-cpu-baseline --steps 32 --warmup 8 > $OUT/bench_banded.log 2>&1; tail -1 $OUT/bench_banded.log > $OUT/bench_banded.json
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --scope banded --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  rm -f $OUT/pmc_$c/*kernel_trace.csv
done
python3 bench.py --cnn-math fp32 --no-cpu-baseline --steps 8 --warmup 6 > $OUT/bench_fp32.log 2>&1; tail -1 $OUT/bench_fp32.log > $OUT/bench_fp32.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time.log 2>&1
python3 tools/cnn_layers.py $(find $OUT/k3trace -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time.log | head -1 | cut -d" " -f2) > $OUT/k3_layers.txt
rm -rf $OUT/k3trace
python3 tools/gpu_cnn_time.py 64 20000 f16x3,bf16x6,fp32 2>&1 | grep "^math" > $OUT/k3_math_modes.txt
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/k3pmc_$c -o p -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3pmc_$c.log 2>&1
  rm -f $OUT/k3pmc_$c/*kernel_trace.csv
done
ls $OUT | head -40; tail -1 $OUT/bench_default.json | cut -c1-300
