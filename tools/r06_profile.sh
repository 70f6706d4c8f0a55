#!/bin/bash
# Round-6 evidence, everything into gpurun_out/r06/ (tools/r06_collect.py turns it into profiles/r06_*):
#   (1) the default bench exactly as the driver runs it (BASELINE configs[2]: 10 000 x 50 kb, full pipeline, CPU baseline + fp32 leg)
#   (2) the same command line under rocprofv3 --kernel-trace --stats (the per-kernel averages must agree with the bench line's own HIP-event means)
#   (3) counter passes AT THE BENCH'S LAUNCH SHAPE (500 x 50 kb per step, 8 Mi-row CNN passes) and at the bench's DEPTH (8 batches in flight:
#       round-3 verdict / advisor -- the committed passes were taken with one), each its own run:
#       WRITE_SIZE | FETCH_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
#   (4) configs[1] (banded scope) bench, and the same three counter passes for it
#   (5) K3 alone on 64 x 20 kb reads (1.2 M positions): per-layer table from the kernel trace, math modes
# The program goes straight after `--` (python3 bench.py ...): no env / sh -c hop under rocprofv3.
cd "${GRAFT_REPO_ROOT:?run under gpurun (it exports GRAFT_REPO_ROOT)}" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r06"; rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --fp32-steps 0 --bf16-steps 0 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for c in WRITE_SIZE FETCH_SIZE MFMA; do
  ctr=$c; [ $c = MFMA ] && ctr="$PMC_MFMA"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_full_$c -o p -- python3 bench.py --steps 8 --warmup 1 --inflight 6 --no-cpu-baseline --fp32-steps 0 --bf16-steps 0 > $OUT/pmc_full_$c.log 2>&1
  rm -f $OUT/pmc_full_$c/*kernel_trace.csv $OUT/pmc_full_$c/*/*kernel_trace.csv
done
python3 bench.py --scope banded --no-cpu-baseline --steps 32 --warmup 8 > $OUT/bench_banded.log 2>&1; tail -1 $OUT/bench_banded.log > $OUT/bench_banded.json
for c in WRITE_SIZE FETCH_SIZE MFMA; do
  ctr=$c; [ $c = MFMA ] && ctr="$PMC_MFMA"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_banded_$c -o p -- python3 bench.py --scope banded --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/pmc_banded_$c.log 2>&1
  rm -f $OUT/pmc_banded_$c/*kernel_trace.csv $OUT/pmc_banded_$c/*/*kernel_trace.csv
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time.log 2>&1
python3 tools/cnn_layers.py $(find $OUT/k3trace -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time.log | head -1 | cut -d" " -f2) > $OUT/k3_layers.txt
rm -rf $OUT/k3trace
python3 tools/gpu_cnn_time.py 64 20000 f16x3,bf16x6,fp32 2>&1 | grep "^math" > $OUT/k3_math_modes.txt
#   (6) BASELINE configs[4]'s read-length law on one GPU (bench.py --scope mixed), plan order and long-first order
python3 bench.py --scope mixed --no-cpu-baseline --fp32-steps 0 --bf16-steps 0 --warmup 3 > $OUT/bench_mixed.log 2>&1; tail -1 $OUT/bench_mixed.log > $OUT/bench_mixed.json
python3 bench.py --scope mixed --order long-first --no-cpu-baseline --fp32-steps 0 --bf16-steps 0 --warmup 3 > $OUT/bench_mixed_longfirst.log 2>&1; tail -1 $OUT/bench_mixed_longfirst.log > $OUT/bench_mixed_longfirst.json
#   (7) the PRODUCT driver at configs[2] size: container -> run_detect -> .detect (tools/time_run_detect.py)
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats.json --keep > $OUT/run_detect.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats_warm.json --reuse > $OUT/run_detect_warm.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --ranks 2 --inflight 3 --sha --reuse --stats $OUT/run_detect_2ranks_gloo_stats.json > $OUT/run_detect_2ranks.log 2>&1     # two ranks share ONE GPU here: 3 contexts each (8 each do not fit 288 GB)
python3 tools/time_run_detect.py --reads 10000 --sha --reuse > $OUT/run_detect_1rank_sha.log 2>&1
DN_RUN_DETECT_SLOW_EXIT=1 python3 tools/time_run_detect.py --reads 10000 --reuse > $OUT/run_detect_slow_exit.log 2>&1
#   (8) the GPU test suite, as the driver runs it
python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
ls $OUT | head -60; tail -1 $OUT/bench_default.json | cut -c1-300
