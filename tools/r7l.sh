#!/bin/bash
# the parts of tools/r05_profile.sh that depend on the rows per CNN pass (8 Mi since the last change): bench line, rocprof stats, counter passes, mixed scope, product driver
OUT=gpurun_out/r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --fp32-steps 0 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for c in WRITE_SIZE FETCH_SIZE MFMA; do
  ctr=$c; [ $c = MFMA ] && ctr="$PMC_MFMA"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_full_$c -o p -- python3 bench.py --steps 8 --warmup 1 --inflight 6 --no-cpu-baseline --fp32-steps 0 > $OUT/pmc_full_$c.log 2>&1
  rm -f $OUT/pmc_full_$c/*kernel_trace.csv $OUT/pmc_full_$c/*/*kernel_trace.csv
done
#   (6) BASELINE configs[4]'s read-length law on one GPU (bench.py --scope mixed), plan order and long-first order
python3 bench.py --scope mixed --no-cpu-baseline --fp32-steps 0 --warmup 3 > $OUT/bench_mixed.log 2>&1; tail -1 $OUT/bench_mixed.log > $OUT/bench_mixed.json
python3 bench.py --scope mixed --order long-first --no-cpu-baseline --fp32-steps 0 --warmup 3 > $OUT/bench_mixed_longfirst.log 2>&1; tail -1 $OUT/bench_mixed_longfirst.log > $OUT/bench_mixed_longfirst.json
#   (7) the PRODUCT driver at configs[2] size: container -> run_detect -> .detect (tools/time_run_detect.py)
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats.json --keep > $OUT/run_detect.log 2>&1
python3 tools/time_run_detect.py --reads 10000 --stats $OUT/run_detect_stats_warm.json --reuse > $OUT/run_detect_warm.log 2>&1
ls $OUT | head -60; tail -1 $OUT/bench_default.json | cut -c1-300
