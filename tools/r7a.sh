#!/bin/bash
# RCCL executed on a 1-GPU box: the group-of-one test, then bench.py's N > 1 code path with a process group of one (DN_BENCH_FORCE_DIST=1) under torch.distributed.run
OUT=gpurun_out/r7a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_rccl_group_of_one.py -x -q -m gpu > $OUT/test.log 2>&1; echo "test rc $?" >> $OUT/test.log
DN_BENCH_FORCE_DIST=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 4 --warmup 2 > $OUT/bench_force_dist.log 2>&1; echo "bench rc $?" >> $OUT/bench_force_dist.log
tail -5 $OUT/test.log; tail -c 3000 $OUT/bench_force_dist.log
