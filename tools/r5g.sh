#!/bin/bash
# round 5, verdict item 4: k2b_eventalign under a register budget (dynamic LDS + __launch_bounds__ waves per SIMD): digests and the pipeline, same session
OUT=gpurun_out/r5g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_cnn_fuzz.py -x -q -s > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log; grep -E "seed 1[12]" $OUT/fuzz.log | sort -u > $OUT/cnn_fuzz_table.txt
for v in base k2b_eu3 k2b_eu4; do
  if [ $v = base ]; then cp tools/_bin/lib_base.so dnascent_amd/lib/libdnascent_hip.so; else cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so; fi
  echo "== $v: $(python tools/variant_check.py --child 2>&1 | grep DIGEST)" | tee -a $OUT/ab.txt
done
for rep in 1 2; do for v in base k2b_eu3 k2b_eu4; do
  if [ $v = base ]; then cp tools/_bin/lib_base.so dnascent_amd/lib/libdnascent_hip.so; else cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so; fi
  timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --fp32-steps 0 > $OUT/bench_$v.log 2>&1
  tail -1 $OUT/bench_$v.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_launch']; s=d['kernel_ms_solo']
print('$v value %.1f Msamples/s  %.2f ms/step   k2b_viterbi in flight %.1f ms, alone %.1f ms; hbm %s' % (d['value'], d['ms_per_step'], k.get('k2b_viterbi',0), s.get('k2b_viterbi',0), d.get('hbm')))" | tee -a $OUT/ab.txt
done; done
cp tools/_bin/lib_base.so dnascent_amd/lib/libdnascent_hip.so
