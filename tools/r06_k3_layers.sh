#!/bin/bash
# K3 alone on 64 x 20 kb reads (1.2 M positions), layer by layer from the kernel trace, WITHOUT the canary (DN_CNN_CANARY=0: the per-layer table and the math-mode
# figures are then comparable with earlier rounds'), and the f16x3 figure once more with it.  -> gpurun_out/r06b/
cd "${GRAFT_REPO_ROOT:?run under gpurun}" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r06b"; mkdir -p "$OUT"
export DN_CNN_CANARY=0
rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time.log 2>&1
python3 tools/cnn_layers.py $(find $OUT/k3trace -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time.log | head -1 | cut -d" " -f2) > $OUT/k3_layers.txt 2> $OUT/k3_layers.err
rm -rf $OUT/k3trace
python3 tools/gpu_cnn_time.py 64 20000 f16x3,bf16x6,fp32 2>&1 | grep "^math" > $OUT/k3_math_modes.txt
unset DN_CNN_CANARY
python3 tools/gpu_cnn_time.py 64 20000 f16x3 2>&1 | grep "^math" > $OUT/k3_math_modes_with_canary.txt
cat $OUT/k3_layers.txt $OUT/k3_layers.err $OUT/k3_math_modes.txt $OUT/k3_math_modes_with_canary.txt
