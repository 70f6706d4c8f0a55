#!/bin/bash
# round 5, first GPU session of k3_block64: bit-identity + time of the block alone, then the CNN tests and the variant digests
OUT=gpurun_out/r5a; mkdir -p $OUT
for args in "4096 2" "25600 2 3" "262144 3" "1200128 5"; do
  echo "== k3_block64_check $args" >> $OUT/check.txt
  timeout 300 tools/_bin/k3_block64_check $args >> $OUT/check.txt 2>&1; echo "exit $?" >> $OUT/check.txt
done
cat $OUT/check.txt
timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_cnn_fuzz.py -x -q > $OUT/pytest_cnn.log 2>&1; tail -3 $OUT/pytest_cnn.log
timeout 1200 python tools/variant_check.py > $OUT/variants.txt 2>&1; cat $OUT/variants.txt
timeout 300 python tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/cnn_time.txt 2>&1; DN_CNN_BLOCK64=0 timeout 300 python tools/gpu_cnn_time.py 64 20000 f16x3 >> $OUT/cnn_time.txt 2>&1; cat $OUT/cnn_time.txt
