#!/bin/bash
# reads (wavefronts) per workgroup of the one-wavefront-per-read kernels, re-measured with the whole-CU kernels in the network: fewer CUs touched by the long-lived per-read wavefronts
OUT=gpurun_out/r7v; mkdir -p $OUT; rm -f $OUT/ab.txt
cp dnascent_amd/lib/libdnascent_hip.so /tmp/lib_keep.so
for v in base f8 f16 f8b6 b6 f8c4 base f8; do
  cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
  line=$(timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1)
  echo "$v $(echo "$line" | grep -o '"value": [0-9.]*' | head -1) $(echo "$line" | grep -o '"k2_fill": [0-9.]*') $(echo "$line" | grep -o '"k2b_viterbi": [0-9.]*') $(echo "$line" | grep -o '"k3_cnn": [0-9.]*')" | tee -a $OUT/ab.txt
done
cp /tmp/lib_keep.so dnascent_amd/lib/libdnascent_hip.so
