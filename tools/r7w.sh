#!/bin/bash
OUT=gpurun_out/r7w; mkdir -p $OUT; rm -f $OUT/ab.txt
cp dnascent_amd/lib/libdnascent_hip.so /tmp/lib_keep.so
for v in base f2 b2 b3 base b3; do
  cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
  line=$(timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1)
  echo "$v $(echo "$line" | grep -o '"value": [0-9.]*' | head -1)" | tee -a $OUT/ab.txt
done
cp /tmp/lib_keep.so dnascent_amd/lib/libdnascent_hip.so
