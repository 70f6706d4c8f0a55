#!/bin/bash
# k3_sep_ws weight prefetch depth in the PIPELINE (four lanes contending for L2 / HBM), same session
OUT=gpurun_out/r7s; mkdir -p $OUT; rm -f $OUT/ab.txt
cp dnascent_amd/lib/libdnascent_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in bd1 bd2; do
  cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
  echo "$v rep $rep $(timeout 600 python bench.py --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | head -1)" | tee -a $OUT/ab.txt
done; done
cp /tmp/lib_keep.so dnascent_amd/lib/libdnascent_hip.so
