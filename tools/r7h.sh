#!/bin/bash
# k3_sep_ws: weight fragments one k16 step ahead (bd1) against a whole channel block ahead (bd2, K3_WS_BDEPTH=2): digests + the network per layer, same session
OUT=gpurun_out/r7h; mkdir -p $OUT; rm -f $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
cp dnascent_amd/lib/libdnascent_hip.so /tmp/lib_keep.so
for v in bd1 bd2; do
  cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
  echo "== $v: $(python tools/variant_check.py --child 2>&1 | grep DIGEST)" | tee -a $OUT/ab.txt
done
for rep in 1 2 3; do for v in bd1 bd2; do
  cp tools/_bin/lib_$v/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
  rm -rf $OUT/k3trace_$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT/k3trace_$v -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > $OUT/k3_time_$v.log 2>&1
  python3 tools/cnn_layers.py $(find $OUT/k3trace_$v -name "*kernel_trace.csv" | head -1) $(grep -o "positions [0-9]*" $OUT/k3_time_$v.log | head -1 | cut -d" " -f2) > $OUT/k3_layers_${v}_$rep.txt
  echo "-- $v rep $rep: $(grep -E '^total' $OUT/k3_layers_${v}_$rep.txt)  $(grep -E 'sepws' $OUT/k3_layers_${v}_$rep.txt | awk '{print $1,$2,$3,$4,$5,$6}' | tr '\n' ';')" | tee -a $OUT/ab.txt
  rm -rf $OUT/k3trace_$v
done; done
cp /tmp/lib_keep.so dnascent_amd/lib/libdnascent_hip.so
